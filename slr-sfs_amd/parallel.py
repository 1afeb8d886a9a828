"""Frame sharding across the GPUs of one node + clip assembly.

Frames of a clip are independent given the frame-invariant state: rank r renders
frames {t : t % world == r}; every rank runs the (cheap) per-clip work redundantly; the only
communication is the all-gather of the finished frames (RCCL over xGMI; gloo in the CPU tests):
either ONE collective after the last frame (gather_clip) or one small asynchronous collective per
round of `world` frames, hidden under the next round's rendering (ClipAssembler; what bench.py
uses -- a 60-frame fp32 clip is 708 MB, too much to leave exposed at the end of a 60 ms job).  The reference has no counterpart (it runs one process per scene on one GPU,
test_animating/CLAW/test_all_CLAW_scenes.py:86-96).
"""
import torch
import torch.distributed as dist


def shard_frames(N, rank, world):
    """Frame indices rendered by ``rank`` (round-robin: equal Euler depth mix per rank)."""
    return list(range(rank, N, world))


def frames_per_rank(N, world):
    return (N + world - 1) // world


def gather_clip(local_frames, N, rank, world, group=None, always_collective=False):
    """local_frames [len(shard_frames(N,rank,world)), ...] -> [N, ...] on every rank (any dtype: fp32 frames [n,3,H,W], or the
    uint8 [n,h,w,3] frames of frames_for_assembly).  ONE all_gather_into_tensor of shards of ceil(N/world) frames: a full shard is
    sent as it is (no staging copy); a short one (N not divisible by world) is padded with rows that land behind frame N - 1 and are
    cut off -- they are never read, so they are not initialised either.  always_collective: enter the collective at world size 1 too
    (the single-GPU pre-flight of the RCCL path, tests/test_gpu_parity.py::test_multi_gpu_preflight_on_rccl)."""
    if world == 1 and not always_collective:
        return local_frames
    per = frames_per_rank(N, world)
    n, tail = local_frames.shape[0], tuple(local_frames.shape[1:])
    if n == per:
        send = local_frames.contiguous()
    else:
        send = local_frames.new_empty((per,) + tail)
        send[:n] = local_frames
    recv = local_frames.new_empty((world * per,) + tail)
    dist.all_gather_into_tensor(recv, send, group=group)
    # recv[r*per + i] is frame r + i*world  ->  clip order (the one reordering copy of the assembly; 2.95 MB per frame as uint8)
    clip = recv.view((world, per) + tail).transpose(0, 1).reshape((per * world,) + tail)
    return clip[:N]


def frames_for_assembly(frames, raw_hw=None):
    """What a rank hands to the clip assembly when the clip's destination is image files (the reference's writer,
    test_animating/test_baseline_4eval_rawsize.py:246-274): its frames resized to the raw size, * 0.5 + 0.5, * 255, rounded, as uint8
    [n,h,w,3] -- done BY EVERY RANK on its own frames (io.frames_to_uint8), so the all-gather moves 1 byte per sample instead of 4
    (2.95 instead of 11.8 MB per 768x1280 frame) and rank 0 has nothing left to do but write."""
    from . import io
    return io.frames_to_uint8(frames, raw_hw).contiguous()


def communicator_report(device, frames, seconds, group=None):
    """What the communicator itself says about the job, gathered from every rank (all_gather_object): backend, world size, and per
    rank its device index, the GPU's name / PCI bus id / UUID, the frames it rendered and its own frames/s.  bench.py prints it in the
    N > 1 line, so a scaling run proves which GPUs and how many RCCL ranks took part."""
    world = dist.get_world_size(group)
    props = torch.cuda.get_device_properties(device) if device.type == "cuda" else None
    mine = {"rank": dist.get_rank(group), "device_index": device.index if device.type == "cuda" else None,
            "device_name": getattr(props, "name", None), "pci_bus_id": getattr(props, "pci_bus_id", None),
            "pci_domain_id": getattr(props, "pci_domain_id", None), "uuid": str(getattr(props, "uuid", None)) if props else None,
            "frames": int(frames), "fps_this_rank": round(frames / seconds, 3) if seconds > 0 else None}
    everybody = [None] * world
    dist.all_gather_object(everybody, mine, group=group)
    rccl = None
    if dist.get_backend(group) == "nccl":
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl = None
    return {"backend": dist.get_backend(group), "rccl_version": rccl, "world_size": world, "ranks": everybody,
            "distinct_devices": len({(r["pci_domain_id"], r["pci_bus_id"], r["uuid"], r["device_index"]) for r in everybody})}


class ClipAssembler:
    """Round-wise clip assembly: in round j every rank renders frame j*world + rank; as soon as a rank's frame is
    finished it enters an ASYNCHRONOUS all_gather_into_tensor whose output is the clip's slice [j*world, (j+1)*world)
    -- already in clip order, no transpose afterwards -- and the rank goes on rendering round j+1 while RCCL moves the
    11.8 MB per peer (768x1280 fp32) on its own stream.  Only the last round's collective is exposed.

        asm = ClipAssembler(N, rank, world)
        for frame in my_frames_in_order: asm.push(frame)        # [3,H,W], must stay untouched until finish()
        clip = asm.finish()                                      # [N,3,H,W] on every rank

    Every rank issues exactly frames_per_rank(N, world) collectives: a rank without a frame in the last round (N not
    divisible by world) contributes zeros that land behind frame N-1 and are cut off."""

    def __init__(self, N, rank, world, group=None, always_collective=False):
        self.N, self.rank, self.world, self.group = N, rank, world, group
        self.always_collective = always_collective       # tests: exercise the RCCL path with world == 1
        self.rounds = frames_per_rank(N, world)
        self.mine = len(shard_frames(N, rank, world))
        self.recv = None
        self.work = []
        self.keep = []                                   # inputs stay alive (and unmodified) until finish()
        self.j = 0

    def push(self, frame):
        assert self.j < self.mine, "more frames than this rank's shard"
        self._gather(frame)

    def _gather(self, frame):
        if self.recv is None:
            self.recv = frame.new_empty(self.rounds * self.world, *frame.shape)
        frame = frame.contiguous()
        out = self.recv[self.j * self.world:(self.j + 1) * self.world]
        if self.world == 1 and not self.always_collective:
            out[0].copy_(frame)
        else:
            self.keep.append(frame)
            # input [1,...] -> output [world,...]: concatenation along dim 0 (the form gloo accepts too)
            self.work.append(dist.all_gather_into_tensor(out, frame.unsqueeze(0), group=self.group, async_op=True))
        self.j += 1

    def finish(self, like=None):
        """Wait for the collectives (the caller's stream waits; the host does not block on RCCL) -> [N,...]."""
        assert self.j == self.mine, f"rank {self.rank}: {self.j} of {self.mine} frames pushed"
        if self.j < self.rounds:                         # the idle slot of the last round
            ref = self.keep[-1] if self.keep else like
            assert ref is not None, "a rank without frames needs `like` (a tensor of the frame's shape/device)"
            self._gather(torch.zeros_like(ref))
        for w in self.work:
            w.wait()
        self.work, self.keep = [], []
        return self.recv[:self.N]


# ----------------------------------------------------------------------------------------------
# The per-clip encoder, split by rows.  Every rank needs the frame-invariant features, and at 8 GPUs
# repeating the encoder on every rank is 4.8 of the ~56 ms a rank spends on its 8 frames.  The
# encoders on the path (ResNetEncoder / ResNetEncoder_with_Z, architectures.py:121-197) are 8
# residual blocks of 3x3 / 1x1 convolutions at full resolution with pointwise BN + ReLU in between:
# output row y depends on input rows y-16 .. y+16 only, so rank r runs the encoder on its band of
# ceil(H/world) rows plus a 16-row halo (rows beyond the image are the convolution's own zero padding,
# i.e. exact), cuts the halo off, and one all-gather (65 planes x H x W = 256 MB in total) gives every
# rank the full tensors -- bit-identical to the unsplit encoder (every output pixel sums its channels and
# taps in the same order wherever its tile lies; tests/test_gpu_parity.py::test_banded_encoder_is_exact).


def band_rows(H, rank, world):
    """Rows [y0, y1) of the image that ``rank`` encodes."""
    rows = -(-H // world)
    return min(rank * rows, H), min((rank + 1) * rows, H)


def encoder_halo(encoder):
    """Receptive-field radius in rows: two 3x3 convolutions per residual block."""
    return 2 * len(encoder.blocks)


def encode_band(encoder, image, rank, world, halo=None, guard=None):
    """This rank's band of the encoder outputs -> ([1,C,y1-y0,W] all outputs concatenated along C, [C_i]).
    guard: optional callable(thunk) -> thunk's result, wrapped around the network call (nets.guarded: what a rank does
    about its own band's activations stays local -- every rank still enters the same all-gather)."""
    H = image.shape[2]
    halo = encoder_halo(encoder) if halo is None else halo
    y0, y1 = band_rows(H, rank, world)
    assert y1 > y0, f"rank {rank} of {world} has no rows of a {H}-row image"
    a0, a1 = max(0, y0 - halo), min(H, y1 + halo)
    band_in = image[:, :, a0:a1].contiguous()
    outs = encoder(band_in) if guard is None else guard(lambda: encoder(band_in))
    outs = outs if isinstance(outs, tuple) else (outs,)
    band = torch.cat([o[:, :, y0 - a0:y1 - a0] for o in outs], 1)
    return band, [o.shape[1] for o in outs]


def encode_banded(encoder, image, rank, world, group=None, halo=None, guard=None, always_collective=False):
    """encoder(image) computed in row bands across the ranks + one all-gather; same return type as encoder(image).
    always_collective: band + collective at world size 1 too (pre-flight of the RCCL path on one GPU)."""
    if world == 1 and not always_collective:
        return encoder(image) if guard is None else guard(lambda: encoder(image))
    H, W = image.shape[2:]
    rows = -(-H // world)
    if (world - 1) * rows >= H:          # checked on EVERY rank before any work: a rank without rows must not leave the
        raise ValueError(               # others waiting in the collective
            f"encode_banded: {H} rows cannot be split into {world} non-empty bands of {rows} rows; use fewer ranks "
            f"(or the redundant encoder)")
    band, split = encode_band(encoder, image, rank, world, halo, guard)
    C = band.shape[1]
    send = band.new_zeros(1, C, rows, W)
    send[0, :, :band.shape[2]] = band[0]
    recv = band.new_empty(world, C, rows, W)
    dist.all_gather_into_tensor(recv, send, group=group)
    full = recv.permute(1, 0, 2, 3).reshape(1, C, world * rows, W)[:, :, :H]
    outs, c0 = [], 0
    for c in split:
        outs.append(full[:, c0:c0 + c].contiguous())
        c0 += c
    return outs[0] if len(outs) == 1 else tuple(outs)
