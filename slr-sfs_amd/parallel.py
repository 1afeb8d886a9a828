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


def gather_clip(local_frames, N, rank, world, group=None):
    """local_frames [len(shard_frames(N,rank,world)),3,H,W] -> [N,3,H,W] on every rank.
    Shards are padded to ceil(N/world) frames so a single all_gather_into_tensor suffices."""
    if world == 1:
        return local_frames
    per = frames_per_rank(N, world)
    C, H, W = local_frames.shape[1:]
    send = local_frames.new_zeros(per, C, H, W)
    send[:local_frames.shape[0]] = local_frames
    recv = local_frames.new_empty(world * per, C, H, W)
    dist.all_gather_into_tensor(recv, send, group=group)
    # recv[r*per + i] is frame r + i*world  ->  clip order
    clip = recv.view(world, per, C, H, W).transpose(0, 1).reshape(per * world, C, H, W)
    return clip[:N]


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
