"""Frame synthesis between encoder and decoder -- the data-flow of the reference's
``forward_flow`` (models/animating_softmax_splating.py:777-981 and
models/animating_softmax_splating_2layers_alpha_seperate.py:843-1108), restructured for MI355X:

  * Euler integration: ONE all-frames pass per direction per clip (the reference re-integrates
    from scratch for every frame: t + (N-t) Python-loop steps per frame, N^2 per clip);
  * per frame: bin the two displacement maps once, then one fused kernel per weight group does
    exp-weighting, both splat directions, and the normalisation (the reference: 2 cats, ~25
    elementwise torch kernels over 65-67 planes, 2 atomic-scatter launches).

Index semantics: batch["index"] = [start, middle, end] = [0, t, N-1]:
forward steps = middle-start = t, backward steps = end-middle+1 = N-t,
alpha = 1 - (middle-start)/(end-start+1) = 1 - t/N.
"""
import ctypes
import os

import torch

from ._lib import check, lib, ptr, require_device, stream_of, workspace
from .euler_integration_manipulator import euler_integration_all


# bench.py sets this to a list to collect (start, stop) torch events around the tile kernel of
# every synth_group call with timed=True (through slr_splat_time_next); None = no timing.
kernel_timing = None
# bench.py sets this to a list to collect ("prep" | "prep+" | "frame", start, stop, frames) torch events around the WHOLE splat
# stage: "prep" = the per-clip motion work (Euler passes, binning, planning), "prep+" = the clip's feature planes packed by 4, "frame" = one features(t) call or one
# features_batch group (`frames` of them in one launch).
stage_timing = None


class _stage:
    def __init__(self, kind, device, frames=1):
        self.kind, self.device, self.frames = kind, device, frames

    def __enter__(self):
        if stage_timing is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream(self.device))

    def __exit__(self, *exc):
        if stage_timing is not None and exc[0] is None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream(self.device))
            stage_timing.append((self.kind, self.e0, e1, self.frames))
        return False


def _arm_timer(t, frames=1):
    """Ask the library to record events around the next tile-kernel launch; kernel_timing gets (start, stop, frames in
    that launch) -- a batched launch does the work of several frames."""
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    for e in ev:
        e.record(torch.cuda.current_stream(t.device))      # materialises the hipEvent_t handle
    lib().slr_splat_time_next(ev[0].cuda_event, ev[1].cuda_event)
    kernel_timing.append((ev[0], ev[1], frames))


def bin_flow(flow, C, role):
    """Sort the source pixels of ``flow`` [1,2,H,W] into output-tile bins (slr_splat_bin).
    Returns the workspace; valid until the next bin_flow with the same role/shape/stream."""
    require_device(flow)
    N, _, H, W = flow.shape
    ws = workspace(flow, role, N, C, H, W)
    with torch.cuda.device(flow.device):
        check(lib().slr_splat_bin(ptr(flow), N, H, W, ptr(ws), ws.numel(), stream_of(flow)), "slr_splat_bin")
    return ws


def bin_flow_pair(flow_a, flow_b, C):
    """bin_flow for the forward and the backward displacement map of a frame in one go."""
    require_device(flow_a, flow_b)
    assert flow_a.shape == flow_b.shape
    N, _, H, W = flow_a.shape
    ws_a, ws_b = workspace(flow_a, "f", N, C, H, W), workspace(flow_a, "p", N, C, H, W)
    with torch.cuda.device(flow_a.device):
        check(lib().slr_splat_bin_pair(ptr(flow_a), ptr(flow_b), N, H, W, ptr(ws_a), ptr(ws_b), ws_a.numel(),
                                       stream_of(flow_a)), "slr_splat_bin_pair")
    return ws_a, ws_b


def global_max(x):
    """x.max() as a 1-element device tensor, no host sync (animating_softmax_splating.py:855)."""
    require_device(x)
    res = x.new_empty(1)
    scratch = x.new_empty(1024)
    with torch.cuda.device(x.device):
        check(lib().slr_global_max(ptr(x), x.numel(), ptr(res), ptr(scratch), stream_of(x)), "slr_global_max")
    return res


def synth_group(values, wlogit, disp_f, disp_p, alpha, ws_f, ws_p, wmax=None, exp_weights=True,
                eps=1e-8, return_norm=False, timed=False):
    """out = [splat(values*w*alpha, disp_f) + splat(values*w*(1-alpha), disp_p)] / max(same for w, eps)
    with w = exp(wlogit - wmax) | exp(wlogit) | wlogit.  values [1,C,H,W], wlogit [1,1,H,W]."""
    require_device(values, wlogit, disp_f, disp_p, wmax)
    assert values.shape[0] == 1 and wlogit.shape[1] == 1 and disp_f.shape[1] == 2 and disp_p.shape[1] == 2
    _, C, H, W = values.shape
    out = torch.empty_like(values)
    norm = values.new_empty(1, 1, H, W) if return_norm else None
    with torch.cuda.device(values.device):
        if timed and kernel_timing is not None:
            _arm_timer(values)
        check(lib().slr_synth_group(ptr(values), ptr(wlogit), ptr(wmax), 1 if exp_weights else 0,
                                    ptr(disp_f), ptr(disp_p), float(alpha), ptr(out), ptr(norm),
                                    C, H, W, float(eps), ptr(ws_f), ptr(ws_p), ws_f.numel(),
                                    stream_of(values)), "slr_synth_group")
    return (out, norm) if return_norm else out


VALUES_B4 = 2         # include/slr_splat.h: SLR_SYNTH_VALUES_B4
USE_B4 = os.environ.get("SLR_SFS_AMD_VALUES_B4", "1") != "0"
PLAN_CHUNK = 64      # frames per clip plan (bounds the plan buffer: ~16 MB of bin lists per map at 768x1280, worst case)


class MotionPlan:
    """Everything of a clip that depends on the motion field only: the two all-frames Euler passes and, for the frames
    that will be rendered, the tile bins + work plans of both directions -- built by ONE set of launches per chunk of
    PLAN_CHUNK frames (slr_clip_plan_build) instead of six latency-bound launches per frame.  Independent of the image,
    so the animators build it BEFORE the encoder runs: the per-frame totals (work items, multi-segment tiles) are
    copied to the host asynchronously and have long arrived when the first frame is launched.

    frame t of an N-frame clip: t forward steps of +motion, N - t backward steps of -motion."""

    def __init__(self, motion, N, frames=None):
        require_device(motion)
        assert motion.shape[0] == 1 and motion.shape[1] == 2
        self.N = int(N)
        self.H, self.W = motion.shape[2:]
        self.frames = list(range(self.N)) if frames is None else [int(t) for t in frames]
        self._where = {}                                   # t -> (chunk record, index inside the chunk)
        with _stage("prep", motion.device):
            # all-frames Euler passes: forward t = 0..N-1 steps of +motion, backward 1..N steps of -motion
            self.disp_f, _ = euler_integration_all(motion, self.N - 1, +1.0, want_visible=False)
            self.disp_p, _ = euler_integration_all(motion, self.N, -1.0, want_visible=False)
            # frames per plan: list offsets are 32-bit (8 * frames * H * W < 2^32, include/slr_splat.h), so large grids
            # get shorter chunks instead of no plan at all
            chunk = max(1, min(PLAN_CHUNK, (2 ** 32 - 1) // (8 * self.H * self.W)))
            for c0 in range(0, len(self.frames), chunk):
                self._build(self.frames[c0:c0 + chunk])

    def _build(self, ts):
        n, dev = len(ts), self.disp_f.device
        L = lib()
        nbytes = int(L.slr_clip_plan_bytes(n, self.H, self.W))
        if nbytes == 0:
            raise RuntimeError(f"slr_sfs_amd: no clip plan for {n} frames of {self.H}x{self.W}")
        plan = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        idx_f = torch.tensor(ts, dtype=torch.int32, device=dev)
        idx_p = (self.N - idx_f).to(torch.int32)
        with torch.cuda.device(dev):
            check(L.slr_clip_plan_build(ptr(self.disp_f), ctypes.c_void_p(idx_f.data_ptr()), ptr(self.disp_p),
                                        ctypes.c_void_p(idx_p.data_ptr()), n, self.H, self.W, ptr(plan), nbytes,
                                        stream_of(plan)), "slr_clip_plan_build")
        off, stride = ctypes.c_size_t(), ctypes.c_int()
        check(L.slr_clip_plan_totals(n, self.H, self.W, ctypes.byref(off), ctypes.byref(stride)), "slr_clip_plan_totals")
        totals = plan[off.value:off.value + n * stride.value * 4].view(torch.int32).view(n, stride.value)
        host = torch.empty(totals.shape, dtype=torch.int32, pin_memory=True)
        host.copy_(totals, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        rec = {"plan": plan, "n": n, "host": host, "event": ev, "keep": (idx_f, idx_p)}
        for i, t in enumerate(ts):
            self._where[t] = (rec, i)

    def chunk_of(self, t):
        """The plan chunk frame t lives in (built on demand for an unannounced frame)."""
        if t not in self._where:
            self._build([t])
        return self._where[t][0]

    def lookup(self, t):
        """-> (plan buffer, frames in it, index of t, work items of t's plan)."""
        if t not in self._where:
            self._build([t])                               # a frame outside the announced set: its own one-frame plan
        rec, i = self._where[t]
        if rec["event"] is not None:
            rec["event"].synchronize()                     # long done unless this is the clip's very first launch
            rec["event"] = None
        row = rec["host"][i]
        return rec["plan"], rec["n"], i, int(row[0])


def synth_group_clip(values, wlogit, mp, t, alpha, wmax=None, exp_weights=True, eps=1e-8, return_norm=False, timed=False,
                     out=None, values_b4=None):
    """synth_group for frame t of a MotionPlan (bins and work plan prepared per clip).
    out: optional [1,C,H,W] destination (e.g. one sample of a batch buffer the decoder will read)."""
    require_device(values, wlogit, wmax)
    assert values.shape[0] == 1 and wlogit.shape[1] == 1
    _, C, H, W = values.shape
    plan, n, i, n_items = mp.lookup(t)
    disp_f, disp_p = mp.disp_f[t], mp.disp_p[mp.N - t]
    if out is None:
        out = torch.empty_like(values)
    else:
        require_device(out)
        assert out.shape == values.shape and out.device == values.device
    norm = values.new_empty(1, 1, H, W) if return_norm else None
    with torch.cuda.device(values.device):
        if timed and kernel_timing is not None:
            _arm_timer(values)
        flags = (1 if exp_weights else 0) | (VALUES_B4 if values_b4 is not None else 0)
        check(lib().slr_synth_group_clip(ptr(values if values_b4 is None else values_b4), ptr(wlogit), ptr(wmax), flags,
                                         ptr(disp_f), ptr(disp_p), float(alpha), ptr(out), ptr(norm), C, H, W,
                                         float(eps), ptr(plan), plan.numel(), n, i, n_items, stream_of(values)), "slr_synth_group_clip")
    return (out, norm) if return_norm else out


MAX_BATCH = min(16, max(1, int(os.environ.get("SLR_SFS_AMD_SPLAT_BATCH", "16"))))   # frames per launch of slr_synth_group_clip_batch (csrc: SLR_CLIP_MAXB = 16; 8 / 12 / 16: 158 / 153 / 151-153 us per frame of work)




def pack_planes4(values):
    """[N,C,H,W] -> the same planes blocked by 4 in memory ([N,C/4,H,W,4], carried in a tensor of the logical shape [N,C,H,W]): what the
    clip kernels read with one 16-byte load per chunk and source pixel (slr_pack_planes4; once per clip)."""
    require_device(values)
    assert values.is_contiguous() and values.shape[1] % 4 == 0
    out = torch.empty_like(values)
    N, C, H, W = values.shape
    with torch.cuda.device(values.device):
        check(lib().slr_pack_planes4(ptr(values), ptr(out), N, C, H, W, stream_of(values)), "slr_pack_planes4")
    return out


def synth_group_clip_batch(values, wlogit, mp, ts, alphas, outs, wmax=None, exp_weights=True, eps=1e-8, timed=False,
                           group2=None, values_b4=None):
    """synth_group_clip for up to MAX_BATCH frames `ts` of ONE chunk of a MotionPlan in one launch of the tile kernel:
    outs[k] ([1,C,H,W], e.g. the samples of a decoder batch) receives frame ts[k].
    group2 = (values2 [1,1,H,W], wlogit2 [1,1,H,W], outs2): a second weight group (exp weights) splatted by the same launch
    with the same records -- the 2-layer model's alpha plane (slr_synth_two_groups_clip_batch).
    values_b4: pack_planes4(values) -- the kernel then reads that copy (same results, a quarter of the plane loads)."""
    require_device(values, wlogit, wmax, *outs)
    assert values.shape[0] == 1 and wlogit.shape[1] == 1 and 1 <= len(ts) <= MAX_BATCH and len(outs) == len(ts)
    _, C, H, W = values.shape
    look = [mp.lookup(t) for t in ts]
    plan, n = look[0][0], look[0][1]
    assert all(lk[0] is plan for lk in look), "frames of one launch must come from one chunk of the plan"
    nb = len(ts)
    L = lib()
    PP = ctypes.c_void_p * nb
    df = PP(*[mp.disp_f[t].data_ptr() for t in ts])
    dp = PP(*[mp.disp_p[mp.N - t].data_ptr() for t in ts])
    op = PP(*[o.data_ptr() for o in outs])
    al = (ctypes.c_float * nb)(*[float(a) for a in alphas])
    fr = (ctypes.c_int * nb)(*[lk[2] for lk in look])
    n_items = (ctypes.c_int * nb)(*[lk[3] for lk in look])
    for o in outs:
        assert o.shape == values.shape and o.device == values.device
    flags = (1 if exp_weights else 0) | (VALUES_B4 if values_b4 is not None else 0)
    vsrc = values if values_b4 is None else values_b4
    assert vsrc.shape == values.shape and vsrc.device == values.device
    with torch.cuda.device(values.device):
        if timed and kernel_timing is not None:
            _arm_timer(values, nb)
        if group2 is None:
            check(L.slr_synth_group_clip_batch(ptr(vsrc), ptr(wlogit), ptr(wmax), flags, df, dp, al, op,
                                               None, C, H, W, float(eps), ptr(plan), plan.numel(), n, fr, nb, n_items,
                                               stream_of(values)), "slr_synth_group_clip_batch")
        else:
            v2, w2, outs2 = group2
            require_device(v2, w2, *outs2)
            assert v2.shape == (1, 1, H, W) and w2.shape == (1, 1, H, W) and len(outs2) == nb
            assert all(o.shape == (1, 1, H, W) and o.is_contiguous() for o in outs2)
            op2 = PP(*[o.data_ptr() for o in outs2])
            check(L.slr_synth_two_groups_clip_batch(ptr(vsrc), ptr(wlogit), ptr(wmax), flags, ptr(v2), ptr(w2), 1,
                                                    df, dp, al, op, op2, C, H, W, float(eps), ptr(plan), plan.numel(), n, fr, nb,
                                                    n_items, stream_of(values)),
                  "slr_synth_two_groups_clip_batch")
    return outs


class ClipSynthesizer:
    """Frame-invariant state of one clip + per-frame decoder-input synthesis.

    baseline:  ClipSynthesizer(fs, Z, motion, N).features(t)                     -> gen_fs
    SLR v1:    ClipSynthesizer(fs, Z, motion, N, alpha_fluid_logit=..., alpha_bg=...,
                               use_alpha0=True, clamp_alpha=True).features(t)   -> gen_fs, alpha_fluid
    ``plan``: a MotionPlan built earlier for the same motion / N (the animators build it before the encoder);
    ``frames``: the frames that will be asked for (default all; a rank of a sharded job passes its share).
    ``blocked_planes``: keep a second copy of the feature planes blocked by 4 ([C/4][H][W][4]) for the clip kernels -- one 16-byte load
    per chunk and source pixel, 150 -> 134 us per frame -- at the cost of the planes' size once more for the life of the clip (+251 MB
    at 64 x 768 x 1280; the planar tensor stays: the single-frame / normaliser paths and callers read it).  None: on, unless the
    environment says SLR_SFS_AMD_VALUES_B4=0; False for many live clips on a tight memory budget.
    """

    def __init__(self, fs, Z, motion, N, alpha_fluid_logit=None, alpha_bg=None, use_alpha0=True,
                 clamp_alpha=None, softmax_v1=False, softmax_v2=False, clamp_z=None, plan=None, frames=None, blocked_planes=None):
        require_device(fs, Z, motion)
        assert fs.shape[0] == 1 and Z.shape[1] == 1 and motion.shape[1] == 2
        self.N = int(N)
        self.fs = fs.contiguous()
        Z = Z.contiguous()
        self.v1 = alpha_fluid_logit is not None
        # alpha clamp: only the 2-layer model has it (..._2layers_alpha_seperate.py:952)
        self.clamp_alpha = self.v1 if clamp_alpha is None else clamp_alpha
        # Z_f_norm = Z - Z.max() unless use_softmax_splatter_v1 / _v2 (animating_softmax_splating.py:849-855)
        self.softmax_v2 = bool(softmax_v2)       # per-frame shift by the maximum-warp-norm splat of Z (:849-851)
        self.clamp_z = clamp_z
        if self.softmax_v2:
            self.Z, self.zmax = Z, None
        elif clamp_z is not None:                                        # :856-859 (checkpoints without no_clamp_Z)
            Zn = Z if softmax_v1 else Z - Z.max()
            self.Z, self.zmax = torch.clamp(Zn, min=clamp_z[0], max=clamp_z[1]).contiguous(), None
        else:
            self.Z, self.zmax = Z, (None if softmax_v1 else global_max(Z))
        if plan is None:
            plan = MotionPlan(motion, self.N, frames)
        assert plan.N == self.N and (plan.H, plan.W) == tuple(fs.shape[2:])
        self.plan = plan
        self.disp_f, self.disp_p = plan.disp_f, plan.disp_p
        self.C = self.fs.shape[1]
        if self.v1:
            af = alpha_fluid_logit.contiguous()
            self.use_alpha0 = bool(use_alpha0)
            if self.use_alpha0:                                        # ..._2layers_alpha_seperate.py:963-965
                sg = torch.sigmoid(af)
                self.A0 = (sg / torch.clamp(sg + alpha_bg, min=1e-8)).contiguous()
                self.af = af
            else:                                                      # :974-976: same weights as the features
                self.fs = torch.cat([self.fs, af], 1).contiguous()
                self.C += 1
        # the feature planes once more, blocked by 4 in memory: the clip kernels read a chunk's 4 planes of a source pixel with ONE 16-byte
        # load (251 MB and 0.1 ms per clip at 768x1280; env SLR_SFS_AMD_VALUES_B4=0: the planar tensor as in round 4)
        self.fs4 = None
        if (USE_B4 if blocked_planes is None else blocked_planes) and self.C % 4 == 0 and self.fs.numel() * 4 < 2 ** 31:
            with _stage("prep+", self.fs.device):           # (more per-clip work of the splat stage: counted in bench.py's stage_us)
                self.fs4 = pack_planes4(self.fs)

    def alpha(self, t):
        a = torch.tensor(1.0, dtype=torch.float32) - torch.tensor(float(t), dtype=torch.float32) / \
            torch.tensor(float(self.N), dtype=torch.float32)          # fp32 arithmetic, as the reference (:860)
        if self.clamp_alpha:
            a = torch.clamp(a, min=1.0 / 600.0, max=599.0 / 600.0)
        return float(a)

    def features(self, t, return_norm=False, out=None, out_alpha=None):
        """Decoder input for frame t (index = [0, t, N-1]).  out / out_alpha: optional [1,C,H,W] / [1,1,H,W]
        destinations (samples of the batch buffers the decoders read; not for the 2-layer model without alpha0,
        whose alpha plane is a slice of the feature splat)."""
        t = int(t)
        assert 0 <= t < self.N
        assert (out is None and out_alpha is None) or not (self.v1 and not self.use_alpha0)
        with _stage("frame", self.fs.device):
            return self._features(t, return_norm, out, out_alpha)

    def features_batch(self, ts, out, out_alpha=None):
        """Decoder inputs of the frames `ts` into the samples of the batch tensors out [b,C,H,W] (and out_alpha [b,1,H,W]
        for the 2-layer model): the frames of one plan chunk go through ONE launch of the tile kernel per weight group
        (up to MAX_BATCH at a time).  Falls back to frame-by-frame calls where a frame needs its own weights
        (use_softmax_splatter_v2) or the alpha plane rides in the feature splat (no alpha0)."""
        ts = [int(t) for t in ts]
        assert out.shape[0] == len(ts)
        if self.softmax_v2 or (self.v1 and not self.use_alpha0):
            assert not (self.v1 and not self.use_alpha0), "no batch buffers for the 2-layer model without alpha0"
            for k, t in enumerate(ts):
                self.features(t, out=out[k:k + 1], out_alpha=None if out_alpha is None else out_alpha[k:k + 1])
            return
        k0 = 0
        while k0 < len(ts):
            chunk = self.plan.chunk_of(ts[k0])
            k1 = k0 + 1
            # (a launch renders each of its frames once -- slr_synth_group_clip_batch rejects a repeated frame index -- so a frame list
            #  with repeats, e.g. a ping-pong loop [.., N-2, N-1, N-1, N-2, ..], closes the group at the repeat: the reference's frame
            #  loop takes any index list, test_*_4eval_rawsize.py:234-245)
            while k1 < len(ts) and k1 - k0 < MAX_BATCH and self.plan.chunk_of(ts[k1]) is chunk and ts[k1] not in ts[k0:k1]:
                k1 += 1
            grp = ts[k0:k1]
            al = [self.alpha(t) for t in grp]
            with _stage("frame", self.fs.device, frames=len(grp)):
                # (2-layer model: the alpha plane, weighted by alpha0, rides in the same launch as a second weight group)
                synth_group_clip_batch(self.fs, self.Z, self.plan, grp, al, [out[k:k + 1] for k in range(k0, k1)],
                                       wmax=self.zmax, timed=True, values_b4=self.fs4,
                                       group2=(self.af, self.A0, [out_alpha[k:k + 1] for k in range(k0, k1)]) if self.v1 else None)
            k0 = k1

    def _features(self, t, return_norm, out=None, out_alpha=None):
        a = self.alpha(t)
        Zt = self.Z
        if self.softmax_v2:                      # Z_f_max = maximum_warp_norm_splater(Z_f, forward_flow)  (:849-851)
            from .softsplat import _FunctionMaximumWarpNormsplat
            Zt = self.Z - _FunctionMaximumWarpNormsplat(self.Z, self.disp_f[t:t + 1].contiguous())
            if self.clamp_z is not None:
                Zt = torch.clamp(Zt, min=self.clamp_z[0], max=self.clamp_z[1])
        if self.v1 and self.use_alpha0 and not return_norm:      # both weight groups in one launch
            gen = out if out is not None else torch.empty_like(self.fs)
            afl = out_alpha if out_alpha is not None else self.fs.new_empty(1, 1, *self.fs.shape[2:])
            synth_group_clip_batch(self.fs, Zt, self.plan, [t], [a], [gen], wmax=self.zmax, timed=True, group2=(self.af, self.A0, [afl]),
                                   values_b4=self.fs4)
            return gen, afl
        res = synth_group_clip(self.fs, Zt, self.plan, t, a, wmax=self.zmax, return_norm=return_norm, timed=True, out=out, values_b4=self.fs4)
        gen, norm = res if return_norm else (res, None)
        if not self.v1:
            return (gen, norm) if return_norm else gen
        if self.use_alpha0:
            afl = synth_group_clip(self.af, self.A0, self.plan, t, a, wmax=None, exp_weights=True, out=out_alpha)
        else:
            gen, afl = gen[:, :-1], gen[:, -1:]
        return (gen, afl, norm) if return_norm else (gen, afl)
