"""The networks and the FRAMES at 256 x 256 against digests of the reference's own classes / models
(tests/golden/large_nets_e2e.npz, tools/make_golden_large.py): the grid on which the convolution kernels run their
multi-tile, channel-blocked 128-channel variants (conv3x3_split_kernel<1,4,*,true>: two thirds of a timed clip) -- the
small fixtures (16x24 / 32x48 nets, 64x64 frames) never reach those.  Per tensor: 4096 seeded positions, per-plane sums
and the max-abs of the reference's output.
CPU (not gpu): the package's torch definition of the decoder (pins the definition the kernels are tested against);
GPU (-m gpu): the HIP kernels / the animators, within 5e-5 of the range (nets) and 1e-4 max-abs (frames: north_star)."""
import numpy as np
import pytest
import torch

import nets_fixture as NF
from test_e2e_golden import _baseline, _net, _v1


def _check(g, tag, x, tol_rel=None, tol_abs=None):
    x = np.ascontiguousarray(x.detach().cpu().numpy() if torch.is_tensor(x) else x, dtype=np.float32)
    assert list(x.shape) == [int(v) for v in g[f"{tag}_shape"]], tag
    pos = NF.digest_positions(tag, x.size, int(g["npos"]))
    scale = float(g[f"{tag}_absmax"])
    tol = tol_abs if tol_abs is not None else tol_rel * scale
    err = float(np.abs(x.ravel()[pos] - g[f"{tag}_val"]).max())
    assert err <= tol, (tag, err, tol)
    sums = x.reshape(-1, x.shape[-2] * x.shape[-1]).astype(np.float64).sum(1)
    hw = x.shape[-2] * x.shape[-1]
    assert float(np.abs(sums - g[f"{tag}_plane_sums"]).max()) <= tol * hw * 0.05, tag      # (mean error per pixel << tol)
    return err


def _mine(name):
    from slr_sfs_amd import nets
    return {"encoder": nets.EncoderWithZ, "projector": lambda: nets.DecoderPconv2(64, 3),
            "net_alpha_decoder": lambda: nets.DecoderPconv2(65, 1)}[name]()


def test_torch_definition_of_the_decoder_at_256(golden_dir):
    from slr_sfs_amd import nets
    g = np.load(f"{golden_dir}/large_nets_e2e.npz")
    net = _net(golden_dir, "projector", _mine("projector"))
    with nets.cpu_reference(), torch.no_grad():
        out = net(NF.net_input_large("projector", int(g["S"])))
    _check(g, "projector_out0", out, tol_rel=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["encoder", "projector", "net_alpha_decoder"])
def test_hip_networks_at_256_vs_reference_digests(golden_dir, name):
    import slr_sfs_amd
    slr_sfs_amd._lib.lib()
    g = np.load(f"{golden_dir}/large_nets_e2e.npz")
    net = _net(golden_dir, name, _mine(name)).cuda()
    with torch.no_grad():
        out = net(NF.net_input_large(name, int(g["S"])).cuda())
    out = out if isinstance(out, tuple) else (out,)
    assert len(out) == int(g[f"{name}_nout"])
    for i, o in enumerate(out):
        _check(g, f"{name}_out{i}", o, tol_rel=5e-5)


@pytest.mark.gpu
def test_frames_at_256_vs_reference_models(golden_dir):
    """Frames of both animators at 256 x 256 (HIP kernels throughout) within 1e-4 max-abs of the frames the reference's own
    models produce from the same seeded weights, image and motion."""
    g = np.load(f"{golden_dir}/large_nets_e2e.npz")
    S, N = int(g["S"]), int(g["N"])
    img, motion, _ = NF.e2e_inputs(S, N)
    img, motion = torch.from_numpy(img).cuda(), torch.from_numpy(motion).cuda()
    ts = [int(t) for t in g["ts"]]
    frames = _baseline(golden_dir).cuda().synthesize(img, motion, N, frames=ts)
    for k, t in enumerate(ts):
        _check(g, f"baseline_PredImg_t{t}", frames[k:k + 1], tol_abs=1e-4)
    t = N // 2
    outs = _v1(golden_dir).cuda().synthesize(img, motion, N, frames=[t], keys=("PredImg", "FluidImg", "CompositeFluidAlpha"))
    for k in ("PredImg", "FluidImg", "CompositeFluidAlpha"):
        _check(g, f"v1_{k}_t{t}", outs[k], tol_abs=1e-4)


def _check_envelope(g, tag, x, tol_abs):
    """Distance of x to the interval spanned by the reference's two own runs (plain and oneDNN CPU convolutions) at the sampled positions."""
    x = np.ascontiguousarray(x.detach().cpu().numpy() if torch.is_tensor(x) else x, dtype=np.float32)
    assert list(x.shape) == [int(v) for v in g[f"{tag}_shape"]], tag
    pos = NF.digest_positions(tag, x.size, int(g["npos"]))
    a, b = g[f"{tag}_val"], g[f"{tag}_val_onednn"]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    v = x.ravel()[pos]
    dist = np.maximum(np.maximum(lo - v, v - hi), 0.0)
    err = float(dist.max())
    assert err <= tol_abs, (tag, err, tol_abs, float(np.abs(a - b).max()))
    sums = x.reshape(-1, x.shape[-2] * x.shape[-1]).astype(np.float64).sum(1)
    hw = x.shape[-2] * x.shape[-1]
    assert float(np.abs(sums - g[f"{tag}_plane_sums"]).max()) <= tol_abs * hw * 0.05, tag          # (mean error per pixel << tol)
    return err, float(min(np.abs(v - a).max(), np.abs(v - b).max()))


def test_native_fixture_holds_both_reference_runs(golden_dir):
    """The fixture itself: both runs of the reference present for every tensor, and the spread between them is what the header of
    tools/make_golden_large.py says (max 2.0e-4 on frame 30 of the baseline model; everything else below 1e-4)."""
    g = np.load(f"{golden_dir}/native_frames_768.npz")
    tags = [k[:-4] for k in g.files if k.endswith("_val")]
    assert len(tags) == 12
    spread = {t: float(np.abs(g[t + "_val"] - g[t + "_val_onednn"]).max()) for t in tags}
    assert 1.5e-4 < spread["baseline_PredImg_t30"] < 2.5e-4, spread
    assert all(v < 1e-4 for t, v in spread.items() if t != "baseline_PredImg_t30"), spread


@pytest.mark.gpu
def test_frames_at_native_768_vs_reference_models(golden_dir):
    """The reference's own working size (test_animating/CLAW/test_v1.sh:19: W = 768, N = 60; test_v1_4eval_rawsize.py:233-239):
    frames t = 1, 30, 59 of both animators (HIP kernels throughout -- the multi-tile 128 / 256-channel convolution variants that make
    up two thirds of a timed clip, the fused splat kernel on 96 x 12 tiles) against the frames the reference's own models produce on
    the CPU from the same seeded weights, image and motion (tools/make_golden_large.py --native).  The reference's fp32 frames are
    themselves only reproducible to 2.0e-4 at this size (its oneDNN and its plain CPU convolutions, both stored); every sampled value
    has to lie within 1e-4 of the interval the two reference runs span."""
    g = np.load(f"{golden_dir}/native_frames_768.npz")
    S, N = int(g["S"]), int(g["N"])
    assert (S, N) == (768, 60)
    img, motion, _ = NF.e2e_inputs(S, N)
    img, motion = torch.from_numpy(img).cuda(), torch.from_numpy(motion).cuda()
    ts = [int(t) for t in g["ts"]]
    frames = _baseline(golden_dir).cuda().synthesize(img, motion, N, frames=ts)
    errs = [_check_envelope(g, f"baseline_PredImg_t{t}", frames[k:k + 1], 1e-4) for k, t in enumerate(ts)]
    v1_ts = [int(t) for t in g["v1_ts"]]
    keys = ("PredImg", "FluidImg", "CompositeFluidAlpha")
    outs = _v1(golden_dir).cuda().synthesize(img, motion, N, frames=v1_ts, keys=keys)
    for i, t in enumerate(v1_ts):
        for k in keys:
            errs.append(_check_envelope(g, f"v1_{k}_t{t}", outs[k][i:i + 1], 1e-4))
    print("native 768 frames: max distance to the reference's own interval", max(e[0] for e in errs),
          "| max distance to the nearer single run", max(e[1] for e in errs))


def test_native_fixture_holds_the_fp64_arbiter(golden_dir):
    """The single-valued arbiter: the reference's own classes run in float64 (tools/make_golden_large.py --native --fp64).
    What the fixture says about the reference itself: its two fp32 runs sit 3e-6 ... 1.9e-4 from the fp64 frames -- on frame 30 of
    the baseline model BOTH are further than 1e-4 from it (1.9e-4 plain, 1.2e-4 oneDNN): fp32 convolution rounding at this depth,
    not a property of any implementation."""
    g = np.load(f"{golden_dir}/native_frames_768.npz")
    tags = [k[:-4] for k in g.files if k.endswith("_val")]
    assert len(tags) == 12 and all(t + "_val_fp64" in g.files and t + "_plane_sums_fp64" in g.files for t in tags)
    d = {t: (float(np.abs(g[t + "_val"] - g[t + "_val_fp64"]).max()), float(np.abs(g[t + "_val_onednn"] - g[t + "_val_fp64"]).max())) for t in tags}
    assert 1.5e-4 < d["baseline_PredImg_t30"][0] < 2.5e-4 and 1.0e-4 < d["baseline_PredImg_t30"][1] < 1.5e-4, d
    assert all(max(v) < 1e-4 for t, v in d.items() if t != "baseline_PredImg_t30"), d


def _check_fp64(g, tag, x):
    """-> max distance of x to the reference's fp64 frame at the sampled positions (+ the mean error per pixel of every plane)."""
    x = np.ascontiguousarray(x.detach().cpu().numpy() if torch.is_tensor(x) else x, dtype=np.float32)
    assert list(x.shape) == [int(v) for v in g[f"{tag}_shape"]], tag
    pos = NF.digest_positions(tag, x.size, int(g["npos"]))
    err = float(np.abs(x.ravel()[pos].astype(np.float64) - g[f"{tag}_val_fp64"]).max())
    hw = x.shape[-2] * x.shape[-1]
    sums = x.reshape(-1, hw).astype(np.float64).sum(1)
    return err, float(np.abs(sums - g[f"{tag}_plane_sums_fp64"]).max()) / hw


@pytest.mark.gpu
@pytest.mark.parametrize("convs", ["auto", "fp32", "fp32-winograd"])
def test_frames_at_native_768_vs_the_fp64_reference(golden_dir, convs):
    """Frames t = 1, 30, 59 of both animators at the reference's native 768 x 768, N = 60 (HIP kernels throughout) against the frames of
    the reference's own classes run in FLOAT64 -- one value per sample, no envelope.  Both convolution rungs: the default
    (split-f16 matrix-core kernels, convs="auto") and the fp32 matrix-core rung.  Bound: 1e-4 (north_star) on every sampled value --
    measured on MI355X: 1.4e-5 at worst on either rung, i.e. closer to the fp64 frames than the reference's own fp32 CPU runs are
    (1.2e-4 / 1.9e-4 on frame 30 of the baseline model; test_native_fixture_holds_the_fp64_arbiter) -- and 4e-5 asserted so that a
    regression shows long before the contract is at risk.  The mean error per pixel of every plane stays below 5e-6.
    convs="fp32-winograd" (the fast fp32 rung) is NOT held to that bound and says so: its layers carry 2 - 4x the direct kernel's
    rounding error, and these seeded random-weight networks amplify a layer's perturbation ~400x at a few ill-conditioned pixels
    (where the reference's own two fp32 runs are 1.2e-4 / 1.9e-4 from the fp64 frames): measured 2.6e-4 at worst over the sampled
    values (v1 FluidImg t = 30), 6.1e-5 on the baseline model -- asserted <= 4e-4, with the same 5e-6 on the mean error."""
    g = np.load(f"{golden_dir}/native_frames_768.npz")
    S, N = int(g["S"]), int(g["N"])
    img, motion, _ = NF.e2e_inputs(S, N)
    img, motion = torch.from_numpy(img).cuda(), torch.from_numpy(motion).cuda()
    ts = [int(t) for t in g["ts"]]
    keys = ("PredImg", "FluidImg", "CompositeFluidAlpha")
    base, v1 = _baseline(golden_dir).cuda(), _v1(golden_dir).cuda()
    base.convs = v1.convs = convs
    frames = base.synthesize(img, motion, N, frames=ts)
    errs = {f"baseline_PredImg_t{t}": _check_fp64(g, f"baseline_PredImg_t{t}", frames[k:k + 1]) for k, t in enumerate(ts)}
    v1_ts = [int(t) for t in g["v1_ts"]]
    outs = v1.synthesize(img, motion, N, frames=v1_ts, keys=keys)
    for i, t in enumerate(v1_ts):
        for k in keys:
            errs[f"v1_{k}_t{t}"] = _check_fp64(g, f"v1_{k}_t{t}", outs[k][i:i + 1])
    print(f"native 768 frames vs the fp64 reference (convs={convs}):", {k: f"{v[0]:.2e}" for k, v in errs.items()})
    bound = 4e-4 if convs == "fp32-winograd" else 4e-5
    for tag, (err, mean_err) in errs.items():
        assert err <= bound, (tag, err)
        assert mean_err <= 5e-6, (tag, mean_err)
