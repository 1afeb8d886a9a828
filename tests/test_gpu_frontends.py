"""The front ends of the one-flow operator (include/slr_splat.h: slr_splat_set_front_end) against the oracle:
`scan` (source-tile destination boxes, the tile kernel lists and walks the candidate rows itself), `rows` (row segments binned per
tile + the plan in one launch, the tile kernel walks the listed rows) and `prebinned` (slr_splat_bin once, then the call with
SLR_WS_PREBINNED: the rows machinery on a shared binning).  Every case runs with the front end FORCED, so all are covered at every
size -- including
BASELINE.json's config C2 as it is stated (FunctionSoftsplat(..., 'softmax'), 64 channels, 256x480, incoherent U(-8,8)
and smooth flow; models/softsplat.py:665-690)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
INT_MAX = 2 ** 31 - 1


@pytest.fixture(scope="module")
def S():
    import slr_sfs_amd
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    slr_sfs_amd._lib.lib()
    return slr_sfs_amd


FRONT_ENDS = {"auto": -1, "scan": 1, "rows": 2}


@pytest.fixture(params=list(FRONT_ENDS))
def frontend(request, S):
    L = S._lib.lib()
    prev = L.slr_splat_set_front_end(FRONT_ENDS[request.param])
    yield request.param
    L.slr_splat_set_front_end(prev)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def smooth_motion(H, W, seed=0, amp=1.5):
    rng = np.random.default_rng(seed)
    p1, p2 = rng.uniform(0, 2 * np.pi, 2)
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    u = amp * np.sin(2 * np.pi * (2 * x / W + y / H) + p1)
    v = amp * np.cos(2 * np.pi * (x / W - 1.5 * y / H) + p2)
    m = (x >= 0.35 * W).astype(np.float32)
    return np.stack([u * m, v * m])[None].astype(np.float32)


def test_default_threshold(S):
    L = S._lib.lib()
    prev = L.slr_splat_set_scan_max_tiles(7)
    assert L.slr_splat_set_scan_max_tiles(prev) == 7
    assert prev == 1024
    prev = L.slr_splat_set_front_end(2)                      # explicit choice: returns the previous one (-1 = by grid size)
    assert prev == -1 and L.slr_splat_set_front_end(17) == 2 and L.slr_splat_set_front_end(-1) == -1


def test_rows_pieces_and_the_pass_by_pass_launch(S, oracle):
    """The rows front end cuts a heavy tile into column ranges by an ESTIMATED histogram; a piece that still holds more than a
    segment (1024 entries) is handed to a second launch that walks it in passes of 2048.  Three flows that force every
    path: everything into one 8-column strip (one octant over 2048 entries: several passes), into one pixel, and a
    moderate squeeze (pieces of 2 - 4 octants, nothing deferred); every mode."""
    L = S._lib.lib()
    H, W, C = 64, 256, 9
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, C, H, W)).astype(np.float32)
    met = rng.standard_normal((2, 1, H, W)).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    strip = np.stack([(100.3 + (xx % 7)) - xx, (20.6 + (yy % 8)) - yy])                  # 16384 sources -> 7 x 8 pixels
    point = np.stack([130.5 - xx, 33.25 - yy])
    squeeze = np.stack([(xx - 128) * -0.6, (yy - 32) * -0.3])
    prev = L.slr_splat_set_front_end(2)
    try:
        for name, fl in (("strip", strip), ("point", point), ("squeeze", squeeze)):
            flow = np.stack([fl, fl[:, ::-1, ::-1].copy()]).astype(np.float32)          # (sample 1: mirrored)
            for mode, m in (("summation", None), ("softmax", met), ("average", None), ("linear", np.abs(met) + 0.1)):
                out = host(S.FunctionSoftsplat(dev(x), dev(flow), None if m is None else dev(m), mode))
                ref = oracle.function_softsplat(x, flow, m, mode)
                scale = max(1.0, float(np.abs(ref).max()))
                assert float(np.abs(out - ref).max()) < 2e-4 * scale, (name, mode, float(np.abs(out - ref).max()), scale)
                assert np.array_equal((out == 0).all(axis=1), (ref == 0).all(axis=1)), (name, mode)
    finally:
        L.slr_splat_set_front_end(prev)


@pytest.mark.parametrize("flowkind", ["incoherent", "smooth_t30", "smooth_t59"])
def test_config_c2_literal(S, oracle, frontend, flowkind):
    """BASELINE.json configs[1]: random 64-channel 256x480 feature + flow, FunctionSoftsplat softmax."""
    H, W, C = 256, 480, 64
    rng = np.random.default_rng(2)
    x = rng.standard_normal((1, C, H, W)).astype(np.float32)
    met = rng.standard_normal((1, 1, H, W)).astype(np.float32)
    if flowkind == "incoherent":
        flow = rng.uniform(-8, 8, (1, 2, H, W)).astype(np.float32)
    else:
        flow = oracle.euler_integration(smooth_motion(H, W), int(flowkind[-2:]))[0]
    out = host(S.FunctionSoftsplat(dev(x), dev(flow), dev(met), "softmax"))
    ref = oracle.function_softsplat(x, flow, met, "softmax")
    err = float(np.abs(out - ref).max())
    assert err < 1e-4, err                                  # north_star's bound; measured ~1e-6
    assert np.array_equal((out == 0).all(axis=1), (ref == 0).all(axis=1))        # same holes


@pytest.mark.parametrize("shape", [(1, 65, 256, 480), (2, 7, 45, 131), (1, 16, 100, 64), (3, 1, 17, 70), (1, 5, 300, 700)])
def test_sum_vs_oracle_piled_up_euler_flow(S, oracle, frontend, shape):
    """Strong Euler-integrated flow: tiles with several times SEG entries (rows: column pieces + the pass-by-pass launch; scan: deferred column pieces)."""
    N, C, H, W = shape
    rng = np.random.default_rng(C)
    flow = np.concatenate([oracle.euler_integration(smooth_motion(H, W, n, amp=3.0), 40 + n)[0] for n in range(N)])
    x = rng.standard_normal(shape).astype(np.float32)
    out = host(S.FunctionSoftsplat(dev(x), dev(flow), None, "summation"))
    ref = oracle.softsplat_forward(x, flow)
    bound = 4e-6 * oracle.softsplat_forward(np.abs(x), flow) + 1e-6
    assert (np.abs(out - ref) <= bound).all(), float((np.abs(out - ref) - bound).max())


@pytest.mark.parametrize("kind", ["row", "column", "shrink", "point"])
def test_collapsing_flows(S, oracle, frontend, kind):
    H, W = 200, 328
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    z = np.zeros_like(x)
    flow = {"row": np.stack([z, (H / 2 - y) * 0.999 - 0.2]),
            "column": np.stack([(W / 2 - x) * 0.999 + 0.3, z]),
            "shrink": np.stack([(W / 2 - x) * 0.75, (H / 2 - y) * 0.75]),
            "point": np.stack([(W / 2 - x) * 0.97 + 0.3, (H / 2 - y) * 0.97 - 0.2])}[kind][None].astype(np.float32)
    rng = np.random.default_rng(7)
    v = rng.standard_normal((1, 10, H, W)).astype(np.float32)
    met = (rng.standard_normal((1, 1, H, W)) * 0.5).astype(np.float32)
    ref = oracle.softsplat_forward(v, flow)
    out = host(S.FunctionSoftsplat(dev(v), dev(flow), None, "summation"))
    bound = 4e-6 * oracle.softsplat_forward(np.abs(v), flow) + 1e-6
    assert (np.abs(out - ref) <= bound).all(), float((np.abs(out - ref) - bound).max())
    refn = oracle.function_softsplat(v, flow, met, "softmax")
    outn = host(S.FunctionSoftsplat(dev(v), dev(flow), dev(met), "softmax"))
    np.testing.assert_allclose(outn, refn, rtol=1e-3, atol=1e-4)
    mx = host(S.ModuleMaximumsplat()(dev(v), dev(flow)))
    assert np.array_equal(mx, oracle.maxsplat_forward(v, flow))                  # a maximum has no summation order


def test_nonfinite_and_far_flows(S, oracle, frontend):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 4, 40, 72)).astype(np.float32)
    flow = rng.uniform(-2, 2, (2, 2, 40, 72)).astype(np.float32)
    flow[0, 0, 3, 5] = np.nan
    flow[0, 1, 7, 9] = np.inf
    flow[1, 0, 8, 1] = 3e9
    flow[1, :, 10:20, 30:40] = 1e6               # a whole block that leaves the image
    flow[0, :, 0:8, 0:64] = -1e4                 # a whole source tile that leaves the image: an empty box
    out = host(S.FunctionSoftsplat(dev(x), dev(flow), None, "summation"))
    np.testing.assert_allclose(out, oracle.softsplat_forward(x, flow), rtol=1e-5, atol=1e-5)
    x[0, 1, 20, 63] = np.inf                     # lands across a tile edge: must pollute only its own corners
    flow[0, :, 20, 63] = (0.5, 0.5)
    out = host(S.FunctionSoftsplat(dev(x), dev(flow), None, "summation"))
    ref = oracle.softsplat_forward(x, flow)
    assert np.array_equal(np.isfinite(out), np.isfinite(ref))


def test_everything_everywhere(S, oracle, frontend):
    """A flow that sends every source block all over the image: every box covers everything (scan: every source
    tile is a candidate of every output tile)."""
    rng = np.random.default_rng(11)
    N, C, H, W = 1, 3, 96, 330
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    flow = np.stack([rng.uniform(-W, W, (H, W)), rng.uniform(-H, H, (H, W))])[None].astype(np.float32)
    met = (rng.standard_normal((N, 1, H, W)) * 0.5).astype(np.float32)
    for mode in ("summation", "average", "linear", "softmax"):
        m = np.abs(met) + 0.1 if mode == "linear" else met
        out = host(S.FunctionSoftsplat(dev(x), dev(flow), dev(m), mode))
        ref = oracle.function_softsplat(x, flow, m, mode)
        assert np.allclose(out, ref, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(ref).max()))), mode


def test_randomised_sweep(S, oracle, frontend):
    rng = np.random.default_rng(20260929)
    modes = ["summation", "average", "linear", "softmax"]
    for case in range(36):
        N, C = int(rng.integers(1, 4)), int(rng.integers(1, 21))
        H, W = int(rng.integers(1, 90)), int(rng.integers(1, 210))
        y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
        kind = case % 6
        if kind == 0:
            fl = rng.uniform(-3, 3, (N, 2, H, W))
        elif kind == 1:
            fl = rng.uniform(-60, 60, (N, 2, H, W))
        elif kind == 2:
            fl = np.stack([(W / 2 - x) * rng.uniform(0.5, 1.0), (H / 2 - y) * rng.uniform(0.5, 1.0)])[None].repeat(N, 0)
        elif kind == 3:
            fl = np.stack([np.sin(x / 7 + y / 11) * 6, np.cos(x / 9 - y / 5) * 6])[None].repeat(N, 0)
        elif kind == 4:
            fl = rng.integers(-5, 6, (N, 2, H, W)).astype(np.float32)
        else:
            fl = rng.uniform(-2, 2, (N, 2, H, W))
            fl[rng.random(fl.shape) < 0.02] = np.nan
        fl = fl.astype(np.float32)
        v = rng.standard_normal((N, C, H, W)).astype(np.float32)
        met = (rng.standard_normal((N, 1, H, W)) * 0.7).astype(np.float32)
        mode = modes[case % 4]
        m = np.abs(met) + 0.1 if mode == "linear" else met
        ref = oracle.function_softsplat(v, fl, m, mode)
        out = host(S.FunctionSoftsplat(dev(v), dev(fl), dev(m), mode))
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.allclose(out, ref, rtol=2e-4, atol=2e-5 * scale), (case, N, C, H, W, kind, mode,
                                                                       float(np.abs(out - ref).max()))


def test_full_size_planes_vs_oracle(S, oracle, frontend):
    """768x1280 (1920 tiles: four rounds of the box test per workgroup in the scan front end), Euler t = 59."""
    H, W, C = 768, 1280, 3
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, C, H, W)).astype(np.float32)
    flow = oracle.euler_integration(smooth_motion(H, W), 59)[0]
    out = host(S.FunctionSoftsplat(dev(x), dev(flow), None, "summation"))
    ref = oracle.softsplat_forward(x, flow)
    bound = 4e-6 * oracle.softsplat_forward(np.abs(x), flow) + 1e-6
    assert (np.abs(out - ref) <= bound).all(), float((np.abs(out - ref) - bound).max())


def test_more_than_2048_source_tiles_per_sample(S, oracle, frontend):
    """560 x 2048 = 70 x 32 = 2240 tiles per sample: the scan front end tests the source tiles' boxes in rounds of 2048
    (two rounds here, the second one ragged), two samples."""
    H, W, C = 560, 2048, 2
    rng = np.random.default_rng(13)
    x = rng.standard_normal((2, C, H, W)).astype(np.float32)
    flow = np.concatenate([oracle.euler_integration(smooth_motion(H, W, n, amp=2.0), 12 + n)[0] for n in range(2)])
    out = host(S.FunctionSoftsplat(dev(x), dev(flow), None, "summation"))
    ref = oracle.softsplat_forward(x, flow)
    bound = 4e-6 * oracle.softsplat_forward(np.abs(x), flow) + 1e-6
    assert (np.abs(out - ref) <= bound).all(), float((np.abs(out - ref) - bound).max())


def test_many_deferred_pieces(S, oracle, frontend):
    """Partial collapses in a batch: dozens of tiles of every sample hold several thousand entries, so the rows front end hands
    more pieces to its pass-by-pass launch than that launch has workgroups (found by tools/dev/fuzz_frontends.py: the launch
    used to stop after one piece per workgroup)."""
    H, W, C = 200, 480, 5
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    for seed, N in ((2, 2), (6, 3)):
        rng = np.random.default_rng(seed)
        flow = np.stack([np.stack([(W * rng.uniform(0, 1) - xx) * rng.uniform(0.5, 1.0), (H * rng.uniform(0, 1) - yy) * rng.uniform(0.5, 1.0)])
                         for _ in range(N)]).astype(np.float32)
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        out = host(S.FunctionSoftsplat(dev(x), dev(flow), None, "summation"))
        ref = oracle.function_softsplat(x, flow, None, "summation")
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(out - ref).max()) < 2e-4 * scale, (seed, N, float(np.abs(out - ref).max()), scale)


@pytest.mark.parametrize("case", ["overflow_c64", "rounds_c65", "one_list_c33", "groups_c72", "shape_7x4", "shape_2x16"])
def test_sink_launch_task_list(S, oracle, case):
    """The scan front end's sink launch takes its (piece, task slot, channel group) from an ordered task list (csrc/splat_op.hip:
    op_sink_kernel): `overflow_c64` -- more tasks than the launch has workgroups (the C2 grid under an Euler t=30 flow: the early
    finishers draw the rest); `rounds_c65` -- more than 32 deferred pieces (two rounds of the list) with 65 planes (groups of 8 + one
    of 9); `one_list_c33` -- 4 channel groups (one list instead of one per XCD); `groups_c72` -- 9 units of 8 planes over 8 groups;
    `shape_7x4` / `shape_2x16` -- slr_splat_set_scan_shape: 7 piece slots (rounds of 7 pieces) x 4 groups; 2 piece slots and a group
    count the piece headers cannot hold (16: the library falls back to 8).
    Against the oracle, three calls each: the result must not depend on who drew which task."""
    L = S._lib.lib()
    prev = L.slr_splat_set_front_end(1)
    try:
        rng = np.random.default_rng(5)
        if case.startswith("shape_"):
            L.slr_splat_set_scan_shape(0, 0, *{"shape_7x4": (7, 4), "shape_2x16": (2, 16)}[case])
            case = "rounds_c65"
        if case == "overflow_c64":
            N, C, H, W = 1, 64, 256, 480
            flow = oracle.euler_integration(smooth_motion(H, W), 30)[0]
        else:
            N, C, H, W = {"rounds_c65": (3, 65, 200, 480), "one_list_c33": (2, 33, 200, 300), "groups_c72": (1, 72, 128, 256)}[case]
            yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
            flow = np.stack([np.stack([(W * rng.uniform(0.2, 0.8) - xx) * rng.uniform(0.5, 0.9), (H * rng.uniform(0.2, 0.8) - yy) * rng.uniform(0.5, 0.9)])
                             for _ in range(N)]).astype(np.float32)
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        xd, fd = dev(x), dev(flow)
        ref = oracle.softsplat_forward(x, flow)
        bound = 4e-6 * oracle.softsplat_forward(np.abs(x), flow) + 1e-6
        for _ in range(3):                                  # (who draws which task differs from call to call; entry slots are handed out by atomics: agreement to rounding)
            out = host(S.FunctionSoftsplat(xd, fd, None, "summation"))
            assert (np.abs(out - ref) <= bound).all(), float((np.abs(out - ref) - bound).max())
    finally:
        L.slr_splat_set_scan_shape(0, 0, 0, 0)
        L.slr_splat_set_front_end(prev)


def test_differential_fuzz_of_the_front_ends(S):
    """tools/dev/fuzz_frontends.py, a short run with a fixed seed: random shapes (ragged edges, one-pixel images, batches, a
    768x1280 case), ten flow families (incoherent, collapsing onto points / lines, far outside, non-finite sprinkles, ...) and
    all modes incl. the maximum splat; the scan and the rows front end against the CPU oracle on the same inputs."""
    import os
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dev", "fuzz_frontends.py")
    r = subprocess.run([sys.executable, tool, "150", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_prebinned_plan_reused_after_deferred_pieces_with_empty_channel_groups(S, oracle):
    """the pass-by-pass launch empties the deferred list for the plan's next use when its LAST workgroup arrives;
    with C = 72 (8 channel groups of 16 planes: three of them empty) the workgroups without planes used to return before
    arriving, the list stayed in place, and the next prebinned call on the same plan walked the same pieces again (ACCUM passes: wrong
    sums).  One binning, three calls on it, a flow with deferred pieces; every call must equal the oracle."""
    from slr_sfs_amd._lib import lib, ptr, stream_of, WS_PREBINNED
    L = lib()
    H, W = 64, 256
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    flow = np.stack([(100.3 + (xx % 7)) - xx, (20.6 + (yy % 8)) - yy])[None].astype(np.float32)       # 16384 sources -> 7 x 8 pixels
    dfl = dev(flow)
    st = stream_of(dfl)
    nbytes = int(L.slr_splat_workspace_bytes(1, H, W))
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    assert L.slr_splat_bin(ptr(dfl), 1, H, W, ptr(ws), nbytes, st) == 0
    rng = np.random.default_rng(72)
    for C in (72, 36, 72):
        x = rng.standard_normal((1, C, H, W)).astype(np.float32)
        dx, out = dev(x), torch.empty(1, C, H, W, device="cuda")
        assert L.slr_softsplat_forward(ptr(dx), ptr(dfl), ptr(out), 1, C, H, W, ptr(ws), nbytes, WS_PREBINNED, st) == 0
        ref = oracle.softsplat_forward(x, flow)
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(host(out) - ref).max()) < 2e-4 * scale, (C, float(np.abs(host(out) - ref).max()), scale)


def test_plane_stack_of_more_than_2_gib(S, oracle):
    """FunctionSoftsplat on 65 x 2160 x 3840 (a 4K frame through the same model: 2.16 GB per tensor, C*H*W*4 >= 2^31).  The reference
    indexes up to 2^31 ELEMENTS (softsplat.py:163, 408-416); here a sample's planes are addressed through 32-bit buffer offsets, so
    the stack is rendered in plane groups (plane_group: 64 + 1 planes).  Forward (summation and softmax) and both gradients against
    the oracle on sampled planes -- the first and last plane of each group among them."""
    C, H, W = 65, 2160, 3840
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(1, C, H, W, device="cuda", generator=g)
    met = torch.randn(1, 1, H, W, device="cuda", generator=g)
    m = smooth_motion(H, W, 3, amp=6.0)
    flow = torch.from_numpy(m).cuda() + (torch.rand(1, 2, H, W, device="cuda", generator=g) - 0.5)
    assert x.numel() * 4 >= 2 ** 31
    planes = [0, 31, 63, 64]
    fl = host(flow)
    xs = host(x[:, planes])
    out = S.FunctionSoftsplat(x, flow, None, "summation")
    ref = oracle.softsplat_forward(xs, fl)
    err = float(np.abs(host(out[:, planes]) - ref).max())
    assert err < 1e-4 * max(1.0, float(np.abs(ref).max())), err
    del out
    # softmax mode on the same planes: out = splat(x e^m) / splat(e^m)
    out = S.FunctionSoftsplat(x, flow, met, "softmax")
    ref = oracle.function_softsplat(xs, fl, host(met), "softmax")
    err = float(np.abs(host(out[:, planes]) - ref).max())
    assert err < 1e-4 * max(1.0, float(np.abs(ref).max())), err
    del out
    # backward: gradInput per plane is independent of the other planes; gradFlow sums over all of them -> a stack whose other planes
    # have zero gradOutput reduces to the sampled ones
    gout = torch.zeros(1, C, H, W, device="cuda")
    gs = torch.randn(1, len(planes), H, W, device="cuda", generator=g)
    gout[:, planes] = gs
    xr, fr = x.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    S.softsplat._FunctionSoftsplat.apply(xr, fr).backward(gout)
    gin_ref, gflow_ref = oracle.softsplat_backward(xs, fl, host(gs))
    assert np.array_equal(host(xr.grad[:, planes]), gin_ref)
    assert float(np.abs(host(xr.grad[:, [1, 62]])).max()) == 0.0
    d = np.abs(host(fr.grad) - gflow_ref)
    assert float(d.max()) <= 1e-5 * max(1.0, float(np.abs(gflow_ref).max())), float(d.max())
