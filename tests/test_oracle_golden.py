"""Pin the CPU oracle (oracle/) against fixtures generated from the reference itself
(tools/make_golden.py).  Euler integration must be bit-exact; the splat family is compared
at 1e-5 (the fixtures come from a sequential execution of the reference kernel text in the
same element order, so the observed difference is 0 -- the slack only covers libm exp)."""
import numpy as np
import pytest


def _load(golden_dir, name):
    return np.load(f"{golden_dir}/{name}.npz")


def test_euler_bit_exact(oracle, golden_dir):
    g = _load(golden_dir, "euler")
    for i in range(int(g["count"])):
        d, v = oracle.euler_integration(g[f"c{i}_motion"], int(g[f"c{i}_n"]))
        tag = str(g[f"c{i}_tag"])
        assert np.array_equal(d, g[f"c{i}_disp"]), tag
        assert np.array_equal(v, g[f"c{i}_vis"]), tag


def test_euler_all_frames_equals_per_frame(oracle, golden_dir):
    g = _load(golden_dir, "euler")
    by_motion = {}
    for i in range(int(g["count"])):
        by_motion.setdefault(g[f"c{i}_motion"].tobytes(), []).append(i)
    for idxs in by_motion.values():
        m = g[f"c{idxs[0]}_motion"]
        dall, vall = oracle.euler_integration_all(m, 60)
        for i in idxs:
            n = int(g[f"c{i}_n"])
            assert np.array_equal(dall[n], g[f"c{i}_disp"][0]), str(g[f"c{i}_tag"])
            assert np.array_equal(vall[n], g[f"c{i}_vis"][0])


def test_euler_module_batch(oracle, golden_dir):
    g = _load(golden_dir, "euler")
    m, dest = g["module_motion"], g["module_dest"]
    for b in range(m.shape[0]):
        d, v = oracle.euler_integration(m[b:b + 1], int(dest[b]))
        assert np.array_equal(d[0], g["module_disp"][b])
        assert np.array_equal(v[0], g["module_vis"][b])


def test_euler_batch_of_16(oracle, golden_dir):
    """The reference's EulerIntegration module on 16 samples with their own step counts (tools/make_golden_euler_batch.py)."""
    g = _load(golden_dir, "euler_batch")
    for tag in ("a", "b"):
        m, steps = g[f"{tag}_motion"], g[f"{tag}_steps"]
        assert m.shape[0] == 16
        for b in range(m.shape[0]):
            d, v = oracle.euler_integration(m[b:b + 1], int(steps[b]))
            assert np.array_equal(d[0], g[f"{tag}_disp"][b]), (tag, b)
            assert np.array_equal(v[0], g[f"{tag}_vis"][b]), (tag, b)


def test_splat_summation_forward_backward(oracle, golden_dir):
    g = _load(golden_dir, "splat_sum")
    for i in range(int(g["count"])):
        tag = str(g[f"c{i}_tag"])
        x, fl, go = g[f"c{i}_in"], g[f"c{i}_flow"], g[f"c{i}_gout"]
        out = oracle.softsplat_forward(x, fl)
        np.testing.assert_allclose(out, g[f"c{i}_out"], rtol=0, atol=1e-6, err_msg=tag)
        gin, gfl = oracle.softsplat_backward(x, fl, go)
        np.testing.assert_allclose(gin, g[f"c{i}_gin"], rtol=0, atol=1e-6, err_msg=tag)
        np.testing.assert_allclose(gfl, g[f"c{i}_gflow"], rtol=1e-6, atol=1e-5, err_msg=tag)


def test_function_softsplat_modes(oracle, golden_dir):
    g = _load(golden_dir, "splat_modes")
    for i in range(int(g["count"])):
        tag = str(g[f"c{i}_tag"])
        out = oracle.function_softsplat(g[f"c{i}_in"], g[f"c{i}_flow"], g[f"c{i}_metric"], str(g[f"c{i}_mode"]))
        np.testing.assert_allclose(out, g[f"c{i}_out"], rtol=1e-5, atol=1e-5, err_msg=tag)
    out = oracle.function_softsplat(g["module_in"], g["module_flow"], np.ones_like(g["module_in"][:, :1]), "summation")
    np.testing.assert_allclose(out, g["module_out"], rtol=0, atol=1e-6)


def test_max_splat_family(oracle, golden_dir):
    g = _load(golden_dir, "splat_max")
    for i in range(int(g["count"])):
        tag = str(g[f"c{i}_tag"])
        mx = oracle.maxsplat_forward(g[f"c{i}_in"], g[f"c{i}_flow"])
        assert np.array_equal(mx, g[f"c{i}_max"]), tag
        wn = oracle.maximum_warp_norm_splat(g[f"c{i}_in"], g[f"c{i}_flow"])
        assert np.array_equal(wn, g[f"c{i}_warpnorm"]), tag


@pytest.mark.parametrize("t", [0, 1, 30, 59])
def test_forward_flow_decoder_input(oracle, golden_dir, t):
    """a6: the tensor the reference forward_flow feeds its decoder (baseline and SLR v1)."""
    g = _load(golden_dir, "pipeline_a6")
    N = int(g["N"])
    gen = oracle.synth_baseline(g["fs"], g["Z"], g["motion"], t, N)
    np.testing.assert_allclose(gen, g[f"baseline_t{t}_gen_fs"], rtol=1e-5, atol=2e-6)
    a = g["alpha_out"]
    abg = (1.0 / (1.0 + np.exp(-a[:, 0:1]))).astype(np.float32)
    for tag, a0 in (("v1", True), ("v1noa0", False)):
        if f"{tag}_t{t}_gen_fs" not in g:
            continue
        gen, afl, _ = oracle.synth_v1(g["fs"], g["Z"], a[:, 1:2], abg, g["motion"], t, N, use_alpha0=a0)
        np.testing.assert_allclose(gen, g[f"{tag}_t{t}_gen_fs"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(np.concatenate([gen, afl], 1), g[f"{tag}_t{t}_dec_alpha_in"],
                                   rtol=1e-5, atol=5e-6)
        # holes are exact zeros: the partial-conv decoder masks on x != 0 (architectures.py:369)
        assert np.array_equal(gen == 0, g[f"{tag}_t{t}_gen_fs"] == 0)


@pytest.mark.parametrize("tag", ["c2", "c3"])
def test_full_size_digests(oracle, golden_dir, tag):
    """Oracle vs digests of the reference's outputs at the C2 (256x480) and C3 (768x1280) grids."""
    from conftest import large_case
    g, motion, inp, steps = large_case(golden_dir, tag)
    disp, vis = oracle.euler_integration(motion, steps)
    assert np.array_equal(disp.ravel()[g[f"{tag}_disp_pos"]], g[f"{tag}_disp_val"])
    assert np.array_equal(disp.astype(np.float64).sum(axis=(2, 3)), g[f"{tag}_disp_sum"])
    assert float(vis.sum()) == float(g[f"{tag}_vis_sum"])
    out = oracle.softsplat_forward(inp, disp)
    assert np.array_equal(out.ravel()[g[f"{tag}_out_pos"]], g[f"{tag}_out_val"])      # same summation order: exact
    np.testing.assert_allclose(out.astype(np.float64).sum(axis=(2, 3)), g[f"{tag}_out_sum"], rtol=1e-12)
    assert int((out == 0).sum()) == int(g[f"{tag}_holes"])


def check_a6_digest(g, tag, kind, t, gen, alpha=None, rtol=1e-5, atol=2e-6, hole_slack=0):
    """Compare a decoder input [1,64,H,W] (and, for SLR v1, the warped alpha plane [1,1,H,W]) with the digests of
    the REFERENCE's forward_flow output stored in pipeline_a6_large.npz.  Returns the largest sampled error."""
    val = gen.ravel()[g[f"{tag}_pos"]]
    ref = g[f"{tag}_{kind}_t{t}_val"]
    np.testing.assert_allclose(val, ref, rtol=rtol, atol=atol, err_msg=f"{tag} {kind} t={t}")
    np.testing.assert_allclose(gen.astype(np.float64).sum(axis=(2, 3)), g[f"{tag}_{kind}_t{t}_sum"],
                               rtol=1e-5, atol=0.5)
    # holes are exactly 0, and the same ones.  hole_slack (GPU path only): an element that is 0.0 in one summation
    # order through exact cancellation of its contributions and ~1e-8 in another is not a hole (1 of 37.7 M observed)
    assert abs(int((gen == 0).sum()) - int(g[f"{tag}_{kind}_t{t}_holes"])) <= hole_slack
    err = float(np.abs(val - ref).max())
    if alpha is not None:
        av = alpha.ravel()[g[f"{tag}_apos"]]
        np.testing.assert_allclose(av, g[f"{tag}_{kind}_t{t}_alpha_val"], rtol=rtol, atol=5e-6)
        np.testing.assert_allclose(float(alpha.astype(np.float64).sum()), float(g[f"{tag}_{kind}_t{t}_alpha_sum"]),
                                   rtol=1e-5, atol=0.5)
        err = max(err, float(np.abs(av - g[f"{tag}_{kind}_t{t}_alpha_val"]).max()))
    return err


@pytest.mark.parametrize("tag,t", [("c3", 30), ("sq", 59)])
def test_forward_flow_decoder_input_full_size_digests(oracle, golden_dir, tag, t):
    """Oracle vs the reference's own forward_flow (baseline + SLR v1) at 768x1280 / 768x768, 64 features, N=60."""
    from conftest import a6_large_inputs
    g = _load(golden_dir, "pipeline_a6_large")
    _, _, H, W = [int(v) for v in g[f"{tag}_shape"]]
    fs, Z, motion, a = a6_large_inputs(H, W)
    N = int(g["N"])
    check_a6_digest(g, tag, "baseline", t, oracle.synth_baseline(fs, Z, motion, t, N))
    abg = (1.0 / (1.0 + np.exp(-a[:, 0:1]))).astype(np.float32)
    gen, afl, _ = oracle.synth_v1(fs, Z, a[:, 1:2], abg, motion, t, N)
    check_a6_digest(g, tag, "v1", t, gen, afl)


def test_euler_backward_vs_reference_autograd(oracle, golden_dir):
    """d euler_integration / d motion: the oracle's restatement vs torch autograd through the reference's own loop
    (tests/golden/euler_grad.npz).  Accumulation order differs (index_put backward), hence 1e-5."""
    g = _load(golden_dir, "euler_grad")
    for i in range(int(g["count"])):
        gm = oracle.euler_backward(g[f"c{i}_motion"], int(g[f"c{i}_n"]), g[f"c{i}_gout"])
        np.testing.assert_allclose(gm, g[f"c{i}_gmotion"], rtol=1e-5, atol=1e-5, err_msg=str(g[f"c{i}_tag"]))
