"""BASELINE.json configs C1 and C2 AS STATED against digests of the reference's own operators
(tests/golden/config_literal.npz, tools/make_golden_configs.py):
  C1  one 1x3x256x480 frame, motion Euler-integrated over N = 5 steps, FunctionSoftsplat(..., 'softmax');
  C2  random 64-channel 256x480 feature + flow (incoherent U(-8,8), and a smooth field at t = 30), 'softmax'.
CPU (not gpu): the oracle (pins it at these sizes / this mode);  GPU (-m gpu): the HIP operators, every front end forced."""
import numpy as np
import pytest
import torch

from config_inputs import config_inputs, digest_positions

TAGS = ["c1", "c2_inc", "c2_smooth"]


def _check(g, tag, out, tol):
    out = np.ascontiguousarray(out, dtype=np.float32)
    assert list(out.shape) == [int(v) for v in g[f"{tag}_shape"]]
    pos = digest_positions(tag, out.size)
    err = float(np.abs(out.ravel()[pos] - g[f"{tag}_val"]).max())
    assert err <= tol, (tag, err)
    assert int((out == 0).all(axis=1).sum()) == int(g[f"{tag}_holes"])                   # the same pixels stay empty
    sums = out.astype(np.float64).sum(axis=(2, 3))
    assert float(np.abs(sums - g[f"{tag}_plane_sums"]).max()) <= tol * out.shape[2] * out.shape[3] * 0.02, tag
    return err


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_vs_reference_operators(oracle, golden_dir, tag):
    g = np.load(f"{golden_dir}/config_literal.npz")
    x, metric, motion, steps, flow = config_inputs(tag)
    if flow is None:
        flow = oracle.euler_integration(motion, steps)[0]
        assert np.array_equal(flow.astype(np.float64).sum(axis=(2, 3)), g[f"{tag}_flow_sum"])     # Euler integration: bit-exact
    _check(g, tag, oracle.function_softsplat(x, flow, metric, "softmax"), 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("front_end", ["auto", "scan", "rows"])
@pytest.mark.parametrize("tag", TAGS)
def test_hip_operators_vs_reference_operators(golden_dir, tag, front_end):
    import slr_sfs_amd as S
    L = S._lib.lib()
    g = np.load(f"{golden_dir}/config_literal.npz")
    x, metric, motion, steps, flow = config_inputs(tag)
    d = lambda a: torch.from_numpy(a).cuda()
    prev = L.slr_splat_set_front_end({"auto": -1, "scan": 1, "rows": 2}[front_end])
    try:
        fl = d(flow) if flow is not None else S.euler_integration(d(motion), steps)[0]
        if flow is None:
            assert np.array_equal(fl.cpu().numpy().astype(np.float64).sum(axis=(2, 3)), g[f"{tag}_flow_sum"])
        out = S.FunctionSoftsplat(d(x), fl, d(metric), "softmax")
        mod = S.ModuleSoftsplat("softmax")(d(x), fl, d(metric))                          # the module form the models use
        assert torch.equal(out, mod) or float((out - mod).abs().max()) < 1e-5
    finally:
        L.slr_splat_set_front_end(prev)
    _check(g, tag, out.cpu().numpy(), 1e-4)                                              # north_star's bound; measured ~1e-6
