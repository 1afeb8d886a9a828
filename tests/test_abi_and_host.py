"""CPU-side checks: the C-ABI library builds for gfx950, loads, and exports every symbol
include/slr_splat.h declares (no compute without a GPU); host-side glue."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import slr_sfs_amd
    if not os.path.exists(slr_sfs_amd._lib.LIB_PATH):
        slr_sfs_amd._lib.build()
    return slr_sfs_amd._lib.lib()


def test_header_symbols_exported(L):
    import slr_sfs_amd
    hdr = open(os.path.join(ROOT, "include", "slr_splat.h")).read()
    declared = set(re.findall(r"\b(slr_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(slr_sfs_amd._lib.SYMBOLS)
    raw = ctypes.CDLL(slr_sfs_amd._lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    m = re.search(r"#define\s+SLR_ABI_VERSION\s+(\d+)", hdr)
    assert L.slr_abi_version() == slr_sfs_amd._lib.ABI_VERSION == int(m.group(1))


def test_workspace_size_and_argument_errors(L):
    # (row-segment lists: 256 records of 8 bytes per tile and list, three lists + plans -- 12.7 MB at 768x1280; the per-pixel bins of
    #  rounds 1-3 took 12 bytes per source pixel + partial tiles: ~300 MB)
    assert 3 * 1920 * 256 * 8 < L.slr_splat_workspace_bytes(1, 768, 1280) < 32 << 20
    assert L.slr_splat_workspace_bytes(0, 768, 1280) == 0
    assert L.slr_splat_workspace_bytes(1, 16, 16) > 0
    # scratch of the backward's channel groups (host arithmetic only): (groups - 1) partial gradFlow planes pairs -- two groups on grids larger
    # than the chip, up to four on the training crops, none below 16 channels or where the tiled kernel does not take the plane stack (>= 2 GiB)
    assert L.slr_softsplat_backward_ws_bytes(1, 65, 768, 1280) == 1 * 2 * 768 * 1280 * 4
    assert L.slr_softsplat_backward_ws_bytes(2, 65, 256, 256) == 3 * 2 * 2 * 256 * 256 * 4
    assert L.slr_softsplat_backward_ws_bytes(1, 8, 768, 1280) == 0 and L.slr_softsplat_backward_ws_bytes(1, 65, 0, 1280) == 0
    assert L.slr_softsplat_backward_ws_bytes(1, 600, 768, 1280) == 0
    # argument validation happens before anything touches the device
    rc = L.slr_softsplat_forward(None, None, None, 1, 1, 8, 8, None, 0, 0, None)
    assert rc == -1 and b"null" in L.slr_last_error()
    rc = L.slr_euler_integrate(None, 8, 8, 1, 1.0, None, None, None)
    assert rc == -1


def test_operators_refuse_cpu_tensors_and_have_no_fallback():
    import slr_sfs_amd as S
    z = torch.zeros
    with pytest.raises(NotImplementedError):
        S.FunctionSoftsplat(z(1, 3, 8, 8), z(1, 2, 8, 8), None, "summation")
    with pytest.raises(NotImplementedError):
        S.euler_integration(z(1, 2, 8, 8), 3)
    with pytest.raises(NotImplementedError):
        S.ModuleMaximumsplat()(z(1, 3, 8, 8), z(1, 2, 8, 8))
    # the networks either side of the path as well: CPU tensors raise unless a TEST opts into the torch
    # definition (nets.cpu_reference(), used to validate the modules against the reference's classes)
    from slr_sfs_amd import nets
    with pytest.raises(NotImplementedError), torch.no_grad():
        nets.DecoderPconv2(64, 3).eval()(z(1, 64, 8, 8))
    with pytest.raises(NotImplementedError), torch.no_grad():
        nets.Conv(16, 64, 3)(z(1, 16, 8, 8))
    # the product never imports the oracle
    import sys
    assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules if "slr" in m)
    for f in os.listdir(os.path.join(ROOT, "slr-sfs_amd")):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "slr-sfs_amd", f)).read(), f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import slr_sfs_amd
    monkeypatch.setattr(slr_sfs_amd._lib, "_lib", None)
    monkeypatch.setattr(slr_sfs_amd._lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="missing"):
        slr_sfs_amd._lib.lib()


def test_dropin_module_names():
    import sys
    import slr_sfs_amd as S
    ss, eim = S.install_into_reference()
    assert sys.modules["models.softsplat"] is ss
    assert sys.modules["models.projection.euler_integration_manipulator"] is eim
    for name in ("FunctionSoftsplat", "ModuleSoftsplat", "_FunctionSoftsplat", "ModuleMaximumsplat",
                 "ModuleMaximumWarpNormsplat"):
        assert hasattr(ss, name)
    assert hasattr(eim, "euler_integration") and hasattr(eim, "EulerIntegration")
    del sys.modules["models.softsplat"], sys.modules["models.projection.euler_integration_manipulator"]


def test_prepare_motion_and_alpha_semantics():
    from slr_sfs_amd import pipeline
    flow = torch.ones(1, 2, 4, 8)
    m = pipeline.prepare_motion(flow, 8, 24, speed=2.0, align=30, N=60)
    assert m.shape == (1, 2, 8, 24)
    assert torch.allclose(m[:, 0], torch.full((1, 8, 24), 24 / 8 * 2.0 * 0.5))
    assert torch.allclose(m[:, 1], torch.full((1, 8, 24), 8 / 4 * 2.0 * 0.5))


def test_device_code_has_only_the_safe_packed_fp32_form(tmp_path):
    """csrc/Makefile builds without compiler-chosen packed-fp32 VALU instructions: with them the splat tile kernel returned wrong
    low halves next to a concurrently running MFMA kernel on MI355X (tools/ubench/pkfma_repro.hip: the form
    that fails is the weight broadcast from the HIGH register of a pair, op_sel:[0,1,0]).  The one packed instruction the library
    may contain is the hand-placed v_pk_fma_f32 of the gather with the weight broadcast from the LOW register
    (op_sel_hi:[1,0,1], splat_tile.hpp: accum4).  Checked on the ISA of every gfx950 code object bundled in the built library."""
    import struct
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(f"{llvm}/llvm-objcopy") and os.path.exists(f"{llvm}/llvm-objdump")):
        pytest.skip("llvm-objcopy / llvm-objdump not available")
    from slr_sfs_amd import _lib
    fat = tmp_path / "fat.bin"
    subprocess.check_call([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", _lib.LIB_PATH, str(tmp_path / "stripped")])
    data = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    objects = mfma = 0
    start = data.find(magic)
    while start >= 0:
        n = struct.unpack_from("<Q", data, start + 24)[0]
        p = start + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                co = tmp_path / f"dev{objects}.co"
                co.write_bytes(data[start + off:start + off + size])
                isa = subprocess.run([f"{llvm}/llvm-objdump", "-d", str(co)], capture_output=True, text=True, check=True).stdout
                assert "s_endpgm" in isa
                packed = re.findall(r"v_pk_(?:fma|add|mul)_f32[^\n]*", isa)
                bad = [q for q in packed if not (q.startswith("v_pk_fma_f32") and "op_sel_hi:[1,0,1]" in q and "op_sel:" not in q
                                                 and "neg_" not in q)]
                assert not bad, f"{len(bad)} packed-fp32 instructions of another form in code object {objects} ({triple}): {bad[:3]}"
                mfma += isa.count("v_mfma_f32_32x32x16_f16")
                # no kernel of the library spills to scratch memory (round 1 shipped a convolution variant with 60 bytes
                # of scratch per lane): every kernel descriptor's private segment size is 0
                # ... and no scratch instruction exists anywhere.  (One exception in the descriptors: the pass-by-pass launch of the
                # rows front end -- splat_tile_kernel<.., WHOLE = true, FE = 2>, normally empty -- gets a 20-byte frame reserved by
                # the register allocator for its loop over deferred pieces, without a single instruction that touches it.)
                assert not re.search(r"\bscratch_(?:load|store)", isa), f"scratch instructions in code object {objects} ({triple})"
                # no register array is indexed with a run-time value: that compiles to s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off per
                # element (round 5: 192 such sequences were 2.8 of the Winograd workgroup's 46 us -- accumulator rows indexed with the
                # wave's half; both halves are instantiated now)
                assert "s_set_gpr_idx_on" not in isa and "v_movrel" not in isa, f"GPR index mode in code object {objects} ({triple})"
                # the clip kernels on plane-blocked values load 16 bytes per entry and chunk: this compiler narrows
                # __builtin_amdgcn_raw_buffer_load_b128 to ONE dword (and splats it) when the elements of its integer result are bit-cast one
                # by one (csrc/splat_core.hpp: buf_ld4) -- 12 such loads in each of the four blocked instantiations
                if "clip_tile_kernel" in isa:
                    assert isa.count("buffer_load_dwordx4") >= 48, isa.count("buffer_load_dwordx4")
                notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", str(co)], capture_output=True, text=True, check=True).stdout
                kernels = re.findall(r"\.name:\s*(\S+)[\s\S]*?\.private_segment_fixed_size:\s*(\d+)", notes)
                sizes = [int(v) for v in re.findall(r"\.private_segment_fixed_size:\s*(\d+)", notes)]
                assert sizes
                spilled = [(k, int(v)) for k, v in kernels if int(v) and not (re.search(r"splat_tile_kernelILb[01]ELb[01]ELi\d+ELi\d+ELb1ELi2E", k) and int(v) <= 32)]
                assert not spilled and (kernels or not any(sizes)), f"scratch in code object {objects} ({triple}): {spilled or sizes}"
                objects += 1
        start = data.find(magic, start + 24)
    assert objects >= 6                      # one per source file of csrc/Makefile
    assert mfma > 0                          # the disassembly really is the convolution's device code too


def test_synthesize_refuses_grids_that_are_not_multiples_of_8():
    from slr_sfs_amd import pipeline
    an = pipeline.BaselineAnimator()
    with pytest.raises(ValueError, match="multiples of 8"):
        an.synthesize(torch.zeros(1, 3, 150, 136), torch.zeros(1, 2, 150, 136), 4)


class _FixedOut(torch.nn.Module):
    def __init__(self, value):
        super().__init__()
        self.value = value

    def forward(self, x):
        return self.value


V1_VARIANTS = {"plain": {}, "region": {}, "clamp": {"clamp_alpha": 0.6}, "softmax": {"use_alpha_softmax": True},
               "fluidonly": {"use_fluid_alpha_only": True}, "bgonly": {"use_bg_alpha_only": True},
               "v1weights": {"softmax_v1": True}}


@pytest.mark.parametrize("tag", sorted(V1_VARIANTS))
def test_v1_compositing_host_logic_vs_reference(oracle, golden_dir, tag):
    """Host side of SLRv1Animator (everything after the splat: sigmoid / normalise / compositing variants /
    alpha_region blur / return dict) against the return dict of the REFERENCE's forward_flow
    (tests/golden/pipeline_v1_surface.npz), fed with the oracle's decoder inputs.  CPU only."""
    import types
    from conftest import v1_surface_inputs
    from slr_sfs_amd import pipeline
    g = np.load(f"{golden_dir}/pipeline_v1_surface.npz")
    W, N, t = int(g["W"]), int(g["N"]), int(g["t"])
    d = v1_surface_inputs(W)
    T = lambda a: torch.from_numpy(a)
    kw = dict(V1_VARIANTS[tag])
    an = pipeline.SLRv1Animator(decoder=_FixedOut(T(d["dec_out"])), alpha_decoder=_FixedOut(T(d["adec_out"])),
                                alpha_encoder=_FixedOut(T(d["alpha_out"])), **kw)
    a = d["alpha_out"]
    abg = (1.0 / (1.0 + np.exp(-a[:, 0:1]))).astype(np.float32)
    gen, afl, _ = oracle.synth_v1(d["fs"], d["Z"], a[:, 1:2], abg, d["motion"], t, N,
                                  variant="v1" if tag == "v1weights" else None)
    clip = types.SimpleNamespace(bg=torch.tanh(T(d["bg_raw"])), alpha_bg=torch.sigmoid(T(a[:, 0:1])),
                                 alpha_bg_raw=T(a[:, 0:1]),
                                 alpha_region=pipeline.blur_alpha_region(T(d["alpha_region"]), W) if tag == "region" else None)
    out = an._decode(clip, T(gen), T(afl))
    assert sorted(out.keys()) == [str(k) for k in g[f"{tag}_keys"]]
    for k, v in out.items():
        ref = g[f"{tag}_{k}"] if f"{tag}_{k}" in g else g[f"plain_{k}"]
        np.testing.assert_allclose(v.numpy(), ref, rtol=2e-5, atol=2e-6, err_msg=f"{tag} {k}")


def test_splat_options_from_checkpoint_opts():
    """How the reference's forward_flow reads the splat-weight options of a checkpoint's pickled Namespace
    (animating_softmax_splating.py:849-859: clamp unless the Namespace HAS no_clamp_Z; 2-layer model: never)."""
    import argparse
    from slr_sfs_amd import pipeline
    old = argparse.Namespace(use_softmax_splatter_v1=False)                 # written before --no_clamp_Z existed
    new = argparse.Namespace(no_clamp_Z=False, use_softmax_splatter_v1=False, use_softmax_splatter_v2=False)
    assert pipeline.splat_options(old, two_layer=False)["clamp_z"] == (-20.0, 20.0)
    assert pipeline.splat_options(new, two_layer=False)["clamp_z"] is None
    assert pipeline.splat_options(old, two_layer=True)["clamp_z"] is None
    v2 = argparse.Namespace(no_clamp_Z=True, use_softmax_splatter_v1=True, use_softmax_splatter_v2=True)
    kw = pipeline.splat_options(v2, two_layer=False)
    assert kw["softmax_v2"] and not kw["softmax_v1"]                       # v2 is tested first (:849-853)
    an = pipeline.SLRv1Animator(opts=argparse.Namespace(use_alpha0_as_blending_weight=True, clamp_alpha=0.25,
                                                        use_alpha_softmax=False))
    assert an.use_alpha0 and an.clamp_alpha == 0.25 and not an.use_fluid_alpha_only
    assert pipeline.BaselineAnimator(opts=old).splat_kw["clamp_z"] == (-20.0, 20.0)


def test_conv_rung_retry_and_reset():
    """convs="auto": a raised rung is not for good (pipeline._ConvRung) -- after RUNG_RETRY_CLIPS clips on it the next clip starts at
    rung 0 again, and reset_conv_rung() goes back at once.  Host logic only."""
    from slr_sfs_amd import pipeline

    class Owner(pipeline._ConvRung):
        pass
    a = Owner()
    a._clip_begins()                                       # rung 0: nothing is counted
    assert getattr(a, "_conv_rung", 0) == 0 and getattr(a, "_rung_clips", 0) == 0
    a._conv_rung = 2
    for _ in range(pipeline.RUNG_RETRY_CLIPS):
        a._clip_begins()
        assert a._conv_rung == 2
    a._clip_begins()                                       # the clip after RUNG_RETRY_CLIPS raised ones: back to rung 0
    assert a._conv_rung == 0 and a._rung_clips == 0
    a._conv_rung = 1
    a._clip_begins()
    assert a._rung_clips == 1
    a.reset_conv_rung()
    assert a._conv_rung == 0 and a._rung_clips == 0
    saved = pipeline.RUNG_RETRY_CLIPS
    try:
        pipeline.RUNG_RETRY_CLIPS = 0                     # 0 = never retry
        a._conv_rung = 2
        for _ in range(40):
            a._clip_begins()
        assert a._conv_rung == 2
    finally:
        pipeline.RUNG_RETRY_CLIPS = saved


def test_conv_policies_are_the_package_kernels_only():
    """The torch / MIOpen composition of the networks is a validation aid (nets.torch_convolutions), not a selectable product route."""
    from slr_sfs_amd import pipeline
    assert pipeline.CONV_POLICIES == ("auto", "split", "fp32", "fp32-winograd")
    with pytest.raises(AssertionError):
        pipeline.BaselineAnimator(convs="torch")


def test_failed_call_drops_cached_workspaces():
    """A hipError from a library call drops the cached workspaces (their counters may be dirty; later calls say SLR_WS_CLEAN)."""
    from slr_sfs_amd import _lib
    _lib._ws_cache[("fake",)] = object()
    with pytest.raises(RuntimeError):
        _lib.check(1, "test")
    assert not _lib._ws_cache
    _lib._ws_cache[("fake",)] = object()
    with pytest.raises(RuntimeError):
        _lib.check(-1, "test")                           # an argument error launched nothing: the cache stays
    assert _lib._ws_cache
    _lib.clear_workspaces()
