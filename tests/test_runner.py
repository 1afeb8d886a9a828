"""The runners (slr_sfs_amd/runner.py, tools/animate.py, tools/animate_scenes.py): scene selection like the reference's
test_all_CLAW_scenes.py (CPU), one scene through the whole path with a reference-format checkpoint, and the same
directory rendered by two ranks (frames sharded, one all-gather per clip) -- config C5's control flow on one GPU."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import nets_fixture as NF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(d, name, h=50, w=70, seed=0):
    from PIL import Image
    from slr_sfs_amd import io
    rng = np.random.default_rng(seed)
    Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(os.path.join(d, name + "_input.jpg"))
    y, x = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    flow = np.stack([1.5 * np.sin(x / 9 + seed), 1.2 * np.cos(y / 7)], -1) * (x > 0.3 * w)[..., None]
    io.write_flo(os.path.join(d, name + ".flo"), flow.astype(np.float32))


def test_scene_selection_like_the_reference(tmp_path):
    from slr_sfs_amd import runner
    d = str(tmp_path)
    for i, n in enumerate(("00003", "00001", "00002", "00010")):
        _scene(d, n, seed=i)
    open(os.path.join(d, "notes.txt"), "w").write("x")
    names = lambda **kw: [s[0] for s in runner.list_scenes(d, **kw)]
    assert names() == ["00001", "00002", "00003", "00010"]                      # sorted *_input.jpg
    assert names(start=1, end=2) == ["00002", "00003"]                          # inclusive window over the sorted list
    align = os.path.join(d, "align.json")
    json.dump({"00002": 300, "00010": 600}, open(align, "w"))
    assert names(align=align) == ["00002", "00010"]                             # scenes missing from the table are skipped
    s = runner.list_scenes(d)[0]
    assert s[1].endswith("00001_input.jpg") and s[2].endswith("00001.flo") and os.path.exists(s[2])
    other = tmp_path / "flows"
    other.mkdir()
    os.rename(os.path.join(d, "00001.flo"), str(other / "00001.flo"))
    assert runner.list_scenes(d, str(other))[0][2] == str(other / "00001.flo")  # flow_dir as the fallback


def _checkpoint(path, golden_dir, nets=("encoder", "projector")):
    g = np.load(f"{golden_dir}/nets_reference.npz")
    sd = {}
    for name in nets:
        keys = [str(k) for k in g[f"{name}_keys"]]
        sd.update({NF.NETS[name][0] + k: v for k, v in NF.state_dict(name, keys, g[f"{name}_shapes"]).items()})
    # a Namespace written by the current option parser HAS no_clamp_Z (pipeline.splat_options)
    torch.save({"state_dict": sd, "opts": argparse.Namespace(no_clamp_Z=False, use_softmax_splatter_v1=False,
                                                             use_softmax_splatter_v2=False)}, path)


@pytest.mark.gpu
def test_one_scene_with_a_reference_format_checkpoint(tmp_path, golden_dir):
    from PIL import Image
    from slr_sfs_amd import io, pipeline, runner
    d = str(tmp_path)
    _scene(d, "00007")
    ck = os.path.join(d, "ck.pth")
    _checkpoint(ck, golden_dir)
    dev = torch.device("cuda:0")
    model = runner.load_model(ck, False, dev)
    align = os.path.join(d, "align.json")
    json.dump({"00007": 300}, open(align, "w"))
    H = W = 64
    N = 6
    dt, fdir = runner.animate_scene(model, os.path.join(d, "00007_input.jpg"), os.path.join(d, "00007.flo"),
                                    os.path.join(d, "out"), "00007", H, W, N, 0.5, align, video=False)
    files = sorted(os.listdir(fdir))
    assert files == ["%06d.png" % t for t in range(N)] and dt > 0
    image, (rw, rh) = io.load_image(os.path.join(d, "00007_input.jpg"), H, W)
    assert (rw, rh) == (70, 50)
    motion = pipeline.prepare_motion(io.load_motion(os.path.join(d, "00007.flo")), H, W, 0.5, 300, N)
    ref = io.frames_to_uint8(model.synthesize(image.to(dev), motion.to(dev), N), (rh, rw)).cpu().numpy()
    for t in range(N):
        got = np.asarray(Image.open(os.path.join(fdir, files[t])))
        assert got.shape == (50, 70, 3)
        assert np.abs(got.astype(int) - ref[t].astype(int)).max() <= 1           # (record order -> last-bit differences)
    assert np.abs(np.diff(ref.astype(int), axis=0)).max() > 0                    # the clip moves
    # the non-rawsize scripts write half-size frames (test_baseline_4eval.py:160-161)
    _, hdir = runner.animate_scene(model, os.path.join(d, "00007_input.jpg"), os.path.join(d, "00007.flo"),
                                   os.path.join(d, "half"), "00007", H, W, N, 0.5, align, video=False, half_size=True)
    assert np.asarray(Image.open(os.path.join(hdir, "000003.png"))).shape == (25, 35, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("v1,shard", [(False, "frames"), (True, "frames"), (False, "scenes")])
def test_directory_on_two_ranks_matches_one_rank(tmp_path, golden_dir, v1, shard):
    from PIL import Image
    d = str(tmp_path)
    for i, n in enumerate(("00001", "00002")):
        _scene(d, n, seed=i)
    ck = os.path.join(d, "ck.pth")
    _checkpoint(ck, golden_dir, ("encoder", "projector", "net_bg", "net_alpha_encoder", "net_alpha_decoder") if v1 else
                ("encoder", "projector"))
    args = [d, d, None, ck, "Demo", "64", "7", "0.5", "None", "-1", "-1", "--no-video", "--shard", shard] + (["--v1"] if v1 else [])
    env = dict(os.environ, SLR_ONE_GPU_GLOO="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29713")
    outs = []
    for world in (1, 2):
        out = os.path.join(d, f"out{world}")
        a = list(args)
        a[2] = out
        if world == 1:
            cmd = [sys.executable, os.path.join(ROOT, "tools", "animate_scenes.py")] + a
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                   "--master-port", "29713", os.path.join(ROOT, "tools", "animate_scenes.py")] + a
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert "2 scenes, 14 frames" in r.stdout
        outs.append(out)
    keys = ["PredImg"] + (["FluidImg", "CompositeFluidAlpha"] if v1 else [])
    for scene in ("00001", "00002"):
        for k in keys:
            a_dir, b_dir = (os.path.join(o, scene, scene, k) for o in outs)
            assert sorted(os.listdir(a_dir)) == sorted(os.listdir(b_dir)) == ["%06d.png" % t for t in range(7)]
            for f in os.listdir(a_dir):
                x, y = (np.asarray(Image.open(os.path.join(q, f))).astype(int) for q in (a_dir, b_dir))
                assert x.shape == y.shape and np.abs(x - y).max() <= 1, (scene, k, f)
        if v1:
            x, y = (np.asarray(Image.open(os.path.join(o, scene, scene, "BGImg.png"))).astype(int) for o in outs)
            assert np.abs(x - y).max() <= 1
