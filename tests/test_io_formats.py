"""Wire formats either side of the path: .flo, motion loading, frame export."""
import os
import struct

import numpy as np
import pytest
import torch

from slr_sfs_amd import io, pipeline


def test_flo_roundtrip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    fl = rng.standard_normal((5, 7, 2)).astype(np.float32)
    p = str(tmp_path / "a.flo")
    io.write_flo(p, fl)
    raw = open(p, "rb").read()
    assert struct.unpack("<f", raw[:4])[0] == 202021.25          # Middlebury magic 'PIEH'
    assert struct.unpack("<ii", raw[4:12]) == (7, 5)             # width, height
    assert len(raw) == 12 + 5 * 7 * 2 * 4
    assert np.array_equal(io.read_flo(p), fl)
    m = io.load_motion(p)
    assert m.shape == (1, 2, 5, 7) and np.array_equal(m[0, 0].numpy(), fl[..., 0])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not present")
def test_flo_matches_reference_reader(tmp_path):
    """The reference's own read_flo (utils/utils.py:252-261) executed on a file we wrote."""
    import re
    src = open("/root/reference/utils/utils.py").read()
    fn = re.search(r"def read_flo\(strFile\):.*?reshape\(\[ intHeight, intWidth, 2 \]\)", src, re.S).group(0)
    ns = {"np": np}
    exec(fn, ns)
    fl = np.random.default_rng(1).standard_normal((4, 9, 2)).astype(np.float32)
    p = str(tmp_path / "b.flo")
    io.write_flo(p, fl)
    assert np.array_equal(ns["read_flo"](p), fl)
    assert np.array_equal(io.read_flo(p), ns["read_flo"](p))


def test_plain_pth_motion(tmp_path):
    arr = np.random.default_rng(2).standard_normal((1, 2, 6, 8)).astype(np.float32)
    p = str(tmp_path / "m.pth")
    torch.save(arr, p)
    assert np.array_equal(io.load_motion(p).numpy(), arr)


def test_image_and_frame_export(tmp_path):
    from PIL import Image
    rgb = (np.random.default_rng(3).uniform(0, 255, (30, 50, 3))).astype(np.uint8)
    p = str(tmp_path / "i.png")
    Image.fromarray(rgb).save(p)
    t, raw = io.load_image(p, 16, 24)
    assert t.shape == (1, 3, 16, 24) and raw == (50, 30) and -1.0 <= float(t.min()) and float(t.max()) <= 1.0
    same, _ = io.load_image(p, 30, 50)
    back = io.frames_to_uint8(same)
    assert np.array_equal(back[0].numpy(), rgb)                 # normalise / de-normalise is lossless at raw size
    up = io.frames_to_uint8(t, (30, 50))
    assert up.shape == (1, 30, 50, 3) and up.dtype == torch.uint8
    d = io.save_frames(up, str(tmp_path / "out"))
    assert sorted(os.listdir(d)) == ["000000.png"]


def test_prepare_motion_matches_script_arithmetic():
    # test_baseline_4eval_rawsize.py:173-184: flow *= [W/w*speed, W/h*speed]; nearest resize; *= frame/FRAME
    flow = torch.arange(2 * 3 * 4, dtype=torch.float32).view(1, 2, 3, 4)
    W, speed, frame, N = 8, 1.5, 40, 60
    ref = flow * torch.tensor([W / 4 * speed, W / 3 * speed]).view(1, 2, 1, 1)
    ref = torch.nn.functional.interpolate(ref, (W, W)) * frame / N
    out = pipeline.prepare_motion(flow, W, W, speed, frame, N)
    assert torch.allclose(out, ref)
