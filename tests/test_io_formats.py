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
    # SLR-v1 runner outputs (test_v1_4eval_rawsize.py:250-252,279-284): grey alpha frames, one background image
    alpha = torch.linspace(0, 1, 2 * 16 * 24).view(2, 1, 16, 24)
    a8 = io.alpha_to_uint8(alpha, (30, 50))
    assert a8.shape == (2, 30, 50) and a8.dtype == torch.uint8 and int(a8.max()) == 255 and int(a8.min()) == 0
    da = io.save_frames(a8, str(tmp_path / "out"), key="CompositeFluidAlpha")
    assert sorted(os.listdir(da)) == ["000000.png", "000001.png"]
    assert Image.open(os.path.join(da, "000001.png")).mode == "L"
    pb = io.save_image(up[0], str(tmp_path / "out" / "BGImg.png"))
    assert np.array_equal(np.asarray(Image.open(pb)), up[0].numpy())


def test_prepare_motion_matches_script_arithmetic():
    # test_baseline_4eval_rawsize.py:173-184: flow *= [W/w*speed, W/h*speed]; nearest resize; *= frame/FRAME
    flow = torch.arange(2 * 3 * 4, dtype=torch.float32).view(1, 2, 3, 4)
    W, speed, frame, N = 8, 1.5, 40, 60
    ref = flow * torch.tensor([W / 4 * speed, W / 3 * speed]).view(1, 2, 1, 1)
    ref = torch.nn.functional.interpolate(ref, (W, W)) * frame / N
    out = pipeline.prepare_motion(flow, W, W, speed, frame, N)
    assert torch.allclose(out, ref)


# ------------------------------------------------------------------ LZ4-framed .pth (utils/utils.py:111-115)

def _lz4_block_encode(data):
    """Tiny greedy LZ4 block encoder (test helper): hash of 4-byte windows, min match 4, last 5 bytes literal."""
    out, i, anchor, n, table = bytearray(), 0, 0, len(data), {}

    def emit(lit, mlen, off):
        token_l = min(len(lit), 15)
        token_m = min(mlen - 4, 15) if mlen else 0
        out.append((token_l << 4) | token_m)
        if len(lit) >= 15:
            r = len(lit) - 15
            out.extend(b"\xff" * (r // 255) + bytes([r % 255]))
        out.extend(lit)
        if mlen:
            out.extend(off.to_bytes(2, "little"))
            if mlen - 4 >= 15:
                r = mlen - 4 - 15
                out.extend(b"\xff" * (r // 255) + bytes([r % 255]))

    while i + 4 <= n - 5:
        key = bytes(data[i:i + 4])
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand <= 0xFFFF:
            m = 4
            while i + m < n - 5 and data[cand + m] == data[i + m]:
                m += 1
            emit(data[anchor:i], m, i - cand)
            i += m
            anchor = i
        else:
            i += 1
    emit(data[anchor:], 0, 0)
    return bytes(out)


def _lz4_frame(data, block=1 << 16, stored=False, content_size=True, block_checksum=False):
    flg = (1 << 6) | (1 << 5) | (int(block_checksum) << 4) | (int(content_size) << 3)
    hdr = bytearray(b"\x04\x22\x4d\x18") + bytes([flg, 4 << 4])
    if content_size:
        hdr += len(data).to_bytes(8, "little")
    hdr += b"\x00"                                            # header checksum (not verified by the reader)
    body = bytearray()
    for s in range(0, len(data), block):
        chunk = data[s:s + block]
        enc = chunk if stored else _lz4_block_encode(chunk)
        body += (len(enc) | (0x80000000 if stored else 0)).to_bytes(4, "little") + enc
        if block_checksum:
            body += b"\x00\x00\x00\x00"
    return bytes(hdr + body + b"\x00\x00\x00\x00")


@pytest.mark.parametrize("kind", ["random", "runs", "zeros", "text"])
@pytest.mark.parametrize("opts", [{}, {"stored": True}, {"content_size": False, "block_checksum": True}, {"block": 1000}])
def test_lz4_frame_decoder(kind, opts):
    rng = np.random.default_rng(5)
    data = {"random": rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
            "runs": np.repeat(rng.integers(0, 256, 300, dtype=np.uint8), rng.integers(1, 700, 300)).tobytes(),
            "zeros": bytes(100000),
            "text": (b"the quick brown fox jumps over the lazy dog. " * 3000)[:123457]}[kind]
    frame = _lz4_frame(data, **opts)
    if kind in ("zeros", "text") and not opts.get("stored"):
        assert len(frame) < len(data) // 4                    # the helper really compresses: matches are exercised
    assert io.lz4_frame_decompress(frame) == data


def test_lz4_framed_pth_motion(tmp_path):
    """The reference's load_compressed_tensor format: LZ4 frame around pickle.dumps(ndarray)."""
    import pickle
    rng = np.random.default_rng(9)
    flow = np.round(rng.standard_normal((1, 2, 40, 64)), 1).astype(np.float32)
    p = str(tmp_path / "motion.pth")
    open(p, "wb").write(_lz4_frame(pickle.dumps(flow)))
    got = io.load_motion(p)
    assert got.shape == (1, 2, 40, 64) and np.array_equal(got.numpy(), flow)
    with pytest.raises(ValueError):
        io.lz4_frame_decompress(b"\x04\x22\x4d\x18\x60\x40\x00" + b"\x10\x00\x00\x00" + b"\x00" * 3)   # truncated block


def test_lz4_frame_from_system_liblz4(golden_dir):
    """A frame written by the system liblz4 (tools/make_golden_lz4.py), i.e. by an implementation independent of
    the reader: decodes to the pickled motion field the generator compressed."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(golden_dir), "..", "tools", "make_golden_lz4.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)                                # (liblz4 is only loaded when the tool runs as a script)
    want = mk.motion()
    got = io.load_motion(os.path.join(golden_dir, "motion_lz4.pth"))
    assert got.shape == want.shape and np.array_equal(got.numpy(), want)


def test_encode_video_command_line(tmp_path, monkeypatch):
    """The ffmpeg step of the reference's test scripts (test_baseline_4eval_rawsize.py:289): same arguments; None
    when ffmpeg is not installed (as in this image)."""
    import shutil
    import subprocess
    from slr_sfs_amd import io
    monkeypatch.setattr(shutil, "which", lambda name: None)
    assert io.encode_video(str(tmp_path), str(tmp_path / "x.mp4")) is None
    calls = []
    monkeypatch.setattr(shutil, "which", lambda name: "/usr/bin/ffmpeg")
    monkeypatch.setattr(subprocess, "check_call", lambda cmd: calls.append(cmd))
    assert io.encode_video("/d/PredImg", "/d/out.mp4") == "/d/out.mp4"
    assert calls == [["/usr/bin/ffmpeg", "-loglevel", "quiet", "-framerate", "30", "-i", "/d/PredImg/%06d.png",
                      "-framerate", "30", "/d/out.mp4", "-y"]]

