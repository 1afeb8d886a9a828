"""Input / output formats either side of the path: what the reference's test
scripts read and write (test_animating/test_baseline_4eval_rawsize.py:51-68,156-184,246-274).
torch + numpy + PIL only (cv2 / torchvision / lz4framed are not in the image; the LZ4 frame decoder is here)."""
import json
import os
import pickle

import numpy as np
import torch
import torch.nn.functional as F

FLO_MAGIC = 202021.25


def read_flo(path):
    """Middlebury .flo -> float32 [H,W,2] (utils/utils.py:252-261, test_baseline_4eval_rawsize.py:51-61)."""
    with open(path, "rb") as f:
        buf = f.read()
    assert np.frombuffer(buf, np.float32, 1, 0)[0] == np.float32(FLO_MAGIC), "not a .flo file"
    w = int(np.frombuffer(buf, np.int32, 1, 4)[0])
    h = int(np.frombuffer(buf, np.int32, 1, 8)[0])
    return np.frombuffer(buf, np.float32, h * w * 2, 12).reshape(h, w, 2).copy()


def write_flo(path, flow_hw2):
    flow_hw2 = np.ascontiguousarray(flow_hw2, dtype=np.float32)
    h, w, c = flow_hw2.shape
    assert c == 2
    with open(path, "wb") as f:
        np.array([FLO_MAGIC], np.float32).tofile(f)
        np.array([w, h], np.int32).tofile(f)
        flow_hw2.tofile(f)


def lz4_frame_decompress(raw):
    """LZ4 *frame* format -> bytes (what ``lz4framed.decompress`` does in the reference's
    ``load_compressed_tensor``, utils/utils.py:111-115; the package is not in the MI355X image).
    Written from the public frame / block format description: magic 0x184D2204, FLG / BD descriptor
    (+ optional content size and dictionary id, header checksum byte), data blocks of
    [u32 size | bit 31 = stored uncompressed][payload][optional block checksum], end mark 0,
    optional content checksum.  Checksums are skipped, not verified.  A block is a series of
    sequences: token (literal length << 4 | match length - 4), 255-extended lengths, literals,
    u16 little-endian match offset into the output produced so far (copies may overlap)."""
    mv = memoryview(raw)
    if len(mv) < 7 or int.from_bytes(mv[0:4], "little") != 0x184D2204:
        raise ValueError("not an LZ4 frame")
    flg = mv[4]
    if (flg >> 6) != 1:
        raise ValueError("unsupported LZ4 frame version")
    block_checksum, has_size, content_checksum, has_dict = (flg >> 4) & 1, (flg >> 3) & 1, (flg >> 2) & 1, flg & 1
    pos = 6 + 8 * has_size + 4 * has_dict + 1                # FLG, BD, [size], [dict id], header checksum
    expected = int.from_bytes(mv[6:14], "little") if has_size else None
    out = bytearray()
    while True:
        if pos + 4 > len(mv):
            raise ValueError("truncated LZ4 frame")
        bsz = int.from_bytes(mv[pos:pos + 4], "little")
        pos += 4
        if bsz == 0:                                          # end mark
            break
        stored = bsz >> 31
        bsz &= 0x7FFFFFFF
        if pos + bsz > len(mv):
            raise ValueError("truncated LZ4 block")
        if stored:
            out += mv[pos:pos + bsz]
        else:
            _lz4_block(mv[pos:pos + bsz], out)
        pos += bsz + 4 * block_checksum
    if expected is not None and expected != len(out):
        raise ValueError(f"LZ4 frame: content size {expected} != decoded {len(out)}")
    return bytes(out)


def _lz4_block(src, out):
    i, n = 0, len(src)
    while i < n:
        token = src[i]
        i += 1
        lit = token >> 4
        if lit == 15:
            while True:
                b = src[i]
                i += 1
                lit += b
                if b != 255:
                    break
        out += src[i:i + lit]
        i += lit
        if i >= n:                                            # the last sequence has literals only
            break
        off = src[i] | (src[i + 1] << 8)
        i += 2
        if off == 0 or off > len(out):
            raise ValueError("corrupt LZ4 block (bad match offset)")
        mlen = (token & 15) + 4
        if (token & 15) == 15:
            while True:
                b = src[i]
                i += 1
                mlen += b
                if b != 255:
                    break
        start = len(out) - off
        if off >= mlen:
            out += out[start:start + mlen]
        else:                                                 # overlapping copy: the pattern repeats
            pat = bytes(out[start:])
            out += (pat * (mlen // off + 1))[:mlen]


def load_motion(path):
    """Motion field as the test scripts load it -> float32 tensor [1,2,h,w] (:168-172).
    .flo, or .pth: the reference's LZ4-frame-compressed pickle of a numpy array
    (``load_compressed_tensor``, utils/utils.py:111-115), decoded here without the lz4framed package;
    a plain torch.save / pickle of the same array is accepted as well.
    (Like the reference, this unpickles the file: only load motion files you trust.)"""
    if path.endswith(".flo"):
        return torch.from_numpy(read_flo(path)).permute(2, 0, 1).contiguous().unsqueeze(0)
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:4] == b"\x04\x22\x4d\x18":
        arr = pickle.loads(lz4_frame_decompress(raw))
    else:
        arr = torch.load(path, map_location="cpu", weights_only=False)
    t = torch.as_tensor(np.asarray(arr), dtype=torch.float32)
    return t if t.dim() == 4 else t.unsqueeze(0)


def load_image(path, H, W):
    """PIL image -> ([1,3,H,W] in [-1,1], (raw_W, raw_H)) -- Resize + ToTensor + Normalize(.5,.5)
    (:156-163; the reference resizes to a square W x W grid, this build also takes H != W)."""
    from PIL import Image
    img = Image.open(path).convert("RGB")
    raw = img.size
    img = img.resize((W, H), Image.BILINEAR)               # torchvision.transforms.Resize default
    t = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
    return ((t - 0.5) / 0.5).unsqueeze(0).contiguous(), raw


def frames_to_uint8(frames, raw_hw=None):
    """[n,3,H,W] in [-1,1] -> uint8 [n,h,w,3] RGB: bilinear resize to the raw size, *0.5+0.5, *255,
    saturating round (what cv2.imwrite does to the float image, :246-249,266-274)."""
    if raw_hw is not None and tuple(frames.shape[2:]) != tuple(raw_hw):
        frames = F.interpolate(frames, raw_hw, mode="bilinear")
    x = (frames.permute(0, 2, 3, 1) * 0.5 + 0.5) * 255.0
    return torch.clamp(torch.round(x), 0, 255).to(torch.uint8)


def alpha_to_uint8(alpha, raw_hw=None):
    """[n,1,H,W] in [0,1] -> uint8 [n,h,w] grey: bilinear resize, *255 (test_v1_4eval_rawsize.py:250-252,279-280)."""
    if raw_hw is not None and tuple(alpha.shape[2:]) != tuple(raw_hw):
        alpha = F.interpolate(alpha, raw_hw, mode="bilinear")
    return torch.clamp(torch.round(alpha[:, 0] * 255.0), 0, 255).to(torch.uint8)


def save_image(img_u8, path):
    """uint8 [h,w,3] or [h,w] -> one PNG."""
    from PIL import Image
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    Image.fromarray(img_u8.cpu().numpy()).save(path)
    return path


def save_frames(frames_u8, out_dir, key="PredImg"):
    """uint8 [n,h,w,3] (or [n,h,w] grey) -> out_dir/key/%06d.png (:252-274)."""
    from PIL import Image
    d = os.path.join(out_dir, key)
    os.makedirs(d, exist_ok=True)
    arr = frames_u8.cpu().numpy()
    for t in range(arr.shape[0]):
        Image.fromarray(arr[t]).save(os.path.join(d, "%06d.png" % t))
    return d


def speed_align(align_json, name):
    """Per-scene frame count of the speed-alignment tables (data/CLAW/CLAW_align_max_frame_001_max600.json,
    test_baseline_4eval_rawsize.py:222-226); None when no table is given."""
    if not align_json or align_json == "None":
        return None
    with open(align_json) as f:
        return json.load(f)[name]


def encode_video(frame_dir, out_path, framerate=30):
    """%06d.png -> mp4 / gif with the command line of the reference's test scripts
    (test_baseline_4eval_rawsize.py:289: ``ffmpeg -loglevel quiet -framerate 30 -i DIR/%06d.png -framerate 30 OUT -y``).
    Returns ``out_path``, or None when there is no ``ffmpeg`` on PATH (the reference's os.system call then fails
    silently; here the caller is told)."""
    import shutil
    import subprocess
    exe = shutil.which("ffmpeg")
    if exe is None:
        return None
    subprocess.check_call([exe, "-loglevel", "quiet", "-framerate", str(framerate), "-i", os.path.join(frame_dir, "%06d.png"),
                           "-framerate", str(framerate), out_path, "-y"])
    return out_path

