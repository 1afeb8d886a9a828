"""Input / output formats either side of the path: what the reference's test
scripts read and write (test_animating/test_baseline_4eval_rawsize.py:51-68,156-184,246-274).
torch + numpy + PIL only (cv2 / torchvision / lz4framed are not in the image)."""
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


def load_motion(path):
    """Motion field as the test scripts load it -> float32 tensor [1,2,h,w] (:168-172).
    .flo, or .pth: the reference's lz4framed-compressed pickle (utils/utils.py:111-115) when
    lz4framed is importable, else a plain torch.save / pickle of the same array."""
    if path.endswith(".flo"):
        return torch.from_numpy(read_flo(path)).permute(2, 0, 1).contiguous().unsqueeze(0)
    with open(path, "rb") as f:
        raw = f.read()
    try:
        import lz4framed                                   # not in the MI355X image
        arr = pickle.loads(lz4framed.decompress(raw))
    except ImportError:
        try:
            arr = torch.load(path, map_location="cpu", weights_only=False)
        except Exception as e:                             # pragma: no cover
            raise RuntimeError(f"{path}: lz4framed-compressed motion needs the lz4framed package") from e
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


def save_frames(frames_u8, out_dir, key="PredImg"):
    """uint8 [n,h,w,3] -> out_dir/key/%06d.png (:252-274)."""
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
