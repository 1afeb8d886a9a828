"""ctypes binding of libslrsplat.so (C ABI: include/slr_splat.h).  No fallback of any kind."""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLR_SFS_AMD_LIB") or os.path.join(_HERE, "lib", "libslrsplat.so")   # env: dev only
ABI_VERSION = 10
WS_PREBINNED, WS_CLEAN = 1, 2       # include/slr_splat.h: flags of the `prebinned` argument

# every symbol include/slr_splat.h declares
SYMBOLS = (
    "slr_abi_version", "slr_last_error", "slr_splat_time_next",
    "slr_euler_integrate", "slr_euler_integrate_all", "slr_euler_backward", "slr_euler_integrate_batch", "slr_euler_backward_batch",
    "slr_splat_workspace_bytes", "slr_splat_workspace_init", "slr_splat_bin", "slr_splat_bin_pair", "slr_splat_set_scan_max_tiles",
    "slr_splat_set_front_end",
    "slr_splat_set_scan_shape",
    "slr_softsplat_forward", "slr_softsplat_mode_forward", "slr_splat_normalize",
    "slr_synth_group", "slr_global_max",
    "slr_clip_plan_bytes", "slr_clip_plan_totals", "slr_clip_plan_build", "slr_synth_group_clip",
    "slr_synth_group_clip_batch", "slr_synth_two_groups_clip_batch", "slr_pack_planes4",
    "slr_softsplat_backward", "slr_softsplat_backward_ws_bytes", "slr_softsplat_backward_ws", "slr_maxsplat_forward", "slr_max_warp_norm",
    "slr_bn_relu_mask", "slr_pconv_epilogue", "slr_conv_saturation_count", "slr_conv_saturation_record",
    "slr_conv3x3_weight_bytes", "slr_conv3x3_split_weights", "slr_conv3x3_f32_weights", "slr_conv3x3_wino_weight_bytes", "slr_conv3x3_wino_weights", "slr_conv3x3_forward", "slr_pconv3x3_forward",
    "slr_conv3x3_forward_skip", "slr_pconv3x3_forward_skip", "slr_conv_pool_ws_bytes", "slr_conv_up_ws_bytes", "slr_conv3x3_forward_skipout", "slr_pconv3x3_forward_skipout",
    "slr_conv1x1_weight_bytes", "slr_conv1x1_split_weights", "slr_conv1x1_f32_weights", "slr_conv1x1_forward",
    "slr_avgpool3x3s2", "slr_upsample_bilinear2x", "slr_conv1x1_small",
)

_lib = None
_lock = threading.Lock()


def build(verbose=False):
    """hipcc --offload-arch=gfx950 build of the library (cross-compiles without a GPU)."""
    import subprocess
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc")], stdout=out)
    return LIB_PATH


def lib():
    """The loaded library; raises RuntimeError (loudly) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"slr_sfs_amd: HIP library {LIB_PATH} is missing -- build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C slr-sfs_amd/csrc`). "
                f"There is no CPU/PyTorch fallback.")
        L = ctypes.CDLL(LIB_PATH)
        vp, fp, i, f, sz = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
        L.slr_abi_version.restype = i
        L.slr_last_error.restype = ctypes.c_char_p
        L.slr_splat_time_next.restype = None
        L.slr_splat_time_next.argtypes = [vp, vp]
        L.slr_splat_set_scan_max_tiles.restype = i
        L.slr_splat_set_scan_max_tiles.argtypes = [i]
        L.slr_splat_set_front_end.restype = i
        L.slr_splat_set_front_end.argtypes = [i]
        L.slr_splat_set_scan_shape.restype = None
        L.slr_splat_set_scan_shape.argtypes = [i, i, i, i]
        L.slr_splat_workspace_bytes.restype = sz
        L.slr_splat_workspace_bytes.argtypes = [i, i, i]
        L.slr_softsplat_backward_ws_bytes.restype = sz
        L.slr_softsplat_backward_ws_bytes.argtypes = [i, i, i, i]
        L.slr_clip_plan_bytes.restype = sz
        L.slr_clip_plan_bytes.argtypes = [i, i, i]
        L.slr_conv3x3_weight_bytes.restype = sz
        L.slr_conv3x3_weight_bytes.argtypes = [i, i]
        L.slr_conv3x3_wino_weight_bytes.restype = sz
        L.slr_conv3x3_wino_weight_bytes.argtypes = [i, i]
        L.slr_conv1x1_weight_bytes.restype = sz
        L.slr_conv1x1_weight_bytes.argtypes = [i, i]
        L.slr_conv_pool_ws_bytes.restype = sz
        L.slr_conv_pool_ws_bytes.argtypes = [i, i, i, i]
        L.slr_conv_up_ws_bytes.restype = sz
        L.slr_conv_up_ws_bytes.argtypes = [i, i, i, i]
        sig = {
            "slr_euler_integrate": [fp, i, i, i, f, fp, fp, vp],
            "slr_euler_integrate_all": [fp, i, i, i, f, fp, fp, vp],
            "slr_euler_backward": [fp, i, i, i, f, fp, fp, vp],
            "slr_euler_integrate_batch": [fp, vp, i, i, i, f, fp, fp, vp],
            "slr_euler_backward_batch": [fp, vp, i, i, i, f, fp, fp, vp],
            "slr_splat_workspace_init": [vp, sz, i, i, i, vp],
            "slr_splat_bin": [fp, i, i, i, vp, sz, vp],
            "slr_splat_bin_pair": [fp, fp, i, i, i, vp, vp, sz, vp],
            "slr_softsplat_forward": [fp, fp, fp, i, i, i, i, vp, sz, i, vp],
            "slr_softsplat_mode_forward": [fp, fp, fp, fp, i, i, i, i, i, vp, sz, i, vp],
            "slr_splat_normalize": [fp, fp, i, i, i, i, i, f, vp],
            "slr_synth_group": [fp, fp, fp, i, fp, fp, f, fp, fp, i, i, i, f, vp, vp, sz, vp],
            "slr_global_max": [fp, sz, fp, fp, vp],
            "slr_clip_plan_totals": [i, i, i, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)],
            "slr_clip_plan_build": [fp, vp, fp, vp, i, i, i, vp, sz, vp],
            "slr_synth_group_clip": [fp, fp, fp, i, fp, fp, f, fp, fp, i, i, i, f, vp, sz, i, i, i, vp],
            "slr_synth_group_clip_batch": [fp, fp, fp, i, vp, vp, vp, vp, vp, i, i, i, f, vp, sz, i, vp, i, vp, vp],
            "slr_pack_planes4": [fp, fp, i, i, i, i, vp],
            "slr_synth_two_groups_clip_batch": [fp, fp, fp, i, fp, fp, i, vp, vp, vp, vp, vp, i, i, i, f, vp, sz, i, vp, i, vp, vp],
            "slr_softsplat_backward": [fp, fp, fp, fp, fp, i, i, i, i, vp],
            "slr_softsplat_backward_ws": [fp, fp, fp, fp, fp, i, i, i, i, vp, sz, vp],
            "slr_maxsplat_forward": [fp, fp, fp, f, i, i, i, i, vp, sz, i, vp],
            "slr_max_warp_norm": [fp, fp, fp, fp, i, i, i, i, vp, sz, i, vp],
            "slr_bn_relu_mask": [fp, fp, fp, fp, i, fp, i, i, i, i, vp],
            "slr_pconv_epilogue": [fp, fp, fp, f, fp, fp, fp, fp, fp, f, i, i, i, i, vp],
            "slr_conv_saturation_count": [ctypes.POINTER(ctypes.c_ulonglong), i, vp],
            "slr_conv_saturation_record": [vp, vp],
            "slr_conv3x3_split_weights": [fp, vp, i, i, f, vp],
            "slr_conv1x1_split_weights": [fp, vp, i, i, f, vp],
            "slr_conv3x3_f32_weights": [fp, vp, i, i, vp],
            "slr_conv3x3_wino_weights": [fp, vp, i, i, vp],
            "slr_conv1x1_f32_weights": [fp, vp, i, i, vp],
            "slr_conv1x1_forward": [fp, vp, fp, fp, i, i, i, i, i, f, f, i, vp],
            "slr_conv3x3_forward": [fp, vp, fp, fp, fp, i, i, i, i, i, f, f, fp, fp, i, vp],
            "slr_pconv3x3_forward": [fp, fp, fp, fp, vp, f, f, fp, fp, fp, fp, fp, fp, i, i, i, i, i, i, vp],
            "slr_conv3x3_forward_skip": [fp, vp, fp, fp, i, i, i, i, i, f, f, fp, fp, fp, vp, fp, i, f, vp, sz, i, vp],
            "slr_pconv3x3_forward_skip": [fp, fp, fp, fp, vp, f, f, fp, fp, fp, i, i, i, i, i, fp, vp, i, f, vp, sz, i, vp],
            "slr_conv3x3_forward_skipout": [fp, vp, fp, fp, fp, i, i, i, i, i, f, f, fp, fp, fp, fp, fp, i, vp],
            "slr_pconv3x3_forward_skipout": [fp, fp, fp, fp, vp, f, f, fp, fp, fp, fp, fp, fp, i, i, i, i, i, fp, fp, i, vp],
            "slr_avgpool3x3s2": [fp, fp, i, i, i, i, i, vp],
            "slr_upsample_bilinear2x": [fp, fp, i, i, i, i, i, vp],
            "slr_conv1x1_small": [fp, fp, fp, fp, i, i, i, i, i, i, vp],
        }
        for name, argtypes in sig.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = i
        if L.slr_abi_version() != ABI_VERSION:
            raise RuntimeError(f"slr_sfs_amd: {LIB_PATH} has ABI {L.slr_abi_version()}, expected {ABI_VERSION}")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().slr_last_error()
        # A failed call may have left the counters of its workspace dirty (a launch error in flight): the calls that follow say
        # SLR_WS_CLEAN and would trust them.  Drop every cached workspace -- the next call gets a freshly initialised one.
        if rc > 0:                                       # (a hipError_t; argument errors, rc < 0, launch nothing)
            clear_workspaces()
        raise RuntimeError(f"slr_sfs_amd: {what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_of(t):
    """HIP stream handle torch is currently using on the tensor's device."""
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_device(*tensors):
    """The reference raises NotImplementedError for CPU tensors (models/softsplat.py:418-419)
    and asserts contiguity (:401-402); same here -- there is no CPU path."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NotImplementedError("slr_sfs_amd operators run on ROCm device tensors only (no CPU path)")
        if t.dtype != torch.float32:
            raise TypeError(f"slr_sfs_amd: float32 tensors required, got {t.dtype}")
        assert t.is_contiguous() is True


# ---- scratch: caller-owned workspaces, cached per (device, stream, role, shape), least-recently-used eviction ----
# A workspace is tied to the stream it was first used on (two streams must never share one), so the stream handle is
# part of the key; the cache is bounded -- a service that animates many resolutions or creates and destroys streams
# would otherwise pin ~300 MB of HBM per (stream, shape) for ever.  Evicted tensors go back to torch's stream-ordered
# caching allocator (safe: they were only ever used on the stream they were allocated on).
# HIP graphs: a workspace that was handed out while its stream was CAPTURING has its address baked into the captured
# graph, whose replays nobody can see from here -- such workspaces are pinned: never evicted (clear_workspaces(
# include_captured=True) drops them once the graphs that use them are gone), and nothing is evicted during a capture.
import collections

WS_CACHE_MAX = max(1, int(os.environ.get("SLR_SFS_AMD_WS_CACHE", "8")))
_ws_cache = collections.OrderedDict()
_ws_captured = {}


def workspace(t, role, N, C, H, W, nbytes=None):
    """A torch-allocated (stream-ordered) workspace for splatting [N,<=C,H,W] on t's device."""
    stream = torch.cuda.current_stream(t.device)
    key = (t.device.index, stream.cuda_stream, role, N, C, H, W, nbytes)
    ws = _ws_captured.get(key)
    if ws is not None:
        return ws
    with torch.cuda.device(t.device):                    # (the capture status of t's device, not of the current one)
        capturing = torch.cuda.is_current_stream_capturing()
    ws = _ws_cache.get(key)
    if ws is None:
        if nbytes is None:
            nbytes = int(lib().slr_splat_workspace_bytes(N, H, W))
        while not capturing and len(_ws_cache) >= WS_CACHE_MAX:
            _ws_cache.popitem(last=False)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=t.device)
        # a splat workspace starts zeroed; the kernels leave its counters zero again, so the calls may say WS_CLEAN (no zero kernel)
        with torch.cuda.device(t.device):
            check(lib().slr_splat_workspace_init(ptr(ws), ws.numel(), N, H, W, ctypes.c_void_p(stream.cuda_stream)), "slr_splat_workspace_init")
        _ws_cache[key] = ws
    else:
        _ws_cache.move_to_end(key)
    if capturing:                                       # its address is now part of a graph: keep it for good
        _ws_captured[key] = _ws_cache.pop(key)
    return ws


def clear_workspaces(include_captured=False):
    _ws_cache.clear()
    if include_captured:
        _ws_captured.clear()
