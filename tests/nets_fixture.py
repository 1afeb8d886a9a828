"""Deterministic state dicts / inputs of the network fixtures (tests/golden/nets_reference.npz).

The reference's networks have 0.36 - 3.7 M parameters each: too much to commit.  Instead every tensor of a
reference state dict is DEFINED by this module from its key name and shape (numpy Generator streams are stable
across platforms): tools/make_golden_nets.py (build container, needs /root/reference) constructs the reference's own
classes, overwrites their state dict with these values, runs the reference forward and stores the outputs together
with the key/shape list; the tests (CPU and GPU box, no reference) regenerate the same state dict, load it through
slr_sfs_amd.nets.load_reference_state_dict and compare.  Test infrastructure only."""
import zlib

import numpy as np
import torch

NETS = {  # name: (state-dict prefix in a checkpoint, input shape)
    "encoder": ("model.module.encoder.", (1, 3, 16, 24)),
    "projector": ("model.module.projector.", (1, 64, 32, 48)),
    "net_bg": ("model.module.net_bg.", (1, 3, 32, 48)),
    "net_alpha_encoder": ("model.module.net_alpha_encoder.", (1, 3, 16, 24)),
    "net_alpha_decoder": ("model.module.net_alpha_decoder.", (1, 65, 32, 48)),
}


def _rng(*what):
    return np.random.default_rng(zlib.crc32("/".join(str(w) for w in what).encode()))


def state_dict(net, keys, shapes):
    """keys / shapes of the reference module's state_dict() -> {key: tensor}, every value a function of (net, key)."""
    sd = {}
    shp = {k: tuple(int(v) for v in s if v >= 0) for k, s in zip(keys, shapes)}
    for k in keys:
        s, r = shp[k], _rng(net, k)
        if k.endswith("weight_orig") or k.endswith(".weight"):
            fan_in = int(np.prod(s[1:])) if len(s) > 1 else 1
            v = r.standard_normal(s) * np.sqrt(2.0 / max(fan_in, 1))
        elif k.endswith("weight_u") or k.endswith("weight_v"):
            continue                                       # second pass (they depend on weight_orig)
        elif k.endswith("stored_mean"):
            v = r.standard_normal(s) * 0.2
        elif k.endswith("stored_var"):
            v = r.uniform(0.5, 1.5, s)
        elif k.endswith("accumulation_counter") or k.endswith("num_batches_tracked"):
            v = np.ones(s)
        elif k.endswith("bias"):
            v = r.standard_normal(s) * 0.1
        else:
            v = r.standard_normal(s) * 0.1
        sd[k] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(s))
    for k in keys:                                         # legacy spectral norm: u random unit, v = normalize(W^T u)
        if k.endswith("weight_u"):
            base = k[:-len("weight_u")]
            w = sd[base + "weight_orig"].double().reshape(shp[base + "weight_orig"][0], -1).numpy()
            u = _rng(net, k).standard_normal(shp[k])
            u = u / np.linalg.norm(u)
            v = w.T @ u
            v = v / np.linalg.norm(v)
            sd[k] = torch.from_numpy(u.astype(np.float32))
            sd[base + "weight_v"] = torch.from_numpy(v.astype(np.float32))
    missing = [k for k in keys if k not in sd]
    assert not missing, missing
    return sd


def net_input(net):
    shape = NETS[net][1]
    r = _rng(net, "input")
    if shape[1] == 3:
        return torch.from_numpy(r.uniform(-1, 1, shape).astype(np.float32))
    x = r.standard_normal(shape).astype(np.float32)        # decoder inputs: a hole, a single-channel zero, a zero column
    x[:, :, 5:20, 10:30] = 0.0
    x[:, 3, 0, 0] = 0.0
    x[:, :, :, 47] = 0.0
    return torch.from_numpy(x)


def net_input_large(net, S=256):
    """Inputs of the 256 x 256 digests (tests/golden/large_nets_e2e.npz): like net_input, with a hole, a zero column and a
    zero row band (the partial convolutions' masks matter on several tiles of the convolution kernels)."""
    cin = NETS[net][1][1]
    r = _rng(net, "input_large", S)
    if cin == 3:
        return torch.from_numpy(r.uniform(-1, 1, (1, 3, S, S)).astype(np.float32))
    x = r.standard_normal((1, cin, S, S)).astype(np.float32)
    x[:, :, 40:90, 100:190] = 0.0
    x[:, 3, 0, 0] = 0.0
    x[:, :, :, S - 1] = 0.0
    x[:, :, 200:203, :] = 0.0
    return torch.from_numpy(x)


def digest_positions(tag, size, n=4096):
    return _rng("digest", tag, size).integers(0, size, n)


def e2e_inputs(W=64, N=8):
    """Inputs of the end-to-end fixture (tests/golden/pipeline_e2e.npz): an image in [-1, 1] and a motion field
    (px / frame at the working resolution, zero outside the 'fluid' region like the data set's masks), seeded."""
    r = _rng("e2e", W, N)
    img = r.uniform(-1, 1, (1, 3, W, W)).astype(np.float32)
    y, x = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    u = 1.2 * np.sin(2 * np.pi * (1.5 * x / W + y / W) + 0.5)
    v = 1.2 * np.cos(2 * np.pi * (x / W - 1.2 * y / W) + 1.3)
    m = ((x >= 0.3 * W) & (y >= 0.2 * W)).astype(np.float32)
    motion = np.stack([u * m, v * m])[None].astype(np.float32)
    return img, motion, N
