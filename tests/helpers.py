"""shared test helpers: seeded parameters as torch tensors, probes, golden loading."""
import os

import numpy as np
import torch

from xggm_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_cfg(g):
    return {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}


def seeded_params(shapes, seed, dtype=torch.float32, device="cpu"):
    return {k: torch.from_numpy(synth.seeded_param(k, s, seed)).to(device=device, dtype=dtype)
            for k, s in shapes.items()}


def probe(name, shape, seed, dtype=torch.float32, device="cpu"):
    v = synth._rng(seed, "probe:" + name).standard_normal(tuple(shape), dtype=np.float32)
    return torch.from_numpy(v).to(device=device, dtype=dtype)


def batch_tensors(b, device="cpu", dtype=torch.float32):
    out = {}
    for k, v in b.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).to(device)
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
