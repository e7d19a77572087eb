"""Seeded synthetic weights and VQA batches (numpy only, platform independent).

There is no dataset, vocabulary or LXMERT snapshot offline, so every test, the golden
generator and ``bench.py`` draw inputs from this one recipe:
identical tensors here, on the GPU box and inside the reference when the goldens
were generated.
"""
import zlib

import numpy as np


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def seeded_param(name, shape, seed=0):
    """Deterministic value for a parameter called ``name`` (reference state_dict key).

    2-D weights ~ N(0, 0.02) (0.05 for the GAT layers), 1-D ``*.weight`` (LayerNorm
    gains) ~ 1 + 0.1 N(0,1), biases ~ 0.05 N(0,1), GIN ``eps`` ~ 0.1 N(0,1).  Non-trivial
    gains/biases make the parity checks sensitive to every term.
    """
    r = _rng(seed, name)
    shape = tuple(int(s) for s in shape)
    z = r.standard_normal(shape, dtype=np.float32)
    if name.endswith("eps"):
        return 0.1 * z
    if len(shape) >= 2:
        return (0.05 if "gat_layers" in name else 0.02) * z
    if name.endswith("weight"):
        return (1.0 + 0.1 * z).astype(np.float32)
    return (0.05 * z).astype(np.float32)


def seeded_state(named_shapes, seed=0):
    """``{name: shape}`` -> ``{name: float32 ndarray}``."""
    return {k: seeded_param(k, s, seed) for k, s in named_shapes.items()}


def vqa_batch(B, A=2274, N=36, T=20, F=2048, vocab=30522, seed=0):
    """One synthetic VQA batch shaped like the reference loader's output
    (src/vqa/vqacpv2_data.py:95-127) after host tokenisation (src/lxrt/entry.py:37-72).

    feats ~ U[0,3) [B,N,F]; boxes ~ U[0,1) [B,N,4]; ``[CLS] ids [SEP]`` padded to T with
    lengths U{5..T-1}; one-hot target over A; adj_true = (triu(U)+triu(U)^T)/max,
    mirroring data/preprocess/vqa/compute_adjacency.py:38-45,90; standard-normal draws
    for the two denoising branches.
    """
    r = _rng(seed, "vqa_batch")
    feats = (3.0 * r.random((B, N, F), dtype=np.float32)).astype(np.float32)
    boxes = r.random((B, N, 4), dtype=np.float32)
    ids = np.zeros((B, T), dtype=np.int64)
    mask = np.zeros((B, T), dtype=np.int64)
    lo = min(1000, vocab // 2)
    for b in range(B):
        L = int(r.integers(5, T))  # 5..T-1 tokens incl. [CLS]/[SEP]
        body = r.integers(lo, vocab, size=L - 2)
        ids[b, 0] = min(101, vocab - 2)
        ids[b, 1:L - 1] = body
        ids[b, L - 1] = min(102, vocab - 1)
        mask[b, :L] = 1
    seg = np.zeros((B, T), dtype=np.int64)
    target = np.zeros((B, A), dtype=np.float32)
    target[np.arange(B), r.integers(0, A, size=B)] = 1.0
    u = np.triu(r.random((B, N, N), dtype=np.float32))
    a = u + np.transpose(u, (0, 2, 1))
    a = (a / a.max(axis=(1, 2), keepdims=True)).astype(np.float32)
    randn_adj = r.standard_normal((B, N, N), dtype=np.float32)
    return dict(feats=feats, boxes=boxes, input_ids=ids, input_mask=mask, segment_ids=seg,
                target=target, adj_true=a, randn_adj=randn_adj)


def randn_nodes(B, N, H, seed=0):
    """standard-normal draw for the node branch (size depends on the hidden width)."""
    return _rng(seed, "randn_node").standard_normal((B, N, H), dtype=np.float32)


def generator_inputs(tag, kind, B, N, H, seed):
    """node features ~ N(0,1) and a noisy symmetric adjacency (zero diagonal; GAT gets
    ~30% exact zeros to exercise its ``adj == 0`` mask) for the generator parity cases."""
    r = _rng(seed, "gen_case:" + tag)
    x = r.standard_normal((B, N, H), dtype=np.float32)
    u = np.triu(r.random((B, N, N), dtype=np.float32), 1)
    a = u + u.transpose(0, 2, 1) + 0.3 * np.triu(r.standard_normal((B, N, N), dtype=np.float32), 1)
    if kind == "GAT":
        a = a * (r.random((B, N, N)) > 0.3)
    return x, np.ascontiguousarray(a.astype(np.float32))
