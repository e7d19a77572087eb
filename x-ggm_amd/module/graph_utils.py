"""Noise injection of the denoising-score-matching step (src/module/graph_utils.py:144-168).

Both functions keep the reference signature ``(tensor, sigma) -> (noisy, grad_log_noise)``.
The draw comes from the in-kernel Philox generator unless ``randn`` supplies it (parity
tests feed the oracle and the HIP path the same draw)."""
import itertools

from .. import functional as XF
from .. import ops

_rng_state = {}
_sid = itertools.count(1)


def _rng_for(t):
    st = _rng_state.get(t.device)
    if st is None:
        st = _rng_state[t.device] = ops.make_rng(9595, t.device)
    return st


def manual_seed(seed, device):
    _rng_state[device] = ops.make_rng(seed, device)


def advance(device):
    """new draws for the next call (one kernel; graph capturable)."""
    st = _rng_state.get(device)
    if st is not None:
        ops.rng_advance(st, 1)


def add_edge_noise_v2(adjs, sigma=0.2, randn=None):
    """noise = triu(randn,1)*sigma mirrored; returns (adjs + noise, -noise/sigma^2).
    ``adjs`` fp32 [B,N,N].  ref: src/module/graph_utils.py:162-168"""
    B, N, _ = adjs.shape
    rng = None if randn is not None else _rng_for(adjs)
    noise, g = ops.adj_init_fwd(None, N, sigma, randn=randn, rng=rng, sid=7001, B=B)
    if randn is None:
        advance(adjs.device)
    return adjs + noise, g


def add_feature_noise_v2(feats, sigma=0.2, randn=None):
    """ref: src/module/graph_utils.py:144-149"""
    rng = None if randn is not None else _rng_for(feats)
    out, g = XF.FeatureNoiseFn.apply(feats, sigma, randn, rng, 7002)
    if randn is None:
        advance(feats.device)
    return out, g
