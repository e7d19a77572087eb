"""Binding of sub-modules to the root model's parameter arena and execution state.

The root (``VQAModel``/``GQAModel``, or a stand-alone encoder / generator) owns ONE
``ParamArena`` and ONE ``Runtime``; ``bind_root`` gives every sub-module a back reference and
a unique dropout stream id.  The arena is created lazily at the first forward, after the
caller has moved the model to the GPU (the reference does ``model = model.cuda()`` after
construction, src/vqa/vqacpv2.py:105).
"""
import weakref

import torch

from .arena import ParamArena
from .functional import Runtime

DEFAULT_DTYPE = torch.bfloat16


def _mark_shadow_dirty(module, incompatible_keys):
    """load_state_dict post-hook (any sub-module): the bf16 shadow weights are refreshed lazily
    at the next forward."""
    ref = getattr(module, "_xg_root", None)
    root = ref() if ref is not None else None
    if root is not None:
        object.__setattr__(root, "_xg_shadow_dirty", True)


def _gather_before_state_dict(module, prefix, keep_vars):
    """state_dict pre-hook (any sub-module): under the sharded update (ZeRO-1) the fp32 masters of the other ranks'
    slices are stale on this rank -- gather them first (a collective: call state_dict() on every rank)."""
    ref = getattr(module, "_xg_root", None)
    root = ref() if ref is not None else None
    rt = getattr(root, "_xg_rt", None) if root is not None else None
    if rt is not None:
        rt.arena.gather_sharded_state()


def bind_root(root, compute_dtype=None):
    """(re)bind all sub-modules of ``root``; the outermost model calls this last, so nested
    roots (an encoder inside a VQAModel) end up pointing at the outermost one."""
    ref = weakref.ref(root)
    for i, m in enumerate(root.modules()):
        object.__setattr__(m, "_xg_root", ref)
        object.__setattr__(m, "_sid", 16 * (i + 1))
        if not getattr(m, "_xg_hooked", False):
            m.register_load_state_dict_post_hook(_mark_shadow_dirty)
            m.register_state_dict_pre_hook(_gather_before_state_dict)
            object.__setattr__(m, "_xg_hooked", True)
    if compute_dtype is not None:
        object.__setattr__(root, "compute_dtype", compute_dtype)
    elif not hasattr(root, "compute_dtype"):
        object.__setattr__(root, "compute_dtype", DEFAULT_DTYPE)
    object.__setattr__(root, "_xg_arena", None)
    object.__setattr__(root, "_xg_rt", None)
    object.__setattr__(root, "_xg_shadow_dirty", False)
    return root


def root_of(module):
    ref = getattr(module, "_xg_root", None)
    root = ref() if ref is not None else None
    if root is None:
        root = bind_root(module)
    return root


def runtime_of(module):
    root = root_of(module)
    rt = root._xg_rt
    if rt is None or not rt.arena.valid() or rt.arena.compute_dtype != root.compute_dtype:
        arena = ParamArena(root, root.compute_dtype)
        seed = getattr(root, "seed", 9595)
        old = rt
        rt = Runtime(arena, seed)
        if old is not None:
            rt.training = old.training
        object.__setattr__(root, "_xg_arena", arena)
        object.__setattr__(root, "_xg_rt", rt)
    z = rt.arena.zero1
    if z is not None and z.pending and not torch.cuda.is_current_stream_capturing():
        z.wait_pending()  # all-gathers of the last sharded update still running beside the captured forward stages
    if getattr(root, "_xg_shadow_dirty", False):
        rt.arena.sync_shadow()
        object.__setattr__(root, "_xg_shadow_dirty", False)
    rt.training = root.training
    return rt


def set_compute_dtype(model, dtype):
    """switch between bf16 (default, MFMA bf16) and fp32 (exact-fp32 MFMA) execution."""
    root = root_of(model)
    object.__setattr__(root, "compute_dtype", dtype)
    object.__setattr__(root, "_xg_rt", None)
    return model


def sync_weights(model):
    """refresh bf16 shadow weights after parameters were modified outside the optimiser
    (load_state_dict, manual edits)."""
    root = root_of(model)
    if root._xg_rt is not None and root._xg_rt.arena.valid():
        root._xg_rt.arena.sync_shadow()
        object.__setattr__(root, "_xg_shadow_dirty", False)
