"""Attribute-class cosine adjacency of the object regions (``adj_true``), on the GPU.

Reference: data/preprocess/vqa/compute_adjacency.py -- per image, BERT pooled embeddings of the 36 detected
objects' class names and attribute names (:77-85), then ``compute_cosin_sim_v2`` (:38-45: a Python double loop
of 666 ``torch.cosine_similarity`` calls filling the upper triangle incl. the diagonal, ``adj + adj^T``) and
``matrix / matrix.max()`` (:90).  The GQA twin (data/preprocess/gqa/compute_adjacency.py) is the same arithmetic.
The embedding extraction itself needs ``bert-base-uncased`` (a download) and stays outside: this module takes
the embeddings -- in practice a [vocabulary, 768] table per label file, gathered by the objects' ids -- and runs
the similarity for a whole split in one launch (``xggm_cosine_adjacency_f32``), one workgroup per image."""
import torch

from .. import _lib
from .._lib import call, ptr, stream


def compute_cosin_sim_v2(matrix1, matrix2, normalize=False):
    """matrix1 / matrix2: fp32 device tensors [N, D] (one image, the reference's signature, :38) or [n, N, D]
    (a batch) -> [N, N] / [n, N, N].  ``normalize`` also applies the ``matrix / matrix.max()`` of :90 -- the
    kernel always computes it; without it the result is scaled back."""
    single = matrix1.dim() == 2
    a = matrix1.unsqueeze(0) if single else matrix1
    b = matrix2.unsqueeze(0) if single else matrix2
    if not a.is_cuda:
        raise RuntimeError("xggm_amd: embeddings must live on the GPU (no CPU fallback)")
    a, b = a.contiguous().float(), b.contiguous().float()
    n, N, D = a.shape
    assert b.shape == a.shape
    out = torch.empty((n, N, N), device=a.device, dtype=torch.float32)
    call("xggm_cosine_adjacency_f32", ptr(a), ptr(b), ptr(out), n, N, D, 1e-6, stream())
    if not normalize:
        # un-normalised form: the maximum of adj + adj^T is max over (i <= j) of (1 + [i == j]) cos_ij; recover it
        # from the diagonal-doubled matrix the kernel divided by it
        an = a / a.norm(dim=-1, keepdim=True).clamp_min(1e-6)
        bn = b / b.norm(dim=-1, keepdim=True).clamp_min(1e-6)
        c = torch.einsum("nid,njd->nij", an, bn).triu()
        mx = (c + c.transpose(1, 2)).flatten(1).max(dim=1)[0]
        out = out * mx.view(n, 1, 1)
    return out[0] if single else out


def build_adjacency(class_table, attr_table, objects_id, attrs_id, chunk=4096):
    """adjacency of every image of a split.  class_table / attr_table: fp32 [vocabulary, D] pooled embeddings of the
    label strings (objects_vocab.txt / attributes_vocab.txt, :66-69); objects_id / attrs_id: int64 [n_img, N] -- the
    ``objects_id`` / ``attrs_id`` datasets of the detection h5 (:75-76).  Returns fp32 [n_img, N, N] on the device
    of the tables (what ``*_obj36_adj_v2.h5`` holds, one dataset per image)."""
    n = objects_id.shape[0]
    outs = []
    for s in range(0, n, chunk):
        oc = objects_id[s:s + chunk].to(class_table.device)
        ac = attrs_id[s:s + chunk].to(attr_table.device)
        outs.append(compute_cosin_sim_v2(class_table[oc], attr_table[ac], normalize=True))
    return torch.cat(outs)
