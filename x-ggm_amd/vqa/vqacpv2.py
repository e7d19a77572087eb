"""The VQA-CP v2 training iteration (src/vqa/vqacpv2.py:164-254) on the HIP blocks.

``loss_func`` / ``compute_kl_loss`` keep the reference names and signatures
(src/vqa/vqacpv2.py:48-61).  ``plain_pass`` / ``ggm_pass`` are the two optimiser passes of
one iteration; ``train_iteration`` strings them together in the VQA order (plain first) or
the GQA order (GGM first, src/gqa/gqa_ood.py:165-292).  The only PyTorch arithmetic left is
scalar glue on 0-dim loss tensors and the removal of the adjacency diagonal of the INPUT.
"""
import random

import torch
import torch.nn as nn

from .. import functional as XF
from ..lxrt.optimization import clip_grad_norm_
from ..runtime import runtime_of


# ``scale`` (not in the reference signatures, default 1): a constant factor folded into the loss kernel and its
# backward.  The training passes use it for the loss weights (x num_answers, x 6 (8 KL + DSM), ...): every
# ``scalar * loss`` written in torch is a kernel of its own in forward and another one in backward, ~5 us each for
# one float.
# ``slot``: a zeroed 1-element fp32 tensor the kernel accumulates into (Runtime.scalar_slot: the loss terms of a pass
# share one buffer zeroed by one launch); None = the op zeroes its own.
def loss_func(score, grad_log_q_noise, sigma=0.2, scale=1.0, slot=None):
    """0.5 sigma^2 mean_b sum_ij (score - g)^2 / (d1 d2).  ref: src/vqa/vqacpv2.py:48-51"""
    return XF.DSMFn.apply(score, grad_log_q_noise, sigma, scale, slot)


def compute_kl_loss(x, y, scale=1.0, slot=None):
    """symmetric KL of the last-dim softmaxes, mean over all elements.
    ref: src/vqa/vqacpv2.py:54-61"""
    if x.dtype != y.dtype:
        x, y = x.float(), y.float()
    return XF.SymKLFn.apply(x, y, scale, slot)


class BCEWithLogitsLoss(nn.Module):
    """nn.BCEWithLogitsLoss() of src/vqa/vqacpv2.py:131 on fp32 logits."""

    def forward(self, logit, target, scale=1.0, slot=None):
        return XF.BCEFn.apply(logit.float(), target.float(), scale, slot)


class _Scaled:
    """a loss term that was computed with its weight folded in; ``float()`` gives the unweighted value"""

    def __init__(self, t, c):
        self.t, self.c = t.detach(), float(c)

    def __float__(self):
        return float(self.t) / self.c

    def detach(self):
        return self


def remove_diagonal(adj_true):
    """adj_true.triu(1) + adj_true.tril(-1)   (src/vqa/vqacpv2.py:188): input preparation"""
    from .. import ops
    return ops.zero_diag(adj_true.float().contiguous())


def enable_data_parallel(model, group=None, wire_dtype=None, overlap=None, zero1=False):
    """one process per GPU: average the flat gradient arena over ``group`` between backward and
    the fused clip + BertAdam of every pass (xggm_amd.dist.GradSync); replicas start equal.
    ``overlap`` (default on; XGGM_DP_OVERLAP=0 turns it off): cut the backward between the single-modality
    and the cross-modality layers so the all-reduce of the upper 60 % of the gradients runs under the
    backward of the lower layers (engine.CapturedTrainer).  ``zero1``: shard the update (dist.ShardedUpdate:
    reduce-scatter of the matrix gradients, BertAdam on this rank's 1/world of every matrix range, all-gather
    of the bf16 weights the GEMMs read).
    Every rank draws its OWN dropout masks and denoising noise (the Philox seed is folded with the rank): the
    averaged gradient is then that of one batch of world x B samples with independent noise; only the host-side
    branch decision is shared (``pick_branch``)."""
    import os
    import torch.distributed as dist
    from ..dist import GradSync, broadcast_params
    rt = runtime_of(model)
    broadcast_params(rt.arena, group)
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if rank:
        # seed word of the device-resident {seed, offset} pair; saved / restored with the training state
        rt.rng[0] = (int(rt.rng[0].item()) + rank * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF
    inplace = wire_dtype == torch.bfloat16 and rt.arena.shadow is not None and os.environ.get("XGGM_DP_WIRE_ARENA", "1") != "0"
    if inplace:
        # bf16 on the wire + bf16 storage: matrix gradients are BORN in the wire arena (weight-gradient GEMM epilogue),
        # reduced in place and read by the update: no cast before, no copy after the exchange
        rt.arena.enable_wire()
    if zero1 and not inplace:
        raise RuntimeError("the sharded update needs the bf16 wire arena (bf16 storage, wire_dtype=torch.bfloat16)")
    if zero1:
        from ..dist import ShardedUpdate
        gs = rt.arena.zero1 = ShardedUpdate(rt.arena, group)
    else:
        gs = GradSync(rt.arena.grads, group, wire_dtype, arena=rt.arena if inplace else None)
    if inplace and os.environ.get("XGGM_DP_SPARSE_EMB", "1") != "0":
        # the word-embedding gradient: exchange the rows of this step's tokens, not the 30522-row table
        wt = model.lxrt_encoder.model.bert.embeddings.word_embeddings.weight
        xg = getattr(wt, "_xg", None)
        if xg is not None and xg[0] is rt.arena:
            gs.set_sparse_table(xg[1], wt.shape[0], wt.shape[1], lambda: rt.emb_ids)
    object.__setattr__(model, "_grad_sync", gs)
    rt.arena.sq_enabled = False  # the clip norm is that of the AVERAGED gradients: read them after the exchange
    rt.arena.row_list_enabled = False  # ... and the word table's gradient holds the OTHER ranks' rows too
    if overlap is None:
        overlap = os.environ.get("XGGM_DP_OVERLAP", "1") != "0"
    rt.cut_enabled = bool(overlap)
    return model


def _sync_grads(model):
    gs = getattr(model, "_grad_sync", None)
    if gs is not None:
        from ..dist import active_ranges
        gs.sync(active_ranges(runtime_of(model).arena))


def forward_backward_plain(model, bce_loss, feats, boxes, sent, target, between=None):
    """step A up to backward: src/vqa/vqacpv2.py:170-174.  ``between``: callback between the two backward
    stages when the runtime cuts the graph (Runtime.backward)."""
    model.zero_grad()
    rt = runtime_of(model)
    _, _, x = model(feats, boxes, sent)
    logit = model.logit_fc(x)
    loss = bce_loss(logit, target, scale=target.size(1))
    rt.backward(loss, between)
    return loss.detach(), logit.detach()


def forward_backward_ggm(model, bce_loss, feats, boxes, sent, target, adj_true, branch, sigma=1.0, kl_weight=8.0,
                         randn=None, between=None):
    """step B up to backward: relation generation (branch 'rel', src/vqa/vqacpv2.py:195-222) or
    representation generation ('node', :228-251).  ``randn`` injects the Gaussian draw
    (parity tests); None = in-kernel Philox."""
    model.zero_grad()
    rt = runtime_of(model)
    feat_seq, _, x = model(feats, boxes, sent)
    adj_true = remove_diagonal(adj_true)
    N = feat_seq[1].shape[1]
    A = target.size(1)
    rt.begin_losses(4)  # the three loss kernels accumulate into slots of one buffer zeroed by one launch
    # tensors the reference uses more than once go through XF.fan_out: same values, same gradients, but the sum of
    # the consumers' gradients is one launch of ours instead of the autograd engine's at::add per extra consumer
    x, x_fuse = XF.fan_out(x, 2)
    if branch == "rel":
        e = model.encoder_adj(x)
        adj_noise, grad_log_noise = XF.AdjInitFn.apply(e, N, sigma, randn, None if randn is not None else rt.rng, 9001)
        node_feats, adj_noise = model.generator(feat_seq[1], adj_noise)
        adj_noise, adj_kl = XF.fan_out(adj_noise, 2)
        # loss = bce * A + 6 * (kl_weight * (kl * A) + dsm), the weights folded into the loss kernels
        w_kl, w_dsm = 6.0 * kl_weight * A, 6.0
        loss_grad = loss_func(adj_noise, grad_log_noise, sigma=sigma, scale=w_dsm, slot=rt.scalar_slot())
        d_loss = compute_kl_loss(adj_true, adj_kl, scale=w_kl, slot=rt.scalar_slot())
    elif branch == "node":
        node_feats = XF.BcastRowsFn.apply(model.node_fc(x), N)  # == node_fc(x.unsqueeze(1).repeat(1, N, 1))
        node_feats, feat_grad = XF.FeatureNoiseFn.apply(node_feats, sigma, randn,
                                                        None if randn is not None else rt.rng, 9002)
        node_feats, _ = model.generator(node_feats, adj_true)
        node_feats, node_kl, node_dsm = XF.fan_out(node_feats, 3)
        # loss = bce * A + 1.1 * (0.15 * (kl * A) + 6 * dsm)
        w_kl, w_dsm = 1.1 * 0.15 * A, 1.1 * 6.0
        d_loss = compute_kl_loss(node_kl, feat_seq[1], scale=w_kl, slot=rt.scalar_slot())
        loss_grad = loss_func(node_dsm, feat_grad, sigma=sigma, scale=w_dsm, slot=rt.scalar_slot())
    else:
        raise ValueError(branch)
    x_gen = model.fusion_fc(XF.PoolConcatFn.apply(x_fuse, node_feats))
    logit = model.logit_fc(x_gen)
    loss = XF.LossSumFn.apply(bce_loss(logit, target, scale=A, slot=rt.scalar_slot()), d_loss, loss_grad)
    rt.backward(loss, between)
    # reported as the reference logs them: d_loss = KL * A, loss_grad = the unweighted DSM term
    return loss.detach(), logit.detach(), dict(d_loss=_Scaled(d_loss, w_kl / A), loss_grad=_Scaled(loss_grad, w_dsm))


def clip_and_step(model, optim, clip=5.0, advance=False):
    """nn.utils.clip_grad_norm_(params, 5.) + optim.step() + optim.zero_grad()
    (src/vqa/vqacpv2.py:175-177), fused: one norm reduction, one update pass.  ``advance``: also end the pass
    (Runtime.advance: new dropout masks / noise for the next one) -- the RNG step then rides on the norm's last launch."""
    rt = runtime_of(model)
    total = clip_grad_norm_(model.parameters(), clip, tail=(optim, rt if advance else None))
    optim.step()
    z = rt.arena.zero1
    if z is not None:
        z.gather()  # sharded update: the other ranks' slices of the bf16 weights
    optim.zero_grad()
    if advance:
        rt.advance()
    return total


def plain_pass(model, optim, bce_loss, feats, boxes, sent, target, clip=5.0, advance=False):
    out = forward_backward_plain(model, bce_loss, feats, boxes, sent, target)
    _sync_grads(model)
    clip_and_step(model, optim, clip, advance)
    return out


def ggm_pass(model, optim, bce_loss, feats, boxes, sent, target, adj_true, branch, sigma=1.0, kl_weight=8.0,
             randn=None, clip=5.0, advance=False):
    out = forward_backward_ggm(model, bce_loss, feats, boxes, sent, target, adj_true, branch, sigma, kl_weight, randn)
    _sync_grads(model)
    clip_and_step(model, optim, clip, advance)
    return out


def pick_branch(delta, rng=random, model=None):
    """random.randint(1, 10) <= args.delta -> relation generation (src/vqa/vqacpv2.py:192-193).  With data
    parallelism (``model`` carries a gradient exchange) every rank takes RANK 0's draw: ranks on different
    branches would exchange different parameter ranges (encoder_adj vs node_fc) and hang or mix gradients."""
    rel = rng.randint(1, 10) <= delta
    gs = getattr(model, "_grad_sync", None) if model is not None else None
    if gs is not None and gs.world > 1:
        from ..dist import sync_branch
        rel = sync_branch(rel, gs.g.device, gs.group)
    return "rel" if rel else "node"


def train_iteration(model, optim, bce_loss, batch, delta=5, sigma=1.0, order="vqa", branch=None, clip=5.0):
    """one iteration = two fwd+bwd+clip+BertAdam passes.  ``batch``: dict with feats, boxes,
    sent, target, adj_true (device tensors).  order 'vqa': plain then GGM (KL weight 8);
    'gqa': GGM then plain (KL weight 12, src/gqa/gqa_ood.py:197)."""
    rt = runtime_of(model)
    model.train()
    if branch is None:
        branch = pick_branch(delta, model=model)
    args = (batch["feats"], batch["boxes"], batch["sent"], batch["target"])
    out = {}
    if order == "vqa":
        out["loss_plain"], out["logit"] = plain_pass(model, optim, bce_loss, *args, clip=clip, advance=True)
        out["loss_ggm"], _, ex = ggm_pass(model, optim, bce_loss, *args, batch["adj_true"], branch, sigma, 8.0,
                                          clip=clip, advance=True)
    else:
        out["loss_ggm"], _, ex = ggm_pass(model, optim, bce_loss, *args, batch["adj_true"], branch, sigma, 12.0,
                                          clip=clip, advance=True)
        out["loss_plain"], out["logit"] = plain_pass(model, optim, bce_loss, *args, clip=clip, advance=True)
    out.update(ex)
    out["branch"] = branch
    return out


def make_optimizer(model, lr, t_total, warmup=0.1):
    """the two parameter groups of src/vqa/vqacpv2.py:113-128: heads/generator at 4*lr,
    encoder at lr; BertAdam(warmup=0.1, t_total=2*iters)."""
    from ..lxrt.optimization import BertAdam
    lxrt_ids = set(map(id, model.lxrt_encoder.parameters()))
    base_params = [p for p in model.parameters() if id(p) not in lxrt_ids]
    groups = [{"params": base_params, "lr": lr * 4}, {"params": list(model.lxrt_encoder.parameters())}]
    return BertAdam(groups, lr=lr, warmup=warmup, t_total=t_total)


def predict(model, eval_tuple, dump=None, predictor=None):
    """``VQA.predict`` (src/vqa/vqacpv2.py:315-339; GQA twin src/gqa/gqa_ood.py:379-403): eval mode, encoder ->
    ``logit_fc`` -> arg-max -> ``{question_id: answer}``.  ``eval_tuple`` = (dset, loader, evaluator) as in the
    reference; only the first four fields of a loader item are looked at (never the ground truth).  ``predictor``:
    an ``engine.CapturedPredictor`` to replay one captured forward per batch instead of launching eagerly."""
    dset, loader, evaluator = eval_tuple
    dev = next(model.parameters()).device
    was_training = model.training
    model.eval()
    quesid2ans = {}
    try:
        for datum_tuple in loader:
            ques_id, feats, boxes, sent = datum_tuple[:4]
            feats, boxes = feats.to(dev, non_blocking=True), boxes.to(dev, non_blocking=True)
            if predictor is not None:
                label, _ = predictor(feats, boxes, sent)
            else:
                with torch.no_grad():
                    _, _, x = model(feats, boxes, sent)
                    label = model.logit_fc(x).max(1)[1]
            for qid, l in zip(ques_id, label.cpu().numpy()):
                quesid2ans[qid.item() if hasattr(qid, "item") else qid] = dset.label2ans[l]
    finally:
        model.train(was_training)
    if dump is not None:
        evaluator.dump_result(quesid2ans, dump)
    return quesid2ans


def evaluate(model, eval_tuple, dump=None, predictor=None):
    """``VQA.evaluate`` (src/vqa/vqacpv2.py:341-344)"""
    return eval_tuple[2].evaluate(predict(model, eval_tuple, dump, predictor))


def save_training_state(path, model, optim, **extra):
    """everything an exact resume needs, which ``VQA.save`` (src/vqa/vqacpv2.py:361-363, model weights only) leaves
    out: BertAdam moments and step counters (reference state layout), the dropout / noise Philox state, the host
    branch-choice generator, and whatever the caller adds (epoch, iteration, best score)."""
    rt = runtime_of(model)
    # under the sharded update (ZeRO-1) this is a collective: every rank calls it (they all hold the gathered state
    # afterwards; let one of them pass a real ``path`` and the others ``None`` to write a single file)
    rt.arena.gather_sharded_state()
    ck = {"model": model.state_dict(), "optimizer": optim.state_dict(), "rng": rt.rng.cpu(),
          "python_random": random.getstate(), "extra": extra}
    if path is not None:
        torch.save(ck, path)
    return ck


def load_training_state(path, model, optim):
    """restores what ``save_training_state`` wrote, in place (captured graphs stay valid); returns ``extra``"""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    model.load_state_dict(ck["model"])
    rt = runtime_of(model)  # creates the arena if this model has not run yet, refreshes the bf16 shadows
    optim.load_state_dict(ck["optimizer"])
    rt.rng.copy_(ck["rng"])
    st = ck["python_random"]
    random.setstate((st[0], tuple(st[1]), st[2]))
    return ck["extra"]
