"""Autograd glue: one ``torch.autograd.Function`` per fused block of the hot path.

PyTorch supplies the tape, memory and streams; every arithmetic step of forward AND backward
is a HIP kernel behind the C ABI (``ops``).  Parameter gradients are not returned to
autograd: each backward writes them straight into the flat fp32 gradient arena
(``arena.target`` / ``arena.atomic_target``) and publishes ``p.grad`` as a view of it, so no
AccumulateGrad copies happen and the optimiser / all-reduce see contiguous memory.
Parameters are still passed as inputs (``*params``) so autograd schedules the backward of
blocks whose data inputs need no gradient.
"""
import os

import torch
from torch.autograd import Function

from . import ops

F32 = torch.float32


class ScalarSlot:
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class Runtime:
    """per-model execution state shared by all blocks: RNG words, training flag, dropout."""

    def __init__(self, arena, seed=9595):
        self.arena = arena
        self.rng = ops.make_rng(seed, arena.device)
        ops.sum_ws(arena.device)  # the ordered-sum workspace must exist before a graph capture can need it
        self.training = True
        self.p_hidden = 0.1   # config.hidden_dropout_prob (src/lxrt/modeling.py:196)
        self.p_attn = 0.1     # config.attention_probs_dropout_prob
        self.p_readout = 0.5  # GCN/GIN read-out dropout (src/module/gcn.py:33)
        self.pending = []     # deferred second stages of the LN backwards of the running backward pass
        self._hold_flush = False  # staged backward: the flushes of the stages below stage 0 are merged into one (``backward``)
        self._task = -1       # autograd graph task the pending jobs belong to
        # two-stage backward (data parallelism): the autograd graph is cut between the single-modality layers
        # and the cross-modality layers so the gradients of everything above the cut can be on the wire
        # while the backward below it still runs
        self.cut_enabled = False
        self.n_stages = 1
        self._cuts = []       # cuts recorded by the running forward, in forward order: (tag, outputs, leaves)
        self.fwd_hook = None  # called with the cut's index when the forward passes a cut (engine: one graph per stage)
        self.cut_layout = {}  # filled by LXRTEncoder.forward: where the cuts of this model sit (dist.stage_ranges)
        self.one = torch.ones((), device=arena.device, dtype=F32)  # d loss / d loss: the root of every backward
        self._slots = None    # zeroed scalar slots for the loss kernels of the running pass (one launch for all)

    def begin_losses(self, n=4):
        """the loss kernels of one pass accumulate into slots of ONE buffer zeroed by one launch"""
        self._slots = [ScalarSlot(t) for t in ops.zeros_f32(n, self.arena.device).split(1)]

    def scalar_slot(self):
        """a zeroed 1-element accumulator (wrapped so autograd does not see an input tensor that the loss function
        would then return a view of), or None outside ``begin_losses``"""
        if self._slots:
            return self._slots.pop()
        return None

    def make_cut(self, tag, *tensors):
        """called by LXRTEncoder.forward at a cut: returns detached leaves to continue with"""
        leaves = [t.detach().requires_grad_(True) for t in tensors]
        self._cuts.append((tag, list(tensors), leaves))
        if self.fwd_hook is not None:
            self.fwd_hook(len(self._cuts) - 1)
        return leaves

    def backward(self, loss, between=None):
        """loss.backward(), in stages when the forward recorded cuts: stage 0 stops at the leaves of the last
        cut, ``between(k)`` runs after stage k (every gradient above that cut is final), the next stage
        continues below it, ... -- ``between`` is not called after the final stage."""
        cuts, self._cuts = self._cuts, []
        self.n_stages = len(cuts) + 1
        self._slots = None
        # explicit root gradient: autograd would otherwise launch a fill kernel for ones_like(loss)
        loss.backward(self.one if loss.dim() == 0 and loss.dtype == F32 and loss.device == self.one.device else None)
        # Stage 0's deferred parameter-gradient sums were flushed by its autograd callback: the groups that are final after
        # stage 0 (enc_tail, heads, generator) go on the wire WITH their vectors.  Every later stage only adds jobs whose
        # targets lie in enc_main's vector region, which is exchanged after the LAST stage (dist.stage_ranges): their
        # flushes are held and run once at the end -- 2 reduce rounds per pass instead of one per stage (5 stages: the
        # one-rank RCCL profile showed 12 launches of 29 us per iteration where the plain step has 4).
        self._hold_flush = len(cuts) > 1
        try:
            for k, (_, outs, leaves) in enumerate(reversed(cuts)):
                if between is not None:
                    between(k)
                pairs = [(o, l.grad) for o, l in zip(outs, leaves) if l.grad is not None]
                if pairs:
                    torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
        finally:
            held, self._hold_flush = self._hold_flush, False
        if held:
            self.flush()

    def defer_list(self):
        """list the LN backwards of the running autograd backward append their reduce jobs to; the
        engine calls ``flush`` once when that backward has finished (before anything reads .grad)."""
        task = torch._C._current_graph_task_id()
        if task == -1:
            return None  # not inside a backward pass: the caller reduces immediately
        if task != self._task:
            if not self._hold_flush:
                self.pending = []  # jobs of a backward that died before its callback ran are dropped
            self._task = task
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)
        return self.pending

    def flush(self):
        self._task = -1
        if self._hold_flush:
            return  # staged backward below stage 0: ``backward`` flushes once after the last stage
        jobs, self.pending = self.pending, []
        # One writer per gradient vector and launch: a second job on the same targets (the shared cross-attention
        # module is applied twice per layer) has to run in a LATER launch than the first.  A launch takes
        # ops.REDUCE_JOBS_PER_LAUNCH jobs; a pass leaves ~90, i.e. two launches anyway -- so the first launch takes the
        # first writers of every shared target (and fills up with others), the second the rest together with the second
        # writers: two launches per pass instead of three.
        rounds = []  # rounds[k] = jobs that are the (k+1)-th writer of some target
        count = {}
        for j in jobs:
            key = [t.data_ptr() for t in j[4] if t is not None]
            k = max([count.get(p, 0) for p in key], default=0)
            for p in key:
                count[p] = k + 1
            while len(rounds) <= k:
                rounds.append([])
            rounds[k].append(j)
        if not rounds:
            return
        shared = {p for p, c in count.items() if c > 1}
        first = sorted(rounds[0], key=lambda j: 0 if any(t is not None and t.data_ptr() in shared for t in j[4]) else 1)
        cap = ops.REDUCE_JOBS_PER_LAUNCH
        n_first = max(sum(1 for j in first if any(t is not None and t.data_ptr() in shared for t in j[4])), min(len(first), cap))
        n_first = min(n_first, len(first)) if len(first) <= cap else max(n_first, cap)
        launches = [first[:n_first]]
        rest = first[n_first:]
        for k in range(1, len(rounds)):
            launches.append(rest + rounds[k])
            rest = []
        if rest:
            launches.append(rest)
        for batch in launches:
            ops.reduce_batch(batch)

    def p(self, p):
        return p if self.training else 0.0

    def advance(self):
        """new dropout masks / noise for the next pass (graph-capturable)."""
        if getattr(self, "rng_advanced", False):
            self.rng_advanced = False  # the norm's finishing launch of this pass has advanced the state (clip_grad_norm_)
        else:
            ops.rng_advance(self.rng, 1)
        if self.arena.fp8 is not None:
            self.arena.fp8.step()  # delayed scaling: the activation maxima of this pass set the next pass's scales


def _w(rt, p):
    return rt.arena.w(p)


def _wgrad(rt, dy, x, ps):
    gw, acc = rt.arena.target(ps)
    ops.linear_wgrad(dy, x, gw, acc)


def _colsum(rt, dy, ps):
    ops.colsum(dy, rt.arena.atomic_target(ps))


# --------------------------------------------------------------------------------- embeddings
class EmbedFn(Function):
    """BertEmbeddings.forward (src/lxrt/modeling.py:298-313)."""

    @staticmethod
    def forward(ctx, rt, mod, ids, seg, *params):
        ctx.np = len(params)
        a = rt.arena
        # fp8 forward: the embedding output is the A operand of the first language layer's e4m3 products -- it leaves
        # this kernel as e4m3 too (no quantisation launch)
        f8 = a.fp8 if a.w(mod.word_embeddings.weight).dtype == torch.bfloat16 else None
        e8 = f8.emit(("emb", id(mod))) if f8 is not None else (None, None)
        side, rt.side_jobs = getattr(rt, "side_jobs", None), None  # input glue LXRTModel.forward queued for this launch
        res = ops.embed_fwd(ids, seg, a.w(mod.word_embeddings.weight),
                            a.w(mod.position_embeddings.weight),
                            a.w(mod.token_type_embeddings.weight), mod.LayerNorm.weight.data,
                            mod.LayerNorm.bias.data, 1e-12, rt.p(rt.p_hidden), rt.rng, mod._sid, emit8=e8[0], side=side)
        out, z, stats = res[:3]
        if f8 is not None:
            f8.put(out, res[3], e8[1])
        ctx.rt, ctx.mod, ctx.p = rt, mod, rt.p(rt.p_hidden)
        ctx.saved = (ids, seg, z, stats)
        rt.emb_ids = ids  # the rows of the word table this pass touches (data parallel: dist.GradSync.set_sparse_table)
        a.emb_uses = getattr(a, "emb_uses", 0) + 1  # ... valid only if this is the ONE look-up since zero_grad()
        return out

    @staticmethod
    def backward(ctx, dy):
        rt, mod = ctx.rt, ctx.mod
        ids, seg, z, stats = ctx.saved
        a = rt.arena
        ops.embed_bwd(ids, seg, dy.contiguous(), z, stats, mod.LayerNorm.weight.data,
                      a.atomic_target(mod.word_embeddings.weight), a.atomic_target(mod.position_embeddings.weight),
                      a.atomic_target(mod.token_type_embeddings.weight), a.atomic_target(mod.LayerNorm.weight),
                      a.atomic_target(mod.LayerNorm.bias), ctx.p, rt.rng, mod._sid,
                      row_list=a.row_list_for(mod.word_embeddings.weight, dy.numel() // dy.shape[-1]))
        return (None,) * (4 + ctx.np)


class VisnEmbedFn(Function):
    """VisualFeatEncoder.forward (src/lxrt/modeling.py:546-556); feats/boxes already T."""

    @staticmethod
    def forward(ctx, rt, mod, feats, boxes, *params):
        ctx.np = len(params)
        u, _ = ops.linear_fwd(feats, _w(rt, mod.visn_fc.weight), None)
        p = rt.p(rt.p_hidden)
        f8 = rt.arena.fp8 if u.dtype == torch.bfloat16 else None
        e8 = f8.emit(("visn_emb", id(mod))) if f8 is not None else (None, None)
        res = ops.visn_embed_fwd(
            u, mod.visn_fc.bias.data, boxes, mod.box_fc.weight.data, mod.box_fc.bias.data,
            mod.visn_layer_norm.weight.data, mod.visn_layer_norm.bias.data, mod.box_layer_norm.weight.data,
            mod.box_layer_norm.bias.data, 1e-12, p, rt.rng, mod._sid, emit8=e8[0])
        out, z1, z2, stats = res[:4]
        if f8 is not None:
            f8.put(out, res[4], e8[1])
        ctx.rt, ctx.mod, ctx.p = rt, mod, p
        ctx.saved = (feats, boxes, z1, z2, stats)
        return out

    @staticmethod
    def backward(ctx, dy):
        rt, mod = ctx.rt, ctx.mod
        feats, boxes, z1, z2, stats = ctx.saved
        a = rt.arena
        grads = dict(dbf=a.atomic_target(mod.visn_fc.bias), dg1=a.atomic_target(mod.visn_layer_norm.weight),
                     db1=a.atomic_target(mod.visn_layer_norm.bias), dWb=a.atomic_target(mod.box_fc.weight),
                     dbb=a.atomic_target(mod.box_fc.bias), dg2=a.atomic_target(mod.box_layer_norm.weight),
                     db2=a.atomic_target(mod.box_layer_norm.bias))
        du = ops.visn_embed_bwd(dy.contiguous(), z1, z2, stats, boxes, mod.visn_layer_norm.weight.data,
                                mod.box_layer_norm.weight.data, grads, ctx.p, rt.rng, mod._sid)
        _wgrad(rt, du, feats, mod.visn_fc.weight)
        return (None,) * (4 + ctx.np)  # inputs are data: no gradient w.r.t. feats / boxes


# --------------------------------------------------------------------------------- transformer blocks
# Every block is written as a GENERATOR that yields lists of GEMM problems (ops.GemmProblem) at the
# points where it needs dense products done, and runs its other kernels (attention core, row
# kernels) inline.  ``drive`` advances one or several generators in lockstep and launches what
# they yielded in the same round as ONE grouped GEMM: dgrad + wgrad of a layer, or the same layer
# of the language and the vision stream, share a grid -- these products are too skinny
# (M = 640 / 1152 rows) to fill 256 CUs alone.
def drive(dt, gens):
    """advance the generators in lockstep; returns their return values."""
    n = len(gens)
    out = [None] * n
    alive = list(range(n))
    while alive:
        probs, rows, nxt = [], [], []
        for i in alive:
            try:
                p = next(gens[i])
                if isinstance(p, ops.ROW_REQUESTS):
                    rows.append(p)  # a row-kernel request: shares its launch with the other stream's
                elif p:
                    probs.extend(p)
                nxt.append(i)
            except StopIteration as e:
                out[i] = e.value
        if rows:
            ops.launch_row_requests(rows)
        if probs:
            p8 = [p for p in probs if getattr(p, "is8", False)]
            if p8:
                ops.gemm_group8(p8)
                probs = [p for p in probs if not getattr(p, "is8", False)]
            if probs:
                ops.gemm_group(dt, probs)
        alive = nxt
    return out


def _p_wgrad(rt, dy, x, ps):
    gw, acc = rt.arena.target(ps)
    return ops.p_wgrad(dy, x, gw, acc, rt.arena.sq_target(ps, dy))


def g_attn_fwd(rt, att, outm, xq, xkv, mask, B, Sq, Sk, salt):
    """BertSelfattLayer / BertCrossattLayer forward (src/lxrt/modeling.py:391-414):
    y = LN(dropout(W_o attn(W_q x_q, W_k x_kv, W_v x_kv) + b_o) + x_q).
    ``xkv is None`` = self-attention: one fused [3H,H] projection."""
    a = rt.arena
    H = xq.shape[1]
    heads = att.num_attention_heads
    wq, wk, wv = att.query.weight, att.key.weight, att.value.weight
    bq, bk, bv = att.query.bias, att.key.bias, att.value.bias
    self_att = xkv is None
    # fp8 forward (xggm_amd.fp8): ``emit`` = producers write e4m3 copies and record maxima (also while calibrating),
    # ``use8`` = the products read them
    f8 = a.fp8 if xq.dtype == torch.bfloat16 else None
    emit, use8 = f8 is not None, f8 is not None and f8.active

    if emit:
        xq8, sq = f8.get(xq, ("in", id(att), salt, 0))
        if not self_att:
            xkv8, skv = f8.get(xkv, ("in", id(att), salt, 1))
    if self_att:
        if use8:
            w8, sw = f8.w8([wq, wk, wv])
            p1, qkv, _ = ops.p_fwd8(xq8, w8, sq, sw, a.fused([bq, bk, bv]))
            p1.is8 = True
        else:
            p1, qkv, _ = ops.p_fwd(xq, a.fused([wq, wk, wv]), a.fused([bq, bk, bv]))
        kv = None
        yield [p1]
        q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    else:
        if use8:
            w8q, sw = f8.w8(wq)
            w8kv, _ = f8.w8([wk, wv])
            p1, qkv, _ = ops.p_fwd8(xq8, w8q, sq, sw, bq.data)
            p2, kv, _ = ops.p_fwd8(xkv8, w8kv, skv, sw, a.fused([bk, bv]))
            p1.is8 = p2.is8 = True
        else:
            p1, qkv, _ = ops.p_fwd(xq, a.w(wq), bq.data)
            p2, kv, _ = ops.p_fwd(xkv, a.fused([wk, wv]), a.fused([bk, bv]))
        yield [p1, p2]
        q, k, v = qkv, kv[:, :H], kv[:, H:]
    p_att, p_hid = rt.p(rt.p_attn), rt.p(rt.p_hidden)
    e_ctx = f8.emit(("ctx", id(att), salt)) if emit else (None, None)
    core = ops.AttnFwdReq(q, k, v, mask, B, heads, Sq, Sk, p_att, rt.rng, att._sid + salt, emit8=e_ctx[0])
    # weight prefetch beside the attention core (ops.prefetch_next): behind W_v in the arena lie W_o (H^2 elements) and
    # this layer's W_1 and W_2 (4 H^2 each) -- the operands of the next three products.  The attention launches are
    # the long ones of the chain; the LayerNorm launches, half as long, take half a next layer's W_q / W_k / W_v each
    core.prefetch = (a.after(wv, 9 * H * H, use8),)
    yield core
    c = core.out
    if use8 and core.out8 is not None:
        w8o, swo = f8.w8(outm.dense.weight)
        p3, h, _ = ops.p_fwd8(core.out8, w8o, f8.dscale[e_ctx[1]:e_ctx[1] + 1], swo, None)
        p3.is8 = True
    else:
        p3, h, _ = ops.p_fwd(c, a.w(outm.dense.weight), None)
    yield [p3]
    e_ln = f8.emit(("ln", id(outm), salt)) if emit else (None, None)
    ln = ops.LnFwdReq(h, outm.dense.bias.data, xq, outm.LayerNorm.weight.data, outm.LayerNorm.bias.data, 1e-12,
                      p_pre=p_hid, rng=rt.rng, sid_pre=outm._sid + salt, emit8=e_ln[0])
    ln.prefetch = (a.after(outm.dense.weight, 3 * H * H // 2, use8, skip=8 * H * H),)  # behind W_1, W_2: the next layer's W_q and half of W_k
    yield ln
    if emit:
        f8.put(ln.out, ln.out8, e_ln[1])
    return ln.out, (att, outm, (B, Sq, Sk, heads, H, p_att, p_hid, self_att, salt),
                    (xq, xkv, mask, qkv, kv, c, ln.z, ln.stats))


def g_attn_bwd(rt, saved, dy, defer_wgrad=False, link=None):
    """backward of g_attn_fwd -> (dxq, dxkv).  ``defer_wgrad``: issue the weight-gradient products
    one round later (second use of SHARED weights in a lockstep pair: its read-modify-write of the
    gradient must not share a launch with the first use's write).  ``link``: dict shared by the two
    directions of one cross-attention layer; the direction that runs second adds its input gradients into
    the first one's buffers (x_l receives a gradient as query of one direction and as key/value of the
    other) in the GEMM epilogue instead of a separate add kernel."""
    att, outm, dims, tens = saved
    B, Sq, Sk, heads, H, p_att, p_hid, self_att, salt = dims
    xq, xkv, mask, qkv, kv, c, z, stats = tens
    a = rt.arena
    wq, wk, wv = att.query.weight, att.key.weight, att.value.weight
    bq, bk, bv = att.query.bias, att.key.bias, att.value.bias
    lb = ops.LnBwdReq(dy.contiguous(), z, stats, outm.LayerNorm.weight.data, a.atomic_target(outm.LayerNorm.weight),
                      a.atomic_target(outm.LayerNorm.bias), a.atomic_target(outm.dense.bias), p_pre=p_hid, rng=rt.rng,
                      sid_pre=outm._sid + salt, defer=rt.defer_list())
    if a.shadow is not None:
        lb.prefetch = (a.w(outm.dense.weight).view(-1),)  # the dgrad right behind this launch reads W_o
    yield lb
    d_h, d_res = lb.d_in, lb.d_res
    pd, d_c = ops.p_dgrad(d_h, a.w(outm.dense.weight))
    if defer_wgrad:
        yield [pd]
        late = [_p_wgrad(rt, d_h, c, outm.dense.weight)]
    else:
        yield [_p_wgrad(rt, d_h, c, outm.dense.weight), pd]
        late = []
    if self_att:
        dqkv = torch.empty_like(qkv)
        gb = a.atomic_target([bq, bk, bv])  # q/k/v bias gradients come out of the attention backward
        ab = ops.AttnBwdReq(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], mask, d_c, dqkv[:, :H], dqkv[:, H:2 * H],
                            dqkv[:, 2 * H:], B, heads, Sq, Sk, p_att, rt.rng, att._sid + salt, gb[:H], gb[H:2 * H],
                            gb[2 * H:], defer=rt.defer_list())
        if a.shadow is not None:
            # the dgrad behind this launch reads W_q, W_k, W_v; in front of them in the arena lie the FFN weights of the
            # layer below (8 H^2 elements), which the backward reaches two launches later
            ab.prefetch = (a.fused([wq, wk, wv]).view(-1), a.before(wq, 8 * H * H))
        yield ab
        pdx, dxq = ops.p_dgrad(dqkv, a.fused([wq, wk, wv]), residual=d_res)
        dxkv = None
        if defer_wgrad:
            yield late + [pdx]
            yield [_p_wgrad(rt, dqkv, xq, [wq, wk, wv])]
        else:
            yield [_p_wgrad(rt, dqkv, xq, [wq, wk, wv]), pdx]
    else:
        dq = torch.empty_like(qkv)
        dkv = torch.empty_like(kv)
        gbq, gbkv = a.atomic_target(bq), a.atomic_target([bk, bv])
        yield ops.AttnBwdReq(qkv, kv[:, :H], kv[:, H:], mask, d_c, dq, dkv[:, :H], dkv[:, H:], B, heads, Sq, Sk, p_att,
                             rt.rng, att._sid + salt, gbq, gbkv[:H], gbkv[H:], defer=rt.defer_list())
        if defer_wgrad and link is not None and "dxq" in link:
            # second direction: its gradient w.r.t. its query input is the other direction's key/value input
            # and vice versa -- accumulate there, one round after the first direction wrote them
            yield late
            pdq, dxq = ops.p_dgrad(dq, a.w(wq), residual=d_res, into=link["dxkv"])
            pdk, dxkv = ops.p_dgrad(dkv, a.fused([wk, wv]), into=link["dxq"])
            yield [_p_wgrad(rt, dq, xq, wq), _p_wgrad(rt, dkv, xkv, [wk, wv]), pdq, pdk]
            return dxq, dxkv
        pdq, dxq = ops.p_dgrad(dq, a.w(wq), residual=d_res)
        pdk, dxkv = ops.p_dgrad(dkv, a.fused([wk, wv]))
        if link is not None:
            link["dxq"], link["dxkv"] = dxq, dxkv
        if defer_wgrad:
            yield late + [pdq, pdk]
            yield [_p_wgrad(rt, dq, xq, wq), _p_wgrad(rt, dkv, xkv, [wk, wv])]
        else:
            yield [_p_wgrad(rt, dq, xq, wq), _p_wgrad(rt, dkv, xkv, [wk, wv]), pdq, pdk]
    return dxq, dxkv


# split-K of the FFN output product (XGGM_SPLIT_K = 0 / 1 turns it off; same-box A/B hook).  Three slabs until round 4;
# with the weights prefetched and the role tiles two measure better at every batch (32 samples: 10.57 / 10.58 ms against
# 10.61 / 10.63, 64: 14.41 against 14.45-14.57, 92: 17.94 / 17.95 against 18.17 / 18.21 -- the LayerNorm behind reads a
# third less fp32; profiles/r04_experiments/split_k_*.txt)
_SPLIT_K = int(os.environ.get("XGGM_SPLIT_K", "2"))


def g_ffn_fwd(rt, inter, outm, x):
    """BertIntermediate + BertOutput (src/lxrt/modeling.py:428-445):
    y = LN(dropout(W_2 gelu(W_1 x + b_1) + b_2) + x)."""
    a = rt.arena
    f8 = a.fp8 if x.dtype == torch.bfloat16 else None
    emit, use8 = f8 is not None, f8 is not None and f8.active
    act8 = e_act = None
    if emit:
        x8, sx = f8.get(x, ("in", id(inter), 0, 0))
        e_act = f8.emit(("act", id(inter)))
        act8 = torch.empty((x.shape[0], inter.dense.weight.shape[0]), device=x.device, dtype=torch.uint8)
        e8 = (act8,) + e_act[0]
    if use8:
        w8, sw = f8.w8(inter.dense.weight)
        p1, act, u = ops.p_fwd8(x8, w8, sx, sw, inter.dense.bias.data, act=ops.ACT_GELU, want_preact=True, emit8=e8)
        p1.is8 = True
    else:
        p1, act, u = ops.p_fwd(x, a.w(inter.dense.weight), inter.dense.bias.data, act=ops.ACT_GELU, want_preact=True,
                               emit8=e8 if emit else None)
    yield [p1]
    w2 = a.w(outm.dense.weight)
    S = _SPLIT_K if (x.dtype == torch.bfloat16 and w2.shape[1] >= 2048 and w2.shape[1] % (128 * max(_SPLIT_K, 1)) == 0) else 0
    if use8:
        w8, sw = f8.w8(outm.dense.weight)
        p2, h, _ = ops.p_fwd8(act8, w8, f8.dscale[e_act[1]:e_act[1] + 1], sw, None, split=S if S > 1 else 0)
        p2.is8 = True
    elif S > 1:
        # K = 3072, N = 768: too few tiles for their long k-loop -> S slices of K, summed by the LayerNorm below
        p2, h = ops.p_fwd_splitk(act, w2, S)
    else:
        p2, h, _ = ops.p_fwd(act, w2, None)
    yield [p2]
    p_hid = rt.p(rt.p_hidden)
    e_ln = f8.emit(("ln", id(outm), 0)) if emit else (None, None)
    ln = ops.LnFwdReq(h, outm.dense.bias.data, x, outm.LayerNorm.weight.data, outm.LayerNorm.bias.data, 1e-12,
                      p_pre=p_hid, rng=rt.rng, sid_pre=outm._sid, dtype=x.dtype, emit8=e_ln[0])
    ln.prefetch = (a.after(outm.dense.weight, 3 * x.shape[1] * x.shape[1] // 2, use8, skip=3 * x.shape[1] * x.shape[1] // 2),)  # the other half of the next layer's W_k, and W_v
    yield ln
    if emit:
        f8.put(ln.out, ln.out8, e_ln[1])
    return ln.out, (inter, outm, p_hid, (x, u, act, ln.z, ln.stats))


def g_ffn_bwd(rt, saved, dy):
    inter, outm, p_hid, (x, u, act, z, stats) = saved
    a = rt.arena
    lb = ops.LnBwdReq(dy.contiguous(), z, stats, outm.LayerNorm.weight.data, a.atomic_target(outm.LayerNorm.weight),
                      a.atomic_target(outm.LayerNorm.bias), a.atomic_target(outm.dense.bias), p_pre=p_hid, rng=rt.rng,
                      sid_pre=outm._sid, defer=rt.defer_list())
    yield lb
    d_h, d_res = lb.d_in, lb.d_res
    # d_u = (d_h W_2) * gelu'(u); its column sums (= grad of b_1) are taken in the same epilogue
    pd, d_u = ops.p_dgrad(d_h, a.w(outm.dense.weight), gelu_aux=u, colsum=a.atomic_target(inter.dense.bias),
                          defer=rt.defer_list())
    yield [_p_wgrad(rt, d_h, act, outm.dense.weight), pd]
    pdx, dx = ops.p_dgrad(d_u, a.w(inter.dense.weight), residual=d_res)
    yield [_p_wgrad(rt, d_u, x, inter.dense.weight), pdx]
    return dx


class AttnBlockFn(Function):
    """one attention block (self or cross); see g_attn_fwd."""

    @staticmethod
    def forward(ctx, rt, att, outm, xq, xkv, mask, B, Sq, Sk, salt, *params):
        ctx.np = len(params)
        (res,) = drive(xq.dtype, [g_attn_fwd(rt, att, outm, xq, xkv, mask, B, Sq, Sk, salt)])
        y, ctx.saved = res
        ctx.rt = rt
        return y

    @staticmethod
    def backward(ctx, dy):
        ((dxq, dxkv),) = drive(dy.dtype, [g_attn_bwd(ctx.rt, ctx.saved, dy)])
        return (None, None, None, dxq, dxkv, None, None, None, None, None) + (None,) * ctx.np


class FFNFn(Function):
    """one feed-forward block; see g_ffn_fwd."""

    @staticmethod
    def forward(ctx, rt, inter, outm, x, *params):
        ctx.np = len(params)
        (res,) = drive(x.dtype, [g_ffn_fwd(rt, inter, outm, x)])
        y, ctx.saved = res
        ctx.rt = rt
        return y

    @staticmethod
    def backward(ctx, dy):
        (dx,) = drive(dy.dtype, [g_ffn_bwd(ctx.rt, ctx.saved, dy)])
        return (None, None, None, dx) + (None,) * ctx.np


_NO_LINK = bool(__import__("os").environ.get("XGGM_NO_LINK"))  # A/B hook: separate add kernels as before


class PairFn(Function):
    """two independent blocks (the same layer of the language and of the vision stream, or the two
    directions of a cross-attention layer) advanced in lockstep so that their GEMMs share launches.
    ``kind``: 'self' (two self-attention blocks), 'cross' (ONE shared attention module applied
    lang<-visn and visn<-lang, src/lxrt/modeling.py:485-492) or 'ffn'."""

    @staticmethod
    def forward(ctx, rt, kind, mods, x_l, x_v, mask_l, B, T, N, *params):
        ctx.np = len(params)
        ctx.set_materialize_grads(False)  # an unused output must arrive as None, not as zeros
        if kind == "self":
            (att_l, out_l), (att_v, out_v) = mods
            gens = [g_attn_fwd(rt, att_l, out_l, x_l, None, mask_l, B, T, T, 0),
                    g_attn_fwd(rt, att_v, out_v, x_v, None, None, B, N, N, 0)]
        elif kind == "cross":
            att, outm = mods
            gens = [g_attn_fwd(rt, att, outm, x_l, x_v, None, B, T, N, 1),
                    g_attn_fwd(rt, att, outm, x_v, x_l, mask_l, B, N, T, 2)]
        else:
            (in_l, out_l), (in_v, out_v) = mods
            gens = [g_ffn_fwd(rt, in_l, out_l, x_l), g_ffn_fwd(rt, in_v, out_v, x_v)]
        (y_l, s_l), (y_v, s_v) = drive(x_l.dtype, gens)
        ctx.rt, ctx.kind, ctx.saved = rt, kind, (s_l, s_v)
        return y_l, y_v

    @staticmethod
    def backward(ctx, dy_l, dy_v):
        rt, kind = ctx.rt, ctx.kind
        s_l, s_v = ctx.saved
        # a stream whose output is not used downstream (the vision side of the last cross layer
        # in the plain-VQA pass) receives no gradient: autograd hands a None / never calls us
        gens, who = [], []
        both = dy_l is not None and dy_v is not None
        link = {} if (kind == "cross" and both and not _NO_LINK) else None
        if dy_l is not None:
            gens.append(g_ffn_bwd(rt, s_l, dy_l) if kind == "ffn" else g_attn_bwd(rt, s_l, dy_l, link=link))
            who.append("l")
        if dy_v is not None:
            gens.append(g_ffn_bwd(rt, s_v, dy_v) if kind == "ffn"
                        else g_attn_bwd(rt, s_v, dy_v, defer_wgrad=(kind == "cross" and dy_l is not None), link=link))
            who.append("v")
        res = dict(zip(who, drive((dy_l if dy_l is not None else dy_v).dtype, gens)))
        dx_l = dx_v = None
        if kind == "ffn":
            dx_l, dx_v = res.get("l"), res.get("v")
        elif kind == "self":
            dx_l = res["l"][0] if "l" in res else None
            dx_v = res["v"][0] if "v" in res else None
        else:  # cross: lang block's dxkv is a gradient of x_v and vice versa
            if "l" in res:
                dx_l, dx_v = res["l"]
            if "v" in res:
                dv_q, dv_kv = res["v"]
                if link is not None:  # the vision direction accumulated into the language direction's buffers
                    assert dv_q is dx_v and dv_kv is dx_l
                else:
                    dx_v = dv_q if dx_v is None else ops.add_n([dx_v, dv_q], out=dx_v)
                    dx_l = dv_kv if dx_l is None else ops.add_n([dx_l, dv_kv], out=dx_l)
        return (None, None, None, dx_l, dx_v, None, None, None, None) + (None,) * ctx.np


# --------------------------------------------------------------------------------- fan-out
class FanOutFn(Function):
    """``n`` aliases of one tensor for ``n`` consumers.  Where a tensor of the training loop is used more than once
    (x, feat_seq[1], node_feats, adj_noise in src/vqa/vqacpv2.py:195-251) torch's autograd engine adds the consumers'
    gradients with at::add, one framework kernel per extra consumer; routed through here the sum is ONE launch of
    xggm_add_n (fp32 arithmetic, rounded once).  Values and gradients are those of plain multiple use."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g.contiguous() for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        if len(gs) <= 4:
            return ops.add_n(gs), None
        acc = ops.add_n(gs[:4])
        for k in range(4, len(gs), 3):
            acc = ops.add_n([acc] + gs[k:k + 3], out=acc)
        return acc, None


def fan_out(x, n):
    return FanOutFn.apply(x, n) if (n > 1 and x.requires_grad) else (x,) * n


class FirstTokenFn(Function):
    """``hidden_states[:, 0]`` (src/lxrt/modeling.py:616): the strided view forward; backward writes the gradient rows
    into a buffer of our own zeroing (autograd's select backward is a framework fill plus a strided copy)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return x[:, 0]

    @staticmethod
    def backward(ctx, g):
        B, S, H = ctx.shape
        if g.dtype != torch.bfloat16:  # fp32 parity mode: the framework's way
            full = g.new_zeros(B, S, H)
            full[:, 0] = g
            return full
        # rows of width H in a row stride of S * H, zeros behind them: exactly xggm_pad_rows
        return _first_token_grad(g.contiguous(), B, S, H)


def _first_token_grad(g, B, S, H):
    out = torch.empty((B, S * H), device=g.device, dtype=g.dtype)
    ops.call("xggm_pad_rows_bf16", ops.ptr(g), 0, ops.ptr(out), B, H, S * H, ops.stream())
    return out.view(B, S, H)



# --------------------------------------------------------------------------------- heads
class LinearActFn(Function):
    """Linear followed by none / tanh / sigmoid.  ``out_f32`` keeps the result in fp32
    (answer logits, encoder_adj probabilities)."""

    @staticmethod
    def forward(ctx, rt, lin, x, act, out_f32, *params):
        ctx.np = len(params)
        a = rt.arena
        x2 = x if x.dim() == 2 else x.reshape(-1, x.shape[-1])
        bias = lin.bias.data if lin.bias is not None else None
        y, _ = ops.linear_fwd(x2, a.w(lin.weight), bias, act=act, out_f32=out_f32)
        ctx.rt, ctx.lin, ctx.act, ctx.out_f32 = rt, lin, act, out_f32
        ctx.saved = (x2, y)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        rt, lin, act = ctx.rt, ctx.lin, ctx.act
        x2, y = ctx.saved
        a = rt.arena
        dy = dy.reshape(y.shape).contiguous()
        dt = x2.dtype
        if act == ops.ACT_SIGMOID:
            g = ops.sigmoid_bwd(dy if dy.dtype == F32 else dy.float(), y if y.dtype == F32 else y.float(), dt)
        elif act == ops.ACT_TANH:
            g = ops.tanh_bwd(dy, y)
        else:
            g = dy
        if g.shape[1] % 8 and dt != F32:
            # an output width that is not a multiple of 8 (2274 answers, 630 edges): zero-pad the row stride so the
            # backward products run on the tuned GEMM path (see pick_mode in gemm.hip) instead of the generic kernel
            # (cast + pad in one launch of ours; F.pad was a fill and a strided copy of the framework's)
            g = ops.pad_rows(g, g.shape[1] + (-g.shape[1]) % 8)
        elif g.dtype == F32 and dt != F32:
            g = ops.cast_from_f32(g, dt)
        if lin.bias is not None:
            _colsum(rt, g, lin.bias)
        probs, dx = [_p_wgrad(rt, g, x2, lin.weight)], None
        if ctx.needs_input_grad[2]:
            pd, dx = ops.p_dgrad(g, a.w(lin.weight))
            probs.append(pd)
        ops.gemm_group(dt, probs)
        if dx is not None:
            dx = dx.view(ctx.xshape)
        return (None, None, dx, None, None) + (None,) * ctx.np


class MLPFn(Function):
    """``Sequential(Linear, GeLU, LayerNorm)`` (node_fc / fusion_fc / first half of logit_fc,
    src/vqa/vqacpv2_model.py:63-105)."""

    @staticmethod
    def forward(ctx, rt, lin, ln, eps, x, *params):
        ctx.np = len(params)
        a = rt.arena
        x2 = x if x.dim() == 2 else x.reshape(-1, x.shape[-1])
        act, u = ops.linear_fwd(x2, a.w(lin.weight), lin.bias.data, act=ops.ACT_GELU, want_preact=True)
        y, z, stats = ops.ln_fwd(act, None, None, ln.weight.data, ln.bias.data, eps)
        ctx.rt, ctx.lin, ctx.ln = rt, lin, ln
        ctx.saved = (x2, u, z, stats)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        rt, lin, ln = ctx.rt, ctx.lin, ctx.ln
        x2, u, z, stats = ctx.saved
        a = rt.arena
        d_u, _ = ops.ln_bwd(dy.reshape(z.shape).contiguous(), z, stats, ln.weight.data, a.atomic_target(ln.weight),
                            a.atomic_target(ln.bias), a.atomic_target(lin.bias), gelu_aux=u, defer=rt.defer_list())
        probs, dx = [_p_wgrad(rt, d_u, x2, lin.weight)], None
        if ctx.needs_input_grad[4]:
            pd, dx = ops.p_dgrad(d_u, a.w(lin.weight))
            probs.append(pd)
        ops.gemm_group(d_u.dtype, probs)
        if dx is not None:
            dx = dx.view(ctx.xshape)
        return (None, None, None, None, dx) + (None,) * ctx.np


# --------------------------------------------------------------------------------- graph blocks
class GCNFn(Function):
    """GCN.forward (src/module/gcn.py:64-77): two GCNConv (LN(x + W (adj @ x)), gcn.py:22-29)
    and the jump-knowledge sum of three read-outs dropout_.5(LN(GeLU(W_k h_k + b_k)))."""

    @staticmethod
    def forward(ctx, rt, gcn, x, adj, *params):
        ctx.np = len(params)
        a = rt.arena
        B, N, H = x.shape
        hs = [x.contiguous()]
        convs = []
        # bf16 storage: W (A x) = A (x W^T) -- the product first, then aggregate + residual + LayerNorm as ONE kernel
        # (xggm_agg_residual_ln_bf16) instead of three launches; the backward follows the same order (d y = A^T d t)
        swapped = ops.agg_residual_ln_ok(x, N)
        for conv in gcn.gnn_layers:
            if swapped:
                y, _ = ops.linear_fwd(hs[-1].view(B * N, H), a.w(conv.ctx_layer.weight), None)
                h, z, stats = ops.agg_residual_ln(adj, y.view(B, N, H), hs[-1], conv.layer_norm.weight.data,
                                                  conv.layer_norm.bias.data, 1e-5)
                convs.append((y, z.view(B * N, H), stats))
            else:
                agg = ops.aggregate(adj, hs[-1])
                t, _ = ops.linear_fwd(agg.view(B * N, H), a.w(conv.ctx_layer.weight), None)
                h, z, stats = ops.ln_fwd(t, None, hs[-1].view(B * N, H), conv.layer_norm.weight.data,
                                         conv.layer_norm.bias.data, 1e-5)
                convs.append((agg, z, stats))
            hs.append(h.view(B, N, H))
        ctx.swapped = swapped
        p_ro = rt.p(gcn.dropout_p)
        pf = [ops.p_fwd(h.view(B * N, H), a.w(mlp[0].weight), mlp[0].bias.data, act=ops.ACT_GELU, want_preact=True)
              for mlp, h in zip(gcn.linear_prediction, hs)]
        ops.gemm_group(x.dtype, [t[0] for t in pf])  # the three read-out projections in one launch
        # the three dropout(LayerNorm(.)) terms and their sum: one launch, the sum held in fp32
        ret, stats = ops.ln_sum_fwd([t[1] for t in pf], [mlp[2].weight.data for mlp in gcn.linear_prediction],
                                    [mlp[2].bias.data for mlp in gcn.linear_prediction], 1e-5, p_post=p_ro, rng=rt.rng,
                                    sids=[gcn._sid + k for k in range(len(pf))])
        reads = [(u, act, st) for (_, act, u), st in zip(pf, stats)]
        ctx.rt, ctx.gcn, ctx.p = rt, gcn, p_ro
        ctx.saved = (adj, hs, convs, reads)
        return ret.view(B, N, H)

    @staticmethod
    def backward(ctx, d_ret):
        rt, gcn = ctx.rt, ctx.gcn
        adj, hs, convs, reads = ctx.saved
        a = rt.arena
        B, N, H = hs[0].shape
        d_ret = d_ret.contiguous().view(B * N, H)
        dh, probs = [], []
        # the three read-out LayerNorm backwards read the same d_ret: one launch
        d_us = ops.ln_bwd_group([dict(dy=d_ret, z=z, stats=stats, gamma=mlp[2].weight.data, dgamma=a.atomic_target(mlp[2].weight),
                                      dbeta=a.atomic_target(mlp[2].bias), dbias=a.atomic_target(mlp[0].bias),
                                      sid_post=gcn._sid + k, gelu_aux=u, defer=rt.defer_list())
                                 for k, (mlp, (u, z, stats)) in enumerate(zip(gcn.linear_prediction, reads))],
                                p_post=ctx.p, rng=rt.rng)
        for k, (mlp, h) in enumerate(zip(gcn.linear_prediction, hs)):
            d_u = d_us[k][0]
            pd, dhk = ops.p_dgrad(d_u, a.w(mlp[0].weight))
            probs += [_p_wgrad(rt, d_u, h.view(B * N, H), mlp[0].weight), pd]
            dh.append(dhk)
        ops.gemm_group(d_ret.dtype, probs)  # wgrad + dgrad of the three read-outs: two launches
        d_adj = None
        for k in reversed(range(len(gcn.gnn_layers))):
            conv = gcn.gnn_layers[k]
            agg, z, stats = convs[k]  # (swapped order: ``agg`` holds y = h_k W^T, the product taken first)
            # dh[k+1] is complete here; its residual branch adds into dh[k]
            d_t, _ = ops.ln_bwd(dh[k + 1], z, stats, conv.layer_norm.weight.data,
                                a.atomic_target(conv.layer_norm.weight), a.atomic_target(conv.layer_norm.bias), None,
                                d_res=dh[k], defer=rt.defer_list())
            if ctx.swapped:
                # t = A y, y = h_k W^T (agg holds y): d y = A^T d t; d W = d y^T h_k; d h_k += d y W; d A += d t y^T
                d_y = ops.aggregate(adj, d_t.view(B, N, H), mode=ops.AGG_TRANSPOSE).view(B * N, H)
                pd, _ = ops.p_dgrad(d_y, a.w(conv.ctx_layer.weight), into=dh[k])
                ops.gemm_group(d_t.dtype, [_p_wgrad(rt, d_y, hs[k].view(B * N, H), conv.ctx_layer.weight), pd])
                if ctx.needs_input_grad[3]:
                    d_adj = ops.bmm_nt(d_t.view(B, N, H), agg.view(B, N, H), into=d_adj)
                continue
            pd, d_agg = ops.p_dgrad(d_t, a.w(conv.ctx_layer.weight))
            ops.gemm_group(d_t.dtype, [_p_wgrad(rt, d_t, agg.view(B * N, H), conv.ctx_layer.weight), pd])
            d_agg = d_agg.view(B, N, H)
            ops.aggregate(adj, d_agg, mode=ops.AGG_TRANSPOSE, out=dh[k].view(B, N, H))
            if ctx.needs_input_grad[3]:
                d_adj = ops.bmm_nt(d_agg, hs[k], into=d_adj)  # the second layer's product adds into the first one's
        dx = dh[0].view(B, N, H) if ctx.needs_input_grad[2] else None
        return (None, None, dx, d_adj) + (None,) * ctx.np


class GINFn(Function):
    """GIN.forward with n_layers=1 (src/module/gin.py:68-87): conv = LN(GeLU(W (x + (1+eps) A x) + b)),
    two read-outs on [x, conv]."""

    @staticmethod
    def forward(ctx, rt, gin, x, adj, *params):
        ctx.np = len(params)
        a = rt.arena
        B, N, H = x.shape
        x = x.contiguous()
        conv = gin.gnn_convs[0]
        eps = conv.eps.data
        hin = ops.aggregate(adj, x, scale_ptr=eps, self_w=1.0)
        act, u = ops.linear_fwd(hin.view(B * N, H), a.w(conv.linear[0].weight), conv.linear[0].bias.data,
                                act=ops.ACT_GELU, want_preact=True)
        h1, z1, st1 = ops.ln_fwd(act, None, None, conv.linear[2].weight.data, conv.linear[2].bias.data, 1e-5)
        hs = [x, h1.view(B, N, H)]
        p_ro = rt.p(gin.dropout_p)
        pf = [ops.p_fwd(h.view(B * N, H), a.w(mlp[0].weight), mlp[0].bias.data, act=ops.ACT_GELU, want_preact=True)
              for mlp, h in zip(gin.linear_prediction, hs)]
        ops.gemm_group(x.dtype, [t[0] for t in pf])  # the read-out projections in one launch
        ret, stats = ops.ln_sum_fwd([t[1] for t in pf], [mlp[2].weight.data for mlp in gin.linear_prediction],
                                    [mlp[2].bias.data for mlp in gin.linear_prediction], 1e-5, p_post=p_ro, rng=rt.rng,
                                    sids=[gin._sid + k for k in range(len(pf))])
        reads = [(u, act, st) for (_, act, u), st in zip(pf, stats)]
        ctx.rt, ctx.gin, ctx.p = rt, gin, p_ro
        ctx.saved = (adj, hs, hin, u, z1, st1, reads)
        return ret.view(B, N, H)

    @staticmethod
    def backward(ctx, d_ret):
        rt, gin = ctx.rt, ctx.gin
        adj, hs, hin, u, z1, st1, reads = ctx.saved
        a = rt.arena
        B, N, H = hs[0].shape
        conv = gin.gnn_convs[0]
        d_ret = d_ret.contiguous().view(B * N, H)
        dh, probs = [], []
        d_us = ops.ln_bwd_group([dict(dy=d_ret, z=zk, stats=sk, gamma=mlp[2].weight.data, dgamma=a.atomic_target(mlp[2].weight),
                                      dbeta=a.atomic_target(mlp[2].bias), dbias=a.atomic_target(mlp[0].bias),
                                      sid_post=gin._sid + k, gelu_aux=uk, defer=rt.defer_list())
                                 for k, (mlp, (uk, zk, sk)) in enumerate(zip(gin.linear_prediction, reads))],
                                p_post=ctx.p, rng=rt.rng)
        for k, (mlp, h) in enumerate(zip(gin.linear_prediction, hs)):
            pd, dhk = ops.p_dgrad(d_us[k][0], a.w(mlp[0].weight))
            probs += [_p_wgrad(rt, d_us[k][0], h.view(B * N, H), mlp[0].weight), pd]
            dh.append(dhk)
        ops.gemm_group(d_ret.dtype, probs)  # weight and input gradients of the read-outs in one launch
        d_u, _ = ops.ln_bwd(dh[1], z1, st1, conv.linear[2].weight.data, a.atomic_target(conv.linear[2].weight),
                            a.atomic_target(conv.linear[2].bias), a.atomic_target(conv.linear[0].bias), gelu_aux=u,
                            defer=rt.defer_list())
        _wgrad(rt, d_u, hin.view(B * N, H), conv.linear[0].weight)
        d_hin = ops.linear_dgrad(d_u, a.w(conv.linear[0].weight)).view(B, N, H)
        # hin = x + (1+eps) A x
        ops.agg_dot(adj, hs[0], d_hin, a.atomic_target(conv.eps))
        dx = dh[0].view(B, N, H)
        ops.aggregate(adj, d_hin, mode=ops.AGG_TRANSPOSE, scale_ptr=conv.eps.data, self_w=1.0, out=dx)
        d_adj = None
        if ctx.needs_input_grad[3]:
            d_adj = ops.bmm_nt(ops.scale(d_hin, 1.0, conv.eps.data), hs[0])
        return (None, None, dx if ctx.needs_input_grad[2] else None, d_adj) + (None,) * ctx.np


class RegenFn(Function):
    """adjacency regeneration (src/module/graph_generative_modeling.py:225-228)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        S = ops.bmm_nt(x, x)
        adj, colmax, argmax = ops.adj_regen_fwd(S)
        ctx.saved = (x, S, adj, colmax, argmax)
        return adj

    @staticmethod
    def backward(ctx, d_adj):
        x, S, adj, colmax, argmax = ctx.saved
        dS = ops.adj_regen_bwd(d_adj.contiguous(), S, adj, colmax, argmax)
        return ops.aggregate(dS, x, mode=ops.AGG_SYMMETRIZE)


class AdjInitFn(Function):
    """scatter + symmetrise + Gaussian edge noise (src/vqa/vqacpv2.py:195-202,
    src/module/graph_utils.py:162-168).  Returns (adj_noisy, grad_log_noise)."""

    @staticmethod
    def forward(ctx, e, N, sigma, randn, rng, sid):
        adj, g = ops.adj_init_fwd(e.contiguous(), N, sigma, randn=randn, rng=rng, sid=sid)
        ctx.mark_non_differentiable(g)
        ctx.set_materialize_grads(False)  # no zero-filled "gradient" of g (a framework fill per pass)
        return adj, g

    @staticmethod
    def backward(ctx, d_adj, _):
        return (None if d_adj is None else ops.adj_init_bwd(d_adj.contiguous())), None, None, None, None, None


class FeatureNoiseFn(Function):
    """add_feature_noise_v2 (src/module/graph_utils.py:144-149)."""

    @staticmethod
    def forward(ctx, x, sigma, randn, rng, sid):
        out, g = ops.feature_noise(x.contiguous(), sigma, randn=randn, rng=rng, sid=sid)
        ctx.mark_non_differentiable(g)
        ctx.set_materialize_grads(False)
        return out, g

    @staticmethod
    def backward(ctx, d_out, _):
        return d_out, None, None, None, None


class PoolConcatFn(Function):
    """cat([x, tanh(nodes.mean(1))], -1) (src/vqa/vqacpv2.py:216-218)."""

    @staticmethod
    def forward(ctx, x, nodes):
        out = ops.pool_concat_fwd(x.contiguous(), nodes.contiguous())
        ctx.saved = (out, nodes.shape[1])
        return out

    @staticmethod
    def backward(ctx, d_out):
        out, N = ctx.saved
        return ops.pool_concat_bwd(d_out.contiguous(), out, N)


class BcastRowsFn(Function):
    """x.unsqueeze(1).repeat(1, N, 1) (src/vqa/vqacpv2.py:228)."""

    @staticmethod
    def forward(ctx, x, N):
        return ops.bcast_rows(x.contiguous(), N)

    @staticmethod
    def backward(ctx, g):
        return ops.sum_rows(g.contiguous()), None


# --------------------------------------------------------------------------------- losses
class DSMFn(Function):
    """loss_func (src/vqa/vqacpv2.py:48-51)."""

    @staticmethod
    def forward(ctx, score, g, sigma, scale=1.0, slot=None):
        score = score.contiguous()
        coef = scale * 0.5 * sigma ** 2 / score.numel()  # 0.5 s^2 / (d1 d2) * mean over the batch
        ctx.saved = (score, g, coef)
        return ops.dsm_fwd(score, g.contiguous(), coef, out=slot.t if slot is not None else None)

    @staticmethod
    def backward(ctx, gout):
        score, g, coef = ctx.saved
        return ops.dsm_bwd(score, g, gout.contiguous(), coef), None, None, None, None


class SymKLFn(Function):
    """compute_kl_loss (src/vqa/vqacpv2.py:54-61)."""

    @staticmethod
    def forward(ctx, x, y, scale=1.0, slot=None):
        x, y = x.contiguous(), y.contiguous()
        coef = scale / x.numel()
        ctx.saved = (x, y, coef)
        return ops.symkl_fwd(x, y, coef, out=slot.t if slot is not None else None)

    @staticmethod
    def backward(ctx, gout):
        x, y, coef = ctx.saved
        return ops.symkl_bwd(x, y, gout.contiguous(), coef, ctx.needs_input_grad[0], ctx.needs_input_grad[1]) + (None, None)


class BCEFn(Function):
    """nn.BCEWithLogitsLoss()(logit, target) (mean).  fp32 logits."""

    @staticmethod
    def forward(ctx, logit, target, scale=1.0, slot=None):
        logit, target = logit.contiguous(), target.contiguous()
        coef = scale / logit.numel()
        ctx.saved = (logit, target, coef)
        return ops.bce_fwd(logit, target, coef, out=slot.t if slot is not None else None)

    @staticmethod
    def backward(ctx, gout):
        logit, target, coef = ctx.saved
        return ops.bce_bwd(logit, target, gout.contiguous(), coef, F32), None, None, None


class LossSumFn(Function):
    """loss = sum of the (already weighted) loss terms of a pass (src/vqa/vqacpv2.py:220-221, 249-250): one kernel
    instead of a framework add per ``+``; the backward hands the upstream gradient to every term unchanged."""

    @staticmethod
    def forward(ctx, *terms):
        ctx.n = len(terms)
        return ops.add_scalars([t.contiguous() for t in terms])

    @staticmethod
    def backward(ctx, gout):
        return (gout,) * ctx.n


class GATFn(Function):
    """GAT.forward (src/module/gat.py:72-79): dropout(.5) on the input, n_head GATConv
    (gat.py:25-49) concatenated along the feature axis."""

    @staticmethod
    def forward(ctx, rt, gat, x, adj, *params):
        ctx.np = len(params)
        a = rt.arena
        B, N, H = x.shape
        x = x.contiguous()
        p = rt.p(gat.dropout)
        xd = ops.dropout(x, p, rt.rng, gat._sid) if p > 0 else x
        x2 = xd.view(B * N, H)
        heads = list(gat.gat_layers)
        D = heads[0].dim_hidden
        out = torch.empty((B * N, D * len(heads)), device=x.device, dtype=x.dtype)
        saved = []
        for k, conv in enumerate(heads):
            h, _ = ops.linear_fwd(x2, a.w(conv.linear_layer.weight), None)
            a2 = a.w(conv.attn_layer.weight).view(2, D)
            s, _ = ops.linear_fwd(h, a2, None, out_f32=True)          # [B*N, 2]: a1.h_i, a2.h_j
            att = ops.gat_att_fwd(s, adj, conv.alpha)
            pre = ops.aggregate(att, h.view(B, N, D))
            ops.elu_fwd(pre.view(B * N, D), out, k * D)
            saved.append((h, s, att))
        ctx.rt, ctx.gat, ctx.p = rt, gat, p
        ctx.saved = (x2, adj, out, saved)
        ctx.dims = (B, N, H, D)
        return out.view(B, N, -1)

    @staticmethod
    def backward(ctx, d_out):
        rt, gat = ctx.rt, ctx.gat
        x2, adj, out, saved = ctx.saved
        B, N, H, D = ctx.dims
        a = rt.arena
        d_out = d_out.contiguous().view(B * N, -1)
        dx = None
        for k, conv in enumerate(gat.gat_layers):
            h, s, att = saved[k]
            d_pre = ops.elu_bwd(d_out, out, k * D, D)
            d_att = ops.bmm_nt(d_pre.view(B, N, D), h.view(B, N, D))
            d_h = ops.aggregate(att, d_pre.view(B, N, D), mode=ops.AGG_TRANSPOSE).view(B * N, D)
            ds = ops.gat_att_bwd(d_att, att, s, adj, conv.alpha, h.dtype)
            a2 = a.w(conv.attn_layer.weight).view(2, D)
            ga, acc = a.target(conv.attn_layer.weight)
            ops.linear_wgrad(ds, h, ga.view(2, D), acc)
            d_h = ops.linear_dgrad(ds, a2, residual=d_h)
            _wgrad(rt, d_h, x2, conv.linear_layer.weight)
            g = ops.linear_dgrad(d_h, a.w(conv.linear_layer.weight), residual=dx)
            dx = g
        if ctx.p > 0:
            dx = ops.dropout(dx, ctx.p, rt.rng, gat._sid)
        return (None, None, dx.view(B, N, H) if ctx.needs_input_grad[2] else None, None) + (None,) * ctx.np
