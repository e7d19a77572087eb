"""CPU oracle for the X-GGM training-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional, dictionary-of-tensors restatement (plain PyTorch on the
CPU, fp32 or fp64) of the algorithm the reference executes for one training
iteration: LXMERT cross-modal encoder + graph-generative module + losses + grad
clip + BertAdam.  It is *the checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``xggm_amd``) never imports anything from ``oracle/`` and fails
loudly when its HIP extension is missing.

Parity status: PINNED.  Every function below is checked against outputs of the
reference's own classes/functions, imported from ``/root/reference/src`` inside the
build container by ``tests/golden/make_golden.py`` (committed), whose outputs are the
``tests/golden/*.npz`` fixtures (``tests/test_oracle_golden.py``).  The reference holds
no tests or golden vectors of its own.

Parameters are passed as a flat ``dict[str, Tensor]`` keyed by the reference's
``state_dict`` names, ``pre`` being the key prefix of the
sub-module.  Linear weights are ``[out, in]``.

Stochastic pieces take their randomness as INPUTS so that the HIP path and the
oracle can be compared on identical draws:
  * Gaussian noise of the denoising-score-matching step: ``randn`` argument;
  * dropout: ``drop`` callback ``drop(x, p, tag) -> x'`` (identity by default,
    i.e. eval mode / p = 0).
"""
import math

import torch
import torch.nn.functional as F

IDENT = lambda x, p, tag: x  # noqa: E731  dropout hook: eval mode


# --------------------------------------------------------------------------- basics
def gelu(x):
    """erf-GELU.  ref: src/lxrt/modeling.py:116-124"""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, w, b, eps):
    """nn.LayerNorm over the last dim (biased variance)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def linear(P, pre, x, bias=True):
    y = x @ P[pre + "weight"].t()
    if bias:
        y = y + P[pre + "bias"]
    return y


# --------------------------------------------------------------------------- LXMERT
def bert_embeddings(P, pre, input_ids, token_type_ids, drop=IDENT):
    """word + position + token-type gather-sum -> LN(1e-12) -> dropout(.1);
    padding_idx=0 on all three tables => row 0 gets zero gradient.
    ref: src/lxrt/modeling.py:298-313"""
    T = input_ids.shape[1]
    pos = torch.arange(T, device=input_ids.device).unsqueeze(0).expand_as(input_ids)
    # all three tables are nn.Embedding(..., padding_idx=0): row 0 is looked up like any
    # other row but never receives gradient (modeling.py:284-290).
    e = (F.embedding(input_ids, P[pre + "word_embeddings.weight"], padding_idx=0)
         + F.embedding(pos, P[pre + "position_embeddings.weight"], padding_idx=0)
         + F.embedding(token_type_ids, P[pre + "token_type_embeddings.weight"], padding_idx=0))
    e = layer_norm(e, P[pre + "LayerNorm.weight"], P[pre + "LayerNorm.bias"], 1e-12)
    return drop(e, 0.1, pre + "dropout")


def bert_attention(P, pre, hidden, context, mask, n_heads, drop=IDENT):
    """multi-head attention core incl. Q/K/V projections.
    ref: src/lxrt/modeling.py:344-374"""
    B, Sq, H = hidden.shape
    Sk = context.shape[1]
    d = H // n_heads
    q = linear(P, pre + "query.", hidden).view(B, Sq, n_heads, d).permute(0, 2, 1, 3)
    k = linear(P, pre + "key.", context).view(B, Sk, n_heads, d).permute(0, 2, 1, 3)
    v = linear(P, pre + "value.", context).view(B, Sk, n_heads, d).permute(0, 2, 1, 3)
    s = q @ k.transpose(-1, -2) / math.sqrt(d)
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    p = drop(p, 0.1, pre + "dropout")
    o = (p @ v).permute(0, 2, 1, 3).reshape(B, Sq, H)
    return o


def bert_att_output(P, pre, hidden, input_tensor, drop=IDENT):
    """LN(dropout(W h + b) + residual).  ref: modeling.py:384-388 and :441-445"""
    h = drop(linear(P, pre + "dense.", hidden), 0.1, pre + "dropout")
    return layer_norm(h + input_tensor, P[pre + "LayerNorm.weight"],
                      P[pre + "LayerNorm.bias"], 1e-12)


def bert_selfatt_layer(P, pre, x, mask, n_heads, drop=IDENT):
    """ref: modeling.py:403-414"""
    o = bert_attention(P, pre + "self.", x, x, mask, n_heads, drop)
    return bert_att_output(P, pre + "output.", o, x, drop)


def bert_crossatt_layer(P, pre, x, ctx, ctx_mask, n_heads, drop=IDENT):
    """ref: modeling.py:391-400"""
    o = bert_attention(P, pre + "att.", x, ctx, ctx_mask, n_heads, drop)
    return bert_att_output(P, pre + "output.", o, x, drop)


def bert_intermediate(P, pre, x):
    """ref: modeling.py:428-431"""
    return gelu(linear(P, pre + "dense.", x))


def bert_layer(P, pre, x, mask, n_heads, drop=IDENT):
    """self-attention then FFN.  ref: modeling.py:455-459"""
    a = bert_selfatt_layer(P, pre + "attention.", x, mask, n_heads, drop)
    i = bert_intermediate(P, pre + "intermediate.", a)
    return bert_att_output(P, pre + "output.", i, a, drop)


def lxrtx_layer(P, pre, lang, lang_mask, visn, visn_mask, n_heads, drop=IDENT):
    """cross-modality layer: ONE shared ``visual_attention`` weight set used in both
    directions (both from the pre-update inputs), then per-modality self-attention,
    then per-modality FFN.  ref: modeling.py:485-527"""
    l1 = bert_crossatt_layer(P, pre + "visual_attention.", lang, visn, visn_mask, n_heads, drop)
    v1 = bert_crossatt_layer(P, pre + "visual_attention.", visn, lang, lang_mask, n_heads, drop)
    l2 = bert_selfatt_layer(P, pre + "lang_self_att.", l1, lang_mask, n_heads, drop)
    v2 = bert_selfatt_layer(P, pre + "visn_self_att.", v1, visn_mask, n_heads, drop)
    li = bert_intermediate(P, pre + "lang_inter.", l2)
    vi = bert_intermediate(P, pre + "visn_inter.", v2)
    l3 = bert_att_output(P, pre + "lang_output.", li, l2, drop)
    v3 = bert_att_output(P, pre + "visn_output.", vi, v2, drop)
    return l3, v3


def visual_feat_encoder(P, pre, feats, boxes, drop=IDENT):
    """(LN(W_f feat + b) + LN(W_b box + b)) / 2 -> dropout(.1)
    ref: modeling.py:546-556"""
    x = layer_norm(linear(P, pre + "visn_fc.", feats), P[pre + "visn_layer_norm.weight"],
                   P[pre + "visn_layer_norm.bias"], 1e-12)
    y = layer_norm(linear(P, pre + "box_fc.", boxes), P[pre + "box_layer_norm.weight"],
                   P[pre + "box_layer_norm.bias"], 1e-12)
    return drop((x + y) / 2, 0.1, pre + "dropout")


def lxrt_encoder(P, pre, lang, lang_mask, feats, boxes, visn_mask, cfg, drop=IDENT):
    """visn_fc -> l_layers on lang -> r_layers on visn -> x_layers.
    ref: modeling.py:585-605"""
    nh = cfg["heads"]
    visn = visual_feat_encoder(P, pre + "visn_fc.", feats, boxes, drop)
    for i in range(cfg["l_layers"]):
        lang = bert_layer(P, pre + "layer.%d." % i, lang, lang_mask, nh, drop)
    for i in range(cfg["r_layers"]):
        visn = bert_layer(P, pre + "r_layers.%d." % i, visn, visn_mask, nh, drop)
    for i in range(cfg["x_layers"]):
        lang, visn = lxrtx_layer(P, pre + "x_layers.%d." % i, lang, lang_mask, visn,
                                 visn_mask, nh, drop)
    return lang, visn


def bert_pooler(P, pre, lang):
    """tanh(W lang[:,0] + b).  ref: modeling.py:614-620"""
    return torch.tanh(linear(P, pre + "dense.", lang[:, 0]))


def lxrt_model(P, pre, input_ids, token_type_ids, attention_mask, feats, boxes, cfg,
               visual_attention_mask=None, drop=IDENT):
    """LXRTModel.forward: additive masks (1-m)*-1e4, embeddings, encoder, pooler.
    Returns ((lang, visn), pooled).  ref: modeling.py:904-952 (``pre`` ends in 'bert.')"""
    dt = P[pre + "pooler.dense.weight"].dtype
    ext = (1.0 - attention_mask[:, None, None, :].to(dt)) * -10000.0
    vext = None
    if visual_attention_mask is not None:
        vext = (1.0 - visual_attention_mask[:, None, None, :].to(dt)) * -10000.0
    emb = bert_embeddings(P, pre + "embeddings.", input_ids, token_type_ids, drop)
    lang, visn = lxrt_encoder(P, pre + "encoder.", emb, ext, feats, boxes, vext, cfg, drop)
    return (lang, visn), bert_pooler(P, pre + "pooler.", lang)


# --------------------------------------------------------------------------- heads
def mlp_gelu_ln(P, pre, x, eps):
    """``Sequential(Linear, GeLU, LayerNorm)`` = node_fc / fusion_fc / GNN readouts.
    ref: src/vqa/vqacpv2_model.py:95-105, src/module/gcn.py:43-62"""
    h = gelu(linear(P, pre + "0.", x))
    return layer_norm(h, P[pre + "2.weight"], P[pre + "2.bias"], eps)


def logit_fc(P, pre, x):
    """Linear(768,1536) GeLU LN(1e-12) Linear(1536,A).  ref: vqacpv2_model.py:63-69"""
    return linear(P, pre + "3.", mlp_gelu_ln(P, pre, x, 1e-12))


def encoder_adj(P, pre, x):
    """Linear(768,630) + Sigmoid.  ref: vqacpv2_model.py:91-94"""
    return torch.sigmoid(linear(P, pre + "0.", x))


def node_fc(P, pre, x):
    """ref: vqacpv2_model.py:95-99 (nn.LayerNorm default eps 1e-5)"""
    return mlp_gelu_ln(P, pre, x, 1e-5)


def fusion_fc(P, pre, x):
    """ref: vqacpv2_model.py:101-105"""
    return mlp_gelu_ln(P, pre, x, 1e-5)


# --------------------------------------------------------------------------- graphs
def triu_index_table(n):
    """k-th strict-upper-triangle entry -> (i, j), row-major: (0,1),(0,2)...(n-2,n-1).
    This is the enumeration order of ``adj[ones.triu(1) == 1] = v.view(-1)``.
    ref: src/vqa/vqacpv2.py:195-198"""
    ii, jj = [], []
    for i in range(n):
        for j in range(i + 1, n):
            ii.append(i)
            jj.append(j)
    return torch.tensor(ii), torch.tensor(jj)


def adj_init(e, n):
    """scatter [B, n(n-1)/2] into the strict upper triangle and symmetrise.
    ref: vqacpv2.py:195-199"""
    B = e.shape[0]
    ii, jj = triu_index_table(n)
    a = torch.zeros(B, n, n, dtype=e.dtype, device=e.device)
    a[:, ii, jj] = e
    return a + a.transpose(1, 2)


def add_edge_noise_v2(adjs, randn, sigma):
    """n = triu(randn,1)*sigma; n += n^T; g = -n/sigma^2; adjs + n.
    ref: src/module/graph_utils.py:162-168"""
    noise = randn.triu(diagonal=1) * sigma
    noise = noise + noise.transpose(-1, -2)
    return adjs + noise, -noise / (sigma ** 2)


def add_feature_noise_v2(feats, randn, sigma):
    """ref: graph_utils.py:144-149"""
    noise = randn * sigma
    return feats + noise, -noise / (sigma ** 2)


def regen_adj(x):
    """adjacency regeneration shared by all three generators:
    S = x x^T; S[b,i,:] /= max_r S[b,r,i]; sigmoid; zero the diagonal.
    ``adj.max(dim=1)[0].unsqueeze(-1)`` makes row i divided by the COLUMN-i maximum.
    ref: src/module/graph_generative_modeling.py:225-228"""
    s = torch.bmm(x, x.transpose(1, 2))
    s = s / s.max(dim=1)[0].unsqueeze(-1)
    s = torch.sigmoid(s)
    return s.triu(1) + s.tril(-1)


def gcn_conv(P, pre, x, adj):
    """LN(x + W_ctx (adj @ x)), no bias, dropout p=0.  ref: src/module/gcn.py:22-29"""
    h = x + torch.bmm(adj, x) @ P[pre + "ctx_layer.weight"].t()
    return layer_norm(h, P[pre + "layer_norm.weight"], P[pre + "layer_norm.bias"], 1e-5)


def gcn(P, pre, x, adj, n_conv=2, drop=IDENT):
    """jump-knowledge sum of readouts of [x, conv0(x), conv1(conv0(x))].
    ref: gcn.py:64-77"""
    hs = [x]
    for k in range(n_conv):
        x = gcn_conv(P, pre + "gnn_layers.%d." % k, x, adj)
        hs.append(x)
    ret = 0.0
    for k, h in enumerate(hs):
        ret = ret + drop(mlp_gelu_ln(P, pre + "linear_prediction.%d." % k, h, 1e-5), 0.5,
                         pre + "readout%d" % k)
    return ret


def gcn_generator(P, pre, x, adj, n_layers=2, drop=IDENT):
    """ref: graph_generative_modeling.py:214-233"""
    for l in range(n_layers):
        x = gcn(P, pre + "gnn_layers.%d." % l, x, adj, 2, drop)
        adj = regen_adj(x)
    return x, adj


def gin_conv(P, pre, x, adj):
    """x + ((1+eps) A) @ x -> Linear -> GeLU -> LN.  ref: src/module/gin.py:21-34"""
    h = x + ((1 + P[pre + "eps"]) * adj) @ x
    return mlp_gelu_ln(P, pre + "linear.", h, 1e-5)


def gin(P, pre, x, adj, n_conv=1, drop=IDENT):
    """ref: gin.py:68-87"""
    hs = [x]
    for k in range(n_conv):
        x = gin_conv(P, pre + "gnn_convs.%d." % k, x, adj)
        hs.append(x)
    ret = 0.0
    for k, h in enumerate(hs):
        ret = ret + drop(mlp_gelu_ln(P, pre + "linear_prediction.%d." % k, h, 1e-5), 0.5,
                         pre + "readout%d" % k)
    return ret


def gin_generator(P, pre, x, adj, n_layers=2, drop=IDENT):
    """ref: graph_generative_modeling.py:177-196"""
    for l in range(n_layers):
        x = gin(P, pre + "gnn_layers.%d." % l, x, adj, 1, drop)
        adj = regen_adj(x)
    return x, adj


def gat_conv(P, pre, x, adj, alpha=0.2):
    """h = W x; e_ij = LeakyReLU(a^T [h_i || h_j]); mask adj==0 -> -9e15; row softmax;
    elu(att @ h).  ref: src/module/gat.py:25-49"""
    h = x @ P[pre + "linear_layer.weight"].t()
    D = h.shape[-1]
    a = P[pre + "attn_layer.weight"].view(-1)
    e = (h @ a[:D]).unsqueeze(2) + (h @ a[D:]).unsqueeze(1)  # [B,N,N]: self i, neighbour j
    e = F.leaky_relu(e, alpha)
    e = e.masked_fill(adj == 0, -9e15)
    att = torch.softmax(e, dim=-1)
    return F.elu(torch.bmm(att, h))


def gat(P, pre, x, adj, n_head=2, drop=IDENT):
    """input dropout(.5) then head concat.  ref: gat.py:72-79"""
    x = drop(x, 0.5, pre + "dropout")
    return torch.cat([gat_conv(P, pre + "gat_layers.%d." % k, x, adj) for k in range(n_head)],
                     dim=2)


def gat_generator(P, pre, x, adj, n_layers=1, drop=IDENT):
    """ref: graph_generative_modeling.py:250-269 (only shape-valid for n_layers == 1)"""
    for l in range(n_layers):
        x = gat(P, pre + "gnn_layers.%d." % l, x, adj, 2, drop)
        adj = regen_adj(x)
    return x, adj


GENERATORS = {"GCN": gcn_generator, "GIN": gin_generator, "GAT": gat_generator}


# --------------------------------------------------------------------------- losses
def loss_func(score, grad_log_q_noise, sigma):
    """denoising score matching: 0.5 sigma^2 mean_b sum_ij (s-g)^2 / (d1 d2).
    ref: src/vqa/vqacpv2.py:48-51"""
    cur = 0.5 * sigma ** 2 * ((score - grad_log_q_noise) ** 2).sum(dim=[-1, -2]).mean()
    return cur / (score.shape[-1] * score.shape[-2])


def compute_kl_loss(x, y):
    """symmetric KL of last-dim softmaxes, mean over ALL elements.
    ref: vqacpv2.py:54-61"""
    lpx = torch.log_softmax(x, dim=-1)
    lpy = torch.log_softmax(y, dim=-1)
    px, py = lpx.exp(), lpy.exp()
    return (py * (lpy - lpx) + px * (lpx - lpy)).mean()


def bce_with_logits_mean(logit, target):
    """nn.BCEWithLogitsLoss() (mean over all elements).  ref: vqacpv2.py:131"""
    return (torch.clamp(logit, min=0) - logit * target
            + torch.log1p(torch.exp(-logit.abs()))).mean()


# --------------------------------------------------------------------------- steps
ENC = "lxrt_encoder.model.bert."


def model_forward(P, batch, cfg, drop=IDENT):
    """VQAModel.forward / GQAModel.forward -> ((lang, visn), mask, x).
    ref: src/vqa/vqacpv2_model.py:122-131, src/lxrt/entry.py:190-206"""
    (lang, visn), x = lxrt_model(P, ENC, batch["input_ids"], batch["segment_ids"],
                                 batch["input_mask"], batch["feats"], batch["boxes"], cfg,
                                 None, drop)
    return (lang, visn), batch["input_mask"], x


def plain_step_loss(P, batch, cfg, drop=IDENT):
    """step A.  ref: vqacpv2.py:170-173"""
    _, _, x = model_forward(P, batch, cfg, drop)
    logit = logit_fc(P, "logit_fc.", x)
    A = batch["target"].shape[1]
    return bce_with_logits_mean(logit, batch["target"]) * A, {"logit": logit, "x": x}


def ggm_step_loss(P, batch, cfg, branch, sigma=1.0, kl_weight=8.0, gnn="GCN", n_layers=2,
                  drop=IDENT):
    """step B.  branch 'rel' (relation generation, vqacpv2.py:195-221) or 'node'
    (representation generation, vqacpv2.py:228-250).  ``kl_weight`` is 8 for VQA-CP and
    12 for GQA-OOD (gqa_ood.py:197).  Noise comes from batch['randn_adj'|'randn_node']."""
    (lang, visn), _, x = model_forward(P, batch, cfg, drop)
    A = batch["target"].shape[1]
    N = visn.shape[1]
    adj_true = batch["adj_true"]
    adj_true = adj_true.triu(1) + adj_true.tril(-1)
    gen = GENERATORS[gnn]
    out = {"x": x, "visn": visn}
    if branch == "rel":
        adj0 = adj_init(encoder_adj(P, "encoder_adj.", x), N)
        adj_noise, g = add_edge_noise_v2(adj0, batch["randn_adj"], sigma)
        node_feats, adj_gen = gen(P, "generator.", visn, adj_noise, n_layers, drop)
        loss_grad = loss_func(adj_gen, g, sigma)
        d_loss = compute_kl_loss(adj_true, adj_gen) * A
        loss_sm = kl_weight * d_loss + loss_grad
        w_sm = 6.0
        out.update(adj0=adj0, adj_gen=adj_gen)
    else:
        nf = node_fc(P, "node_fc.", x.unsqueeze(1).repeat(1, N, 1))
        nf, g = add_feature_noise_v2(nf, batch["randn_node"], sigma)
        node_feats, _ = gen(P, "generator.", nf, adj_true, n_layers, drop)
        d_loss = compute_kl_loss(node_feats, visn) * A
        loss_grad = loss_func(node_feats, g, sigma)
        loss_sm = 0.15 * d_loss + 6 * loss_grad
        w_sm = 1.1
    x_gen = fusion_fc(P, "fusion_fc.", torch.cat([x, torch.tanh(node_feats.mean(1))], dim=-1))
    logit = logit_fc(P, "logit_fc.", x_gen)
    bce = bce_with_logits_mean(logit, batch["target"]) * A
    loss = bce + w_sm * loss_sm
    out.update(logit=logit, node_feats=node_feats, bce=bce, d_loss=d_loss, loss_grad=loss_grad,
               loss_sm=loss_sm)
    return loss, out


# --------------------------------------------------------------------------- optimiser
def warmup_linear(x, warmup=0.002):
    """ref: src/lxrt/optimization.py:42-48"""
    if x < warmup:
        return x / warmup
    return max((x - 1.0) / (warmup - 1.0), 0)


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_(params, 5.) semantics (L2, coef clamped to 1).
    Returns (total_norm, scaled grads).  ref call site: vqacpv2.py:175"""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(
        next(iter(grads.values())).dtype)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, {k: g * coef for k, g in grads.items()}


def bert_adam_step(P, G, M, V, step, lr_of, t_total, warmup=0.1, b1=0.9, b2=0.999, e=1e-6,
                   weight_decay=0.01):
    """BertAdam.step for the params that have a gradient (others untouched):
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; u = m/(sqrt(v)+e) + wd p;
    p -= lr*sched(step/t_total, warmup) * u.  No bias correction; wd on every param.
    ``lr_of(name)`` gives the group's base lr.  ``step[name]`` is the per-param counter.
    ref: src/lxrt/optimization.py:116-203"""
    for k, g in G.items():
        M[k] = M[k] * b1 + (1 - b1) * g
        V[k] = V[k] * b2 + (1 - b2) * g * g
        u = M[k] / (V[k].sqrt() + e) + weight_decay * P[k]
        if t_total != -1:
            lr = lr_of(k) * warmup_linear(step[k] / t_total, warmup)
        else:
            lr = lr_of(k)
        P[k] = P[k] - lr * u
        step[k] += 1
    return P, M, V, step


def vqa_lr_of(base_lr):
    """two param groups: non-encoder params at 4*lr, encoder at lr.
    ref: vqacpv2.py:118-128"""
    return lambda name: base_lr if name.startswith("lxrt_encoder.") else 4 * base_lr


def train_pass(P, M, V, step, batch, cfg, kind, base_lr, t_total, **kw):
    """one fwd + bwd + clip_grad_norm_(5.) + BertAdam pass; ``kind`` in
    {'plain','rel','node'}.  Params whose grad is None (unused in the graph) are skipped,
    as torch >= 2 (set_to_none zero_grad) does with the reference loop.
    Returns (loss, total_norm, grads-before-clip, extras)."""
    Pl = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    if kind == "plain":
        loss, out = plain_step_loss(Pl, batch, cfg, kw.get("drop", IDENT))
    else:
        loss, out = ggm_step_loss(Pl, batch, cfg, kind, **kw)
    names = list(Pl)
    gs = torch.autograd.grad(loss, [Pl[k] for k in names], allow_unused=True)
    G = {k: g for k, g in zip(names, gs) if g is not None}
    total, Gc = clip_grad_norm(G, 5.0)
    bert_adam_step(P, Gc, M, V, step, vqa_lr_of(base_lr), t_total)
    return loss.detach(), total, G, out


# --------------------------------------------------------------------------- preprocessing / data (section 8f)
def compute_cosin_sim_v2(matrix1, matrix2):
    """adj_cos[i, j] = cosine_similarity(matrix1[i], matrix2[j]) for j >= i; adj_cos + adj_cos^T.
    ref: data/preprocess/vqa/compute_adjacency.py:38-45 (restated as the same double loop)"""
    n = matrix1.shape[0]
    adj_cos = torch.zeros((n, n), dtype=torch.float32)
    for i in range(n):
        for j in range(n):
            if j >= i:
                adj_cos[i, j] = torch.cosine_similarity(matrix1[i], matrix2[j], dim=0, eps=1e-6)
    return adj_cos + adj_cos.transpose(0, 1)


def adjacency_of(matrix_class, matrix_attribute):
    """ref: compute_adjacency.py:89-90"""
    m = compute_cosin_sim_v2(matrix_class, matrix_attribute)
    return m / m.max()


def normalize_boxes(boxes, img_w, img_h):
    """ref: src/vqa/vqacpv2_data.py:108-117 (numpy; incl. the (img_h,) tuple the reference divides by)"""
    import numpy as np
    boxes = boxes.copy()
    boxes[:, (0, 2)] /= img_w
    boxes[:, (1, 3)] /= (img_h,)
    np.testing.assert_array_less(boxes, 1 + 1e-5)
    np.testing.assert_array_less(-boxes, 0 + 1e-5)
    return boxes


def vqa_target(num_answers, labels, scores):
    """ref: vqacpv2_data.py:120-123"""
    target = torch.zeros(num_answers)
    for ans, score in zip(labels, scores):
        target[ans] = score
    return target


def vqa_score(quesid2ans, id2datum, ans2label):
    """VQAEvaluator.evaluate, ref: vqacpv2_data.py:134-142"""
    score = 0.
    for quesid, ans in quesid2ans.items():
        datum = id2datum[quesid]
        label = dict(zip(datum['label'], datum['score']))
        aid = ans2label[ans]
        if aid in label:
            score += label[aid]
    return score / len(quesid2ans)
