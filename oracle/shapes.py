"""state_dict names and shapes of the reference model (TEST INFRASTRUCTURE ONLY).

Enumerates, without importing the reference, every parameter the reference's
``VQAModel`` / ``GQAModel`` registers (SURVEY.md section 8b "state_dict contract").
Checked against the key lists stored in the golden fixtures, which were read off the
reference's own ``named_parameters()``.
"""

ENC = "lxrt_encoder.model.bert."


def _lin(d, pre, n_out, n_in, bias=True):
    d[pre + "weight"] = (n_out, n_in)
    if bias:
        d[pre + "bias"] = (n_out,)


def _ln(d, pre, n):
    d[pre + "weight"] = (n,)
    d[pre + "bias"] = (n,)


def _att(d, pre, H):
    for k in ("query.", "key.", "value."):
        _lin(d, pre + k, H, H)


def _att_out(d, pre, H, n_in=None):
    _lin(d, pre + "dense.", H, n_in or H)
    _ln(d, pre + "LayerNorm.", H)


def _bert_layer(d, pre, H, I):
    _att(d, pre + "attention.self.", H)
    _att_out(d, pre + "attention.output.", H)
    _lin(d, pre + "intermediate.dense.", I, H)
    _att_out(d, pre + "output.", H, I)


def encoder_shapes(cfg, pre=ENC):
    """ref: src/lxrt/modeling.py:278-620, 894-902"""
    H, I = cfg["hidden"], cfg["inter"]
    d = {}
    d[pre + "embeddings.word_embeddings.weight"] = (cfg["vocab"], H)
    d[pre + "embeddings.position_embeddings.weight"] = (cfg["max_pos"], H)
    d[pre + "embeddings.token_type_embeddings.weight"] = (2, H)
    _ln(d, pre + "embeddings.LayerNorm.", H)
    e = pre + "encoder."
    _lin(d, e + "visn_fc.visn_fc.", H, cfg["feat_dim"])
    _ln(d, e + "visn_fc.visn_layer_norm.", H)
    _lin(d, e + "visn_fc.box_fc.", H, 4)
    _ln(d, e + "visn_fc.box_layer_norm.", H)
    for i in range(cfg["l_layers"]):
        _bert_layer(d, e + "layer.%d." % i, H, I)
    for i in range(cfg["x_layers"]):
        x = e + "x_layers.%d." % i
        _att(d, x + "visual_attention.att.", H)
        _att_out(d, x + "visual_attention.output.", H)
        for m in ("lang", "visn"):
            _att(d, x + m + "_self_att.self.", H)
            _att_out(d, x + m + "_self_att.output.", H)
        for m in ("lang", "visn"):
            _lin(d, x + m + "_inter.dense.", I, H)
            _att_out(d, x + m + "_output.", H, I)
    for i in range(cfg["r_layers"]):
        _bert_layer(d, e + "r_layers.%d." % i, H, I)
    _lin(d, pre + "pooler.dense.", H, H)
    return d


def _mlp(d, pre, n_out, n_in):
    _lin(d, pre + "0.", n_out, n_in)
    _ln(d, pre + "2.", n_out)


def generator_shapes(kind, H, n_layers, pre="generator."):
    """ref: src/module/graph_generative_modeling.py:162-269, gcn.py:33-62, gin.py:11-66,
    gat.py:7-70"""
    d = {}
    for l in range(n_layers):
        g = pre + "gnn_layers.%d." % l
        if kind == "GCN":
            for k in range(2):
                _lin(d, g + "gnn_layers.%d.ctx_layer." % k, H, H, bias=False)
                _ln(d, g + "gnn_layers.%d.layer_norm." % k, H)
            for k in range(3):
                _mlp(d, g + "linear_prediction.%d." % k, H, H)
        elif kind == "GIN":
            d[g + "gnn_convs.0.eps"] = (1,)
            _mlp(d, g + "gnn_convs.0.linear.", H, H)
            for k in range(2):
                _mlp(d, g + "linear_prediction.%d." % k, H, H)
        elif kind == "GAT":
            for k in range(2):
                _lin(d, g + "gat_layers.%d.linear_layer." % k, H, H, bias=False)
                _lin(d, g + "gat_layers.%d.attn_layer." % k, 1, 2 * H, bias=False)
        else:
            raise ModuleNotFoundError(kind)
    return d


def head_shapes(H, A, n_adj=630):
    """ref: src/vqa/vqacpv2_model.py:63-69, 91-105"""
    d = {}
    _mlp(d, "logit_fc.", 2 * H, H)
    _lin(d, "logit_fc.3.", A, 2 * H)
    _lin(d, "encoder_adj.0.", n_adj, H)
    _mlp(d, "node_fc.", H, H)
    _mlp(d, "fusion_fc.", H, 2 * H)
    return d


def model_shapes(cfg, A, gnn="GCN", n_layers=2, n_adj=630):
    d = encoder_shapes(cfg)
    d.update(head_shapes(cfg["hidden"], A, n_adj))
    d.update(generator_shapes(gnn, cfg["hidden"], n_layers))
    return d


TINY = dict(hidden=128, heads=2, inter=256, vocab=64, max_pos=32, feat_dim=64,
            l_layers=2, x_layers=2, r_layers=1)
FULL = dict(hidden=768, heads=12, inter=3072, vocab=30522, max_pos=512, feat_dim=2048,
            l_layers=9, x_layers=5, r_layers=5)
