"""The CPU oracle against the fixtures produced by the reference itself
(tests/golden/make_golden.py).  fp32 on both sides: tolerance 2e-5 relative on
tensors, 1e-4 on (norm, projection) gradient summaries."""
import numpy as np
import pytest
import torch

from oracle import shapes
from oracle import xggm_oracle as O
from xggm_amd import synth
from helpers import load_golden, golden_cfg, seeded_params, probe, batch_tensors, rel_err

TOL = 2e-5


def check_grad_summary(g, G, seed, tol=1e-4, names_key="grad_names", norms="grad_norms",
                       dots="grad_dots"):
    names = [str(n) for n in g[names_key]]
    assert sorted(names) == sorted(G.keys())
    for n, rn, rd in zip(names, g[norms], g[dots]):
        v = G[n].double()
        assert abs(float(v.norm()) - rn) <= tol * rn + 1e-5, n  # key.bias grads are ~0
        d = float((v * probe(n, v.shape, seed).double()).sum())
        assert abs(d - rd) <= 20 * tol * rn + 1e-4, n


@pytest.mark.parametrize("tag", ["enc_tiny", "enc_full"])
def test_encoder(tag):
    g = load_golden(tag)
    cfg, B, seed = golden_cfg(g), int(g["B"]), int(g["seed"])
    sh = shapes.encoder_shapes(cfg)
    assert sorted(sh) == sorted(str(n) for n in g["grad_names"])  # state_dict contract
    P = {k: v.requires_grad_(True) for k, v in seeded_params(sh, seed).items()}
    b = batch_tensors(synth.vqa_batch(B, A=8, F=cfg["feat_dim"], vocab=cfg["vocab"], seed=seed))
    (lang, visn), pooled = O.lxrt_model(P, O.ENC, b["input_ids"], b["segment_ids"],
                                        b["input_mask"], b["feats"], b["boxes"], cfg)
    assert rel_err(lang, torch.from_numpy(g["lang"])) < TOL
    assert rel_err(visn, torch.from_numpy(g["visn"])) < TOL
    assert rel_err(pooled, torch.from_numpy(g["pooled"])) < TOL
    loss = ((lang * probe("lang", lang.shape, seed)).sum()
            + (visn * probe("visn", visn.shape, seed)).sum()
            + (pooled * probe("pooled", pooled.shape, seed)).sum())
    loss.backward()
    check_grad_summary(g, {k: v.grad for k, v in P.items()}, seed)


@pytest.mark.parametrize("tag", ["gen_gcn36", "gen_gin36", "gen_gat36", "gen_gcn64",
                                 "gen_gcn_small"])
def test_generator(tag):
    g = load_golden(tag)
    kind, H, N, B = str(g["kind"]), int(g["H"]), int(g["N"]), int(g["B"])
    nl, seed = int(g["n_layers"]), int(g["seed"])
    sh = shapes.generator_shapes(kind, H, nl)
    P = {k: v.requires_grad_(True) for k, v in seeded_params(sh, seed).items()}
    xn, an = synth.generator_inputs(tag, kind, B, N, H, seed)
    x = torch.from_numpy(xn).requires_grad_(True)
    adj = torch.from_numpy(an).requires_grad_(True)
    xo, ao = O.GENERATORS[kind](P, "generator.", x, adj, nl)
    assert rel_err(xo, torch.from_numpy(g["x_out"])) < TOL
    assert rel_err(ao, torch.from_numpy(g["adj_out"])) < TOL
    loss = (xo * probe("xo", xo.shape, seed)).sum() + (ao * probe("ao", ao.shape, seed)).sum()
    loss.backward()
    assert rel_err(x.grad, torch.from_numpy(g["dx"])) < 1e-4
    if kind != "GAT":
        assert rel_err(adj.grad, torch.from_numpy(g["dadj"])) < 1e-4
    check_grad_summary(g, {k: v.grad for k, v in P.items()}, seed)


def test_pieces():
    g = load_golden("pieces")
    H, A, N, B, seed = (int(g[k]) for k in ("H", "A", "N", "B", "seed"))
    P = seeded_params(shapes.head_shapes(H, A, N * (N - 1) // 2), seed)
    x = torch.from_numpy(g["x"])
    assert rel_err(O.logit_fc(P, "logit_fc.", x), torch.from_numpy(g["logit"])) < TOL
    e = O.encoder_adj(P, "encoder_adj.", x)
    assert rel_err(e, torch.from_numpy(g["enc_adj"])) < TOL
    assert rel_err(O.node_fc(P, "node_fc.", x), torch.from_numpy(g["node_fc"])) < TOL
    assert rel_err(O.fusion_fc(P, "fusion_fc.", torch.from_numpy(g["x2"])),
                   torch.from_numpy(g["fusion"])) < TOL
    # adjacency init: index table bit-exact, values exact given the same e
    ii, jj = O.triu_index_table(N)
    assert np.array_equal(ii.numpy(), g["triu_i"]) and np.array_equal(jj.numpy(), g["triu_j"])
    adj0 = O.adj_init(torch.from_numpy(g["enc_adj"]), N)
    assert torch.equal(adj0, torch.from_numpy(g["adj0"]))
    sigma = float(g["sigma"])
    an, ag = O.add_edge_noise_v2(torch.from_numpy(g["adj0"]), torch.from_numpy(g["randn_adj"]),
                                 sigma)
    assert rel_err(an, torch.from_numpy(g["edge_noisy"])) < 1e-6
    assert rel_err(ag, torch.from_numpy(g["edge_grad"])) < 1e-6
    fn, fg = O.add_feature_noise_v2(torch.from_numpy(g["feats"]),
                                    torch.from_numpy(g["randn_feat"]), sigma)
    assert rel_err(fn, torch.from_numpy(g["feat_noisy"])) < 1e-6
    assert rel_err(fg, torch.from_numpy(g["feat_grad"])) < 1e-6
    # losses and their input gradients
    s = torch.from_numpy(g["dsm_score"]).requires_grad_(True)
    l = O.loss_func(s, torch.from_numpy(g["dsm_g"]), sigma)
    l.backward()
    assert abs(float(l) - float(g["dsm"])) < 1e-5 * abs(float(g["dsm"]))
    assert rel_err(s.grad, torch.from_numpy(g["dsm_dscore"])) < 1e-5
    kx = torch.from_numpy(g["kl_x"]).requires_grad_(True)
    ky = torch.from_numpy(g["kl_y"]).requires_grad_(True)
    l = O.compute_kl_loss(kx, ky)
    l.backward()
    assert abs(float(l) - float(g["kl"])) < 1e-5 * abs(float(g["kl"]))
    assert rel_err(kx.grad, torch.from_numpy(g["kl_dx"])) < 1e-4
    assert rel_err(ky.grad, torch.from_numpy(g["kl_dy"])) < 1e-4
    bl = torch.from_numpy(g["bce_logit"]).requires_grad_(True)
    l = O.bce_with_logits_mean(bl, torch.from_numpy(g["bce_target"])) * A
    l.backward()
    assert abs(float(l) - float(g["bce"])) < 1e-5 * abs(float(g["bce"]))
    assert rel_err(bl.grad, torch.from_numpy(g["bce_dlogit"])) < 1e-5


def test_bert_adam():
    g = load_golden("pieces")
    P = {"a": torch.from_numpy(g["adam_p1_0"]).clone(), "b": torch.from_numpy(g["adam_p2_0"]).clone()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    step = {"a": 0, "b": 0}
    lr_of = lambda n: 4e-3 if n == "a" else 1e-3  # noqa: E731
    for s in range(6):
        if s:
            lrs = [lr_of(n) * O.warmup_linear(step[n] / 10, 0.1) for n in ("a", "b")]
            assert np.allclose(lrs, g["adam_lrs"][s], rtol=1e-12)
        G = {"a": torch.from_numpy(g["adam_g1"][s]), "b": torch.from_numpy(g["adam_g2"][s])}
        O.bert_adam_step(P, G, M, V, step, lr_of, 10, 0.1)
        assert rel_err(P["a"], torch.from_numpy(g["adam_p1"][s])) < 1e-6
        assert rel_err(P["b"], torch.from_numpy(g["adam_p2"][s])) < 1e-6


@pytest.mark.parametrize("tag", ["train_tiny_gcn", "train_tiny_gin"])
def test_train_passes(tag):
    g = load_golden(tag)
    cfg, B, A, seed = golden_cfg(g), int(g["B"]), int(g["A"]), int(g["seed"])
    gnn = str(g["gnn"])
    sh = shapes.model_shapes(cfg, A, gnn, 2)
    assert sorted(sh) == sorted(str(n) for n in g["param_names"])
    P = seeded_params(sh, seed)
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    step = {k: 0 for k in P}
    b = synth.vqa_batch(B, A=A, F=cfg["feat_dim"], vocab=cfg["vocab"], seed=seed)
    b["randn_node"] = synth.randn_nodes(B, 36, cfg["hidden"], seed)
    b = batch_tensors(b)
    for i, kind in enumerate(["plain", "rel", "node", "plain"]):
        kw = {} if kind == "plain" else dict(sigma=float(g["sigma"]), kl_weight=8.0, gnn=gnn)
        loss, total, _, out = O.train_pass(P, M, V, step, b, cfg, kind, float(g["lr"]),
                                           int(g["t_total"]), **kw)
        assert abs(float(loss) - float(g["loss%d" % i])) < 2e-4 * abs(float(g["loss%d" % i]))
        assert abs(float(total) - float(g["norm%d" % i])) < 2e-3 * float(g["norm%d" % i])
        assert rel_err(out["logit"], torch.from_numpy(g["logit%d" % i])) < 2e-3
    for n, rn, rd in zip(g["param_names"], g["param_norms"], g["param_dots"]):
        v = P[str(n)].double()
        assert abs(float(v.norm()) - rn) <= 1e-4 * (rn + 1e-6), n
