"""CPU-only checks of the boundary: the shared library loads, exports every symbol that
include/xggm.h declares, validates arguments before launching anything, and the host-side
mirror keeps the reference's state_dict contract.  No kernel runs here."""
import ctypes
import os
import re

import pytest
import torch

from oracle import shapes


def test_library_exports_every_declared_symbol():
    from xggm_amd import _lib
    decl = _lib.parse_header()
    assert len(decl) >= 60
    for name in decl:
        assert hasattr(_lib.lib, name), name
    assert _lib.lib.xggm_version() == 100
    # every public symbol of the .so is declared in the header (no undocumented entry points)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (xggm_\w+)", out))
    assert exported == set(decl), exported ^ set(decl)


def test_header_cites_reference_call_sites():
    from xggm_amd import _lib
    src = open(_lib.HEADER_PATH).read()
    for cite in ["src/lxrt/modeling.py:355-373", "src/module/gcn.py:28", "src/lxrt/optimization.py:159-193",
                 "src/module/graph_generative_modeling.py:225-228", "src/vqa/vqacpv2.py:195-202"]:
        assert cite in src, cite


def test_argument_validation_fails_before_launch():
    from xggm_amd import _lib
    L = _lib.lib
    # empty GEMM, null pointers: rejected on the host, nothing is enqueued
    rc = L.xggm_gemm_f32(None, None, None, 0, 4, 4, 4, 1, 4, 1, 4, 1, 0, 0, 0, None, None, None, None, None, 0, 0, 0, 1.0,
                         None)
    assert rc != 0 and b"xggm_gemm" in L.xggm_last_error()
    assert L.xggm_ln_bwd_workspace_bytes(1152, 768) == 4 * 144 * 3 * 768  # one [3, H] fp32 partial per 8-row workgroup
    rc = L.xggm_ln_fwd_bf16(None, None, None, None, None, None, None, None, 4, 6, 1e-5, 0.0, 0.0, None, 0, 0, 0, 1.0, None)
    assert rc != 0 and b"multiple of 4" in L.xggm_last_error()
    rc = L.xggm_attn_fwd_f32(None, None, None, None, None, 1, 1, 65, 65, 64, 64, 64, 64, 64, 0.125, 0.0, None, 0, None)
    assert rc != 0 and b"64" in L.xggm_last_error()
    rc = L.xggm_adj_init_fwd(None, None, None, None, 2, 36, 0.0, None, 0, None)
    assert rc != 0


def test_triu_index_map_is_the_reference_enumeration():
    """k-th strict-upper-triangle entry, row-major -- the order of
    ``adj[ones.triu(1) == 1] = v.view(-1)`` (src/vqa/vqacpv2.py:195-198), bit-exact for all k."""
    from xggm_amd import _lib
    from oracle import xggm_oracle as O
    for N in (36, 64, 2, 5):
        ii, jj = O.triu_index_table(N)
        i, j = ctypes.c_int(), ctypes.c_int()
        for k in range(N * (N - 1) // 2):
            assert _lib.lib.xggm_triu_index(k, N, ctypes.byref(i), ctypes.byref(j)) == 0
            assert (i.value, j.value) == (int(ii[k]), int(jj[k]))
        assert _lib.lib.xggm_triu_index(N * (N - 1) // 2, N, ctypes.byref(i), ctypes.byref(j)) != 0


@pytest.mark.parametrize("gnn,nl", [("GCN", 2), ("GIN", 2)])
def test_state_dict_contract(gnn, nl):
    """module tree = the reference's key names and shapes."""
    from xggm_amd import param
    from xggm_amd.lxrt.modeling import BertConfig, VISUAL_CONFIG
    from xggm_amd.vqa.vqacpv2_model import VQAModel
    from xggm_amd.gqa.gqa_ood_model import GQAModel
    cfg = shapes.TINY
    VISUAL_CONFIG.set_visual_dims(cfg["feat_dim"], 4)
    a = param.parse_args(["--llayers", "2", "--xlayers", "2", "--rlayers", "1"])
    bc = BertConfig(cfg["vocab"], hidden_size=cfg["hidden"], num_attention_heads=cfg["heads"],
                    intermediate_size=cfg["inter"], max_position_embeddings=cfg["max_pos"])
    for cls in (VQAModel, GQAModel):
        m = cls(29, gnn=gnn, n_layers=nl, args=a, config=bc)
        sd = m.state_dict()
        sh = shapes.model_shapes(cfg, 29, gnn, nl)
        assert sorted(sd) == sorted(sh)
        assert all(tuple(sd[k].shape) == tuple(v) for k, v in sh.items())
        assert m.lxrt_encoder.dim == cfg["hidden"] and m.lxrt_encoder.max_seq_length == 20
    with pytest.raises(ModuleNotFoundError):
        VQAModel(29, gnn="XYZ", args=a, config=bc)


@pytest.mark.parametrize("gnn,nl", [("GCN", 2), ("GIN", 2)])
def test_init_bert_weights_coverage_and_statistics(gnn, nl):
    """which modules ``init_bert_weights`` reaches and what it leaves there (reference:
    src/lxrt/modeling.py:734-747 applied by LXRTFeatureExtraction.__init__ :1068-1076 to the WHOLE encoder and by
    VQAModel.__init__ to ``logit_fc`` only, src/vqa/vqacpv2_model.py:63-69): Linear / Embedding weights ~ N(0, 0.02)
    (the padding row of the embedding tables included: ``normal_`` overwrites it), Linear biases 0, LayerNorm (1, 0).
    The generator, ``encoder_adj``, ``node_fc`` and ``fusion_fc`` keep torch's default initialisation
    (kaiming-uniform weights bounded by 1/sqrt(fan_in), uniform non-zero biases)."""
    import math
    import torch.nn as nn
    from xggm_amd import param
    from xggm_amd.lxrt.modeling import BertConfig, VISUAL_CONFIG, BertLayerNorm
    from xggm_amd.vqa.vqacpv2_model import VQAModel
    cfg = shapes.TINY
    VISUAL_CONFIG.set_visual_dims(cfg["feat_dim"], 4)
    a = param.parse_args(["--llayers", "2", "--xlayers", "2", "--rlayers", "1"])
    bc = BertConfig(cfg["vocab"], hidden_size=cfg["hidden"], num_attention_heads=cfg["heads"],
                    intermediate_size=cfg["inter"], max_position_embeddings=cfg["max_pos"])
    assert bc.initializer_range == 0.02
    torch.manual_seed(1234)
    m = VQAModel(31, gnn=gnn, n_layers=nl, args=a, config=bc)

    def bert_initialised(root, label):
        ws, n_lin, n_emb, n_ln = [], 0, 0, 0
        for name, mod in root.named_modules():
            if isinstance(mod, nn.Linear):
                n_lin += 1
                ws.append(mod.weight.detach().flatten())
                assert mod.bias is None or float(mod.bias.abs().max()) == 0.0, (label, name)
            elif isinstance(mod, nn.Embedding):
                n_emb += 1
                ws.append(mod.weight.detach().flatten())
                assert float(mod.weight[0].detach().abs().max()) > 0.0, (label, name)  # padding row re-drawn, as in the reference
            elif isinstance(mod, (BertLayerNorm, nn.LayerNorm)):
                n_ln += 1
                assert bool((mod.weight == 1).all()) and bool((mod.bias == 0).all()), (label, name)
        w = torch.cat(ws).double()
        assert abs(float(w.mean())) < 4 * 0.02 / math.sqrt(w.numel()), label
        assert abs(float(w.std()) - 0.02) < 0.02 * 4 / math.sqrt(2 * w.numel()) + 1e-5, (label, float(w.std()))
        # a normal, not a truncated normal or a uniform: the 4-sigma tail is populated, nothing beyond 6 sigma
        assert float(w.abs().max()) > 4 * 0.02 * (w.numel() > 1e5) and float(w.abs().max()) < 6.5 * 0.02, label
        return n_lin, n_emb, n_ln

    n_lin, n_emb, n_ln = bert_initialised(m.lxrt_encoder.model, "encoder")
    # 2 l-layers + 1 r-layer (6 Linear each), 2 x-layers (cross 4 + two self-attention 4 + two FFN 2), visn_fc +
    # box_fc, pooler
    assert (n_lin, n_emb) == (3 * 6 + 2 * 16 + 2 + 1, 3)
    assert bert_initialised(m.logit_fc, "logit_fc")[0] == 2

    for label, root in (("generator", m.generator), ("encoder_adj", m.encoder_adj), ("node_fc", m.node_fc),
                        ("fusion_fc", m.fusion_fc)):
        lins = [mod for mod in root.modules() if isinstance(mod, nn.Linear)]
        assert lins, label
        for lin in lins:
            bound = 1.0 / math.sqrt(lin.in_features)
            w = lin.weight.detach().double()
            assert float(w.abs().max()) <= bound * (1 + 1e-6), label          # kaiming_uniform(a = sqrt 5)
            assert abs(float(w.std()) - bound / math.sqrt(3)) < 0.1 * bound, label  # uniform, not N(0, 0.02)
            if lin.bias is not None:
                assert float(lin.bias.abs().max()) > 0.0 and float(lin.bias.abs().max()) <= bound * (1 + 1e-6), label
        for mod in root.modules():
            if isinstance(mod, (BertLayerNorm, nn.LayerNorm)):
                assert bool((mod.weight == 1).all()) and bool((mod.bias == 0).all()), label


def test_no_cpu_fallback():
    """the product refuses CPU tensors instead of silently computing elsewhere"""
    from xggm_amd import param
    from xggm_amd.lxrt.modeling import BertConfig, VISUAL_CONFIG
    from xggm_amd.vqa.vqacpv2_model import VQAModel
    cfg = shapes.TINY
    VISUAL_CONFIG.set_visual_dims(cfg["feat_dim"], 4)
    a = param.parse_args(["--llayers", "1", "--xlayers", "1", "--rlayers", "1"])
    bc = BertConfig(cfg["vocab"], hidden_size=128, num_attention_heads=2, intermediate_size=256,
                    max_position_embeddings=32)
    m = VQAModel(5, args=a, config=bc)
    ids = torch.zeros(2, 20, dtype=torch.long)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(2, 36, 64), torch.zeros(2, 36, 4), (ids, ids, ids))


def test_schedule_and_tokenised_features():
    from xggm_amd.lxrt.optimization import warmup_linear
    from xggm_amd.lxrt.entry import convert_sents_to_features
    from oracle import xggm_oracle as O
    for x in (0.0, 0.05, 0.1, 0.5, 1.0, 1.2):
        assert warmup_linear(x, 0.1) == O.warmup_linear(x, 0.1)

    class Tok:
        def tokenize(self, s):
            return s.split()

        def convert_tokens_to_ids(self, toks):
            return [{"[CLS]": 101, "[SEP]": 102}.get(t, 7) for t in toks]

    f = convert_sents_to_features(["what is this", " ".join(["w"] * 40)], 20, Tok())
    assert f[0].input_ids[:5] == [101, 7, 7, 7, 102] and sum(f[0].input_mask) == 5 and len(f[0].input_ids) == 20
    assert f[1].input_ids[0] == 101 and f[1].input_ids[19] == 102 and sum(f[1].input_mask) == 20
    assert f[0].segment_ids == [0] * 20


def test_wordpiece_tokenizer(tmp_path):
    from xggm_amd.lxrt.tokenization import BertTokenizer
    from xggm_amd.lxrt.entry import convert_sents_to_features
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "what", "is", "the", "man", "hold", "##ing", "?", "un",
             "##aff", "##able", ",", "cafe"]
    f = tmp_path / "vocab.txt"
    f.write_text("\n".join(vocab) + "\n")
    tok = BertTokenizer(str(f), do_lower_case=True)
    assert tok.tokenize("What is the man HOLDING?") == ["what", "is", "the", "man", "hold", "##ing", "?"]
    assert tok.tokenize("unaffable, café zzz") == ["un", "##aff", "##able", ",", "cafe", "[UNK]"]
    feats = convert_sents_to_features(["What is the man holding?"], 20, tok)
    assert feats[0].input_ids[:9] == [2, 5, 6, 7, 8, 9, 10, 11, 3] and sum(feats[0].input_mask) == 9


def test_tokenizer_and_features_match_reference_golden():
    """host tokeniser + [CLS]/[SEP]/padding (src/lxrt/tokenization.py:72-348, entry.py:37-72) against what the
    reference's own BertTokenizer / convert_sents_to_features produced over a synthetic vocabulary
    (tests/golden/make_golden.py tokenizer_case): lower-casing, accent stripping, punctuation and CJK
    splitting, greedy WordPiece, [UNK] for unknown / over-long words, truncation, empty input."""
    import json
    from xggm_amd.lxrt.tokenization import BertTokenizer
    from xggm_amd.lxrt.entry import convert_sents_to_features
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = json.load(open(os.path.join(here, "tokenizer.json"), encoding="utf-8"))
    tok = BertTokenizer(os.path.join(here, "vocab_small.txt"), do_lower_case=True)
    for sent, want in zip(g["sents"], g["tokens"]):
        assert tok.tokenize(sent.strip()) == want, sent
    from xggm_amd.lxrt.entry import SentenceBatcher
    for L, want in g["features"].items():
        feats = convert_sents_to_features(g["sents"], int(L), tok)
        got = [[f.input_ids, f.input_mask, f.segment_ids] for f in feats]
        assert got == want, L
        # the cached, single-buffer batch path of the training loop gives the same tensors (twice: cold and cached)
        sb = SentenceBatcher(tok, int(L))
        for _ in range(2):
            hb = sb.host_batch(g["sents"])
            assert hb.shape == (3, len(want), int(L))
            assert hb[0].tolist() == [w[0] for w in want] and hb[1].tolist() == [w[1] for w in want]
            assert hb[2].tolist() == [w[2] for w in want]


def test_lxmert_snapshot_loading_matches_reference_golden(tmp_path, monkeypatch):
    """AnswerTable / load_lxmert_qa (src/pretrain/qa_answer_table.py:8-198) against what the reference's own
    function left in a model (tests/golden/make_golden.py answer_table_case): answer normalisation, rows
    copied for known answers, rows zeroed for unknown ones, the rest of the head and the whole encoder
    taken from the snapshot, ``module.`` prefixes and foreign keys ignored."""
    import json
    import numpy as np
    from xggm_amd import param, synth
    from xggm_amd.lxrt.modeling import BertConfig, VISUAL_CONFIG
    from xggm_amd.vqa.vqacpv2_model import VQAModel
    from xggm_amd.pretrain.qa_answer_table import AnswerTable, load_lxmert_qa
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(here, "answer_table.json")))
    g = np.load(os.path.join(here, "answer_table.npz"))
    seed, labels = meta["seed"], meta["labels"]
    cfg = shapes.TINY
    VISUAL_CONFIG.set_visual_dims(cfg["feat_dim"], 4)
    a = param.parse_args(["--llayers", str(cfg["l_layers"]), "--xlayers", str(cfg["x_layers"]), "--rlayers",
                          str(cfg["r_layers"])])
    bc = BertConfig(cfg["vocab"], hidden_size=cfg["hidden"], num_attention_heads=cfg["heads"],
                    intermediate_size=cfg["inter"], max_position_embeddings=cfg["max_pos"])
    model = VQAModel(len(labels), args=a, config=bc)
    # the synthetic snapshot of the golden script, regenerated from its seed recipe
    hid, n_pre = cfg["hidden"], len(meta["table"])
    snap = {"module." + k: torch.from_numpy(synth.seeded_param("snap." + k, tuple(v.shape), seed + 1))
            for k, v in model.lxrt_encoder.model.state_dict().items()}
    for k, shp in {"0.weight": (2 * hid, hid), "0.bias": (2 * hid,), "2.weight": (2 * hid,), "2.bias": (2 * hid,),
                   "3.weight": (n_pre, 2 * hid), "3.bias": (n_pre,)}.items():
        snap["module.answer_head.logit_fc." + k] = torch.from_numpy(
            synth.seeded_param("snap.answer_head." + k, shp, seed + 1))
    snap["module.obj_predict_head.decoder.weight"] = torch.zeros(3, 3)
    os.makedirs(tmp_path / "data" / "lxmert")
    (tmp_path / "data" / "lxmert" / "all_ans.json").write_text(json.dumps(meta["table"]))
    torch.save(snap, str(tmp_path / "snap_LXRT.pth"))
    monkeypatch.chdir(tmp_path)  # the table path is relative to the working directory, as in the reference
    table = AnswerTable()
    assert [table.convert_ans(x) for x in labels] == meta["converted"]
    assert AnswerTable(dsets=["gqa"]).all_answers() == meta["gqa_only"]
    assert table.num_answers == n_pre and table.used("gray") and not table.used("grey")
    load_lxmert_qa(str(tmp_path / "snap"), model, labels)
    sd = model.state_dict()
    for k in ("0.weight", "0.bias", "2.weight", "2.bias", "3.weight", "3.bias"):
        assert np.array_equal(sd["logit_fc." + k].numpy(), g["head." + k]), k
    zero_rows = [i for i, c in enumerate(meta["converted"]) if not table.used(c)]
    assert zero_rows and all(not sd["logit_fc.3.weight"][i].any() for i in zero_rows)
    enc = model.lxrt_encoder.model.state_dict()
    assert sorted(enc) == list(g["enc_names"])
    for k, n, d in zip(g["enc_names"], g["enc_norms"], g["enc_dots"]):
        v = enc[str(k)].double()
        probe = torch.from_numpy(synth._rng(seed, "probe:" + str(k)).standard_normal(tuple(v.shape), dtype=np.float32))
        assert abs(float(v.norm()) - n) <= 1e-9 * max(1.0, n), k
        assert abs(float((v * probe.double()).sum()) - d) <= 1e-9 * max(1.0, abs(d)), k
    # a snapshot that lacks an encoder tensor is refused, as in the reference (:188-190)
    del snap["module.bert.pooler.dense.bias"]
    torch.save(snap, str(tmp_path / "short_LXRT.pth"))
    with pytest.raises(AssertionError):
        load_lxmert_qa(str(tmp_path / "short"), model, labels)


def test_measured_tile_table_is_well_formed():
    """x-ggm_amd/gemm_tiles_gfx950.json (tools/tune_gemm.py): launch signature -> tile pin; pins are tiles the library
    has (1: 64x64, 2: 128x64, 3: 128x128 on four waves, 4: 128x128 on eight waves, 7 / 8 / 9: the role k-loop on 128x128 / 128x64 / 64x64), signatures parse"""
    import json
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "x-ggm_amd", "gemm_tiles_gfx950.json")
    t = json.load(open(path))
    assert t["tiles"] and "tune_gemm" in t["what"]
    for sig, pin in t["tiles"].items():
        assert pin in (1, 2, 3, 4, 7, 8, 9), (sig, pin)
        dtp, probs = sig.split("|")
        assert dtp in ("bf16", "f32", "e4m3")
        for p in probs.split("+"):
            assert re.fullmatch(r"\d+x\d+x\d+(b\d+)?:[01][01][a-zA-Z]*", p), p
