"""LXMERT snapshot -> fine-tuning model, with the answer-head surgery of the reference
(src/pretrain/qa_answer_table.py:8-81 AnswerTable, :84-122 load_lxmert_qa_bert, :125-198 load_lxmert_qa).

The snapshot is read with ``weights_only=True``; after the copy the bf16 shadow weights of the arena are
refreshed (the model's ``load_state_dict`` does it), so the next launch sees the loaded values."""
import json

import torch

_NORMALISED = {"a man": "man", "the man": "man", "a woman": "woman", "the woman": "woman", "grey": "gray"}
_NORMALISED.update({w: str(i + 1) for i, w in enumerate(
    ["one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten"])})
_ARTICLES = ("a ", "an ", "the ")


class AnswerTable:
    """the pre-training answer vocabulary (``data/lxmert/all_ans.json``: [{"ans": str, "dsets": [...]}, ...]),
    optionally restricted to the answers some dataset of ``dsets`` uses.  ref :8-81"""
    ANS_CONVERT = _NORMALISED

    def __init__(self, dsets=None, path="data/lxmert/all_ans.json"):
        with open(path) as f:
            self.all_ans = json.load(f)
        wanted = None if dsets is None else set(dsets)
        self.anss = [e["ans"] for e in self.all_ans if wanted is None or wanted.intersection(e["dsets"])]
        self.ans_set = set(self.anss)
        self._ans2id_map = {a: i for i, a in enumerate(self.anss)}
        if len(self._ans2id_map) != len(self.anss):
            raise AssertionError("duplicate answers in the answer table")

    def convert_ans(self, ans):
        """lower-case, drop one trailing full stop, then each leading article in turn, then the
        number-word / synonym map (ref :46-59; the order matters: "a the man" -> "man")."""
        if not ans:
            return ""
        ans = ans.lower()
        if ans.endswith("."):
            ans = ans[:-1].strip()
        for art in _ARTICLES:
            if ans.startswith(art):
                ans = ans[len(art):].strip()
        return _NORMALISED.get(ans, ans)

    def ans2id(self, ans):
        return self._ans2id_map[ans]

    def id2ans(self, ans_id):
        return self.anss[ans_id]

    def ans2id_map(self):
        return dict(self._ans2id_map)

    def id2ans_map(self):
        return list(self.anss)

    def used(self, ans):
        return ans in self.ans_set

    def all_answers(self):
        return list(self.anss)

    @property
    def num_answers(self):
        return len(self.anss)


def _split_snapshot(path):
    """snapshot keys without the DataParallel ``module.`` prefix, split into encoder and answer head"""
    loaded = torch.load("%s_LXRT.pth" % path, map_location="cpu", weights_only=True)
    bert, head = {}, {}
    for key, value in loaded.items():
        key = key.replace("module.", "")
        if key.startswith("bert."):
            bert[key] = value
        elif key.startswith("answer_head."):
            head[key.replace("answer_head.", "")] = value
    return bert, head


def _load_encoder(model, bert):
    missing = set(model.lxrt_encoder.model.state_dict().keys()) - set(bert.keys())
    assert len(missing) == 0, sorted(missing)[:5]
    model.lxrt_encoder.model.load_state_dict(bert, strict=False)


def load_lxmert_qa_bert(path, model):
    """encoder weights only (ref :84-122)"""
    print("*" * 80)
    print(f"Load QA pre-trained LXMERT from {path} ")
    bert, _ = _split_snapshot(path)
    _load_encoder(model, bert)
    _refresh(model)


def load_lxmert_qa(path, model, label2ans, answer_table=None):
    """encoder weights + answer head; the last linear layer's row of every fine-tuning answer that the
    pre-training vocabulary knows (after ``convert_ans``) is copied from the snapshot, every other row
    (weight and bias) is zeroed (ref :125-198).  ``label2ans``: list or {label: answer}."""
    print("*" * 80)
    print(f"Load QA pre-trained LXMERT from {path} ")
    bert, head = _split_snapshot(path)
    table = answer_table if answer_table is not None else AnswerTable()
    if isinstance(label2ans, list):
        label2ans = dict(enumerate(label2ans))
    own = model.state_dict()
    weight, bias = own["logit_fc.3.weight"].clone(), own["logit_fc.3.bias"].clone()
    labels = sorted(label2ans)
    pre_ids = [table._ans2id_map.get(table.convert_ans(label2ans[l]), -1) for l in labels]
    known = [(l, i) for l, i in zip(labels, pre_ids) if i >= 0]
    unknown = [l for l, i in zip(labels, pre_ids) if i < 0]
    if known:
        dst = torch.tensor([l for l, _ in known])
        src = torch.tensor([i for _, i in known])
        weight[dst] = head["logit_fc.3.weight"][src].to(weight)
        bias[dst] = head["logit_fc.3.bias"][src].to(bias)
    if unknown:
        weight[torch.tensor(unknown)] = 0.
        bias[torch.tensor(unknown)] = 0.
    print("Loaded %d answers from LXRTQA pre-training and %d not" % (len(known), len(unknown)))
    print()
    head["logit_fc.3.weight"], head["logit_fc.3.bias"] = weight, bias
    _load_encoder(model, bert)
    extra = set(head.keys()) - set(own.keys())
    assert len(extra) == 0, sorted(extra)[:5]
    model.load_state_dict(head, strict=False)
    _refresh(model)


def _refresh(model):
    from ..runtime import sync_weights
    sync_weights(model)
