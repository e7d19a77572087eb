"""GQA-OOD datasets (src/gqa/gqa_ood_data.py:19-186) on the shard format: the twins of vqa/vqacpv2_data.py with
the GQA annotation layout -- ``img_id`` / ``sent`` keys, ``label`` = {answer string: score} mapped through
``ans2label``, one datum kept per (answer in the table) occurrence (:88-93)."""
import os

import numpy as np
import torch

from ..vqa.vqacpv2_data import VQADataset, VQATorchDataset, VQAEvaluator, load_json, ShardReader

GQA_DATA_ROOT = 'data/gqa_ood/'
VG_GQA_IMGFEAT_ROOT = 'data/vg_gqa_imgfeat/'


class GQADataset(VQADataset):
    """ref :19-51"""
    img_key = 'img_id'
    sent_key = 'sent'

    def __init__(self, splits: str, root=GQA_DATA_ROOT, data=None, ans2label=None, label2ans=None):
        if data is None:
            data = []
            for split in splits.split(','):
                data.extend(load_json(os.path.join(root, "%s.json" % split)))
        super().__init__(splits, root, data, ans2label, label2ans)
        for ans, label in self.ans2label.items():
            assert self.label2ans[label] == ans


class GQATorchDataset(VQATorchDataset):
    """ref :54-147"""

    def __init__(self, dataset, shard=None, imgfeat_root=VG_GQA_IMGFEAT_ROOT, tiny=False, fast=False):
        torch.utils.data.Dataset.__init__(self)
        self.raw_dataset = dataset
        if shard is None:
            shard = os.path.join(imgfeat_root, '%s_obj36.xgs' % dataset.splits[0])
        self.shard = shard if isinstance(shard, ShardReader) else ShardReader(shard)
        self.data = []
        for datum in dataset.data:  # ref :88-93: once per label that is in the answer table
            for ans, score in datum['label'].items():
                if ans in dataset.ans2label and datum['img_id'] in self.shard.row_of:
                    self.data.append(datum)
        if tiny:
            self.data = self.data[:512]
        self.rows = np.asarray([self.shard.row_of[d['img_id']] for d in self.data], dtype=np.int64)
        print("Use %d data in torch dataset" % (len(self.data)))

    def target_of(self, datum, out=None):
        t = out if out is not None else torch.zeros(self.raw_dataset.num_answers)
        for ans, score in datum['label'].items():
            t[self.raw_dataset.ans2label[ans]] = score
        return t


    def fill_target(self, datum, row):
        for ans, score in datum['label'].items():
            row[self.raw_dataset.ans2label[ans]] = score


class GQAEvaluator(VQAEvaluator):
    """ref :150-186: the score of the predicted answer STRING in the datum's label dict"""

    def evaluate(self, quesid2ans: dict):
        score = 0.
        for quesid, ans in quesid2ans.items():
            datum = self.dataset.id2datum[quesid]
            label = datum['label']
            if ans in label:
                score += label[ans]
        return score / len(quesid2ans)

    @staticmethod
    def dump_result(quesid2ans: dict, path):
        import json
        with open(path, 'w') as f:
            result = [{'questionId': ques_id, 'prediction': ans} for ques_id, ans in quesid2ans.items()]
            json.dump(result, f, indent=4, sort_keys=True)
