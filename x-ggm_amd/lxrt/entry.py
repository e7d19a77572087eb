"""LXRTEncoderFeature: host tokenisation -> ids/mask/segment -> encoder, with the reference's
constructor/attributes (src/lxrt/entry.py:161-238)."""
import os

import torch
import torch.nn as nn

from .modeling import BertConfig, LXRTFeatureExtraction as VisualBertForLXRFeature, VISUAL_CONFIG
from ..runtime import sync_weights


class InputFeatures(object):
    """A single set of features of data.  ref: src/lxrt/entry.py:29-34"""

    def __init__(self, input_ids, input_mask, segment_ids):
        self.input_ids = input_ids
        self.input_mask = input_mask
        self.segment_ids = segment_ids


def convert_sents_to_features(sents, max_seq_length, tokenizer):
    """[CLS] tokens [SEP], zero-padded to max_seq_length.  ref: src/lxrt/entry.py:37-72"""
    features = []
    for sent in sents:
        tokens_a = tokenizer.tokenize(sent.strip())
        if len(tokens_a) > max_seq_length - 2:
            tokens_a = tokens_a[:(max_seq_length - 2)]
        tokens = ["[CLS]"] + tokens_a + ["[SEP]"]
        input_ids = tokenizer.convert_tokens_to_ids(tokens)
        n = len(input_ids)
        padding = [0] * (max_seq_length - n)
        features.append(InputFeatures(input_ids=input_ids + padding, input_mask=[1] * n + padding,
                                      segment_ids=[0] * max_seq_length))
    return features


def set_visual_config(args):
    """ref: src/lxrt/entry.py:75-78"""
    VISUAL_CONFIG.l_layers = args.llayers
    VISUAL_CONFIG.x_layers = args.xlayers
    VISUAL_CONFIG.r_layers = args.rlayers


class LXRTEncoderFeature(nn.Module):
    """ref: src/lxrt/entry.py:161-238.  Differences forced by running offline: the BERT
    vocabulary and ``bert-base-uncased`` weights cannot be downloaded, so the model is built
    from ``config`` (default: bert-base sizes) with ``init_bert_weights`` and the tokenizer
    comes from ``tokenizer`` / ``args.vocab_path`` (None: callers pass token ids)."""

    def __init__(self, args, max_seq_length, mode='x', config=None, tokenizer=None):
        super().__init__()
        self.max_seq_length = max_seq_length
        set_visual_config(args)
        if tokenizer is None and getattr(args, "vocab_path", None):
            from .tokenization import BertTokenizer
            tokenizer = BertTokenizer(args.vocab_path, do_lower_case=True)
        self.tokenizer = tokenizer
        if config is None:
            config = BertConfig(30522)
        self.model = VisualBertForLXRFeature(config, mode=mode)
        self._dim = config.hidden_size

    def multi_gpu(self):
        raise NotImplementedError("single-process nn.DataParallel is replaced by one process per GPU: "
                                  "see xggm_amd.dist.DataParallelTrainer")

    @property
    def dim(self):
        return self._dim

    def forward(self, sents, feats, visual_attention_mask=None):
        device = feats[0].device
        if isinstance(sents, (tuple, list)) and len(sents) == 3 and torch.is_tensor(sents[0]):
            input_ids, input_mask, segment_ids = (t.to(device) for t in sents)
        else:
            if self.tokenizer is None:
                raise RuntimeError("no tokenizer: pass vocab_path/tokenizer, or feed (input_ids, input_mask, "
                                   "segment_ids) tensors instead of strings")
            feats_ = convert_sents_to_features(sents, self.max_seq_length, self.tokenizer)
            input_ids = torch.tensor([f.input_ids for f in feats_], dtype=torch.long, device=device)
            input_mask = torch.tensor([f.input_mask for f in feats_], dtype=torch.long, device=device)
            segment_ids = torch.tensor([f.segment_ids for f in feats_], dtype=torch.long, device=device)
        feat_seq, output = self.model(input_ids, segment_ids, input_mask, visual_feats=feats,
                                      visual_attention_mask=visual_attention_mask)
        return feat_seq, input_mask, output

    def save(self, path):
        torch.save(self.model.state_dict(), os.path.join("%s_LXRT.pth" % path))

    def load(self, path):
        """ref: src/lxrt/entry.py:212-238 (strip ``module.``, non-strict)"""
        print("Load LXMERT pre-trained model from %s" % path)
        state_dict = torch.load("%s_LXRT.pth" % path, map_location="cpu", weights_only=True)
        state_dict = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        load_keys, model_keys = set(state_dict.keys()), set(self.model.state_dict().keys())
        print("\nWeights in loaded but not in model:")
        for key in sorted(load_keys.difference(model_keys)):
            print(key)
        print("\nWeights in model but not in loaded:")
        for key in sorted(model_keys.difference(load_keys)):
            print(key)
        print()
        self.model.load_state_dict(state_dict, strict=False)
        sync_weights(self)
