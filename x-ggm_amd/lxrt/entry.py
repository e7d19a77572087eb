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


class SentenceBatcher:
    """the same conversion for a whole batch, built for a training loop that runs at milliseconds per step:
    questions repeat (every epoch, and many VQA questions are shared between images), so token ids are cached
    per sentence; the batch is assembled in ONE int64 [3, B, T] pinned host buffer and crosses PCIe as one
    asynchronous copy instead of three ``torch.tensor(list, device=...)`` calls.  Results are identical to
    ``convert_sents_to_features`` (tests/test_abi_cpu.py).

    The pinned buffers form a ring (``depth`` of them), each guarded by an event recorded behind the H2D copy
    that read it last: a buffer is rewritten only after that copy has completed, however far the host runs
    ahead of the GPU (every iteration tokenises at least twice: plain pass and GGM pass)."""

    def __init__(self, tokenizer, max_seq_length, max_cached=1 << 20, depth=4):
        self.tok, self.T, self.max_cached = tokenizer, max_seq_length, max_cached
        self.cache = {}
        self.depth = depth
        self._ring = []     # [pinned tensor, numpy view, event of the last copy out of it | None]
        self._next = 0
        self._last = None   # ring entry handed out by the latest host_batch

    def ids_of(self, sent):
        ids = self.cache.get(sent)
        if ids is None:
            toks = self.tok.tokenize(sent.strip())[:self.T - 2]
            ids = self.tok.convert_tokens_to_ids(["[CLS]"] + toks + ["[SEP]"])
            if len(self.cache) < self.max_cached:
                self.cache[sent] = ids
        return ids

    def _entry(self, B):
        if self._ring and self._ring[0][0].shape[1] < B:
            for e in self._ring:  # a larger batch than the ring was built for: drain, then rebuild
                if e[2] is not None:
                    e[2].synchronize()
            self._ring, self._next = [], 0
        if len(self._ring) < self.depth:
            buf = torch.zeros((3, max(B, 32), self.T), dtype=torch.long)
            if torch.cuda.is_available():
                buf = buf.pin_memory()
            self._ring.append([buf, buf.numpy(), None])  # numpy view: row fills without per-row tensor objects
            return self._ring[-1]
        e = self._ring[self._next]
        self._next = (self._next + 1) % self.depth
        if e[2] is not None:
            e[2].synchronize()  # the copy that read this buffer last has finished
            e[2] = None
        return e

    def host_batch(self, sents):
        """int64 [3, B, T] host tensor: ids, mask, segment ids (pinned when a GPU is present).  A caller that ships
        it with its own asynchronous copy calls ``record_copy()`` right behind that copy."""
        B = len(sents)
        e = self._entry(B)
        buf, arr = e[0], e[1]
        arr[:2, :B] = 0
        for b, s in enumerate(sents):
            ids = self.ids_of(s)
            n = len(ids)
            arr[0, b, :n] = ids
            arr[1, b, :n] = 1
        self._last = e
        return buf[:, :B]

    def record_copy(self, stream=None):
        """mark the buffer of the latest ``host_batch`` as in flight on ``stream`` (default: the current one)"""
        if self._last is not None and torch.cuda.is_available():
            ev = torch.cuda.Event()
            ev.record(stream if stream is not None else torch.cuda.current_stream())
            self._last[2] = ev

    def __call__(self, sents, device):
        dev = self.host_batch(sents).to(device, non_blocking=True)
        if dev.is_cuda:
            self.record_copy()
        return dev[0], dev[1], dev[2]


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
                                  "see xggm_amd.vqa.vqacpv2.enable_data_parallel")

    @property
    def dim(self):
        return self._dim

    @property
    def batcher(self):
        """the cached question -> token-id batcher of this encoder's tokenizer"""
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer: pass vocab_path/tokenizer, or feed (input_ids, input_mask, "
                               "segment_ids) tensors instead of strings")
        if getattr(self, "_batcher", None) is None or self._batcher.tok is not self.tokenizer:
            self._batcher = SentenceBatcher(self.tokenizer, self.max_seq_length)
        return self._batcher

    def forward(self, sents, feats, visual_attention_mask=None):
        device = feats[0].device
        if isinstance(sents, (tuple, list)) and len(sents) == 3 and torch.is_tensor(sents[0]):
            input_ids, input_mask, segment_ids = (t.to(device) for t in sents)
        else:
            input_ids, input_mask, segment_ids = self.batcher(sents, device)
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
