"""VQAModel with the reference's constructor, attributes and forward signature
(src/vqa/vqacpv2_model.py:52-131)."""
import torch.nn as nn

from ..heads import MLPHead, SigmoidHead
from ..lxrt.entry import LXRTEncoderFeature
from ..module.graph_generative_modeling import GCNGenerator, GINGenerator, GATGenerator
from ..runtime import bind_root, sync_weights, root_of

# Max length including <bos> and <eos>
MAX_VQA_LENGTH = 20


class XGGMModel(nn.Module):
    """shared body of VQAModel / GQAModel (they are line-for-line twins in the reference:
    src/vqa/vqacpv2_model.py:52-131, src/gqa/gqa_ood_model.py:52-123)."""

    def __init__(self, num_answers, gnn='GCN', n_layers=2, args=None, max_seq_length=MAX_VQA_LENGTH,
                 compute_dtype=None, config=None, tokenizer=None, n_objects=36):
        super().__init__()
        if args is None:
            from ..param import args as default_args
            args = default_args
        self.lxrt_encoder = LXRTEncoderFeature(args, max_seq_length=max_seq_length, mode='lxr', config=config,
                                               tokenizer=tokenizer)
        hid_dim = self.lxrt_encoder.dim
        # VQA answer head (BertLayerNorm eps 1e-12; fp32 logits so answer indices are exact)
        self.logit_fc = MLPHead(hid_dim, hid_dim * 2, 1e-12, d_out=num_answers, out_f32=True)
        self.logit_fc.apply(self.lxrt_encoder.model.init_bert_weights)
        if gnn == 'GCN':
            self.generator = GCNGenerator(hidden_dim=hid_dim, n_layers=n_layers)
        elif gnn == 'GIN':
            self.generator = GINGenerator(hidden_dim=hid_dim, n_layers=n_layers)
        elif gnn == 'GAT':
            self.generator = GATGenerator(hidden_dim=hid_dim, n_layers=n_layers)
        else:
            raise ModuleNotFoundError
        # relation / node initialisation (nn.LayerNorm default eps 1e-5)
        self.encoder_adj = SigmoidHead(hid_dim, n_objects * (n_objects - 1) // 2)
        self.node_fc = MLPHead(hid_dim, hid_dim, 1e-5)
        self.fusion_fc = MLPHead(hid_dim * 2, hid_dim, 1e-5)
        # parameters the plain-VQA pass never reaches (their grad stays None there): the last
        # cross layer's visual self-attention and FFN -> their own optimiser range
        nx = self.lxrt_encoder.model.bert.encoder.num_x_layers
        if nx > 0:
            base = "lxrt_encoder.model.bert.encoder.x_layers.%d." % (nx - 1)
            self._enc_tail_prefixes = tuple(base + s for s in ("visn_self_att.", "visn_inter.", "visn_output."))
        else:
            self._enc_tail_prefixes = ()
        bind_root(self, compute_dtype)

    def forward(self, feat, pos, sent):
        """feat (b, o, f), pos (b, o, 4), sent: list of strings (needs a tokenizer) or a tuple
        (input_ids, input_mask, segment_ids) of int64 tensors -> ((lang, visn), que_mask, x)"""
        feat_seq, que_mask, x = self.lxrt_encoder(sent, (feat, pos))
        return feat_seq, que_mask, x

    # ---- bookkeeping hooks that keep the flat arena consistent with the trainer's calls
    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=True)
        rt = getattr(self, "_xg_rt", None)
        if rt is not None:
            rt.arena.begin_pass()

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        sync_weights(self)
        return out

    def arena(self):
        from ..runtime import runtime_of
        return runtime_of(self).arena


class VQAModel(XGGMModel):
    def __init__(self, num_answers, gnn='GCN', n_layers=2, **kw):
        super().__init__(num_answers, gnn=gnn, n_layers=n_layers, max_seq_length=MAX_VQA_LENGTH, **kw)
