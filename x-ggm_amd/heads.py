"""Head modules of VQAModel / GQAModel (src/vqa/vqacpv2_model.py:63-105).

They are ``nn.Sequential`` subclasses so ``state_dict`` keys stay ``logit_fc.0.weight``,
``logit_fc.2.bias``, ``logit_fc.3.weight`` ... and the trainers can call them exactly as the
reference does (``self.model.logit_fc(x)``); their forward runs the fused HIP blocks."""
import torch.nn as nn

from . import functional as XF
from . import ops
from .lxrt.modeling import GeLU, BertLayerNorm
from .runtime import runtime_of


class MLPHead(nn.Sequential):
    """Linear -> GeLU -> LayerNorm(eps) [-> Linear]"""

    def __init__(self, d_in, d_hidden, eps, d_out=None, out_f32=False):
        mods = [nn.Linear(d_in, d_hidden), GeLU(), BertLayerNorm(d_hidden, eps=eps)]
        if d_out is not None:
            mods.append(nn.Linear(d_hidden, d_out))
        super().__init__(*mods)
        self.eps = eps
        self.out_f32 = out_f32

    def forward(self, x):
        rt = runtime_of(self)
        y = XF.MLPFn.apply(rt, self[0], self[2], self.eps, x, *self[0].parameters(), *self[2].parameters())
        if len(self) > 3:
            y = XF.LinearActFn.apply(rt, self[3], y, ops.ACT_NONE, self.out_f32, *self[3].parameters())
        return y


class SigmoidHead(nn.Sequential):
    """encoder_adj: Linear(768, 630) + Sigmoid, fp32 output (src/vqa/vqacpv2_model.py:91-94)"""

    def __init__(self, d_in, d_out):
        super().__init__(nn.Linear(d_in, d_out), nn.Sigmoid())

    def forward(self, x):
        rt = runtime_of(self)
        return XF.LinearActFn.apply(rt, self[0], x, ops.ACT_SIGMOID, True, *self[0].parameters())
