"""GQAModel (src/gqa/gqa_ood_model.py:52-123): same members as VQAModel."""
from ..vqa.vqacpv2_model import XGGMModel

MAX_GQA_LENGTH = 20


class GQAModel(XGGMModel):
    def __init__(self, num_answers, gnn='GCN', n_layers=2, **kw):
        super().__init__(num_answers, gnn=gnn, n_layers=n_layers, max_seq_length=MAX_GQA_LENGTH, **kw)
