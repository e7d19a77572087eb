"""GINConv / GIN with the reference's parameter names (src/module/gin.py:10-87); the
arithmetic is the fused HIP block ``functional.GINFn``."""
import torch
import torch.nn as nn

from .. import functional as XF
from ..lxrt.modeling import GeLU
from ..runtime import runtime_of


class GINConv(nn.Module):
    """X + (1 + eps) A X -> Linear -> GeLU -> LN; ref: src/module/gin.py:10-34"""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.eps = nn.Parameter(torch.zeros(1))
        self.linear = nn.Sequential(nn.Linear(input_dim, hidden_dim), GeLU(), nn.LayerNorm(hidden_dim))


class GIN(nn.Module):
    """ref: src/module/gin.py:37-87"""

    def __init__(self, input_dim, hidden_dims, n_layers, dropout=0.5):
        super().__init__()
        if n_layers != 1:
            raise NotImplementedError("the reference generators build GIN with n_layers=1 "
                                      "(src/module/graph_generative_modeling.py:171-175)")
        self.dropout_p = dropout
        self.gnn_convs = nn.ModuleList()
        self.linear_prediction = nn.ModuleList()
        self.gnn_convs.append(GINConv(input_dim, hidden_dims[0]))
        self.linear_prediction.append(nn.Sequential(nn.Linear(input_dim, hidden_dims[0]), GeLU(),
                                                    nn.LayerNorm(hidden_dims[0])))
        self.linear_prediction.append(nn.Sequential(nn.Linear(hidden_dims[-2], hidden_dims[-1]), GeLU(),
                                                    nn.LayerNorm(hidden_dims[-1])))

    def forward(self, X, A):
        rt = runtime_of(self)
        return XF.GINFn.apply(rt, self, X, A.float().contiguous(), *self.parameters())
