"""GCNConv / GCN with the reference's parameter names (src/module/gcn.py:10-77); the
arithmetic is the fused HIP block ``functional.GCNFn``."""
import torch.nn as nn

from .. import functional as XF
from ..lxrt.modeling import GeLU
from ..runtime import runtime_of


class GCNConv(nn.Module):
    """LN(x + W_ctx (adj @ x)); ref: src/module/gcn.py:10-29 (dropout p = 0.0 there)."""

    def __init__(self, dim_hidden, dropout=0.0):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("GCNConv dropout is 0.0 in the reference and not built here")
        self.ctx_layer = nn.Linear(dim_hidden, dim_hidden, bias=False)
        self.layer_norm = nn.LayerNorm(dim_hidden)
        self.dropout = nn.Dropout(p=dropout)


class GCN(nn.Module):
    """ref: src/module/gcn.py:32-77"""

    def __init__(self, input_dim, hidden_dims, n_layers, dropout=0.5):
        super().__init__()
        self.dropout_p = dropout
        self.gnn_layers = nn.ModuleList()
        self.linear_prediction = nn.ModuleList()
        for i in range(n_layers):
            d_in = input_dim if i == 0 else hidden_dims[i - 1]
            if d_in != hidden_dims[i]:
                raise NotImplementedError("the fused GCN block assumes equal widths (reference uses 768 throughout)")
            self.gnn_layers.append(GCNConv(d_in))
            self.linear_prediction.append(nn.Sequential(nn.Linear(d_in, hidden_dims[i]), GeLU(),
                                                        nn.LayerNorm(hidden_dims[i])))
        self.linear_prediction.append(nn.Sequential(nn.Linear(hidden_dims[-2], hidden_dims[-1]), GeLU(),
                                                    nn.LayerNorm(hidden_dims[-1])))

    def forward(self, x, adj):
        rt = runtime_of(self)
        return XF.GCNFn.apply(rt, self, x, adj.float().contiguous(), *self.parameters())
