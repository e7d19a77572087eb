"""GATConv / GAT with the reference's parameter names (src/module/gat.py:6-79); arithmetic in
``functional.GATFn``.  As in the reference the two heads are concatenated, so the output is
2 x hidden wide and a GATGenerator is only shape-valid with n_layers == 1."""
import torch.nn as nn

from .. import functional as XF
from ..runtime import runtime_of


class GATConv(nn.Module):
    """ref: src/module/gat.py:6-49"""

    def __init__(self, dim_input, dim_hidden, dropout=0.5, alpha=0.2, concat=True):
        super().__init__()
        if not concat:
            raise NotImplementedError("the reference only builds GATConv with concat=True")
        self.dropout = dropout
        self.concat = concat
        self.dim_hidden = dim_hidden
        self.alpha = alpha
        self.linear_layer = nn.Linear(dim_input, dim_hidden, bias=False)
        self.attn_layer = nn.Linear(2 * dim_hidden, 1, bias=False)
        self.reset_parameters()
        self.leaky_relu = nn.LeakyReLU(alpha)

    def reset_parameters(self):
        gain = nn.init.calculate_gain('relu')
        nn.init.xavier_normal_(self.linear_layer.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_layer.weight, gain=gain)


class GAT(nn.Module):
    """ref: src/module/gat.py:52-79"""

    def __init__(self, input_dim, hidden_dim, n_head, dropout=0.5, alpha=0.2, merge='cat'):
        super().__init__()
        if merge != 'cat':
            raise NotImplementedError("the reference generators use merge='cat'")
        self.dropout = dropout
        self.merge = merge
        self.gat_layers = nn.ModuleList()
        for _ in range(n_head):
            self.gat_layers.append(GATConv(input_dim, hidden_dim, dropout=dropout, alpha=alpha, concat=True))

    def forward(self, x, adj):
        rt = runtime_of(self)
        return XF.GATFn.apply(rt, self, x, adj.float().contiguous(), *self.parameters())
