"""GCNGenerator / GINGenerator / GATGenerator: iterative node update + adjacency
regeneration (src/module/graph_generative_modeling.py:162-269)."""
import torch.nn as nn

from .. import functional as XF
from ..runtime import bind_root
from .gcn import GCN
from .gin import GIN


class _Generator(nn.Module):
    def forward(self, x, adj):
        """(x or adj) sampling from noise.  x [bs, n_node, hidden], adj [bs, n_node, n_node]"""
        for layer in range(self.n_layers):
            x = self.gnn_layers[layer](x, adj)
            x, x_r = XF.fan_out(x, 2)    # x feeds the regeneration AND the next layer: their gradients meet in one launch
            adj = XF.RegenFn.apply(x_r)  # bmm(x, x^T) / column max -> sigmoid -> zero diagonal
        return x, adj


class GCNGenerator(_Generator):
    """ref: src/module/graph_generative_modeling.py:199-233"""

    def __init__(self, hidden_dim, n_layers, dropout=0.5):
        super().__init__()
        self.dropout_p = dropout
        self.n_layers = n_layers
        self.act = nn.Sigmoid()
        self.gnn_layers = nn.ModuleList()
        for _ in range(n_layers):
            self.gnn_layers.append(GCN(input_dim=hidden_dim, hidden_dims=[hidden_dim, hidden_dim], n_layers=2,
                                       dropout=self.dropout_p))
        bind_root(self)


class GINGenerator(_Generator):
    """ref: src/module/graph_generative_modeling.py:162-196"""

    def __init__(self, hidden_dim, n_layers, dropout=0.5):
        super().__init__()
        self.dropout_p = dropout
        self.n_layers = n_layers
        self.act = nn.Sigmoid()
        self.gnn_layers = nn.ModuleList()
        for _ in range(n_layers):
            self.gnn_layers.append(GIN(input_dim=hidden_dim, hidden_dims=[hidden_dim, hidden_dim], n_layers=1,
                                       dropout=self.dropout_p))
        bind_root(self)


class GATGenerator(_Generator):
    """ref: src/module/graph_generative_modeling.py:236-269"""

    def __init__(self, hidden_dim, n_layers, dropout=0.5):
        super().__init__()
        from .gat import GAT
        self.dropout_p = dropout
        self.n_layers = n_layers
        self.act = nn.Sigmoid()
        self.gnn_layers = nn.ModuleList()
        for _ in range(n_layers):
            self.gnn_layers.append(GAT(input_dim=hidden_dim, hidden_dim=hidden_dim, n_head=2))
        bind_root(self)
