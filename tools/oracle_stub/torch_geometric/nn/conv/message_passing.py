"""PyG 2.1.0 `MessagePassing.propagate` for a COO `edge_index`, flow source_to_target,
aggr='add', node_dim=0: gather x_j = x[0][ei[0]], x_i = x[1][ei[1]] -> message -> scatter-sum
over ei[1] -> update (identity)."""
import inspect

import torch
from torch import nn


class MessagePassing(nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2, **kwargs):
        super().__init__()
        assert aggr == "add" and flow == "source_to_target"
        self.aggr = aggr
        self.node_dim = node_dim
        self._msg_params = list(inspect.signature(self.message).parameters)

    def propagate(self, edge_index, size=None, **kwargs):
        assert self.node_dim == 0 and edge_index.dim() == 2 and edge_index.size(0) == 2
        src, dst = edge_index[0], edge_index[1]
        x = kwargs.get("x")
        if isinstance(x, torch.Tensor):
            x = (x, x)
        n_dst = x[1].size(0) if size is None else size[1]
        feed = {}
        for name in self._msg_params:
            if name == "index":
                feed[name] = dst
            elif name == "ptr":
                feed[name] = None
            elif name == "size_i":
                feed[name] = n_dst
            elif name.endswith("_j"):
                base = kwargs[name[:-2]]
                base = base[0] if isinstance(base, (tuple, list)) else base
                feed[name] = base.index_select(0, src)
            elif name.endswith("_i"):
                base = kwargs[name[:-2]]
                base = base[1] if isinstance(base, (tuple, list)) else base
                feed[name] = base.index_select(0, dst)
            else:
                feed[name] = kwargs.get(name)
        msg = self.message(**feed)
        out = torch.zeros((n_dst,) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
        out.index_add_(0, dst, msg)
        return out
