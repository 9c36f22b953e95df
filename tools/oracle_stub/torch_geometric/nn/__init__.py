"""PyG 2.1.0 `HeteroConv(convs, aggr='sum')` (+ a name-only SAGEConv the reference imports)."""
from collections import defaultdict

import torch
from torch import nn

from .conv import MessagePassing  # noqa: F401
from .dense.linear import Linear  # noqa: F401


class SAGEConv(nn.Module):  # imported by heterogclstm.py, never run (layers == 1)
    def __init__(self, *args, **kwargs):
        super().__init__()


class HeteroConv(nn.Module):
    def __init__(self, convs, aggr="sum"):
        super().__init__()
        assert aggr == "sum"
        self.convs = nn.ModuleDict({"__".join(k): v for k, v in convs.items()})
        self.aggr = aggr

    def forward(self, x_dict, edge_index_dict, *args_dict, **kwargs_dict):
        out_dict = defaultdict(list)
        for edge_type, edge_index in edge_index_dict.items():
            src, rel, dst = edge_type
            str_edge_type = "__".join(edge_type)
            if str_edge_type not in self.convs:
                continue
            kwargs = {}
            for arg, value_dict in kwargs_dict.items():
                assert arg.endswith("_dict")
                if edge_type in value_dict:
                    kwargs[arg[:-5]] = value_dict[edge_type]
            conv = self.convs[str_edge_type]
            if src == dst:
                out = conv(x_dict[src], edge_index, **kwargs)
            else:
                out = conv((x_dict[src], x_dict[dst]), edge_index, **kwargs)
            out_dict[dst].append(out)
        return {k: torch.stack(v, dim=0).sum(0) if len(v) > 1 else v[0]
                for k, v in out_dict.items()}
