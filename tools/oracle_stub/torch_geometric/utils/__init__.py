"""PyG 2.1.0 `torch_geometric.utils.softmax` semantics (segment softmax over `index`)."""
import torch


def softmax(src, index=None, ptr=None, num_nodes=None, dim=0):
    assert ptr is None and dim == 0
    N = int(index.max()) + 1 if num_nodes is None else num_nodes
    shape = (N,) + tuple(src.shape[1:])
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    seg_max = torch.full(shape, float("-inf"), dtype=src.dtype, device=src.device)
    seg_max = seg_max.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    out = (src - seg_max.gather(0, idx)).exp()
    seg_sum = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(0, idx, out)
    return out / (seg_sum.gather(0, idx) + 1e-16)
