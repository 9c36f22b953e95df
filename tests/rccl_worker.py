"""Child process of tests/test_training.py::test_ddp_over_rccl_as_the_reference_wraps_it: the RCCL ('nccl')
process group of dist_train.py:76-82 on ONE GPU (world size 1) -- init_process_group('nccl', device_id=cuda:0),
graingraphnn_amd.dist.gather_states on device tensors (one packed all-gather), and
DistributedDataParallel(model, device_ids=[0]) for two iterations whose gradients must equal the unwrapped
model's bit for bit.  Its own process, so that no communicator lives inside the pytest process.
Prints 'RCCL_OK ...' on success; any failure is an exception (non-zero exit)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from torch.nn.parallel import DistributedDataParallel  # noqa: E402

from helpers import load_graph, product_models, tt  # noqa: E402
from graingraphnn_amd import training  # noqa: E402
from graingraphnn_amd.dist import gather_states  # noqa: E402
from test_training import _targets  # noqa: E402


def main():
    port = int(sys.argv[1])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    x, ei, ea = load_graph("40")
    y_np, m_np = _targets(x, ei)
    y, mask = tt(y_np, "cuda"), tt(m_np, "cuda")
    R, _ = product_models(10020, 1.0, "cuda")
    R.train()
    X, EI, EA = tt(x, "cuda"), tt(ei, "cuda"), tt(ea, "cuda")
    training.regressor_loss(y, R(X, EI, EA), mask).backward()
    ref = {n: p.grad.clone() for n, p in R.named_parameters()}
    R.zero_grad()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        # the packed all-gather on DEVICE buffers, through the communicator (world 2 is claimed so that
        # gather_states does not take its single-rank shortcut; the group itself has one rank)
        state = {"joint_xy": X["joint"][:, :2].clone(), "grain_area_v": X["grain"][:, 3:5].clone(),
                 "step": torch.tensor([7], dtype=torch.int64, device=dev)}
        packed = torch.cat([state[k].contiguous().view(-1).view(torch.uint8) for k in sorted(state)])
        bufs = [torch.empty_like(packed)]
        dist.all_gather(bufs, packed)
        torch.cuda.synchronize()
        assert bufs[0].is_cuda and torch.equal(bufs[0], packed)
        got = gather_states(state, dist.get_world_size())     # world 1: identity by contract
        assert len(got) == 1 and all(torch.equal(got[0][k], state[k]) for k in state)
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)
        assert torch.equal(t, torch.ones(4, device=dev))
        model = DistributedDataParallel(R, device_ids=[0])
        for _ in range(2):          # the second iteration is what unused parameters would break
            model.zero_grad()
            training.regressor_loss(y, model(X, EI, EA), mask).backward()
        torch.cuda.synchronize()
        for n, p in R.named_parameters():
            assert torch.equal(p.grad, ref[n]), n
        ver = ".".join(map(str, torch.cuda.nccl.version()))
    finally:
        dist.destroy_process_group()
    print(f"RCCL_OK rccl {ver}: packed all-gather on device buffers, DDP gradients bit-equal over 2 iterations")


if __name__ == "__main__":
    main()
