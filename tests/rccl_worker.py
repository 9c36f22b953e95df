"""Child process of tests/test_training.py::test_ddp_over_rccl_as_the_reference_wraps_it: the RCCL ('nccl')
process group of dist_train.py:76-82 on ONE GPU (world size 1) -- init_process_group('nccl', device_id=cuda:0),
the pieces of graingraphnn_amd.dist.gather_states' multi-rank branch (pack_state -> all_gather -> unpack_state) on
device buffers through the communicator, and DistributedDataParallel(model, device_ids=[0]) for two iterations whose
gradients must equal the unwrapped model's bit for bit.  (Two ranks over RCCL: tests/rccl_worker2.py, which needs
two GPUs.)  Its own process, so that no communicator lives inside the pytest process.
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
from graingraphnn_amd.dist import gather_states, pack_state, unpack_state  # noqa: E402
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
        # gather_states' multi-rank branch piece by piece on DEVICE buffers through the communicator (the group has
        # one rank, for which gather_states itself returns [state] by contract): mixed dtypes, segments that start at
        # 16-byte offsets inside the packed buffer
        state = {"joint_xy": X["joint"][:, :2].clone(), "grain_area_v": X["grain"][:, 3:5].clone(),
                 "step": torch.tensor([7], dtype=torch.int64, device=dev),
                 "flags": torch.tensor([1, 0, 1], dtype=torch.uint8, device=dev),
                 "half": torch.arange(5, device=dev).to(torch.float16)}
        packed, keys, sizes = pack_state(state)
        assert packed.is_cuda and packed.numel() % 16 == 0
        bufs = [torch.empty_like(packed)]
        dist.all_gather(bufs, packed)
        torch.cuda.synchronize()
        back = unpack_state(bufs[0], keys, sizes, state)
        assert all(back[k].is_cuda and back[k].dtype == state[k].dtype and torch.equal(back[k], state[k]) for k in state)
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
