"""World-size-2 `gloo` test of the trajectory sharding + all-gather (the N > 1 path of
bench.py / cfg4).  Each rank rolls its own shard out with the CPU oracle standing in for the
device (the collective logic is backend-independent); the gathered result must equal the
single-process one."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import load_graph, oracle_models, tt
from graingraphnn_amd import synthetic
from graingraphnn_amd.dist import gather_states, run_sharded, shard_trajectories
from oracle import grainnn_oracle as oracle

N_TRAJ = 5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_one(t):
    """cfg4-style trajectory t: perturbed joints, 2 oracle steps -> joint xy [236, 2]."""
    x, ei, ea = load_graph("40")
    x = synthetic.perturbed_copy(x, 1e-3, 1000 + t)
    R, Cm = oracle_models(10020)
    X, EI, EA = tt(x), tt(ei), tt(ea)
    for _ in range(2):
        _, EA = oracle.rollout_step(R, Cm, X, EI, EA, 6)
    return X["joint"][:, :2].clone()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = run_sharded(N_TRAJ, _run_one, rank, world)
    # several keys of different dtypes and shapes travel as ONE packed all-gather
    states = gather_states({"rank_id": torch.full((3,), float(rank)),
                            "xy": torch.arange(8, dtype=torch.float64).view(4, 2) + rank,
                            "step": torch.tensor([10 * rank + 1], dtype=torch.int64)}, world)
    if rank == 0:
        torch.save({"res": res, "ids": [float(s["rank_id"][0]) for s in states],
                    "xy": [s["xy"] for s in states], "step": [int(s["step"]) for s in states]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_assignment():
    assert shard_trajectories(64, 3, 8) == list(range(3, 64, 8))
    got = sorted(t for r in range(8) for t in shard_trajectories(64, r, 8))
    assert got == list(range(64))
    assert shard_trajectories(5, 1, 2) == [1, 3]


def test_two_rank_gloo_gather(tmp_path):
    torch.set_num_threads(1)
    ref = torch.stack([_run_one(t) for t in range(N_TRAJ)])
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["ids"] == [0.0, 1.0] and got["step"] == [1, 11]
    for r in range(2):
        assert got["xy"][r].dtype == torch.float64 and got["xy"][r].shape == (4, 2)
        assert torch.equal(got["xy"][r], torch.arange(8, dtype=torch.float64).view(4, 2) + r)
    assert got["res"].shape == ref.shape
    assert np.array_equal(got["res"].numpy(), ref.numpy())  # same code, same seeds -> same bits


def _run_cfg4(t):
    """BASELINE config 4's trajectory t: the 40 um fixture with joints perturbed by RandomState(1000 + t), ONE oracle
    step (64 of them have to fit the CPU suite) -> joint xy | grain (area, extraV) as one row."""
    x, ei, ea = load_graph("40")
    x = synthetic.perturbed_copy(x, 1e-3, 1000 + t)
    R, Cm = oracle_models(10020)
    X, EI, EA = tt(x), tt(ei), tt(ea)
    oracle.rollout_step(R, Cm, X, EI, EA, 6)
    return torch.cat([X["joint"][:, :2].reshape(-1), X["grain"][:, 3:5].reshape(-1)])


def _cfg4_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert shard_trajectories(64, rank, world) == list(range(rank, 64, world)) and len(shard_trajectories(64, rank, world)) == 8
    res = run_sharded(64, _run_cfg4, rank, world)
    # the state gather reuses its buffers: a second gather of the same layout must not disturb the protocol
    for rep in range(2):
        states = gather_states({"xy": res[rank::world].clone() + rep}, world)
        assert all(torch.equal(states[r]["xy"], res[r::world] + rep) for r in range(world))
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_cfg4_sixty_four_trajectories_over_eight_ranks(tmp_path):
    """BASELINE config 4 as stated -- 64 independent trajectories, trajectory t on rank t mod 8, 8 per rank, one
    all-gather of the results -- on eight gloo ranks: the gathered [64, ...] result equals the single-process one bit
    for bit, in trajectory order, on the path bench.py --workload cfg4 --gpus 8 takes (shard_trajectories, run_sharded,
    gather_states)."""
    torch.set_num_threads(1)
    ref = torch.stack([_run_cfg4(t) for t in range(64)])
    out = str(tmp_path / "cfg4.pt")
    mp.spawn(_cfg4_worker, args=(8, _free_port(), out), nprocs=8, join=True)
    got = torch.load(out)
    assert got.shape == ref.shape == (64, 236 * 2 + 118 * 2)
    assert np.array_equal(got.numpy(), ref.numpy())


# ---------------------------------------------------------------------------------------
# training: DistributedDataParallel exactly as dist_train.py:82 wraps the model
# ---------------------------------------------------------------------------------------
def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.nn.parallel import DistributedDataParallel
    from emulator import TorchEmulatorBackend
    from helpers import product_models
    from graingraphnn_amd import training
    be = TorchEmulatorBackend()
    training.default_backend = lambda: be          # this process only: every C-ABI call emulated
    R, _ = product_models(4, 1.0)
    R.train()
    model = DistributedDataParallel(R)
    x, ei, ea = load_graph("40")
    y, mask = _train_targets(rank)
    for _ in range(2):                              # the second iteration is what unused parameters break
        model.zero_grad()
        loss = training.regressor_loss(tt(y), model(tt(x), tt(ei), tt(ea)), tt(mask))
        loss.backward()
    if rank == 0:
        torch.save({n: p.grad.clone() for n, p in R.named_parameters()}, out)
    dist.barrier()
    dist.destroy_process_group()


def _train_targets(rank):
    rs = np.random.RandomState(500 + rank)
    y = {"joint": rs.uniform(-1, 1, (236, 2)).astype(np.float32), "grain": rs.uniform(-1, 1, (118, 2)).astype(np.float32)}
    return y, {"joint": np.ones((236, 1), np.float32), "grain": np.ones((118, 1), np.float32)}


def test_two_rank_ddp_gradients_are_the_rank_average(tmp_path):
    from helpers import oracle_models as om
    from graingraphnn_amd import training
    torch.set_num_threads(1)
    x, ei, ea = load_graph("40")
    ref = None
    for rank in range(2):                           # single-process reference: the oracle, per rank
        oR, _ = om(4, 1.0)
        oR.train()
        y, mask = _train_targets(rank)
        training.regressor_loss(tt(y), oR(tt(x), tt(ei), tt(ea)), tt(mask)).backward()
        g = {n: p.grad.clone() for n, p in oR.named_parameters()}
        ref = g if ref is None else {n: 0.5 * (ref[n] + g[n]) for n in g}
    out = str(tmp_path / "grads.pt")
    mp.spawn(_ddp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert set(got) == set(ref)
    for n, g in ref.items():
        scale = float(g.abs().max())
        assert float((got[n] - g).abs().max()) <= 2e-4 * scale + 1e-9, n
