"""Multi-GPU: independent grain-graph trajectories shard across ranks (one process per GPU,
`torch.distributed`, backend 'nccl' = RCCL over xGMI on ROCm, 'gloo' in CPU tests).  There is
no exchange during a rollout; the only collective is one all-gather of the per-trajectory
results at the end.  The reference's only parallel code is DDP training
(dist_train.py:76-93); inference there is single-process."""
from typing import Callable, Dict, List

import torch
import torch.distributed as dist


def shard_trajectories(n_traj: int, rank: int, world: int) -> List[int]:
    """Trajectory t runs on rank t mod world (round-robin keeps ranks within one of each other)."""
    return [t for t in range(n_traj) if t % world == rank]


def _check_shardable(n_traj: int, world: int):
    """Raised identically on EVERY rank, before any work or collective: a rank that aborted alone
    (an empty shard) would leave the others blocked in the all-gather."""
    if n_traj < 1:
        raise ValueError("no trajectories to run")
    if n_traj < world:
        raise ValueError(f"more ranks ({world}) than trajectories ({n_traj}): every rank needs at least one")


def pack_state(state: Dict[str, torch.Tensor]):
    """A dict of tensors as ONE byte buffer (keys in sorted order, every segment 16-byte aligned so that it can be
    viewed back as its own dtype) + the layout needed to take it apart again: (packed uint8 tensor, keys, sizes)."""
    keys = sorted(state)
    flat, sizes = [], []
    for k in keys:
        b = state[k].contiguous().view(-1).view(torch.uint8)
        sizes.append(b.numel())
        pad = -b.numel() % 16
        flat.append(torch.cat([b, b.new_zeros(pad)]) if pad else b)
    return (torch.cat(flat) if len(flat) > 1 else flat[0]), keys, sizes


def unpack_state(buf: torch.Tensor, keys, sizes, like: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Inverse of pack_state on a (received) buffer: tensors with the device, dtype and shape of `like`'s."""
    d, off = {}, 0
    for k, n in zip(keys, sizes):
        ref = like[k]
        d[k] = buf[off:off + n].to(ref.device).view(ref.dtype).view(ref.shape)
        off += n + (-n % 16)
    return d


class StateGather:
    """The packed all-gather of a fixed state layout with every buffer allocated ONCE: the send buffer the state is
    packed into, one flat receive buffer of world x n bytes (`all_gather_into_tensor`: one collective, one output) and
    the per-rank result dicts as VIEWS of it (host-staged backends: one pinned host copy each way).  gather_states
    keeps one per layout, so a rollout loop that gathers every few steps allocates nothing inside its timed region."""

    def __init__(self, state: Dict[str, torch.Tensor], world: int):
        self.world = world
        self.keys = sorted(state)
        self.like = {k: (state[k].dtype, tuple(state[k].shape), state[k].device) for k in self.keys}
        self.sizes = [state[k].numel() * state[k].element_size() for k in self.keys]
        self.offsets, off = [], 0
        for n in self.sizes:
            self.offsets.append(off)
            off += n + (-n % 16)   # every segment 16-byte aligned: viewable as its own dtype
        self.nbytes = off
        dev = state[self.keys[0]].device
        self.host = dist.get_backend() == "gloo"   # RCCL ('nccl') moves device buffers; gloo needs host buffers
        self.send = torch.zeros(self.nbytes, dtype=torch.uint8, device=dev)
        self.recv = torch.empty(world * self.nbytes, dtype=torch.uint8, device=dev)
        if self.host:
            pin = dev.type == "cuda"
            self.send_h = torch.zeros(self.nbytes, dtype=torch.uint8, pin_memory=pin)
            self.recv_h = torch.empty(world * self.nbytes, dtype=torch.uint8, pin_memory=pin)
        self.out = [{k: self.recv[r * self.nbytes + o: r * self.nbytes + o + n].view(self.like[k][0]).view(self.like[k][1])
                     for k, o, n in zip(self.keys, self.offsets, self.sizes)} for r in range(world)]

    def matches(self, state, world) -> bool:
        return (world == self.world and sorted(state) == self.keys and
                all((state[k].dtype, tuple(state[k].shape), state[k].device) == self.like[k] for k in self.keys))

    def __call__(self, state):
        for k, o, n in zip(self.keys, self.offsets, self.sizes):
            self.send[o:o + n].view(self.like[k][0]).view(self.like[k][1]).copy_(state[k])
        if self.host:
            self.send_h.copy_(self.send)
            _all_gather_flat(self.recv_h, self.send_h, self.world)
            self.recv.copy_(self.recv_h)
        else:
            _all_gather_flat(self.recv, self.send, self.world)
        return self.out


def _all_gather_flat(recv: torch.Tensor, send: torch.Tensor, world: int):
    try:
        dist.all_gather_into_tensor(recv, send)
    except (RuntimeError, NotImplementedError):   # a backend without the flat form: the list form on views of `recv`
        dist.all_gather([recv[r * send.numel():(r + 1) * send.numel()] for r in range(world)], send)


_gatherers: List[StateGather] = []


def gather_states(state: Dict[str, torch.Tensor], world: int) -> List[Dict[str, torch.Tensor]]:
    """All-gather a dict of tensors whose shapes and dtypes agree across ranks; returns one dict per
    rank (rank order).  ONE collective whatever the number of keys: the tensors travel as one packed
    byte buffer (a collective per key would pay the launch + ring latency of a small message each).
    The buffers of a layout are allocated at its first gather (StateGather) and reused: the returned
    tensors are views of the receive buffer, valid until the next gather of the same layout."""
    if world <= 1 or not dist.is_initialized():
        return [state]
    for g in _gatherers:
        if g.matches(state, world):
            return g(state)
    g = StateGather(state, world)
    _gatherers.append(g)
    del _gatherers[:-8]   # (a handful of layouts per process; old ones go)
    return g(state)


def run_sharded(n_traj: int, run_one: Callable[[int], torch.Tensor], rank: int, world: int,
                device=None) -> torch.Tensor:
    """Run `run_one(t)` (-> fixed-shape result tensor) for this rank's trajectories and
    all-gather everything: returns [n_traj, ...] in trajectory order on every rank.
    Ranks with fewer trajectories pad their shard so the collective stays regular."""
    _check_shardable(n_traj, world)
    mine = shard_trajectories(n_traj, rank, world)
    per_rank = (n_traj + world - 1) // world
    results = [run_one(t) for t in mine]
    proto = results[0]
    local = torch.zeros((per_rank,) + tuple(proto.shape), dtype=proto.dtype, device=proto.device)
    for i, r in enumerate(results):
        local[i] = r
    if world <= 1 or not dist.is_initialized():
        return local[:n_traj]
    # ONE packed collective on buffers that are kept per layout (gather_states / StateGather: send, flat receive and --
    # under gloo -- pinned host staging are allocated at the first call of a shape and reused; round 5 allocated `world`
    # receive tensors per call here)
    shards = gather_states({"shard": local}, world)
    out = torch.empty((n_traj,) + tuple(proto.shape), dtype=proto.dtype, device=proto.device)
    for r in range(world):
        n_r = len(range(r, n_traj, world))   # trajectories t = r, r + world, ...: rows 0 .. n_r - 1 of rank r's shard
        out[r::world] = shards[r]["shard"][:n_r]
    return out


def rollout_trajectories(rmodel, cmodel, graphs, span: int, n_steps: int, rank: int = 0,
                         world: int = 1, device="cuda", use_graph: bool = True,
                         refresh_centres: bool = False):
    """BASELINE config 4: `graphs[t]` = (x, ei, ea) numpy dicts of independent trajectories with
    EQUAL node counts.  Rank r rolls out trajectories t = r (mod world) as ONE disjoint-union
    graph on its GPU (one set of launches for the whole shard), then all ranks all-gather the
    final joint coordinates and grain (area, extraV): returns {'joint_xy': [T, N_j, 2],
'grain_area_v': [T, N_g, 2]} in trajectory order on every rank.  `refresh_centres`: also
    recompute the grain centres every step (unfolded domains: factor 1), as `GrainRollout` does."""
    from . import synthetic
    from .rollout import GrainRollout

    _check_shardable(len(graphs), world)
    mine = shard_trajectories(len(graphs), rank, world)
    x, ei, ea, slices = synthetic.disjoint_union([graphs[t] for t in mine])
    X, EI, EA = synthetic.to_torch(x, ei, ea, device)
    ro = GrainRollout(rmodel, cmodel, X, EI, EA, span, use_graph=use_graph, refresh_centres=refresh_centres)
    ro.run(n_steps)
    local = {"joint_xy": torch.stack([X["joint"][lo:hi, :2] for lo, hi in (s["joint"] for s in slices)]),
             "grain_area_v": torch.stack([X["grain"][lo:hi, 3:5] for lo, hi in (s["grain"] for s in slices)])}
    out = {}
    for k, v in local.items():
        it = iter(range(len(mine)))
        out[k] = run_sharded(len(graphs), lambda t, v=v, it=it: v[next(it)], rank, world)
    return out
