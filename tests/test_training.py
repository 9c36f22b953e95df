"""Training path: losses and parameter gradients.

Pins, in order: the oracle's autograd against digests of the UNMODIFIED reference's gradients
(tests/golden/golden_cfg1_grads.npz, made by make_golden_grads.py); the product's differentiable
forward (graingraphnn_amd/training.py) through the torch emulator of the C ABI against the
oracle on the CPU; and, on the GPU, the HIP sweep + its hand-written backward against both.
Tolerance: per parameter tensor max|g - g_ref| <= 2e-4 * max|g_ref| (fp32 sums in another order).
"""
import os
import sys

import numpy as np
import pytest
import torch

from emulator import TorchEmulatorBackend
from helpers import EDGE_TYPES, GOLDEN, load_graph, oracle_models, product_models, tt
from graingraphnn_amd import synthetic, training

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
GTOL = 2e-4


def _targets(x, ei):
    """Same draw as tests/golden/make_golden_grads.py:targets."""
    n_j, n_g, E = x["joint"].shape[0], x["grain"].shape[0], ei[EDGE_TYPES[2]].shape[1]
    rs = np.random.RandomState(77)
    y = {"joint": rs.uniform(-1, 1, (n_j, 2)).astype(np.float32),
         "grain": rs.uniform(-1, 1, (n_g, 2)).astype(np.float32),
         "edge_event": rs.randint(-1, 2, size=E).astype(np.int64)}
    mask = {"joint": np.ones((n_j, 1), np.float32), "grain": np.ones((n_g, 1), np.float32)}
    mask["joint"][::7] = 0
    mask["grain"][::5] = 0
    return y, mask


def _digest(g):
    g = g.detach().cpu().numpy().astype(np.float64).ravel()
    idx = np.random.RandomState(5).randint(g.size, size=6)
    return np.concatenate([[g.sum(), np.sqrt((g * g).sum()), np.abs(g).max()], g[idx]])


def _grads(R, Cm, x, ei, ea, device="cpu", dtype=torch.float32):
    """dtype=float64 (oracle only): the models must have been cast with .double()."""
    y_np, m_np = _targets(x, ei)
    cast = lambda d: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in tt(d, device).items()}
    y, mask = cast(y_np), cast(m_np)
    R.train(), Cm.train()
    R.zero_grad(), Cm.zero_grad()
    lr = training.regressor_loss(y, R(cast(x), tt(ei, device), cast(ea)), mask)
    lc = training.classifier_loss(y, Cm(cast(x), tt(ei, device), cast(ea)), 1.0)
    lr.backward()
    lc.backward()
    out = {}
    for tag, m in (("R", R), ("C", Cm)):
        for name, p in m.named_parameters():
            out[f"{tag}/{name}"] = (torch.zeros_like(p) if p.grad is None else p.grad).detach().cpu()
    return float(lr.detach()), float(lc.detach()), out


def _check_digests(lr, lc, grads):
    gold = np.load(os.path.join(GOLDEN, "golden_cfg1_grads.npz"))
    assert abs(lr - float(gold["loss_regressor"])) <= 1e-5 * abs(float(gold["loss_regressor"]))
    assert abs(lc - float(gold["loss_classifier"])) <= 1e-5 * abs(float(gold["loss_classifier"]))
    assert len(grads) == len(gold.files) - 2 == 568
    # absolute floor for gradients that are zero in exact arithmetic (the key bias: a softmax is
    # shift-invariant), relative to the largest gradient entry of the whole model
    atol = 1e-6 * max(float(gold[name][2]) for name in grads)
    for name, g in grads.items():
        d, r = _digest(g), gold[name]
        gmax, n = r[2], g.numel()
        assert abs(d[0] - r[0]) <= (GTOL * gmax + atol) * np.sqrt(n), (name, "sum", d[0], r[0])
        assert abs(d[1] - r[1]) <= GTOL * r[1] + atol * np.sqrt(n), (name, "norm", d[1], r[1])
        assert np.abs(d[2:] - r[2:]).max() <= GTOL * gmax + atol, (name, "entries", d[2:], r[2:])


def _check_full(grads, ref):
    worst = 0.0
    atol = 1e-6 * max(float(g.abs().max()) for g in ref.values())
    for name, g in ref.items():
        scale = float(g.abs().max())
        err = float((grads[name] - g).abs().max())
        assert err <= GTOL * scale + atol, (name, err, scale)
        if scale > 100 * atol:
            worst = max(worst, err / scale)
    return worst


def test_oracle_gradients_match_the_reference_digests():
    x, ei, ea = load_graph("40")
    oR, oC = oracle_models(10020, 1.0)
    lr, lc, grads = _grads(oR, oC, x, ei, ea)
    _check_digests(lr, lc, grads)
    # the encoder's forget gate multiplies c = 0: exactly zero gradient in the reference too
    assert float(grads["R/gclstm_encoder.cell_list.0.b_f.joint"].abs().max()) == 0.0


def test_training_path_on_the_emulator_matches_oracle_gradients(monkeypatch):
    """Host logic of training.py (key-free operands from the parameters, gate batching, sweep
    output layout, reverse CSR) with every C-ABI call emulated in torch."""
    x, ei, ea = load_graph("40")
    be = TorchEmulatorBackend()
    monkeypatch.setattr(training, "default_backend", lambda: be)
    R, Cm = product_models(10020, 1.0)
    lr, lc, grads = _grads(R, Cm, x, ei, ea)
    _check_digests(lr, lc, grads)
    oR, oC = oracle_models(10020, 1.0)
    _, _, ref = _grads(oR, oC, x, ei, ea)
    _check_full(grads, ref)
    # inference dispatch is untouched: eval() / no_grad still refuse CPU tensors
    R.eval()
    from graingraphnn_amd import _lib
    with pytest.raises(_lib.GGNNError):
        R(tt(x), tt(ei), tt(ea))


@pytest.mark.parametrize("layer_size", [64, 32])
def test_narrow_layer_training_on_the_emulator_matches_oracle_gradients(monkeypatch, layer_size):
    """parameters.py:19: layer_size 64 / 32.  The training path packs a narrow cell through index tables that address
    its c-wide parameters directly and read the zero slot for every padded row / hidden-state column of the 96-wide layout
    (train_pack.PackPlan, _widen); the score scale is 1 / sqrt(c).  Loss and all 568 parameter gradients of both models
    against the oracle of the same width; the packed matrices equal those of the inference path's zero-padded holder."""
    from test_host_logic import _narrow_models
    from graingraphnn_amd import packing, train_pack
    x, ei, ea = load_graph("40")
    be = TorchEmulatorBackend()
    monkeypatch.setattr(training, "default_backend", lambda: be)
    (R, Cm), (oR, oC) = _narrow_models(layer_size, 41)
    lr, lc, grads = _grads(R, Cm, x, ei, ea)
    olr, olc, ref = _grads(oR, oC, x, ei, ea)
    assert abs(lr - olr) <= 1e-5 * abs(olr) and abs(lc - olc) <= 1e-5 * abs(olc), (lr, olr, lc, olc)
    assert all(grads[k].shape == ref[k].shape for k in ref)
    _check_full(grads, ref)
    for encoder, cell in ((True, R.gclstm_encoder.cell_list[0]), (False, R.gclstm_decoder.cell_list[0])):
        gates = "ico" if encoder else "ifco"
        _, wp, bp, ep, w2 = train_pack.packed_weights(cell, gates, cell.in_channels_dict, not encoder)
        pc = packing.pack_cell(packing.padded_cell(cell, layer_size), cell.in_channels_dict, encoder)
        for nt in wp:
            n = min(wp[nt].size(0), pc.wp[nt].size(0))     # (same rows; the inference layout may drop dead columns)
            if wp[nt].shape == pc.wp[nt].shape:
                assert torch.allclose(wp[nt], pc.wp[nt], rtol=1e-5, atol=1e-7), (encoder, nt)


def test_one_adam_step_on_the_emulator_follows_the_oracle(monkeypatch):
    """train.py:158-166: forward, loss, zero_grad, backward, Adam step -- same new parameters.
    (Where the exact gradient is zero -- key biases: a softmax is shift-invariant; the encoder's
    forget gate -- Adam steps along each implementation's own rounding noise: those tensors are
    held to 2 steps x lr each way only.)"""
    x, ei, ea = load_graph("40")
    be = TorchEmulatorBackend()
    monkeypatch.setattr(training, "default_backend", lambda: be)
    R, _ = product_models(4, 1.0)
    oR, _ = oracle_models(4, 1.0)
    y_np, m_np = _targets(x, ei)
    lr_adam, g_first = 5e-3, {}
    for m in (R, oR):
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=lr_adam)
        for it in range(2):
            loss = training.regressor_loss(tt(y_np), m(tt(x), tt(ei), tt(ea)), tt(m_np))
            opt.zero_grad()
            loss.backward()
            if m is oR and it == 0:
                g_first = {n: float(p.grad.abs().max()) for n, p in m.named_parameters()}
            opt.step()
    gmax = max(g_first.values())
    for (n, p), (_, q) in zip(R.named_parameters(), oR.named_parameters()):
        noise_only = g_first[n] <= 1e-5 * gmax
        bound = 4 * lr_adam * 1.01 if noise_only else 1e-3 * max(float(q.abs().max()), 1e-3)
        assert float((p.detach() - q.detach()).abs().max()) <= bound, (n, noise_only)


@pytest.mark.gpu
def test_hip_training_gradients_match_oracle_and_reference_digests():
    x, ei, ea = load_graph("40")
    R, Cm = product_models(10020, 1.0, "cuda")
    lr, lc, grads = _grads(R, Cm, x, ei, ea, "cuda")
    _check_digests(lr, lc, grads)
    oR, oC = oracle_models(10020, 1.0)
    _, _, ref = _grads(oR, oC, x, ei, ea)
    worst = _check_full(grads, ref)
    print(f"worst per-tensor relative gradient error {worst:.2e}")
    # reproducible: no atomics anywhere in the backward
    _, _, again = _grads(R, Cm, x, ei, ea, "cuda")
    assert all(torch.equal(grads[k], again[k]) for k in grads)
    # the training forward agrees with the fused inference forward
    R.eval()
    X, EI, EA = tt(x, "cuda"), tt(ei, "cuda"), tt(ea, "cuda")
    with torch.no_grad():
        yi = R(X, EI, EA)
    R.train()
    yt = R(X, EI, EA)
    for k in ("joint", "grain", "grain_area"):
        assert float((yi[k] - yt[k].detach()).abs().max()) <= 1e-4 * float(yi[k].abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("layer_size", [64, 32])
def test_hip_training_gradients_of_narrow_layers_match_the_oracle(layer_size):
    """layer_size 64 / 32 on the HIP training path (zero-padded packed weights through the index tables): loss and every
    parameter gradient of both models against the oracle of that width, and one FusedAdam step."""
    from test_host_logic import _narrow_models
    x, ei, ea = load_graph("40")
    (R, Cm), (oR, oC) = _narrow_models(layer_size, 43, "cuda")
    lr, lc, grads = _grads(R, Cm, x, ei, ea, "cuda")
    olr, olc, ref = _grads(oR, oC, x, ei, ea)
    assert abs(lr - olr) <= 1e-5 * abs(olr) and abs(lc - olc) <= 1e-5 * abs(olc), (lr, olr, lc, olc)
    worst = _check_full(grads, ref)
    print(f"layer_size {layer_size}: worst per-tensor relative gradient error {worst:.2e}")
    opt = training.FusedAdam(R.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in R.parameters()]
    opt.step()
    moved = [float((p.detach() - b).abs().max()) for p, b in zip(R.parameters(), before) if p.grad is not None]
    assert moved and max(moved) <= 1.01e-3 and max(moved) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed,hub_deg", [(1, 37), (2, 900)])
def test_hip_sweep_backward_on_ragged_graphs(seed, hub_deg):
    """The sweep's backward alone against autograd of its torch emulation: empty rows, a hub of
    degree `hub_deg`, sources without out-edges; G = 4 with hidden rows and G = 3 without."""
    from graingraphnn_amd.backend import default_backend
    be, emu = default_backend(), TorchEmulatorBackend()
    rs = np.random.RandomState(seed)
    n_src, n_dst, E = 70, 50, 400 + hub_deg
    src = rs.randint(0, n_src - 5, size=E)           # the last 5 sources have no out-edge
    dst = rs.randint(1, n_dst, size=E)               # destination 0 has no in-edge
    dst[:hub_deg] = 7
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    xs = torch.from_numpy(rs.uniform(0, 1, (n_src, 8)).astype(np.float32))
    xd = torch.from_numpy(rs.uniform(0, 1, (n_dst, 8)).astype(np.float32))
    ea = torch.from_numpy(rs.uniform(0.01, 0.1, E).astype(np.float32))
    for G, has_h in ((4, True), (3, False)):
        mk = lambda *s: torch.from_numpy(rs.uniform(-1, 1, s).astype(np.float32))
        p_dst = mk(n_dst, G * 112 if has_h else G * 16)
        v, h, ep = mk(n_src, G * 96), (mk(n_src, 96) if has_h else None), mk(G, 3, 96)
        g_agg = mk(n_dst, G * 128)
        offs = (0, 0, G * 96 if has_h else 0, 0, 128, 96)
        res = {}
        for name, b, dev in (("hip", be, "cuda"), ("emu", emu, "cpu")):
            t = lambda a: None if a is None else a.to(dev)
            csr = b.build_csr(t(ei), n_src, n_dst)
            rcsr = b.build_csr(t(ei).flip(0).contiguous(), n_dst, n_src)
            inv = torch.empty(E, dtype=torch.int32, device=dev)
            inv[csr.perm[:E].long()] = torch.arange(E, dtype=torch.int32, device=dev)
            r_slot = inv[rcsr.perm[:E].long()].contiguous()
            einfo = torch.zeros(E + 3, 20, device=dev)
            b.edge_prepare([(csr, t(ea), t(xs), t(xd), einfo)])
            agg = torch.zeros(n_dst, G * 128, device=dev)
            b.aggregate(csr, einfo, t(v), t(p_dst), t(h), t(ep), agg, *offs, G)
            res[name] = b.aggregate_backward(csr, rcsr, r_slot, einfo, t(v), t(p_dst), t(h), t(ep), agg,
                                             t(g_agg), *offs, G)
        for a, b_, what in zip(res["hip"], res["emu"], ("g_p_dst", "g_p_src", "g_h_src", "g_ep")):
            if b_ is None:
                assert a is None
                continue
            err, scale = float((a.cpu() - b_).abs().max()), float(b_.abs().max())
            assert err <= GTOL * scale, (G, what, err, scale)


@pytest.mark.gpu
def test_bf16_autocast_training_step_stays_close_to_fp32():
    """BASELINE config 5 names bf16: library GEMMs under torch.autocast, the sweep stays fp32."""
    x, ei, ea = load_graph("40")
    y_np, m_np = _targets(x, ei)
    y, mask = tt(y_np, "cuda"), tt(m_np, "cuda")
    R, _ = product_models(10020, 1.0, "cuda")
    R.train()
    X, EI, EA = tt(x, "cuda"), tt(ei, "cuda"), tt(ea, "cuda")
    ref = training.regressor_loss(y, R(X, EI, EA), mask)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        low = training.regressor_loss(y, R(X, EI, EA), mask)
    low.backward()
    assert abs(float(low) - float(ref)) <= 2e-2 * abs(float(ref))
    g = R.gclstm_decoder.cell_list[0].conv_i.convs["joint__connect__joint"].lin_value.weight.grad
    assert g is not None and g.dtype == torch.float32 and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0


BF16_GTOL = 5e-2   # bf16 autocast: 8 significand bits (unit roundoff 2^-9) through the stacked
#                     linears of two cells; per-tensor gradient error against the fp32 oracle


@pytest.mark.gpu
def test_cfg5_collated_minibatch_fp32_and_bf16_against_the_fp32_oracle():
    """BASELINE config 5 as a config: a mini-batch of 4 graphs collated into one disjoint-union
    graph with offset indices (what train.py:365-366's DataLoader(batch_size=4) hands the model),
    (1) fp32 losses and all 568 parameter gradients against the oracle's autograd at 2e-4;
    (2) the same under torch.autocast(bf16) against the FP32 oracle with the bf16 bound above;
    (3) two Adam steps of the reference's loop (train.py:82, 158-166), fp32 and bf16."""
    from graingraphnn_amd import synthetic
    x0, ei0, ea0 = load_graph("40")
    x, ei, ea, _ = synthetic.disjoint_union(
        [(synthetic.perturbed_copy(x0, 1e-3, 1000 + t), ei0, ea0) for t in range(4)])
    assert x["grain"].shape[0] == 4 * x0["grain"].shape[0]
    R, Cm = product_models(10020, 1.0, "cuda")
    oR, oC = oracle_models(10020, 1.0)
    olr, olc, ref32 = _grads(oR, oC, x, ei, ea)
    # The gradients are held to the oracle evaluated in fp64: in fp32 the oracle itself is 1.5e-6 of the model's
    # largest gradient entry off on the decoder's forget-gate value weights of this batch (the product's error
    # against fp64 is 1e-8 there), which is more than the floor of _check_full allows either side.
    _, _, ref = _grads(oR.double(), oC.double(), x, ei, ea, dtype=torch.float64)
    ref = {k: v.float() for k, v in ref.items()}
    gmax = max(float(g.abs().max()) for g in ref.values())
    assert max(float((ref32[k] - ref[k]).abs().max()) for k in ref) <= 5e-6 * gmax     # the fp32 oracle's own noise
    oR.float(), oC.float()
    lr, lc, grads = _grads(R, Cm, x, ei, ea, "cuda")
    assert abs(lr - olr) <= 1e-5 * abs(olr) and abs(lc - olc) <= 1e-5 * abs(olc)
    worst32 = _check_full(grads, ref)
    # (2) bf16 autocast (library GEMMs in bf16, the sweep and the attention logits in fp32)
    y_np, m_np = _targets(x, ei)
    y, mask = tt(y_np, "cuda"), tt(m_np, "cuda")
    X, EI, EA = tt(x, "cuda"), tt(ei, "cuda"), tt(ea, "cuda")
    R.zero_grad(), Cm.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        blr = training.regressor_loss(y, R(X, EI, EA), mask)
        blc = training.classifier_loss(y, Cm(X, EI, EA), 1.0)
    blr.backward()
    blc.backward()
    assert abs(float(blr) - olr) <= 2e-2 * abs(olr) and abs(float(blc) - olc) <= 2e-2 * abs(olc)
    worst16 = 0.0
    for tag, m in (("R", R), ("C", Cm)):
        gmax = max(float(ref[f"{tag}/{n}"].abs().max()) for n, _ in m.named_parameters())
        for name, p in m.named_parameters():
            # (parameters behind the classifier's dead decoder outputs get no gradient: zero, as in _grads)
            g, r = (torch.zeros_like(p) if p.grad is None else p.grad), ref[f"{tag}/{name}"]
            assert g.dtype == torch.float32 and bool(torch.isfinite(g).all()), name
            err, scale = float((g.cpu() - r).abs().max()), float(r.abs().max())
            # floor: tensors whose exact gradient is tiny next to the model's largest one
            assert err <= BF16_GTOL * scale + 2e-3 * gmax, (tag, name, err, scale)
            if scale > 1e-2 * gmax:
                worst16 = max(worst16, err / scale)
    print(f"cfg5 batch of 4: worst per-tensor gradient error fp32 {worst32:.2e}, bf16 autocast {worst16:.2e}")
    # (3) two Adam steps, as train.py does them
    lr_adam = 5e-3                                                   # parameters.py:18-50 regressor lr
    for autocast in (False, True):
        Rt, _ = product_models(4, 1.0, "cuda")
        oRt, _ = oracle_models(4, 1.0)
        g_first = {}
        for m, dev in ((Rt, "cuda"), (oRt, "cpu")):
            m.train()
            opt = torch.optim.Adam(m.parameters(), lr=lr_adam)
            for it in range(2):
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast and dev == "cuda"):
                    loss = training.regressor_loss(tt(y_np, dev), m(tt(x, dev), tt(ei, dev), tt(ea, dev)), tt(m_np, dev))
                opt.zero_grad()
                loss.backward()
                if dev == "cpu" and it == 0:
                    g_first = {n: float(p.grad.abs().max()) for n, p in m.named_parameters()}
                opt.step()
        gmax = max(g_first.values())
        for (n, p), (_, q) in zip(Rt.named_parameters(), oRt.named_parameters()):
            # Adam moves a weight by ~lr per step whatever the gradient's size: where the exact
            # gradient is zero (key biases: a softmax is shift-invariant; the encoder's forget gate)
            # each implementation steps along its own rounding noise, so those tensors are only held
            # to 2 steps x lr each way.  Under bf16 every tensor is held to that bound: a gradient
            # entry near zero may change sign.
            noise_only = g_first[n] <= 1e-5 * gmax
            bound = 4 * lr_adam * 1.01 if (autocast or noise_only) else 1e-3 * max(float(q.abs().max()), 1e-3)
            assert float((p.detach().cpu() - q.detach()).abs().max()) <= bound, (autocast, n, noise_only)


@pytest.mark.gpu
def test_ddp_over_rccl_as_the_reference_wraps_it():
    """dist_train.py:76-82 on one GPU, over RCCL: init_process_group('nccl', device_id=cuda:0), the packed
    all-gather of graingraphnn_amd.dist on device buffers, DistributedDataParallel(model, device_ids=[0]) for
    two iterations with gradients bit-equal to the unwrapped model.  Runs in a child process
    (tests/rccl_worker.py) so that no communicator outlives the test inside pytest."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_worker.py"),
                        str(port)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    print(r.stdout.strip().splitlines()[-1])


def _run_two_ranks(backend, n_dev, world=2):
    """Start tests/rccl_worker2.py `world` times (one process per rank, as torchrun would) and wait for all."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_worker2.py")
    procs = [subprocess.Popen([sys.executable, script, str(port), str(r), backend, str(n_dev), str(world)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK{r}_OK" in so, (r, so[-2000:], se[-4000:])


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_shard_trajectories_and_average_gradients():
    """tests/rccl_worker2.py with both ranks on cuda:0 (RCCL refuses two ranks per device: the group is gloo, the
    collectives are staged through the host): gather_states' multi-rank branch on device tensors, config 4 in small
    sharded over two ranks == one rank bit for bit, DDP gradients == the mean of the ranks' local gradients.  The
    RCCL run of the same file needs two GPUs (next test)."""
    _run_two_ranks("gloo", 1)


@pytest.mark.gpu
def test_two_ranks_over_rccl_when_two_gpus_are_visible():
    """The same two-rank worker over RCCL ('nccl', init_process_group(device_id=...), one GPU per rank, device-buffer
    collectives over xGMI) and `bench.py --gpus 2` reporting two RCCL ranks.  Skipped on a one-GPU box."""
    import json
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("needs two GPUs (one RCCL rank per device)")
    _run_two_ranks("nccl", 2)
    if n_dev > 2:   # every visible GPU: config 4's 64 trajectories over min(n, 8) RCCL ranks
        _run_two_ranks("nccl", min(n_dev, 8), world=min(n_dev, 8))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["scaling"] == "weak" and line["value"] > 0


def _ddp_gpu_worker(rank, world, port, out):
    """One of two ranks that share cuda:0 (the GPU box has one device and RCCL refuses two ranks
    per device, so the process group is gloo; DistributedDataParallel, the bucketed gradient
    all-reduce and the HIP training path are the ones an 8-GPU RCCL job runs)."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.set_num_threads(1)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graingraphnn_amd import synthetic
        x0, ei0, ea0 = load_graph("40")
        x, ei, ea, _ = synthetic.disjoint_union(
            [(synthetic.perturbed_copy(x0, 1e-3, 2000 + 4 * rank + t), ei0, ea0) for t in range(4)])
        y_np, m_np = _targets(x, ei)
        R, _ = product_models(4, 1.0, "cuda")
        R.train()
        model = DistributedDataParallel(R, device_ids=[0])
        X, EI, EA = tt(x, "cuda"), tt(ei, "cuda"), tt(ea, "cuda")
        for _ in range(2):
            model.zero_grad()
            training.regressor_loss(tt(y_np, "cuda"), model(X, EI, EA), tt(m_np, "cuda")).backward()
        if rank == 0:
            torch.save({n: p.grad.cpu() for n, p in R.named_parameters()}, out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_ddp_on_the_hip_path_averages_minibatch_gradients(tmp_path):
    """dist_train.py:76-91 with two ranks: each rank gets its own collated mini-batch of 4 graphs
    (DistributedSampler), DDP averages the gradients; reference = mean of the oracle's
    single-process gradients of the two mini-batches."""
    import socket
    import torch.multiprocessing as mp
    from graingraphnn_amd import synthetic
    x0, ei0, ea0 = load_graph("40")
    ref = None
    for rank in range(2):
        x, ei, ea, _ = synthetic.disjoint_union(
            [(synthetic.perturbed_copy(x0, 1e-3, 2000 + 4 * rank + t), ei0, ea0) for t in range(4)])
        y_np, m_np = _targets(x, ei)
        oR, _ = oracle_models(4, 1.0)
        oR.train()
        training.regressor_loss(tt(y_np), oR(tt(x), tt(ei), tt(ea)), tt(m_np)).backward()
        g = {n: p.grad.clone() for n, p in oR.named_parameters()}
        ref = g if ref is None else {n: 0.5 * (ref[n] + g[n]) for n in g}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "grads.pt")
    mp.spawn(_ddp_gpu_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    atol = 1e-6 * max(float(g.abs().max()) for g in ref.values())
    for n, g in ref.items():
        assert float((got[n] - g).abs().max()) <= GTOL * float(g.abs().max()) + atol, n


@pytest.mark.gpu
def test_sweep_backward_full_size_properties():
    """cfg3 size (oracle autograd would take minutes): the backward is linear in the incoming
    gradient, reproducible, and its destination-side part matches a float64 torch restatement
    on a sample of rows."""
    from graingraphnn_amd import synthetic
    from graingraphnn_amd.backend import default_backend
    be = default_backend()
    x, ei, ea = synthetic.honeycomb(100, 10, 0)
    et = EDGE_TYPES[0]                                   # grain -> joint: 10 000 sources, 20 000 destinations
    n_src, n_dst, E, G = x["grain"].shape[0], x["joint"].shape[0], ei[et].shape[1], 4
    EI = torch.from_numpy(ei[et]).cuda()
    xs, xd = torch.from_numpy(x["grain"]).cuda(), torch.from_numpy(x["joint"]).cuda()
    gen = torch.Generator(device="cuda").manual_seed(3)
    mk = lambda *s: torch.rand(*s, device="cuda", generator=gen) * 2 - 1
    p_dst, v, h, ep = mk(n_dst, G * 112), mk(n_src, G * 96), mk(n_src, 96), mk(G, 3, 96)
    csr = be.build_csr(EI, n_src, n_dst)
    rcsr = be.build_csr(EI.flip(0).contiguous(), n_dst, n_src)
    inv = torch.empty(E, dtype=torch.int32, device="cuda")
    inv[csr.perm[:E].long()] = torch.arange(E, dtype=torch.int32, device="cuda")
    r_slot = inv[rcsr.perm[:E].long()].contiguous()
    einfo = torch.zeros(E + 3, 20, device="cuda")
    be.edge_prepare([(csr, torch.from_numpy(ea[et]).cuda().view(-1), xs, xd, einfo)])
    offs = (0, 0, G * 96, 0, 128, 96)
    agg = torch.zeros(n_dst, G * 128, device="cuda")
    be.aggregate(csr, einfo, v, p_dst, h, ep, agg, *offs, G)
    bwd = lambda g: be.aggregate_backward(csr, rcsr, r_slot, einfo, v, p_dst, h, ep, agg, g, *offs, G)
    g1, g2 = mk(n_dst, G * 128), mk(n_dst, G * 128)
    a, b, c, again = bwd(g1), bwd(g2), bwd(g1 + 0.5 * g2), bwd(g1)
    for ta, tb, tc, td, what in zip(a, b, c, again, ("g_p_dst", "g_p_src", "g_h_src", "g_ep")):
        assert torch.equal(ta, td), what                 # no atomics: bit-reproducible
        scale = float(tc.abs().max())
        assert float((ta + 0.5 * tb - tc).abs().max()) <= 2e-5 * scale, what
    # destination-side gradient of 64 sampled rows in float64
    rows = torch.arange(0, n_dst, n_dst // 64, device="cuda")[:64]
    rp, col = csr.rowptr.long(), csr.col.long()
    for i in rows.tolist():
        sl = slice(int(rp[i]), int(rp[i + 1]))
        j = col[sl]
        for g in range(G):
            uh = p_dst[i, g * 96:(g + 1) * 96].double().requires_grad_(True)
            u4 = p_dst[i, G * 96 + g * 16: G * 96 + (g + 1) * 16].double().requires_grad_(True)
            s = einfo[sl, :16].double() @ u4 + h[j].double() @ uh
            alpha = torch.softmax(s, 0)
            val = torch.relu(v[j, g * 96:(g + 1) * 96].double() + einfo[sl, 16:19].double() @ ep[g].double())
            out = torch.cat([(alpha[:, None] * val).sum(0), alpha.sum()[None], (alpha * einfo[sl, 19].double()).sum()[None]])
            (out * g1[i, g * 128: g * 128 + 98].double()).sum().backward()
            ref = torch.cat([uh.grad, u4.grad])
            got = torch.cat([a[0][i, g * 96:(g + 1) * 96], a[0][i, G * 96 + g * 16: G * 96 + (g + 1) * 16]]).double()
            # ds = alpha (dalpha - S) cancels 96-term sums of size ~5 down to ~1e-3 here: fp32 leaves
            # a few 1e-6 absolute (as it does in the reference's own fp32 autograd)
            assert float((got - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 5e-6, (i, g)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True, "ggnn"])
def test_graphed_train_step_follows_the_eager_step(fused):
    """training.GraphedTrainStep (forward, loss, backward, Adam replayed from one hipGraph) against the same
    steps run eagerly on a second copy of the model: losses and parameters agree (same kernels, same
    order), also with new targets copied into the captured buffers, and the .grad tensors of the captured step
    hold the last step's gradients.  fused: the replayed step with Adam(fused=True) (one optimizer launch) against
    the eager step with the default foreach Adam."""
    import copy
    from graingraphnn_amd import training
    x, ei, ea = load_graph("40")
    dev = "cuda"
    A, _ = product_models(4, 1.0, dev)
    B = copy.deepcopy(A)
    X, EI, EA = tt(x, dev), tt(ei, dev), tt(ea, dev)
    rs = np.random.RandomState(9)
    Ys = [{nt: torch.from_numpy(rs.uniform(-1, 1, (x[nt].shape[0], 2)).astype(np.float32)).to(dev) for nt in x}
          for _ in range(2)]
    mask = {nt: torch.ones(x[nt].shape[0], 1, device=dev) for nt in x}
    loss_fn = lambda pred, y: training.regressor_loss(y, pred, mask)
    if fused == "ggnn":   # (the whole update in one ggnn_adam_step launch)
        optA = training.FusedAdam(A.parameters(), lr=1e-3)
    else:
        optA = torch.optim.Adam(A.parameters(), lr=1e-3, capturable=True, **({"fused": True} if fused else {}))
    optB = torch.optim.Adam(B.parameters(), lr=1e-3, capturable=True)
    step = training.GraphedTrainStep(A, optA, loss_fn, X, EI, EA, Ys[0], warmup=3)
    B.train()

    def eager(y):
        loss = loss_fn(B(X, EI, EA), y)
        optB.zero_grad(set_to_none=False)
        loss.backward()
        optB.step()
        return loss
    for _ in range(3):                      # the warm-up steps updated A (the capture only records)
        eager(Ys[0])
    for k in range(4):
        y = Ys[k % 2]
        la, lb = step(X, EA, y), eager(y)
        assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb)), (k, float(la), float(lb))
    gmax = max(float(p.grad.abs().max()) for p in B.parameters())
    for (n, pa), (_, pb) in zip(A.named_parameters(), B.named_parameters()):
        # Adam moves a weight by ~lr per step whatever the gradient's size: where the gradient is rounding noise
        # around zero (key biases / weights: a softmax is shift-invariant) the fused and the foreach update may step
        # in different directions, so those tensors are only held to 7 steps x lr each way
        d = (pa - pb).abs()
        if fused:
            noise = pb.grad.abs() <= 1e-4 * gmax
            assert float(d[noise].max() if bool(noise.any()) else 0.0) <= 2 * 7 * 1e-3, n
            d = d[~noise]
            assert float(d.max() if d.numel() else 0.0) <= 2e-4, n    # (lr 1e-3: a fifth of one step)
        else:
            assert torch.allclose(pa, pb, rtol=1e-4, atol=1e-6), n
        # (floor relative to the model's largest gradient entry: some gradients are rounding noise around zero, and
        # the two Adam implementations leave the parameters a rounding apart)
        assert pa.grad is not None and bool(torch.isfinite(pa.grad).all()), n
        if not fused:  # (fused: the parameters already differ by up to a step in the noise directions)
            assert torch.allclose(pa.grad, pb.grad, rtol=1e-3, atol=1e-5 * gmax), n


@pytest.mark.gpu
def test_captured_fused_adam_step_follows_a_learning_rate_schedule():
    """train.py:91 puts StepLR(gamma = 0.5) on top of Adam.  A FusedAdam.step() captured in a hipGraph reads its learning
    rates from device memory, refreshed by sync_hyperparams() between replays: ten replayed steps under StepLR(step_size
    = 3) with a second group's weight decay changed half-way equal eager torch.optim.Adam under the same schedule (with
    the rates passed by value the replays would keep the captured ones: the parameters end ~40 % away)."""
    from graingraphnn_amd import training
    dev = "cuda"
    rs = np.random.RandomState(21)
    shapes = [(96, 104), (4097,), (7,)]
    pa = [torch.nn.Parameter(torch.from_numpy(rs.standard_normal(sh).astype(np.float32)).to(dev)) for sh in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    groups = lambda ps: [{"params": ps[:2]}, {"params": ps[2:], "lr": 4e-3, "weight_decay": 0.02}]
    oa = training.FusedAdam(groups(pa), lr=2e-3)
    ob = torch.optim.Adam(groups(pb), lr=2e-3)
    sa = torch.optim.lr_scheduler.StepLR(oa, step_size=3, gamma=0.5)
    sb = torch.optim.lr_scheduler.StepLR(ob, step_size=3, gamma=0.5)
    grads = [torch.zeros_like(p) for p in pa]          # static gradient buffers: what a captured backward writes into
    for p, g in zip(pa, grads):
        p.grad = g
    oa.sync_hyperparams()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            oa.step()
    torch.cuda.current_stream().wait_stream(st)
    start = [p.detach().clone() for p in pa]
    for k in range(10):
        for g, pbk in zip(grads, pb):
            g.copy_(torch.from_numpy((rs.standard_normal(g.shape) * 0.1).astype(np.float32)).to(dev))
            pbk.grad = g.clone()
        if k == 5:
            for o in (oa, ob):
                o.param_groups[1]["weight_decay"] = 0.05
        oa.sync_hyperparams()      # (GraphedTrainStep.__call__ does this before every replay)
        graph.replay()
        ob.step()
        sa.step()
        sb.step()
    assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"] == 2e-3 * 0.5 ** 3
    for i, (a, b, a0) in enumerate(zip(pa, pb, start)):
        moved = float((b - a0).abs().max())
        # (a few roundings of a parameter of magnitude ~3 on top of the update's own tolerance; a replay that kept the
        # captured rates would be off by ~0.4 x moved)
        assert float((a - b).abs().max()) <= 2e-5 * moved + 1e-6 * float(b.abs().max()), (i, float((a - b).abs().max()), moved)
    assert float(oa.state[pa[0]]["step"]) == 10.0


@pytest.mark.gpu
def test_fused_adam_follows_torch_adam():
    """training.FusedAdam (ggnn_adam_step: every tensor in one launch, step count on the device) against torch.optim.Adam
    on copies of the same tensors: two groups with their own learning rate and weight decay, a learning rate changed
    between steps (StepLR does that), a tensor without a gradient in some steps, sizes around the 4096-element chunk; the
    state dict round-trips into a fresh optimizer that continues identically."""
    from graingraphnn_amd import training
    dev = "cuda"
    rs = np.random.RandomState(3)
    shapes = [(1,), (7,), (4096,), (4097,), (96, 104), (3, 96, 224), (2112, 108), (5,)]
    mk = lambda: [torch.nn.Parameter(torch.from_numpy(rs.standard_normal(sh).astype(np.float32)).to(dev)) for sh in shapes]
    rs = np.random.RandomState(3)
    pa = mk()
    rs = np.random.RandomState(3)
    pb = mk()
    groups = lambda ps: [{"params": ps[:3], "lr": 2e-3, "weight_decay": 0.01}, {"params": ps[3:]}]
    oa = training.FusedAdam(groups(pa), lr=1e-3, betas=(0.9, 0.99), eps=1e-8)
    ob = torch.optim.Adam(groups(pb), lr=1e-3, betas=(0.9, 0.99), eps=1e-8)

    def steps(oa, pa, ob, pb, n, seed):
        rg = np.random.RandomState(seed)
        for k in range(n):
            for i, (a, b) in enumerate(zip(pa, pb)):
                if i == 4 and k % 2 == 1:
                    a.grad = b.grad = None
                    continue
                g = torch.from_numpy((rg.standard_normal(a.shape) * 10.0 ** rg.uniform(-6, 0)).astype(np.float32)).to(dev)
                a.grad, b.grad = g.clone(), g.clone()
            if k == 2:
                for o in (oa, ob):
                    o.param_groups[1]["lr"] *= 0.5
            oa.step()
            ob.step()
    steps(oa, pa, ob, pb, 5, 11)
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (i, float((a - b).abs().max()))
        assert torch.allclose(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], rtol=1e-5, atol=1e-30), i
    assert float(oa.state[pa[0]]["step"]) == 5.0 and float(oa.state[pa[4]]["step"]) == 3.0   # (no gradient in steps 1, 3)
    # state dict -> a fresh optimizer over the same parameters continues the same trajectory
    sd = oa.state_dict()
    oc = training.FusedAdam(groups(pa), lr=1e-3, betas=(0.9, 0.99), eps=1e-8)
    oc.load_state_dict(sd)
    assert oc.param_groups[1]["lr"] == oa.param_groups[1]["lr"] == 5e-4
    steps(oc, pa, ob, pb, 3, 12)
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.allclose(a, b, rtol=4e-6, atol=4e-7), (i, float((a - b).abs().max()))
    assert float(oc.state[pa[0]]["step"]) == 8.0 and float(oc.state[pa[4]]["step"]) == 5.0
    # a parameter group added later (train.py:84-91 builds its groups up front; torch allows it afterwards): the moments
    # of the tensors already there are carried over into the rebuilt buffers
    extra_a = torch.nn.Parameter(torch.full((10,), 0.5, device=dev))
    extra_b = torch.nn.Parameter(extra_a.detach().clone())
    oc.add_param_group({"params": [extra_a], "lr": 1e-2})
    ob.add_param_group({"params": [extra_b], "lr": 1e-2})
    pa2, pb2 = pa + [extra_a], pb + [extra_b]
    steps(oc, pa2, ob, pb2, 2, 13)
    for i, (a, b) in enumerate(zip(pa2, pb2)):
        assert torch.allclose(a, b, rtol=6e-6, atol=6e-7), (i, float((a - b).abs().max()))
    assert float(oc.state[pa[0]]["step"]) == 10.0 and float(oc.state[extra_a]["step"]) == 2.0
    with pytest.raises(Exception, match="MI355X|CPU"):
        training.FusedAdam([torch.nn.Parameter(torch.zeros(3))], lr=1e-3).step()
    # more tensors than one call carries (GGNN_ADAM_MAX_TENSORS = 384): two calls per update, each advancing its own counts
    rs = np.random.RandomState(8)
    many_a = [torch.nn.Parameter(torch.from_numpy(rs.standard_normal(1 + k % 37).astype(np.float32)).to(dev)) for k in range(500)]
    many_b = [torch.nn.Parameter(p.detach().clone()) for p in many_a]
    oa, ob = training.FusedAdam(many_a, lr=3e-3), torch.optim.Adam(many_b, lr=3e-3)
    for k in range(3):
        for a, b in zip(many_a, many_b):
            g = torch.from_numpy(rs.standard_normal(a.shape).astype(np.float32)).to(dev)
            a.grad, b.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    assert len(oa._built["launches"]) == 2
    for i, (a, b) in enumerate(zip(many_a, many_b)):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), i
    assert float(oa.state[many_a[499]]["step"]) == 3.0


@pytest.mark.gpu
@pytest.mark.parametrize("row_mask", [True, False])
def test_fused_regressor_loss_equals_the_recorded_expression(row_mask):
    """training.regressor_loss on the GPU (ggnn_masked_mse: loss and gradient in one launch, a fixed summation order)
    against the same expression as recorded torch ops (train.py:31-37), with a mask per row (the reference's) and per
    element, an upstream gradient other than one, and bit-identical repeats."""
    from graingraphnn_amd import training
    dev = "cuda"
    rs = np.random.RandomState(5)
    n = {"joint": 20011, "grain": 9973}
    f = lambda *sh: torch.from_numpy(rs.uniform(-1, 1, sh).astype(np.float32)).to(dev)
    y = {nt: f(n[nt], 2) for nt in n}
    mask = {nt: (f(n[nt], 1) > -0.5).float() if row_mask else (f(n[nt], 2) > -0.5).float() for nt in n}
    res = []
    for fn in (training.regressor_loss, training.regressor_loss_recorded, training.regressor_loss):
        rs2 = np.random.RandomState(6)
        pred = {nt: torch.from_numpy(rs2.uniform(-1, 1, (n[nt], 2)).astype(np.float32)).to(dev).requires_grad_() for nt in n}
        loss = fn(y, pred, mask)
        (loss * 0.37).backward()
        res.append((loss.detach(), pred["joint"].grad, pred["grain"].grad))
    (la, gja, gga), (lb, gjb, ggb), (lc, gjc, ggc) = res
    assert abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb)), (float(la), float(lb))
    assert torch.allclose(gja, gjb, rtol=1e-5, atol=1e-9) and torch.allclose(gga, ggb, rtol=1e-5, atol=1e-9)
    assert torch.equal(la, lc) and torch.equal(gja, gjc) and torch.equal(gga, ggc)


@pytest.mark.gpu
@pytest.mark.parametrize("batch,rows,cols", [(3, 768, 1152), (1, 313, 400), (2, 1, 4), (1, 31, 36), (1, 257, 8196)])
def test_sum_rows_is_a_fixed_order_column_sum(batch, rows, cols):
    """ggnn_sum_rows (the reduction over the sweep backward's per-workgroup partial sums and over many small ggnn_wgrad
    partials): against the fp64 column sums within fp32 summation error, bit-identical on repeats, row counts that are not a
    multiple of the 32 row groups, column counts that do not fill the last workgroup."""
    from graingraphnn_amd.backend import default_backend
    be = default_backend()
    g = torch.Generator().manual_seed(rows + cols)
    t = (torch.randn(batch, rows, cols, generator=g) * 10.0 ** torch.empty(batch, rows, 1).uniform_(-4, 0, generator=g)).cuda()
    out = be.sum_rows(t)
    ref, mag = t.double().sum(1), t.double().abs().sum(1)
    assert out.shape == (batch, cols)
    assert bool(((out.double() - ref).abs() <= 2e-6 * mag + 1e-30).all()), float((out.double() - ref).abs().max())
    assert torch.equal(out, be.sum_rows(t))
    with pytest.raises(Exception):
        be.sum_rows(t[:, :, :cols - 1].contiguous())      # cols % 4 != 0


@pytest.mark.gpu
def test_graphed_train_step_survives_cache_eviction():
    """The captured step holds raw addresses of tensors that only evictable caches own (constant blocks of
    train_pack, the reverse CSR of training._topo_cache, the CSR tables of engine._graph_cache).  Between two
    replays: eager steps on a dozen OTHER topologies of other sizes (more than every cache holds), the allocator's
    cache emptied and re-filled with garbage.  The replayed steps must still follow the eager steps of a twin
    model -- the step object pins what its graph reads (graingraphnn_amd/_pins.py)."""
    import copy
    from graingraphnn_amd import engine, train_pack, training
    x, ei, ea = load_graph("40")
    dev = "cuda"
    A, _ = product_models(4, 1.0, dev)
    B = copy.deepcopy(A)
    X, EI, EA = tt(x, dev), tt(ei, dev), tt(ea, dev)
    rs = np.random.RandomState(19)
    Y = {nt: torch.from_numpy(rs.uniform(-1, 1, (x[nt].shape[0], 2)).astype(np.float32)).to(dev) for nt in x}
    mask = {nt: torch.ones(x[nt].shape[0], 1, device=dev) for nt in x}
    loss_fn = lambda pred, y: training.regressor_loss(y, pred, mask)
    optA = torch.optim.Adam(A.parameters(), lr=1e-3, capturable=True)
    optB = torch.optim.Adam(B.parameters(), lr=1e-3, capturable=True)
    step = training.GraphedTrainStep(A, optA, loss_fn, X, EI, EA, Y, warmup=2)
    assert len(step._pins) > 0
    B.train()

    def eager(model, opt, Xd, EId, EAd, y, m):
        loss = training.regressor_loss(y, model(Xd, EId, EAd), m)
        opt.zero_grad(set_to_none=False)
        loss.backward()
        opt.step()
        return loss
    for _ in range(2):
        eager(B, optB, X, EI, EA, Y, mask)
    la, lb = step(X, EA, Y), eager(B, optB, X, EI, EA, Y, mask)
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
    # a dozen other topologies and batch shapes through the same caches, on a throw-away model
    T, _ = product_models(5, 1.0, dev)
    T.train()
    optT = torch.optim.Adam(T.parameters(), lr=1e-3)
    for k in range(12):
        gx, gei, gea = synthetic.voronoi(20 + 3 * k, seed=100 + k)
        Xk, EIk, EAk = tt(gx, dev), tt(gei, dev), tt(gea, dev)
        yk = {nt: torch.zeros(gx[nt].shape[0], 2, device=dev) for nt in gx}
        mk = {nt: torch.ones(gx[nt].shape[0], 1, device=dev) for nt in gx}
        eager(T, optT, Xk, EIk, EAk, yk, mk)
    assert len(engine._graph_cache) <= engine._GRAPH_CACHE_MAX and len(training._topo_cache) <= 8
    del T, optT
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(64)]   # re-use freed blocks, poisoned
    del junk
    for k in range(3):
        la, lb = step(X, EA, Y), eager(B, optB, X, EI, EA, Y, mask)
        assert np.isfinite(float(la)) and abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb)), (k, float(la), float(lb))
    for (n, pa), (_, pb) in zip(A.named_parameters(), B.named_parameters()):
        assert torch.allclose(pa, pb, rtol=1e-4, atol=1e-6), n


@pytest.mark.gpu
@pytest.mark.parametrize("K,M,Nc,batch", [(20000, 1984, 108, 1), (20000, 96, 224, 4), (10000, 96, 128, 3),
                                           (472, 480, 12, 1), (77, 96, 12, 1), (15, 16, 4, 2), (1000, 1060, 116, 1),
                                           (3, 4, 4, 1), (20001, 288, 8, 1)])
def test_wgrad_is_the_transposed_product(K, M, Nc, batch):
    """ggnn_wgrad (C = A^T B, reduction over the nodes split across the chip) against the fp64 product: operands
    with row pitches wider than the used columns, batches addressed by element offsets, K not a multiple of the
    4-row MFMA group (and shorter than the register ring), Nc / M below and not a multiple of the wave's block.  The
    bound is that of an fp32 dot product (3e-6 of sum |a||b|); the result is reproducible bit for bit."""
    from graingraphnn_amd.backend import default_backend
    be = default_backend()
    g = torch.Generator().manual_seed(K + M)
    lda, ldb = M + 8, batch * Nc + 4
    a = torch.randn(batch, K, lda, generator=g).cuda()
    b = torch.randn(K, ldb, generator=g).cuda()
    c = be.wgrad(a, b, K, M, Nc, lda, ldb, batch=batch, a_bstride=K * lda, b_bstride=Nc)
    assert c.shape == (batch, M, Nc)
    for k in range(batch):
        a64, b64 = a[k, :, :M].double(), b[:, k * Nc:(k + 1) * Nc].double()
        ref, mag = a64.t() @ b64, a64.abs().t() @ b64.abs()
        assert bool(((c[k].double() - ref).abs() <= 3e-6 * mag + 1e-30).all()), (k, float((c[k] - ref).abs().max()))
    assert torch.equal(c, be.wgrad(a, b, K, M, Nc, lda, ldb, batch=batch, a_bstride=K * lda, b_bstride=Nc))
    # gradients are small numbers: rows scaled down to 1e-12 .. 1e-6 keep the bound (the kernel splits its operands into
    # three bf16 pieces -- fp32's range; the cells' two-piece fp16 split would lose them)
    tiny = a * (10.0 ** torch.empty(batch, K, 1).uniform_(-12, -6, generator=g)).cuda()
    ct = be.wgrad(tiny, b, K, M, Nc, lda, ldb, batch=batch, a_bstride=K * lda, b_bstride=Nc)
    for k in range(batch):
        a64, b64 = tiny[k, :, :M].double(), b[:, k * Nc:(k + 1) * Nc].double()
        ref, mag = a64.t() @ b64, a64.abs().t() @ b64.abs()
        assert bool(((ct[k].double() - ref).abs() <= 3e-6 * mag + 1e-38).all()), (k, float((ct[k] - ref).abs().max()))
    with pytest.raises(Exception):
        be.wgrad(a, b, K + 1, M, Nc, lda, ldb, batch=batch, a_bstride=K * lda, b_bstride=Nc)   # rows beyond the tensor
    with pytest.raises(Exception):
        be.wgrad(a, b, K, M, Nc - 1, lda, ldb, batch=batch, a_bstride=K * lda, b_bstride=Nc)   # Nc % 4 != 0


@pytest.mark.parametrize("encoder", [True, False])
def test_index_table_packing_equals_the_recorded_torch_ops(encoder):
    """train_pack._PackWeights (gathers through index tables + 3 bmm, hand-written backward) against the same
    layout definition run on the parameter values under autograd: identical packed matrices (bit for bit: every
    entry is one parameter, a sum in the same order, or the same bmm), gradients of a random functional of all nine
    outputs to fp32 rounding, exactly zero gradients for the encoder's unused forget gate."""
    from graingraphnn_amd import train_pack
    R, _ = product_models(123, 1.0)
    cell = (R.gclstm_encoder if encoder else R.gclstm_decoder).cell_list[0]
    gates, sees_h = ("ico", False) if encoder else ("ifco", True)
    F = cell.in_channels_dict
    g = torch.Generator().manual_seed(5)
    res = {}
    for name, fn in (("tables", train_pack.packed_weights), ("ops", train_pack.packed_weights_ops)):
        R.zero_grad()
        layout, wp, bp, ep, w2 = fn(cell, gates, F, sees_h)
        outs = train_pack._outputs_in_order(wp, bp, ep, w2)
        if name == "tables":
            probes = [torch.randn(o.shape, generator=g) for o in outs]
        sum((o * p).sum() for o, p in zip(outs, probes)).backward()
        res[name] = ([o.detach().clone() for o in outs],
                     {n: (None if p.grad is None else p.grad.clone()) for n, p in cell.named_parameters()}, layout)
    for a, b in zip(res["tables"][0], res["ops"][0]):
        assert a.shape == b.shape and torch.equal(a, b)
    for nt in ("grain", "joint"):
        assert vars(res["tables"][2][nt]) == vars(res["ops"][2][nt])
    n_checked = 0
    for n, gt in res["tables"][1].items():
        go = res["ops"][1][n]
        if encoder and (n.startswith("conv_f.") or n.startswith("b_f.")):
            assert gt is not None and not bool(gt.any()), n         # read by the reference, contributes nothing
            continue
        assert gt is not None and go is not None, n
        scale = max(float(go.abs().max()), 1e-6)
        assert float((gt - go).abs().max()) <= 2e-6 * scale + 1e-7, n
        n_checked += 1
    assert n_checked >= 100


@pytest.mark.gpu
@pytest.mark.parametrize("encoder,width", [(True, 96), (False, 96)])
def test_pack_kernels_equal_the_recorded_torch_ops_on_the_gpu(encoder, width):
    """ggnn_pack_weights / ggnn_pack_weights_backward (csrc/pack.hip: the packing of a cell's parameters in two launches,
    its backward in four) against the layout definition run as recorded torch ops on the same device: the packed matrices
    (entries that are one parameter or a sum of parameters bit for bit, the product entries to fp32 rounding of a 96-term
    sum) and the gradient of a random functional of all nine outputs with respect to every parameter.  (Narrow layers go
    through the same kernels with zero-padded tables: test_hip_training_gradients_of_narrow_layers_match_the_oracle.)"""
    from graingraphnn_amd import train_pack
    R = product_models(123, 1.0, "cuda")[0] if width == 96 else _narrow_models(width, 123, "cuda")[0][0]
    cell = (R.gclstm_encoder if encoder else R.gclstm_decoder).cell_list[0]
    gates, sees_h = ("ico", False) if encoder else ("ifco", True)
    F = cell.in_channels_dict
    g = torch.Generator().manual_seed(5)
    res = {}
    for name, fn in (("hip", train_pack.packed_weights), ("ops", train_pack.packed_weights_ops)):
        R.zero_grad()
        layout, wp, bp, ep, w2 = fn(cell, gates, F, sees_h)
        outs = train_pack._outputs_in_order(wp, bp, ep, w2)
        if name == "hip":
            probes = [torch.randn(o.shape, generator=g).cuda() for o in outs]
        sum((o * p).sum() for o, p in zip(outs, probes)).backward()
        res[name] = ([o.detach().clone() for o in outs],
                     {n: (None if p.grad is None else p.grad.clone()) for n, p in cell.named_parameters()})
    for a, b in zip(res["hip"][0], res["ops"][0]):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 1e-6 * max(float(b.abs().max()), 1e-6)
    n_checked = 0
    for n, gt in res["hip"][1].items():
        go = res["ops"][1][n]
        if encoder and (n.startswith("conv_f.") or n.startswith("b_f.")):
            assert gt is not None and not bool(gt.any()), n
            continue
        assert gt is not None and go is not None, n
        assert float((gt - go).abs().max()) <= 2e-6 * max(float(go.abs().max()), 1e-6) + 1e-7, n
        n_checked += 1
    assert n_checked >= 100


@pytest.mark.gpu
@pytest.mark.parametrize("G", [4, 3])
def test_lstm_train_kernels_against_autograd_of_the_update(G):
    """ggnn_lstm_train_forward / _backward through the C ABI against torch autograd of heteropgclstm.py:140-183's
    update in fp64: z = gemm + skip is left in the buffer, h', c' to 1e-6; g_z (twice: the [G, N, 96] copy and the
    skip columns of the projection gradient, other columns untouched) and g_c to 1e-5 of their scale; absent
    gradients (None) are zeros; bad arguments are refused."""
    from graingraphnn_amd import _lib
    from graingraphnn_amd.backend import default_backend
    be = default_backend()
    g = torch.Generator().manual_seed(G)
    N, ldp, s_off = 777, 1056, 384
    gemm = torch.randn(G, N, 96, generator=g)
    P = torch.randn(N, ldp, generator=g)
    c_in = torch.randn(N, 96, generator=g) if G == 4 else None
    g_h, g_c = torch.randn(N, 96, generator=g), torch.randn(N, 96, generator=g)
    # reference in fp64
    zr = (gemm.double() + P[:, s_off:s_off + G * 96].double().view(N, G, 96).transpose(0, 1)).requires_grad_(True)
    cr = None if c_in is None else c_in.double().requires_grad_(True)
    i, o, ct = torch.sigmoid(zr[0]), torch.sigmoid(zr[G - 1]), torch.tanh(zr[G - 2])
    c_new = i * ct + (torch.sigmoid(zr[1]) * cr if G == 4 else 0.0)
    h_new = o * torch.tanh(c_new)
    for gh, gc in ((g_h, g_c), (g_h, None), (None, g_c)):
        zr.grad = None
        if cr is not None:
            cr.grad = None
        terms = ([(h_new * gh.double()).sum()] if gh is not None else []) + ([(c_new * gc.double()).sum()] if gc is not None else [])
        sum(terms).backward(retain_graph=True)
        z = gemm.clone().cuda()
        h_out, c_out = torch.empty(N, 96, device="cuda"), torch.empty(N, 96, device="cuda")
        cin_d = None if c_in is None else c_in.cuda()
        be.lstm_train_forward(z, P.cuda(), s_off, cin_d, h_out, c_out)
        assert float((z.double().cpu() - zr.detach()).abs().max()) <= 1e-6
        assert float((h_out.double().cpu() - h_new.detach()).abs().max()) <= 1e-6
        assert float((c_out.double().cpu() - c_new.detach()).abs().max()) <= 2e-6
        g_z = torch.empty_like(z)
        gP = torch.full((N, ldp), 7.0, device="cuda")
        g_cin = torch.empty(N, 96, device="cuda") if G == 4 else None
        be.lstm_train_backward(z, cin_d, c_out, None if gh is None else gh.cuda(), None if gc is None else gc.cuda(),
                               g_z, gP, s_off, g_cin)
        scale = float(zr.grad.abs().max())
        assert float((g_z.double().cpu() - zr.grad).abs().max()) <= 1e-5 * scale
        assert torch.equal(gP[:, s_off:s_off + G * 96].view(N, G, 96).transpose(0, 1), g_z)
        assert bool((gP[:, :s_off] == 7.0).all()) and bool((gP[:, s_off + G * 96:] == 7.0).all())
        if G == 4:
            assert float((g_cin.double().cpu() - cr.grad).abs().max()) <= 1e-5 * float(cr.grad.abs().max())
    with pytest.raises(_lib.GGNNError):
        be.lstm_train_forward(z, P.cuda(), s_off + 2, cin_d, h_out, c_out)                   # s_off % 4 != 0
    with pytest.raises(_lib.GGNNError):
        be.lstm_train_forward(z, P.cuda(), ldp - 96, cin_d, h_out, c_out)                    # skip columns beyond the row
    with pytest.raises(_lib.GGNNError):
        be.lstm_train_forward(z, P.cuda(), s_off, c_out if G == 3 else None, h_out, c_out)   # c_in must match the gate count


def test_second_derivatives_are_refused(monkeypatch):
    """The cell's backward is hand-written and not itself differentiable: asking for a second derivative
    (create_graph=True, e.g. a gradient penalty) must fail loudly instead of returning a wrong number."""
    x, ei, ea = load_graph("40")
    be = TorchEmulatorBackend()
    monkeypatch.setattr(training, "default_backend", lambda: be)
    R, _ = product_models(4, 1.0)
    R.train()
    y_np, m_np = _targets(x, ei)
    loss = training.regressor_loss(tt(y_np), R(tt(x), tt(ei), tt(ea)), tt(m_np))
    w = R.gclstm_decoder.cell_list[0].conv_i.convs["joint__connect__joint"].lin_l2.weight
    g, = torch.autograd.grad(loss, w, create_graph=True)
    with pytest.raises(RuntimeError, match="once_differentiable|differentiate twice"):
        g.pow(2).sum().backward()


@pytest.mark.gpu
@pytest.mark.parametrize("K,M,ins_off,tail", [(20000, 2112, 12, 4), (10000, 1248, 12, 4), (20000, 4, 0, 4), (333, 672, 8, 8),
                                               (1000, 64, 0, 4)])
def test_wgrad_with_an_inserted_operand_equals_the_concatenated_one(K, M, ins_off, tail):
    """ggnn_wgrad_args.b_ins (ABI 25): B = the data rows [x | 0 | 1 0 0 0] with the 96 hidden-state columns inserted at
    column ins_off, read from the two matrices where they lie -- bit for bit the product with the concatenated copy, on both
    kernels (the exact fp32 MFMA for the short results, the shared-B bf16x3 kernel for M >= 512).  Bad insertions are refused."""
    from graingraphnn_amd import _lib
    from graingraphnn_amd.backend import default_backend
    be = default_backend()
    g = torch.Generator().manual_seed(K + M)
    a = torch.randn(K, M, generator=g).cuda()
    outer = torch.randn(K, ins_off + tail, generator=g).cuda()
    h = torch.randn(K, 96, generator=g).cuda()
    Nc = ins_off + tail + 96
    cat = torch.cat([outer[:, :ins_off], h, outer[:, ins_off:]], 1).contiguous()
    want = be.wgrad(a, cat, K, M, Nc, M, Nc)
    got = be.wgrad(a, outer, K, M, Nc, M, outer.size(1), b_ins=h, ins_off=ins_off)
    assert got.shape == want.shape == (1, M, Nc) and torch.equal(got, want)
    ref = a.double().t() @ cat.double()
    assert float((got[0].double() - ref).abs().max()) <= 3e-6 * float((a.double().abs().t() @ cat.double().abs()).max())
    with pytest.raises(_lib.GGNNError):
        be.wgrad(a, outer, K, M, Nc, M, outer.size(1), b_ins=h, ins_off=ins_off + 2)        # ins_off % 4 != 0
    with pytest.raises(_lib.GGNNError):
        be.wgrad(a, outer, K, M, Nc + 4, M, outer.size(1), b_ins=h, ins_off=ins_off)        # b narrower than Nc - ins_w
    with pytest.raises(_lib.GGNNError):
        be.wgrad(a, outer, K, M, Nc, M, outer.size(1), b_ins=h[:, :94], ins_off=ins_off)    # not contiguous / width % 4


@pytest.mark.gpu
@pytest.mark.parametrize("G", [4, 3])
def test_batched_lstm_train_updates_equal_the_single_calls_and_zero_the_padding(G):
    """ggnn_lstm_train_forward_batch / _backward_batch (ABI 25): two node types of different sizes in one launch each way,
    bit for bit the two single calls; the backward also zeroes the padding columns it is given and nothing else."""
    from graingraphnn_amd import _lib
    from graingraphnn_amd.backend import default_backend
    be = default_backend()
    g = torch.Generator().manual_seed(10 + G)
    shapes = [(501, 1248, 384 + 96, 1184, 64), (1203, 672, 96, 640, 32)]   # N, ldp, s_off, pad_off, pad_n
    fwd, bwd, single = [], [], []
    for N, ldp, s_off, pad_off, pad_n in shapes:
        gemm = torch.randn(G, N, 96, generator=g).cuda()
        P = torch.randn(N, ldp, generator=g).cuda()
        c_in = torch.randn(N, 96, generator=g).cuda() if G == 4 else None
        g_h, g_c = torch.randn(N, 96, generator=g).cuda(), torch.randn(N, 96, generator=g).cuda()
        z1, h1, c1 = gemm.clone(), torch.empty(N, 96, device="cuda"), torch.empty(N, 96, device="cuda")
        be.lstm_train_forward(z1, P, s_off, c_in, h1, c1)
        gz1, gP1 = torch.empty_like(z1), torch.full((N, ldp), 7.0, device="cuda")
        gc1 = torch.empty(N, 96, device="cuda") if G == 4 else None
        be.lstm_train_backward(z1, c_in, c1, g_h, g_c, gz1, gP1, s_off, gc1)
        single.append((z1, h1, c1, gz1, gP1, gc1))
        z2, h2, c2 = gemm.clone(), torch.empty(N, 96, device="cuda"), torch.empty(N, 96, device="cuda")
        fwd.append((z2, P, s_off, c_in, h2, c2))
        gz2, gP2 = torch.empty_like(z2), torch.full((N, ldp), 7.0, device="cuda")
        gc2 = torch.empty(N, 96, device="cuda") if G == 4 else None
        bwd.append((z2, c_in, c2, g_h, g_c, gz2, gP2, s_off, gc2, pad_off, pad_n))
    be.lstm_train_forward_batch(fwd, G)
    be.lstm_train_backward_batch(bwd, G)
    for (z1, h1, c1, gz1, gP1, gc1), f, b, (N, ldp, s_off, pad_off, pad_n) in zip(single, fwd, bwd, shapes):
        assert torch.equal(f[0], z1) and torch.equal(f[4], h1) and torch.equal(f[5], c1)
        assert torch.equal(b[5], gz1) and (gc1 is None or torch.equal(b[8], gc1))
        gP2 = b[6]
        assert bool((gP2[:, pad_off:pad_off + pad_n] == 0).all())
        gP1[:, pad_off:pad_off + pad_n] = 0.0
        assert torch.equal(gP2, gP1)                                   # (everything else as the single call left it)
    with pytest.raises(_lib.GGNNError):
        be.lstm_train_backward_batch([bwd[0][:9] + (bwd[0][9] + 2, 32)], G)     # pad_off % 4 != 0
    with pytest.raises(_lib.GGNNError):
        be.lstm_train_backward_batch([bwd[0][:9] + (shapes[0][1] - 32, 64)], G)  # padding beyond the row


@pytest.mark.gpu
def test_train_input_rows_and_sweep_padding_and_accumulated_hidden_gradient():
    """The three small ABI 25 additions of the training step against their torch expressions: ggnn_train_input_rows
    ([x | 0 | 1 0 0 0]); ggnn_aggregate_args.pad_n (the sweep zeroes the padding behind its scalars in every gate row and
    leaves the result otherwise bit-identical); ggnn_aggregate_bwd_args.g_h_accumulate (the second sweep out of a node type
    adds its hidden-state gradient to the first one's in place = the sum of the two separate results)."""
    from graingraphnn_amd.backend import default_backend
    from graingraphnn_amd.engine import alloc_einfo, graph_for
    from graingraphnn_amd.training import train_topology
    be = default_backend()
    g = torch.Generator().manual_seed(77)
    x9, x11 = torch.randn(1000, 9, generator=g).cuda(), torch.randn(517, 14, generator=g).cuda()[:, :11]
    o9, o11 = be.train_input_rows([(x9, 9), (x11, 11)])
    for x, F, o in ((x9, 9, o9), (x11, 11, o11)):
        want = torch.zeros(x.size(0), 16, device="cuda")
        want[:, :F], want[:, 12] = x[:, :F], 1.0
        assert torch.equal(o, want)
    x, ei, ea = load_graph("40")
    X, EI, EA = ({k: v.cuda() for k, v in tt(t).items()} for t in (x, ei, ea))
    graph = graph_for(be, EI, {nt: X[nt].size(0) for nt in X})
    topo = train_topology(be, graph)
    einfo = alloc_einfo(graph, "cuda", zero=False)
    be.edge_prepare([(graph.csr[et], EA[et].reshape(-1), X[et[0]], X[et[-1]], einfo[et]) for et in graph.csr])
    G, C = 4, 96
    nj, ng = X["joint"].size(0), X["grain"].size(0)
    ldp, Kg = 1248, 224
    P = {"joint": torch.randn(nj, ldp, generator=g).cuda() * 0.3, "grain": torch.randn(ng, ldp, generator=g).cuda() * 0.3}
    H = {"joint": torch.randn(nj, C, generator=g).cuda() * 0.3, "grain": torch.randn(ng, C, generator=g).cuda() * 0.3}
    ep = torch.randn(G, 3, C, generator=g).cuda() * 0.1
    et_jj = [et for et in EDGE_TYPES if et[0] == "joint" and et[-1] == "joint"][0]
    et_jg = [et for et in EDGE_TYPES if et[0] == "joint" and et[-1] == "grain"][0]
    # forward: pad_n
    sweep = lambda agg, pad: (graph.csr[et_jj], einfo[et_jj], P["joint"], P["joint"], H["joint"], ep, agg, 0, 384, 768, 96, Kg,
                              2 * C + 2, G, pad)
    a0, a1 = torch.full((nj, G * Kg), 5.0, device="cuda"), torch.full((nj, G * Kg), 5.0, device="cuda")
    be.aggregate_batch([sweep(a0, 0)])
    be.aggregate_batch([sweep(a1, Kg - (2 * C + 4))])
    v0, v1 = a0.view(nj, G, Kg), a1.view(nj, G, Kg)
    assert bool((v0[:, :, 2 * C + 4:] == 5.0).all()) and bool((v1[:, :, 2 * C + 4:] == 0.0).all())
    assert torch.equal(v0[:, :, :2 * C + 4], v1[:, :, :2 * C + 4])
    # backward: the joints are the source of two edge types
    g_agg = {"joint": torch.randn(nj, G * Kg, generator=g).cuda(), "grain": torch.randn(ng, G * Kg, generator=g).cuda()}
    agg = {"joint": a1, "grain": torch.zeros(ng, G * Kg, device="cuda")}
    be.aggregate_batch([(graph.csr[et_jg], einfo[et_jg], P["joint"], P["grain"], H["joint"], ep, agg["grain"], 0, 384, 768, 0, Kg,
                         C, G, 0)])

    def back(et, a_off, sc_off, into):
        d = et[-1]
        return be.aggregate_backward(graph.csr[et], topo.rcsr[et], topo.r_slot[et], einfo[et], P["joint"], P[d], H["joint"], ep,
                                     agg[d], g_agg[d], 0, 384, 768, a_off, Kg, sc_off, G, g_h_into=into)[2]
    gh_jj, gh_jg = back(et_jj, 96, 2 * C + 2, None), back(et_jg, 0, C, None)
    both = back(et_jg, 0, C, back(et_jj, 96, 2 * C + 2, None))
    assert torch.equal(both, gh_jj + gh_jg)


@pytest.mark.gpu
def test_batched_row_sums_equal_the_single_calls():
    """ggnn_sum_rows_batch (ABI 25): problems of different shapes in one launch.  Many rows of a small result: bit for bit
    ggnn_sum_rows (same summation tree); few rows or a large result: the rows added in index order (what ggnn_wgrad does for
    such a reduction itself) -- so a postponed ggnn_wgrad reduction gives the bits of an immediate one."""
    from graingraphnn_amd.backend import default_backend
    be = default_backend()
    g = torch.Generator().manual_seed(9)
    shapes = [(3, 768, 1152), (1, 45, 2112 * 112), (1, 313, 400), (1, 2, 4), (4, 19, 96 * 224), (1, 31, 36)]
    ins = [torch.randn(*s, generator=g).cuda() for s in shapes]
    outs = [torch.full((s[0], s[2]), 7.0, device="cuda") for s in shapes]
    be.sum_rows_batch(list(zip(ins, outs)))
    for t, o in zip(ins, outs):
        if t.size(1) > 48 and t.size(2) // 4 <= 16384:
            assert torch.equal(o, be.sum_rows(t))
        else:
            acc = torch.zeros_like(o)
            for r in range(t.size(1)):
                acc = acc + t[:, r]
            assert torch.equal(o, acc)
    for K, M, Nc, batch in ((20000, 96, 224, 1), (20000, 2112, 112, 1), (20000, 4, 100, 1), (10000, 96, 128, 4)):
        a, b = torch.randn(batch, K, M, generator=g).cuda(), torch.randn(K, batch * Nc, generator=g).cuda()
        red = []
        c = be.wgrad(a, b, K, M, Nc, M, batch * Nc, batch=batch, a_bstride=K * M, b_bstride=Nc, defer=red)
        assert len(red) == 1
        be.sum_rows_batch(red)
        assert torch.equal(c, be.wgrad(a, b, K, M, Nc, M, batch * Nc, batch=batch, a_bstride=K * M, b_bstride=Nc))
