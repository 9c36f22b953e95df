"""CPU tests of the product's host logic: state_dict layout, weight packing, column offsets
and launch plan, checked end-to-end against the oracle through tests/emulator.py (a torch
restatement of each C-ABI call -- test infrastructure, never a product fallback)."""
import json
import os

import numpy as np
import pytest
import torch

from emulator import TorchEmulatorBackend
from helpers import (EDGE_TYPES, GOLDEN, assert_close, etk, fold_120, golden, load_graph,
                     oracle_models, product_models, random_state, tt)
from graingraphnn_amd import _lib, engine, packing, synthetic
from graingraphnn_amd.seeding import seeded_state_dict
from oracle import grainnn_oracle as oracle

TOL = 2e-5


def test_state_dict_layout_matches_reference():
    R, Cm = product_models(3)
    keys = json.load(open(os.path.join(GOLDEN, "keys.json")))
    for name, m in (("regressor", R), ("classifier", Cm)):
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == keys[name]
    assert sum(p.numel() for p in R.parameters()) == 1204612
    assert sum(p.numel() for p in Cm.parameters()) == 1204806
    # a reference-layout checkpoint loads strictly, and round-trips through torch.save
    oR, _ = oracle_models(3)
    R.load_state_dict(oR.state_dict(), strict=True)


def test_top_level_models_module_is_the_drop_in(tmp_path):
    """test.py:16 / train.py:16 / dist_train.py:15 do `from models import GrainNN_regressor,
    GrainNN_classifier`: with the repository root on sys.path (in a fresh interpreter, as the
    reference's scripts run) that resolves to the HIP classes; test.py:177-188's construction,
    .pt loading and threshold assignments work on them unchanged."""
    import subprocess
    import sys
    from helpers import ROOT
    oR, oC = oracle_models(3)
    torch.save(oR.state_dict(), tmp_path / "regressor0.pt")
    torch.save(oC.state_dict(), tmp_path / "classifier1.pt")
    code = f"""
import json, sys, torch
sys.path.insert(0, {ROOT!r})
from models import GrainNN_regressor, GrainNN_classifier          # test.py:16
import models, graingraphnn_amd.models as hip
assert models.GrainNN_regressor is hip.GrainNN_regressor and models.GrainNN_classifier is hip.GrainNN_classifier
from graingraphnn_amd import synthetic
hp = synthetic.default_hyper('cpu')
Rmodel = GrainNN_regressor(hp)                                     # test.py:177-184
Rmodel.load_state_dict(torch.load({str(tmp_path / 'regressor0.pt')!r}, map_location=torch.device('cpu')))
Rmodel.eval()
Cmodel = GrainNN_classifier(hp, Rmodel)
Cmodel.load_state_dict(torch.load({str(tmp_path / 'classifier1.pt')!r}, map_location='cpu'))
Cmodel.eval()
Rmodel.threshold = 1e-4                                            # test.py:187-188
Cmodel.threshold = 0.6
keys = json.load(open({os.path.join(GOLDEN, 'keys.json')!r}))
assert {{k: list(v.shape) for k, v in Rmodel.state_dict().items()}} == keys['regressor']
assert {{k: list(v.shape) for k, v in Cmodel.state_dict().items()}} == keys['classifier']
assert len(keys['regressor']) == len(keys['classifier']) == 284
print('ok')
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_sharding_refuses_empty_shards_on_every_rank():
    """A rank that raised alone would leave the others blocked in the all-gather."""
    from graingraphnn_amd.dist import rollout_trajectories, run_sharded
    for rank in range(4):
        with pytest.raises(ValueError, match="more ranks"):
            run_sharded(3, lambda t: torch.zeros(2), rank, 4)
    with pytest.raises(ValueError, match="no trajectories"):
        run_sharded(0, lambda t: torch.zeros(2), 0, 1)
    with pytest.raises(ValueError, match="more ranks"):
        rollout_trajectories(None, None, [None], 6, 1, rank=0, world=2)


def test_classifier_deepcopies_regressor_cells():
    """models.py:551-552."""
    from graingraphnn_amd.models import GrainNN_classifier, GrainNN_regressor
    hp = synthetic.default_hyper("cpu")
    R = GrainNN_regressor(hp)
    Cm = GrainNN_classifier(hp, R)
    a = R.gclstm_encoder.cell_list[0].conv_i.convs["grain__push__joint"].lin_key.weight
    b = Cm.gclstm_encoder.cell_list[0].conv_i.convs["grain__push__joint"].lin_key.weight
    assert torch.equal(a, b) and a.data_ptr() != b.data_ptr()


def test_unsupported_configs_fail_loudly():
    from graingraphnn_amd.models import GrainNN_classifier, GrainNN_regressor
    hp = synthetic.default_hyper("cpu")
    with pytest.raises(NotImplementedError):
        GrainNN_regressor(hp, history=True)
    with pytest.raises(NotImplementedError):
        GrainNN_classifier(hp, None, history=True)
    hp.layer_size = 128     # (narrower than 96 is supported: packed zero-padded; wider is not)
    with pytest.raises(NotImplementedError):
        GrainNN_regressor(hp)
    hp.layer_size = 64
    GrainNN_regressor(hp)


def test_no_cpu_fallback():
    """The product refuses CPU tensors instead of silently computing on the host."""
    x, ei, ea = load_graph("40")
    R, Cm = product_models(1)
    with pytest.raises(_lib.GGNNError):
        R(tt(x), tt(ei), tt(ea))
    with pytest.raises(_lib.GGNNError):
        Cm(tt(x), tt(ei), tt(ea))


def test_layout_offsets():
    lj = packing.node_layout("joint", 8, 4)
    lg = packing.node_layout("grain", 11, 4)
    # joint: V for j->g, j->j (768) | u_h for g->j, j->j (768) | S (384) | u4 (2 x 64), padded to 96s
    assert (lj.ncols, lg.ncols) == (2112, 1248)
    assert (lj.Ka, lg.Ka) == (196, 100) and (lj.Kg, lg.Kg) == (224, 128)
    assert lj.v_off == {EDGE_TYPES[1]: 0, EDGE_TYPES[2]: 384}
    assert lj.u_off == {EDGE_TYPES[0]: 768, EDGE_TYPES[2]: 1152} and lj.s_off == 1536
    assert lj.u4_off == {EDGE_TYPES[0]: 1920, EDGE_TYPES[2]: 1984}
    # encoder (h = 0): no hidden-state part of u
    le = packing.node_layout("joint", 8, 3, sees_h=False)
    assert le.u_off == {} and le.s_off == 576 and le.u4_off == {EDGE_TYPES[0]: 864, EDGE_TYPES[2]: 912}
    assert (le.ncols, packing.node_layout("grain", 11, 3, sees_h=False).ncols) == (960, 672)
    # a dead destination type keeps only its value columns (classifier decoder, grain)
    dead = packing.node_layout("grain", 11, 4, live=False)
    assert dead.ncols == 384 and dead.dst_ets == [] and dead.v_off == {EDGE_TYPES[0]: 0}


def _run_model_emulated(model, x, ei, ea, fused_decoder=True):
    be = TorchEmulatorBackend()
    be.fused_decoder = fused_decoder
    n_nodes = {nt: v.shape[0] for nt, v in x.items()}
    graph = engine.GraphCSR(be, ei, n_nodes)
    enc = model.gclstm_encoder.cell_list[0].packed(True)
    dec = model.gclstm_decoder.cell_list[0].packed(False, model._live_out)
    ws = engine.Workspace(enc, dec, n_nodes, "cpu")
    h, c = engine.run_encoder_decoder(be, enc, dec, graph, ws, x, ea)
    return be, graph, h


@torch.no_grad()
def test_weights_beyond_fp16_range_fall_back_to_the_unfused_plans():
    """include/ggnn.h, OPERAND RANGE: the fused cells compute with two fp16 pieces per operand, so a weight at or beyond
    65504 (or a non-finite one) cannot be packed for them.  pack_cell then leaves their streams out (no exception) and
    the cell runs on sweep + gate GEMM launches, whose bf16 x 3 split covers fp32's range: same outputs as the oracle."""
    x, ei, ea = load_graph("40")
    R, _ = product_models(4, 1.0)
    Ro, _ = oracle_models(4, 1.0)
    for model in (R, Ro):   # one huge (finite) weight in an encoder value layer and in a decoder skip layer
        enc = model.gclstm_encoder.cell_list[0].conv_i.convs["grain__push__joint"]
        dec = model.gclstm_decoder.cell_list[0].conv_o.convs["joint__connect__joint"]
        enc.lin_value.weight[5, 4] = 1.0e5
        dec.lin_skip.weight[7, 3] = -2.0e5
    X, EI, EA = tt(x), tt(ei), tt(ea)
    enc_pc = R.gclstm_encoder.cell_list[0].packed(True)
    dec_pc = R.gclstm_decoder.cell_list[0].packed(False, R._live_out)
    assert not enc_pc.ecs and not dec_pc.dcs                  # no fused operands: the weights cannot be split
    be, graph, h = _run_model_emulated(R, X, EI, EA)
    assert "encoder_cell_batch" not in getattr(be, "calls", []) and "decoder_cell_batch" not in getattr(be, "calls", [])
    w, b = packing.pack_regressor_heads(R.linear)
    yj, yg, area = torch.empty(X["joint"].size(0), 2), torch.empty(X["grain"].size(0), 2), torch.empty(X["grain"].size(0))
    be.heads_regressor(h["joint"], h["grain"], X["grain"], w, b, yj, yg, area)
    want = Ro(X, EI, EA)
    assert_close(yj, want["joint"], "range fallback joint", TOL)
    assert_close(yg, want["grain"], "range fallback grain", TOL)


def _narrow_models(layer_size, seed, device="cpu"):
    """(product regressor, classifier), (oracle regressor, classifier) of parameters.py:19's narrower layer sizes."""
    from oracle import grainnn_oracle as oracle
    from graingraphnn_amd import synthetic
    from graingraphnn_amd.models import GrainNN_classifier, GrainNN_regressor
    from graingraphnn_amd.seeding import load_seeded
    out = []
    for R_cls, C_cls, dev in ((GrainNN_regressor, GrainNN_classifier, device),
                              (oracle.GrainNN_regressor, oracle.GrainNN_classifier, "cpu")):
        hp = synthetic.default_hyper(dev)
        hp.layer_size = layer_size
        R = R_cls(hp)
        Cm = C_cls(hp, R)
        load_seeded(R, seed, 2.0).eval()
        load_seeded(Cm, seed + 1, 2.0).eval()
        out.append((R.to(dev), Cm.to(dev)))
    return out


@torch.no_grad()
@pytest.mark.parametrize("layer_size", [64, 32])
def test_narrow_layer_sizes_run_zero_padded_on_the_96_wide_path(layer_size):
    """parameters.py:19: the regressor's grid also holds layer_size 64 and 32.  The kernels are 96 wide; a narrower model
    is packed zero-padded (packing.padded_cell: padded rows / hidden columns, query side scaled by sqrt(96 / c)) and its
    padded channels stay exactly zero.  Through the emulator of the C ABI: the 284-tensor state dict has the reference's
    shapes, both models' outputs equal the oracle's of the same width, the hidden state's padding is exactly zero."""
    (R, Cm), (Ro, Co) = _narrow_models(layer_size, 21)
    x, ei, ea = load_graph("40")
    X, EI, EA = tt(x), tt(ei), tt(ea)
    sd, sdo = R.state_dict(), Ro.state_dict()
    assert list(sd) == list(sdo) and all(sd[k].shape == sdo[k].shape for k in sd) and len(sd) == 284
    c = layer_size
    assert sd["gclstm_decoder.cell_list.0.conv_i.convs.joint__connect__joint.lin_key.weight"].shape == (c, 8 + c)
    be, graph, h = _run_model_emulated(R, X, EI, EA)
    for nt in h:
        assert h[nt].shape[1] == 96 and float(h[nt][:, c:].abs().max()) == 0.0 and float(h[nt][:, :c].abs().max()) > 0
    w, b = packing.pack_regressor_heads(R.linear)
    yj, yg, area = torch.empty(X["joint"].size(0), 2), torch.empty(X["grain"].size(0), 2), torch.empty(X["grain"].size(0))
    be.heads_regressor(h["joint"], h["grain"], X["grain"], w, b, yj, yg, area)
    want = Ro(X, EI, EA)
    assert_close(yj, want["joint"], f"layer_size {c} joint", TOL)
    assert_close(yg, want["grain"], f"layer_size {c} grain", TOL)
    assert_close(area, want["grain_area"], f"layer_size {c} area", TOL)
    _, graph, hc = _run_model_emulated(Cm, X, EI, EA)
    w_node, w_edge = packing.pack_classifier_heads(Cm.lin1, Cm.lin2)
    E = EI[EDGE_TYPES[2]].size(1)
    ev, ed, tmp = torch.empty(E), torch.empty(E, 2), torch.empty(X["joint"].size(0), 6)
    be.heads_classifier(hc["joint"], graph.edge_index[EDGE_TYPES[2]], EA[EDGE_TYPES[2]].reshape(-1), w_node, w_edge, tmp, ev, ed)
    wantc = Co(X, EI, EA)
    assert_close(ev, wantc["edge_event"], f"layer_size {c} edge_event", TOL)
    assert_close(ed, wantc["edge"], f"layer_size {c} edge", TOL)


def test_fp16_two_piece_split_is_fp32_equivalent():
    """The arithmetic of the fused decoder cell and of the encoder cell's gate GEMM (csrc/common.h: split_f16x2,
    mfma_x3h; packing.split2_f16): x = hi + lo' / 2^11 with two fp16 pieces, three of the four products, the cross
    terms in their own accumulator.  Emulated here with exact products (float64) rounded to fp32 per accumulator:
    its error against the fp64 product, normalised by sum |x||w|, stays below a plain fp32 fma chain's."""
    rs = np.random.RandomState(3)
    for K, wscale, xscale in ((104, 0.3, 1.0), (196, 0.3, 3.0), (104, 1e-2, 1e-3)):
        X = torch.from_numpy((np.tanh(rs.randn(128, K)) * xscale).astype(np.float32))
        W = torch.from_numpy((rs.randn(96, K) * wscale).astype(np.float32))
        xh, xl = packing.split2_f16(X)
        wh, wl = packing.split2_f16(W)
        assert float((xh.float() + xl.float() / packing.DC_LO_SCALE - X).abs().max()) <= 2.0 ** -21 * float(X.abs().max())
        d = lambda a, b: a.double() @ b.double().t()
        main = d(xh, wh).float()
        cross = (d(xh, wl) + d(xl, wh)).float()
        got = (main + cross / packing.DC_LO_SCALE).double()
        ref, norm = d(X, W), d(X.abs(), W.abs())
        err = float(((got - ref).abs() / norm).max())
        chain = torch.zeros(128, 96)
        for k in range(K):                       # what fp32 hardware arithmetic gives: one rounding per term
            chain = chain + X[:, k:k + 1] * W[:, k][None, :]
        err_chain = float(((chain.double() - ref).abs() / norm).max())
        assert err <= 1.2e-7 and err < err_chain, (K, err, err_chain)
    with pytest.raises(ValueError):
        packing.split2_f16(torch.tensor([1.0e5]))         # beyond fp16's range: refused at packing time
    with pytest.raises(ValueError):
        packing.split2_f16(torch.tensor([float("nan")]))


@torch.no_grad()
def test_decoder_plan_per_model():
    """GGNN_DEC=fused-classifier / fused-regressor (backend.fused_decoder = "classifier" / "regressor"): the fused
    decoder cell for that model only -- told apart by the number of live destination types of its decoder."""
    x, ei, ea = load_graph("40")
    R, Cm = product_models(10020, 1.0)
    X, EI, EA = tt(x), tt(ei), tt(ea)
    for which, fused_for in (("classifier", {"C"}), ("regressor", {"R"}), (True, {"R", "C"}), (False, set())):
        for name, model in (("R", R), ("C", Cm)):
            be, _, _ = _run_model_emulated(model, X, EI, EA, which)
            used_fused = "decoder_cell_batch" in getattr(be, "calls", [])
            assert used_fused == (name in fused_for), (which, name)


@pytest.mark.parametrize("fused_decoder", [False, True])
@pytest.mark.parametrize("tag,seed,scale", [("cfg1_s1", 10020, 1.0), ("cfg1_s3", 10020, 3.0),
                                            ("cfg2_s1", 0, 1.0)])
@torch.no_grad()
def test_packing_and_plan_reproduce_golden_forward(tag, seed, scale, fused_decoder):
    """Both decoder plans through the emulator: the fused decoder cell (the default), whose weight stream the
    emulator decodes back from its two fp16 planes, and projection + sweeps + gate GEMM (GGNN_DEC=split)."""
    if tag.startswith("cfg1"):
        x, ei, ea = load_graph("40")
    else:
        x, ei, ea = load_graph("120")
        x, ea = fold_120(x, ea)
    g = golden(tag)
    R, Cm = product_models(seed, scale)
    X, EI, EA = tt(x), tt(ei), tt(ea)
    be, graph, h = _run_model_emulated(R, X, EI, EA, fused_decoder)
    w, b = packing.pack_regressor_heads(R.linear)
    yj, yg, area = torch.empty(X["joint"].size(0), 2), torch.empty(X["grain"].size(0), 2), torch.empty(X["grain"].size(0))
    be.heads_regressor(h["joint"], h["grain"], X["grain"], w, b, yj, yg, area)
    assert_close(yj, g["R_joint"], f"{tag} R joint", TOL)
    assert_close(yg, g["R_grain"], f"{tag} R grain", TOL)
    assert_close(area, g["R_grain_area"], f"{tag} R area", TOL)
    be, graph, h = _run_model_emulated(Cm, X, EI, EA, fused_decoder)
    wn, we = packing.pack_classifier_heads(Cm.lin1, Cm.lin2)
    E = EI[EDGE_TYPES[2]].size(1)
    ev, ed, tmp = torch.empty(E), torch.empty(E, 2), torch.empty(X["joint"].size(0), 8)
    be.heads_classifier(h["joint"], EI[EDGE_TYPES[2]], EA[EDGE_TYPES[2]].view(-1), wn, we, tmp, ev, ed)
    assert_close(ev, g["C_edge_event"], f"{tag} C edge_event", TOL)
    assert_close(ed, g["C_edge"], f"{tag} C edge", TOL)


@torch.no_grad()
def test_cell_with_state_and_single_conv_packing():
    x, ei, ea = load_graph("40")
    g = golden("cfg1_s1")
    R, _ = product_models(10020)
    be = TorchEmulatorBackend()
    X, EI, EA = tt(x), tt(ei), tt(ea)
    n_nodes = {nt: v.shape[0] for nt, v in x.items()}
    h0, c0 = tt(random_state(n_nodes, 7)), tt(random_state(n_nodes, 8))
    graph = engine.GraphCSR(be, EI, n_nodes)
    ea1 = {et: EA[et].view(-1) for et in EDGE_TYPES}
    for encoder, key in ((True, "cell0"), (False, "cell1")):
        cell = (R.gclstm_encoder if encoder else R.gclstm_decoder).cell_list[0]
        pc = cell.packed(encoder)
        proj = {nt: torch.empty(n_nodes[nt], pc.layout[nt].ncols) for nt in n_nodes}
        agg = {nt: torch.zeros(n_nodes[nt], pc.G * pc.layout[nt].Kg) for nt in n_nodes}
        ho = {nt: torch.empty(n_nodes[nt], 96) for nt in n_nodes}
        co = {nt: torch.empty(n_nodes[nt], 96) for nt in n_nodes}
        einfo = engine.prepare_edges(be, graph, X, ea1, None)
        engine.run_cell(be, pc, graph, X, einfo, None if encoder else h0, None if encoder else c0,
                        proj, agg, ho, co)
        for nt in n_nodes:
            assert_close(ho[nt], g[f"{key}_h_{nt}"], f"{key} h {nt}", TOL)
            assert_close(co[nt], g[f"{key}_c_{nt}"], f"{key} c {nt}", TOL)
    # single PeriodConv through pack_conv + raw epilogue
    for et in EDGE_TYPES:
        conv = R.gclstm_decoder.cell_list[0].conv_i.convs[etk(et)]
        Fs, Fd = X[et[0]].size(1), X[et[-1]].size(1)
        wps, bps, wpd, bpd, ep, w2 = packing.pack_conv(conv, Fs, Fd, 96)
        ps, pd = torch.empty(n_nodes[et[0]], 96), torch.empty(n_nodes[et[-1]], 288)
        be.project(X[et[0]], Fs, h0[et[0]], wps, bps, ps)
        be.project(X[et[-1]], Fd, h0[et[-1]], wpd, bpd, pd)
        agg = torch.zeros(n_nodes[et[-1]], 100)
        be.aggregate(graph.csr[et], einfo[et], ps, pd, h0[et[0]], ep, agg, 0, 0, 192, 0, 100, 96, 1)
        out = torch.empty(n_nodes[et[-1]], 96)
        be.lstm_epilogue(agg, w2, pd, 96, None, None, None, out, 1, 2)
        assert_close(out, g["conv_" + etk(et)], f"conv {et}", TOL)


@torch.no_grad()
def test_step_glue_matches_golden():
    x, ei, ea = load_graph("40")
    g = golden("cfg1_s1")
    be = TorchEmulatorBackend()
    X = tt(x)
    yj, yg = torch.from_numpy(g["R_joint"].copy()), torch.from_numpy(g["R_grain"].copy())
    flags = torch.zeros(2, dtype=torch.int32)
    dz, zmax = float(np.float32(6 / 121)), float(np.float32(120 / 121))
    be.step_update(X["joint"], X["grain"], yj, yg, dz, zmax, flags)
    EI = tt(ei)
    EA = {et: torch.empty(EI[et].size(1)) for et in EDGE_TYPES}
    be.step_refresh(X["joint"], X["grain"], zmax, flags,
                    [(EI[et], X[et[0]], X[et[-1]], EA[et]) for et in EDGE_TYPES])
    for nt in x:
        assert_close(X[nt], g[f"step1_x_{nt}"], f"step1 x {nt}", TOL)
    for et in EDGE_TYPES:
        assert_close(EA[et].view(-1, 1), g["step1_ea_" + etk(et)], f"step1 ea {et}", TOL)


@torch.no_grad()
def test_grain_centre_glue_matches_golden_and_oracle():
    """SURVEY 8f-1 host contract: step_update -> grain_centres -> step_refresh reproduces the
    reference-generated golden; the folded-domain (factor 3) variant matches the oracle."""
    x, ei, ea = load_graph("40")
    g, gc = golden("cfg1_s1"), golden("cfg1_centres")
    be = TorchEmulatorBackend()
    X, EI = tt(x), tt(ei)
    yj, yg = torch.from_numpy(g["R_joint"].copy()), torch.from_numpy(g["R_grain"].copy())
    flags = torch.zeros(2, dtype=torch.int32)
    dz, zmax = float(np.float32(6 / 121)), float(np.float32(120 / 121))
    be.step_update(X["joint"], X["grain"], yj, yg, dz, zmax, flags)
    csr = be.build_csr(EI[EDGE_TYPES[1]], 236, 118)
    be.grain_centres(csr, X["joint"], X["grain"])
    EA = {et: torch.empty(EI[et].size(1)) for et in EDGE_TYPES}
    be.step_refresh(X["joint"], X["grain"], zmax, flags,
                    [(EI[et], X[et[0]], X[et[-1]], EA[et]) for et in EDGE_TYPES])
    for nt in x:
        assert_close(X[nt], gc[f"step1_x_{nt}"], f"centres step1 x {nt}", TOL)
    for et in EDGE_TYPES:
        assert_close(EA[et].view(-1, 1), gc["step1_ea_" + etk(et)], f"centres step1 ea {et}", TOL)

    x, ei, ea = load_graph("120")
    X, EA = tt(x), tt(ea)
    off, _ = oracle.scale_feature_patchs(3.0, X, EA)
    X["joint"][:, :2] += 0.01 * torch.randn(X["joint"].size(0), 2, generator=torch.Generator().manual_seed(1))
    EI = tt(ei)
    oX = {k: v.clone() for k, v in X.items()}
    oracle.refresh_grain_centres(oX, EI, 3.0, off)
    csr = be.build_csr(EI[EDGE_TYPES[1]], X["joint"].size(0), X["grain"].size(0))
    be.grain_centres(csr, X["joint"], X["grain"], 3.0, off)
    d = (X["grain"][:, :2] - oX["grain"][:, :2]).abs()
    assert float(torch.minimum(d, 1 - d).max()) < 1e-5     # frac() may land on either side of 0/1


def test_patch_folding_matches_oracle_and_honeycomb_offset():
    """synthetic.scale_feature_patchs (host data prep, numpy) == the oracle's restatement of
    test.py:29-55; honeycomb's domain_offset undoes its folding."""
    x, ei, ea = load_graph("120")
    oX, oEA = tt(x), tt(ea)
    ooff, _ = oracle.scale_feature_patchs(3.0, oX, oEA)
    x2, ea2 = {k: v.copy() for k, v in x.items()}, {k: v.copy() for k, v in ea.items()}
    off = synthetic.scale_feature_patchs(3.0, x2, ea2)
    assert np.array_equal(off, ooff.numpy())
    for nt in x:
        assert np.array_equal(x2[nt], oX[nt].numpy()), nt
    for et in EDGE_TYPES:
        assert np.array_equal(ea2[et], oEA[et].numpy()), et
    xh, eih, _, offh = synthetic.honeycomb(8, 2, 0, return_offset=True)
    assert set(np.unique(offh)) <= {0.0, 1.0}
    xg = (xh["joint"][:, :2] + offh) / 2
    c = oracle.grain_centres(torch.from_numpy(xg), torch.from_numpy(eih[EDGE_TYPES[0]]), 64).numpy()
    d = np.abs((c * 2) % 1 - xh["grain"][:, :2])
    assert np.minimum(d, 1 - d).max() < 1e-5    # honeycomb grain centres are the polygon means


def test_seeding_is_deterministic_and_order_free():
    shapes = {"b": (4, 9), "a": (5,), "c.weight": (96, 1)}
    s1 = seeded_state_dict(shapes, 5)
    s2 = seeded_state_dict(dict(reversed(list(shapes.items()))), 5)
    assert all(torch.equal(s1[k], s2[k]) for k in shapes)
    assert float(s1["c.weight"].abs().max()) <= 1.0 and float(s1["b"].abs().max()) <= 1 / 3


def test_honeycomb_invariants():
    x, ei, ea = synthetic.honeycomb(20, 2, 0)
    ng, nj = 400, 800
    assert x["grain"].shape == (ng, 11) and x["joint"].shape == (nj, 8)
    for et in EDGE_TYPES:
        assert ei[et].shape == (2, 3 * nj) and ei[et].dtype == np.int64
    assert (np.bincount(ei[EDGE_TYPES[0]][1], minlength=nj) == 3).all()   # 3 grains per junction
    assert (np.bincount(ei[EDGE_TYPES[2]][1], minlength=nj) == 3).all()   # 3 junctions per junction
    assert (np.bincount(ei[EDGE_TYPES[1]][1], minlength=ng) == 6).all()   # hexagons
    jj = set(map(tuple, ei[EDGE_TYPES[2]].T))
    assert all((b, a) in jj for a, b in jj) and all(a != b for a, b in jj)
    gj = set(map(tuple, ei[EDGE_TYPES[0]].T))
    assert all((b, a) in gj for a, b in map(tuple, ei[EDGE_TYPES[1]].T))
    # edge lengths equal the min-image distances of the folded coordinates (test.py:562-575)
    X, EI = tt(x), tt(ei)
    EA = oracle.refresh_edge_attr(X, EI)
    for et in EDGE_TYPES:
        assert_close(torch.from_numpy(ea[et]), EA[et], f"honeycomb ea {et}", 1e-5)
        assert float(ea[et].max()) < 0.25
    x3, ei3, _ = synthetic.honeycomb(100, 10, 0)
    assert x3["grain"].shape[0] == 10000 and x3["joint"].shape[0] == 20000
    assert all(v.shape == (2, 60000) for v in ei3.values())


def test_voronoi_generator_invariants():
    """the scipy-Voronoi initial-structure generator gives a valid grain graph
    (graph_trajectory.py:985-988): 3 + 3 neighbours per junction, N_j = 2 N_g, mutual edge lists,
    unit total area; its grain centres are the junction-polygon means the centre refresh computes."""
    for kw in (dict(n_grains=300, seed=2), dict(n_grains=256, seed=5, lattice_noise=0.2)):
        x, ei, ea = synthetic.voronoi(**kw)
        ng, nj = x["grain"].shape[0], x["joint"].shape[0]
        assert nj == 2 * ng and all(ei[et].shape == (2, 3 * nj) for et in EDGE_TYPES)
        assert (np.bincount(ei[EDGE_TYPES[0]][1], minlength=nj) == 3).all()
        assert (np.bincount(ei[EDGE_TYPES[2]][1], minlength=nj) == 3).all()
        assert set(map(tuple, ei[EDGE_TYPES[2]].T)) == set(map(tuple, ei[EDGE_TYPES[2]][::-1].T))
        assert np.array_equal(ei[EDGE_TYPES[0]], ei[EDGE_TYPES[1]][::-1])
        assert abs(float(x["grain"][:, 3].sum()) - 1.0) < 1e-5
        deg = np.bincount(ei[EDGE_TYPES[1]][1], minlength=ng)
        assert deg.min() >= 3 and deg.sum() == 3 * nj
        c = oracle.grain_centres(torch.from_numpy(x["joint"][:, :2]), torch.from_numpy(ei[EDGE_TYPES[0]]), ng).numpy()
        d = np.abs(c % 1 - x["grain"][:, :2])
        assert np.minimum(d, 1 - d).max() < 1e-5
        assert float(ea[EDGE_TYPES[2]].max()) < 0.5


def _structure_stats(x, ei, ea):
    n_g = x["grain"].shape[0]
    deg = np.bincount(ei[EDGE_TYPES[1]][1], minlength=n_g)
    a = x["grain"][:, 3].astype(np.float64)
    return dict(n_g=n_g, deg_var=float(deg.var()), area_cv=float(a.std() / a.mean()), area_sum=float(a.sum()),
                len_gj=float(ea[EDGE_TYPES[0]].mean()), len_jj=float(ea[EDGE_TYPES[2]].mean()))


def test_span_lookup_matches_the_reference_expression():
    """the reference's generator takes its frame span from a nearest-neighbour lookup in its (G, R)
    training grid (graph_trajectory.py:1308-1316).  `synthetic.span_for` on the shipped table against spans the
    reference's own expression returned (tests/golden/make_gr_span_grid.py): fixtures' parameters, the CLI
    defaults, the grid's corners and 40 random pairs; and `generate(G, R)` writes that span into its features."""
    z = np.load(os.path.join(GOLDEN, "gr_span_pins.npz"))
    assert len(z["GR"]) >= 5
    for (G, R), s in zip(z["GR"], z["span"]):
        assert synthetic.span_for(float(G), float(R)) == int(s), (G, R)
    assert len(set(z["span"].tolist())) > 3          # the pins do exercise different table entries
    x, _, _ = synthetic.generate(lxd=40, seed=3, G=10.0, R=0.2)
    want = np.float32(synthetic.span_for(10.0, 0.2) / 120)
    assert synthetic.span_for(10.0, 0.2) == 60 and np.all(x["grain"][:, 9] == want) and np.all(x["joint"][:, 5] == want)


def _reference_generated(tag):
    g = np.load(os.path.join(GOLDEN, f"generated_{tag}.npz"))
    x = {"grain": g["x_grain"], "joint": g["x_joint"]}
    ei = {et: g["ei_" + etk(et)] for et in EDGE_TYPES}
    ea = {et: g["ea_" + etk(et)] for et in EDGE_TYPES}
    return x, ei, ea, float(g["span"]), float(g["G"]), float(g["R"])


@pytest.mark.parametrize("tag,lxd,seed", [("40_seed1", 40, 1), ("40_seed2", 40, 2), ("80_seed7", 80, 7)])
def test_generate_reproduces_the_reference_sample_bit_for_bit(tag, lxd, seed):
    """`synthetic.generate(lxd, seed, G, R)` IS the reference's `graph_trajectory.py --mode=generate`
    sample of that seed (:1289-1333): the fixtures were pickled by the unmodified reference
    (tests/golden/make_golden_generated.py; the 80 um one with G = 10, R = 0.2, i.e. span 60 from the lookup table).
    Index work is held to bit-exactness -- the three edge lists column for column, hence the junction and grain
    numbering --, and so are the fp32 features and edge lengths (same arithmetic in float64, one cast)."""
    rx, rei, rea, span, G, R = _reference_generated(tag)
    x, ei, ea = synthetic.generate(lxd=lxd, seed=seed, G=G, R=R)      # span: from the (G, R) lookup, as the reference
    assert float(x["grain"][0, 9]) == np.float32(span / 120)
    for et in EDGE_TYPES:
        assert ei[et].dtype == np.int64 and np.array_equal(ei[et], rei[et]), et
        assert ea[et].dtype == np.float32 and np.array_equal(ea[et], rea[et]), et
    for nt in ("grain", "joint"):
        assert x[nt].dtype == np.float32 and np.array_equal(x[nt], rx[nt]), nt


def test_generated_structure_invariants_and_the_polygon_variant():
    """What every generated structure satisfies (reference sample or `lattice_structure`, the round-2 construction
    with exact polygon areas and an own random stream): N_j = 2 N_g, E = 3 N_j, 3 grains + 3 junctions at every
    junction, mean grain degree 6, grain->joint the flip of joint->grain, unit orientation vectors; and the two
    constructions agree statistically (grain count within 3 %, degree variance within a factor 2, area spread within
    30 %, mean edge lengths within 3 %)."""
    refs = [_reference_generated(t)[:3] for t in ("40_seed1", "40_seed2")]
    ref_stats = [_structure_stats(x, ei, ea) for x, ei, ea in refs]
    mean = lambda k: float(np.mean([st[k] for st in ref_stats]))
    ours = []
    for seed in range(6):
        for fn in (synthetic.generate, synthetic.lattice_structure):
            x, ei, ea = fn(lxd=40, seed=seed, G=2.0, R=0.4)
            n_g, n_j = x["grain"].shape[0], x["joint"].shape[0]
            assert x["grain"].shape[1] == 11 and x["joint"].shape[1] == 8 and x["grain"].dtype == np.float32
            assert n_j == 2 * n_g
            for et in EDGE_TYPES:
                assert ei[et].shape == (2, 3 * n_j) and ea[et].shape[0] == 3 * n_j and ei[et].dtype == np.int64
            assert np.array_equal(np.bincount(ei[EDGE_TYPES[0]][1], minlength=n_j), np.full(n_j, 3))   # 3 grains per junction
            assert np.array_equal(np.bincount(ei[EDGE_TYPES[2]][1], minlength=n_j), np.full(n_j, 3))   # 3 junctions per junction
            assert np.bincount(ei[EDGE_TYPES[1]][1], minlength=n_g).mean() == 6.0
            assert np.array_equal(ei[EDGE_TYPES[0]], ei[EDGE_TYPES[1]][::-1])
            assert np.allclose(x["grain"][:, 5] ** 2 + x["grain"][:, 6] ** 2, 1, atol=1e-6)
            assert np.allclose(x["grain"][:, 7] ** 2 + x["grain"][:, 8] ** 2, 1, atol=1e-6)
            if fn is synthetic.lattice_structure:
                assert abs(x["grain"][:, 3].astype(np.float64).sum() - 1.0) < 1e-5   # polygons tile the 40 um domain
                ours.append(_structure_stats(x, ei, ea))
    omean = lambda k: float(np.mean([st[k] for st in ours]))
    assert abs(omean("n_g") - mean("n_g")) <= 0.03 * mean("n_g")
    assert 0.5 * mean("deg_var") <= omean("deg_var") <= 2.0 * mean("deg_var")
    assert abs(omean("area_cv") - mean("area_cv")) <= 0.3 * mean("area_cv")
    assert abs(omean("len_gj") - mean("len_gj")) <= 0.03 * mean("len_gj")
    assert abs(omean("len_jj") - mean("len_jj")) <= 0.03 * mean("len_jj")
    assert all(abs(st["area_sum"] - 1.0) < 0.02 for st in ref_stats)   # the raster areas fill the domain to a pixel
    x, ei, ea = synthetic.lattice_structure(lxd=120, seed=0)            # bigger domains keep the density (9 patches)
    assert abs(x["grain"].shape[0] / 9 - mean("n_g")) <= 0.05 * mean("n_g")
    assert abs(float(x["grain"][:, 3].astype(np.float64).sum()) - 9.0) < 1e-4


def test_disjoint_union_offsets():
    g1 = synthetic.honeycomb(4, 1, 0)
    g2 = synthetic.honeycomb(6, 1, 1)
    x, ei, ea, slices = synthetic.disjoint_union([g1, g2])
    assert x["grain"].shape[0] == 16 + 36 and x["joint"].shape[0] == 32 + 72
    assert slices[1] == {"grain": (16, 52), "joint": (32, 104)}
    for et in EDGE_TYPES:
        n1 = g1[1][et].shape[1]
        assert np.array_equal(ei[et][:, :n1], g1[1][et])
        shift = np.array([[slices[1][et[0]][0]], [slices[1][et[-1]][0]]])
        assert np.array_equal(ei[et][:, n1:], g2[1][et] + shift)
        assert ea[et].shape[0] == ei[et].shape[1]


def test_classifier_loss_ignores_unlabelled_edges_like_the_reference_indexing():
    """train.py:44-47 indexes the unlabelled edges (label -1) away before the loss: a non-finite logit there reaches neither
    the value nor a gradient (the capturable weighted-sum form multiplied inf by 0)."""
    from graingraphnn_amd import training
    g = torch.Generator().manual_seed(4)
    z = torch.randn(40, generator=g)
    y = (torch.rand(40, generator=g) > 0.5).float()
    y[::5] = -1.0
    z[0], z[5], z[10] = float("inf"), float("-inf"), float("nan")
    z.requires_grad_(True)
    loss = training.classifier_loss({"edge_event": y}, {"edge_event": z}, 2.0)
    keep = y > -1
    want = torch.nn.functional.binary_cross_entropy_with_logits(z.detach()[keep], y[keep], pos_weight=torch.tensor(2.0))
    assert torch.isfinite(loss) and abs(float(loss) - float(want)) < 1e-6
    loss.backward()
    assert bool(torch.isfinite(z.grad).all()) and bool((z.grad[~keep] == 0).all())


def test_sampled_parameter_versions_notice_what_changes_a_whole_model():
    """rollout.refresh_weights(sample=True) -- the per-step check of a step_events() loop -- looks at every 24th parameter
    tensor (modules._param_version_sample): an optimizer-style update of all parameters, load_state_dict and a move of the
    storage each change the sample; the full walk (every 16th step) is the one that notices a single edited tensor."""
    from graingraphnn_amd.modules import _param_version, _param_version_sample
    from helpers import product_models
    R, _ = product_models(3, 1.0)
    R2, _ = product_models(4, 1.0)
    full, sample = _param_version(R), _param_version_sample(R)
    assert 8 <= len(sample) <= len(full) // 20 and set(sample) <= set(full)
    with torch.no_grad():
        for p in R.parameters():
            p.add_(0.0)                                   # an optimizer step bumps every version counter
    assert _param_version_sample(R) != sample
    sample = _param_version_sample(R)
    R.load_state_dict(R2.state_dict())
    assert _param_version_sample(R) != sample
    sample, full = _param_version_sample(R), _param_version(R)
    R.double().float()                                    # new storages
    assert _param_version_sample(R) != sample
    sample, full = _param_version_sample(R), _param_version(R)
    name = "gclstm_decoder.cell_list.0.conv_c.convs.joint__connect__joint.lin_value.weight"
    with torch.no_grad():
        dict(R.named_parameters())[name].mul_(1.5)        # one tensor edited in place: the full walk's business
    assert _param_version(R) != full
