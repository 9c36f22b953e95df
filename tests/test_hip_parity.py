"""GPU parity tests: the HIP path, called through the product API (-> C ABI, libggnn.so),
against the CPU oracle and the committed golden vectors.  Tolerance: 1e-4 relative fp32
(max|a-b| <= 1e-4 * max|ref| per tensor), the north_star bar."""
import numpy as np
import pytest
import torch

from helpers import (EDGE_TYPES, assert_close, etk, fold_120, golden, load_graph,
                     oracle_models, product_models, random_state, rel_err, tt)
from graingraphnn_amd import _lib, synthetic
from oracle import grainnn_oracle as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"
GJ, JG, JJ = EDGE_TYPES


def backend():
    from graingraphnn_amd.backend import default_backend
    return default_backend()


def test_library_is_loaded_from_the_tree():
    be = backend()
    assert be.lib.ggnn_version() == _lib.GGNN_ABI_VERSION
    assert _lib.LIB_PATH.endswith("graingraphnn_amd/libggnn.so")


# ---------------------------------------------------------------------------------------
# CSR build: bit-exact against numpy (integer work)
# ---------------------------------------------------------------------------------------
def _csr_numpy(ei, n_dst):
    order = np.argsort(ei[1], kind="stable")
    rowptr = np.zeros(n_dst + 1, np.int64)
    rowptr[1:] = np.cumsum(np.bincount(ei[1], minlength=n_dst))
    return rowptr, ei[0][order], order


@pytest.mark.parametrize("n_src,n_dst,E,seed", [(7, 5, 0, 0), (1, 1, 1, 1), (236, 118, 708, 2),
                                               (5000, 3000, 40000, 3), (100, 2500, 700, 4),
                                               (50, 3, 5000, 5)])
def test_build_csr_bit_exact(n_src, n_dst, E, seed):
    rs = np.random.RandomState(seed)
    ei = np.stack([rs.randint(0, n_src, E), rs.randint(0, n_dst, E)]).astype(np.int64)
    csr = backend().build_csr(torch.from_numpy(ei).to(DEV), n_src, n_dst)
    r_ref, c_ref, p_ref = _csr_numpy(ei, n_dst)
    assert np.array_equal(csr.row.cpu().numpy()[:E], ei[1][p_ref])
    assert np.array_equal(csr.rowptr.cpu().numpy(), r_ref)
    assert np.array_equal(csr.col.cpu().numpy()[:E], c_ref)
    assert np.array_equal(csr.perm.cpu().numpy()[:E], p_ref)
    # unit table: rows cut into <= 3-edge units, empty rows keep one empty unit
    deg = np.diff(r_ref)
    cnt = np.where(deg == 0, 1, (deg + 2) // 3)
    uptr = np.concatenate([[0], np.cumsum(cnt)])
    assert np.array_equal(csr.unit_ptr.cpu().numpy(), uptr)
    units = csr.units.cpu().numpy()[:uptr[-1]]
    for i in np.random.RandomState(seed).randint(0, n_dst, 50):
        for c in range(cnt[i]):
            d = units[uptr[i] + c]
            p0 = r_ref[i] + 3 * c
            nact = int(min(3, max(0, r_ref[i + 1] - p0)))
            assert d[0] == i and d[1] == p0
            assert d[2] == (nact | (int(c == 0) << 8) | (int(c == cnt[i] - 1) << 9))
            want = [c_ref[p0 + t] if t < nact else (c_ref[p0] if nact else 0) for t in range(3)]
            assert list(d[4:7]) == want


def test_build_csr_rejects_out_of_range_indices():
    ei = torch.tensor([[0, 1, 9], [0, 1, 1]], dtype=torch.int64, device=DEV)
    with pytest.raises(IndexError):
        backend().build_csr(ei, 5, 3)
    ei = torch.tensor([[0, 1, 2], [0, -1, 1]], dtype=torch.int64, device=DEV)
    with pytest.raises(IndexError):
        backend().build_csr(ei, 5, 3)


# ---------------------------------------------------------------------------------------
# projection GEMM (fp32 MFMA) against a plain fp32 torch reference
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,F,k2,ncols", [(1, 8, 0, 96), (63, 11, 96, 192), (64, 8, 96, 96),
                                         (65, 11, 0, 288), (1000, 8, 96, 2688), (777, 11, 96, 1536),
                                         (300, 3, 96, 96), (129, 12, 0, 96)])
def test_project_matches_fp32_matmul(M, F, k2, ncols):
    rs = np.random.RandomState(M + F)
    Fp = (F + 3) & ~3
    x = torch.from_numpy(rs.uniform(-1, 1, (M, F)).astype(np.float32))
    h = torch.from_numpy(rs.uniform(-1, 1, (M, 96)).astype(np.float32)) if k2 else None
    wp = torch.from_numpy(rs.uniform(-0.3, 0.3, (ncols, Fp + k2)).astype(np.float32))
    wp[:, F:Fp] = 0
    bp = torch.from_numpy(rs.uniform(-1, 1, ncols).astype(np.float32))
    xin = torch.zeros(M, Fp + k2, dtype=torch.float64)
    xin[:, :F] = x
    if k2:
        xin[:, Fp:] = h
    ref = (xin @ wp.double().t() + bp.double()).float()
    out = torch.full((M, ncols), float("nan"), device=DEV)
    backend().project(x.to(DEV), F, None if h is None else h.to(DEV), wp.to(DEV), bp.to(DEV), out)
    assert_close(out, ref, f"project M={M} F={F} k2={k2} ncols={ncols}", 2e-6)


@pytest.mark.parametrize("M,F,ncols", [(63, 11, 192), (1000, 8, 2112), (777, 11, 1248), (16, 3, 96)])
def test_project_bf16_precision_is_one_rounded_product(M, F, ncols):
    """GGNN_PRECISION_BF16 (the training path under torch.autocast(bfloat16)): both operands rounded to bf16 (round
    to nearest even), ONE MFMA product per k-step, fp32 accumulation -- against the float64 product of the rounded
    operands (every product of two bf16 values is exact, so only the accumulation differs), and refused where it does
    not exist (encoder shape)."""
    rs = np.random.RandomState(M + F)
    Fp = (F + 3) & ~3
    x = torch.from_numpy(rs.uniform(-1, 1, (M, F)).astype(np.float32))
    h = torch.from_numpy(rs.uniform(-1, 1, (M, 96)).astype(np.float32))
    wp = torch.from_numpy(rs.uniform(-0.3, 0.3, (ncols, Fp + 96)).astype(np.float32))
    wp[:, F:Fp] = 0
    bp = torch.from_numpy(rs.uniform(-1, 1, ncols).astype(np.float32))
    r = lambda t: t.to(torch.bfloat16).double()
    xin = torch.zeros(M, Fp + 96, dtype=torch.float64)
    xin[:, :F], xin[:, Fp:] = r(x), r(h)
    ref = (xin @ r(wp).t() + bp.double()).float()
    out = torch.full((M, ncols), float("nan"), device=DEV)
    be = backend()
    if be.lib.ggnn_gemm_mode() == 0:   # GGNN_GEMM=fp32: the native-fp32 kernels have no bf16 mode and say so
        with pytest.raises(_lib.GGNNError):
            be.project_batch([(x.to(DEV), F, h.to(DEV), wp.to(DEV), bp.to(DEV), out, _lib.GGNN_PRECISION_BF16)])
        return
    be.project_batch([(x.to(DEV), F, h.to(DEV), wp.to(DEV), bp.to(DEV), out, _lib.GGNN_PRECISION_BF16)])
    assert_close(out, ref, f"bf16 project M={M} F={F} ncols={ncols}", 2e-6)
    exact = torch.empty_like(out)
    be.project_batch([(x.to(DEV), F, h.to(DEV), wp.to(DEV), bp.to(DEV), exact)])
    assert 1e-4 < rel_err(out, exact) < 2e-2          # it IS a different arithmetic: ~2^-9 per operand
    with pytest.raises(_lib.GGNNError):               # no single-product mode for the K <= 12 encoder projection
        be.project_batch([(x.to(DEV), F, None, wp[:96, :Fp].contiguous().to(DEV), bp[:96].to(DEV), out[:, :96],
                           _lib.GGNN_PRECISION_BF16)])


def test_gemm_arithmetic_is_fp32_equivalent():
    """The decoder GEMMs run by default as 3 x bf16 split / 6 MFMA products (GGNN_GEMM_BF16X6).
    Their error against an fp64 product, normalised by sum_k |x_k||w_k| (the quantity fp32
    rounding scales with), must stay at the fp32 level.  Measured on MI355X with these inputs:
    2.7e-7 for the split kernels, 7.0e-7 for the native fp32 MFMA chain (GGNN_GEMM=fp32), whose
    104 sequential fp32 roundings the split path replaces by 24."""
    rs = np.random.RandomState(5)
    M, F, ncols = 4096, 8, 2688
    x = torch.from_numpy((rs.standard_normal((M, F)) * 10 ** rs.uniform(-3, 1, (M, 1))).astype(np.float32))
    h = torch.from_numpy(np.tanh(rs.standard_normal((M, 96))).astype(np.float32))
    wp = torch.from_numpy((rs.standard_normal((ncols, 104)) * 10 ** rs.uniform(-2, 0, (ncols, 1))).astype(np.float32))
    bp = torch.zeros(ncols)
    xin = torch.cat([x, h], 1).double()
    ref = xin @ wp.double().t()
    scale = xin.abs() @ wp.double().abs().t()
    out = torch.empty(M, ncols, device=DEV)
    backend().project(x.to(DEV), F, h.to(DEV), wp.to(DEV), bp.to(DEV), out)
    err = float(((out.cpu().double() - ref).abs() / scale).max())
    bound = 4e-7 if backend().lib.ggnn_gemm_mode() == 1 else 1e-6
    assert err < bound, f"decoder projection error {err:.2e} of sum|x||w| (mode {backend().lib.ggnn_gemm_mode()})"


def test_three_product_projection_is_fp32_equivalent():
    """GGNN_PRECISION_F16X2 (the fused decoder plan's value projection: two fp16 pieces per operand, three MFMA products)
    against the fp64 product on operands spanning 1e-3 .. 1e1 row by row: within 5e-7 of sum |x||w| + |b| -- two pieces keep
    22 significand bits per operand, 2 x 2^-22 = 4.8e-7 per term at worst (measured 3.4e-7 where the six-product split,
    with its 24 fp32 accumulations per output, gives 8.0e-7) --, never worse than the six-product result; ragged row
    counts and every feature width."""
    if backend().lib.ggnn_gemm_mode() != 1:
        pytest.skip("split GEMM kernels only")
    rs = np.random.RandomState(6)
    for M, F, ncols in ((4096, 8, 768), (4099, 11, 384), (20000, 8, 768), (7, 3, 96)):
        x = torch.from_numpy((rs.standard_normal((M, F)) * 10 ** rs.uniform(-3, 1, (M, 1))).astype(np.float32))
        h = torch.from_numpy(np.tanh(rs.standard_normal((M, 96))).astype(np.float32))
        Fp = (F + 3) & ~3
        wp = torch.zeros(ncols, Fp + 96)
        wp[:, :F] = torch.from_numpy((rs.standard_normal((ncols, F)) * 10 ** rs.uniform(-2, 0, (ncols, 1))).astype(np.float32))
        wp[:, Fp:] = torch.from_numpy((rs.standard_normal((ncols, 96)) * 10 ** rs.uniform(-2, 0, (ncols, 1))).astype(np.float32))
        bp = torch.from_numpy(rs.standard_normal(ncols).astype(np.float32))
        xin = torch.cat([x, torch.zeros(M, Fp - F), h], 1).double()
        ref = xin @ wp.double().t() + bp.double()
        scale = xin.abs() @ wp.double().abs().t() + bp.double().abs()
        out3, out6 = torch.empty(M, ncols, device=DEV), torch.empty(M, ncols, device=DEV)
        args = (x.to(DEV), F, h.to(DEV), wp.to(DEV), bp.to(DEV))
        backend().project_batch([(*args, out3, _lib.GGNN_PRECISION_F16X2)])
        backend().project_batch([(*args, out6)])
        e3 = float(((out3.cpu().double() - ref).abs() / scale).max())
        e6 = float(((out6.cpu().double() - ref).abs() / scale).max())
        assert e3 < 5e-7, f"three-product projection M={M} F={F}: {e3:.2e} of sum|x||w| (six products: {e6:.2e})"
        assert e6 < 1.2e-6 and e3 < e6 + 1e-7


@pytest.mark.parametrize("M,K,n_out,batch,transposed,with_cin", [
    (20000, 224, 96, 4, False, False),      # gate GEMM of the decoder's joints: z_g = agg_g W2_g^T
    (20000, 96, 224, 4, True, False),       # its input gradient: g_agg_g = g_z_g W2_g
    (10000, 1248, 96, 1, True, True),       # g_h += gP Wp[:, h columns] (grains)
    (20000, 2112, 96, 1, True, True),       # ... joints
    (10000, 128, 96, 3, False, False), (10000, 96, 128, 3, True, False),   # encoder, grains
    (1, 32, 16, 1, False, False), (15, 64, 48, 2, True, True), (17, 96, 112, 1, False, False), (130, 160, 208, 1, True, False)])
def test_rowgemm_against_float64(M, K, n_out, batch, transposed, with_cin):
    """ggnn_rowgemm (training path: the gate GEMM, its input gradient, the hidden-state gradient; every instantiated
    tile width and the padded ones, ragged last tiles, M < 16, batches, both weight orientations, accumulate-into)
    against the fp64 product: fp32 mode within 5e-7 of sum |a||w| -- the two-piece split keeps 22 significand bits per
    operand, i.e. 2 x 2^-22 = 4.8e-7 per term at worst (few terms dominate these wide-range rows; measured 3e-8 .. 2.7e-7;
    an fp32 fma chain: 2e-7 .. 7e-7) --, bf16 mode equal to the product of bf16-rounded operands."""
    be = backend()
    rs = np.random.RandomState(M + K + n_out)
    lda, ldc = K + 32, n_out + 16
    a = torch.from_numpy((rs.standard_normal((batch, M, lda)) * 10 ** rs.uniform(-3, 1, (batch, M, 1))).astype(np.float32)).to(DEV)
    wshape = (batch, K + 8, n_out + 4) if transposed else (batch, n_out + 4, K + 8)
    w = torch.from_numpy((rs.standard_normal(wshape) * 10 ** rs.uniform(-2, 0, (batch, wshape[1], 1))).astype(np.float32)).to(DEV)
    cin = torch.from_numpy(rs.standard_normal((batch, M, ldc)).astype(np.float32)).to(DEV) if with_cin else None
    out = torch.full((batch, M, ldc), float("nan"), device=DEV)
    be.rowgemm(a, w, out, K, n_out, batch=batch, c_in=cin, transposed=transposed)
    A64 = a[:, :, :K].cpu().double()
    W64 = (w[:, :K, :n_out].transpose(1, 2) if transposed else w[:, :n_out, :K]).cpu().double()
    ref = A64 @ W64.transpose(1, 2)
    norm = A64.abs() @ W64.abs().transpose(1, 2)
    if with_cin:
        ref = ref + cin[:, :, :n_out].cpu().double()
        norm = norm + cin[:, :, :n_out].cpu().double().abs()
    got = out[:, :, :n_out].cpu().double()
    err = float(((got - ref).abs() / norm).max())
    # (+ the fp32 accumulator's own rounding over K terms: ~sqrt(K) 2^-24, which a K = 2112 reduction shows: 1.5e-6)
    assert err < 5e-7 * max(1.0, (K / 96) ** 0.5), f"rowgemm fp32 mode: {err:.2e} of sum |a||w|"
    assert bool(torch.isnan(out[:, :, n_out:]).all())              # columns behind n_out are not touched
    ob = torch.full((batch, M, ldc), float("nan"), device=DEV)
    be.rowgemm(a, w, ob, K, n_out, batch=batch, c_in=cin, transposed=transposed, bf16=True)
    refb = A64.float().bfloat16().double() @ W64.float().bfloat16().double().transpose(1, 2)
    if with_cin:
        refb = refb + cin[:, :, :n_out].cpu().double()
    errb = float(((ob[:, :, :n_out].cpu().double() - refb).abs() / norm).max())
    assert errb < 2e-6, f"rowgemm bf16 mode: {errb:.2e} of sum |a||w| against the product of bf16-rounded operands"
    with pytest.raises(_lib.GGNNError):
        be.rowgemm(a, w, out, K, n_out + 1, batch=batch, transposed=transposed)    # n_out must be a multiple of 16
    # weight planes packed ahead (ggnn_rowgemm_pack: both precisions of this product in one launch): the same bits
    pf, pb = be.rowgemm_pack([(w, K, n_out, batch, transposed, False), (w, K, n_out, batch, transposed, True)])
    two_pass = n_out > 128 and K > 128          # fp32 products that run as two column halves pack per half
    o2 = torch.full((batch, M, ldc), float("nan"), device=DEV)
    if two_pass:
        with pytest.raises(_lib.GGNNError):
            be.rowgemm(a, w, o2, K, n_out, batch=batch, c_in=cin, transposed=transposed, planes=pf)
    else:
        be.rowgemm(a, w, o2, K, n_out, batch=batch, c_in=cin, transposed=transposed, planes=pf)
        assert torch.equal(o2[:, :, :n_out], out[:, :, :n_out])
    be.rowgemm(a, w, o2, K, n_out, batch=batch, c_in=cin, transposed=transposed, bf16=True, planes=pb)
    assert torch.equal(o2[:, :, :n_out], ob[:, :, :n_out])


@pytest.mark.parametrize("bf16", [False, True])
def test_rowgemm_pair_equals_the_two_single_products(bf16):
    """ggnn_rowgemm_pair (ABI 25): the two node types' hidden-state gradients (10 000 x 1248 and 20 003 x 2112 -> 96, accumulated
    into the sweeps' source-side gradient) side by side in one grid -- bit for bit the two single calls; a pair the C entry
    point would refuse (one product's planes stay in LDS) runs as two calls through the same wrapper."""
    be = backend()
    rs = np.random.RandomState(7)

    def product(M, K):
        a = torch.from_numpy((rs.standard_normal((M, K + 32)) * 10 ** rs.uniform(-6, 0, (M, 1))).astype(np.float32)).to(DEV)
        w = torch.from_numpy(rs.standard_normal((K, 100)).astype(np.float32)).to(DEV)
        cin = torch.from_numpy(rs.standard_normal((M, 96)).astype(np.float32)).to(DEV)
        planes, = be.rowgemm_pack([(w, K, 96, 1, True, bf16)])
        want = be.rowgemm(a, w, torch.empty(M, 96, device=DEV), K, 96, c_in=cin, transposed=True, bf16=bf16, planes=planes)
        return (a, w, torch.full((M, 96), float("nan"), device=DEV), K, 96, cin, True, bf16, planes), want
    (p0, w0), (p1, w1), (p2, w2) = product(10000, 1248), product(20003, 2112), product(777, 128)
    o0, o1 = be.rowgemm_pair(p0, p1)
    assert torch.equal(o0, w0) and torch.equal(o1, w1)
    p0[2].fill_(float("nan"))
    o0, o2 = be.rowgemm_pair(p0, p2)      # (K = 128: resident planes -> two calls)
    assert torch.equal(o0, w0) and torch.equal(o2, w2)


@pytest.mark.parametrize("M,K,n_out,batch,transposed", [
    (4096, 224, 96, 4, False),      # resident planes (the gate GEMM's shapes)
    (4096, 96, 224, 2, True),       # its input gradient g_agg = g_z W2
    (4099, 1248, 96, 1, True)])     # streamed planes: g_h += gP Wp[:, h columns]
def test_rowgemm_keeps_its_resolution_on_gradient_sized_rows(M, K, n_out, batch, transposed):
    """The backward products stream GRADIENT rows (g_z, gP: 1e-4 .. 1e-10 per node, masked nodes exactly 0).  The
    two-piece fp16 split resolves 2^-36 absolutely, which would be 1.5e-4 of a 1e-7 element: ggnn_rowgemm scales every
    row by a power of two from its largest magnitude (include/ggnn.h).  Rows spanning 1e-3 .. 1e-12 row by row (and 1e3
    within a row) against the fp64 product: the same 5e-7 of sum |a||w| as O(1) rows; all-zero rows give exact zeros; a
    row with an inf or a NaN gives NaN outputs in that row only."""
    be = backend()
    rs = np.random.RandomState(M + K)
    a = rs.standard_normal((batch, M, K)) * 10 ** rs.uniform(-12, -3, (batch, M, 1)) * 10 ** rs.uniform(-3, 0, (batch, M, K))
    a[:, 5] = 0.0                                     # a masked node
    a = torch.from_numpy(a.astype(np.float32)).to(DEV)
    wshape = (batch, K, n_out) if transposed else (batch, n_out, K)
    w = torch.from_numpy((rs.standard_normal(wshape) * 0.3).astype(np.float32)).to(DEV)
    out = torch.empty(batch, M, n_out, device=DEV)
    be.rowgemm(a, w, out, K, n_out, batch=batch, transposed=transposed)
    A64 = a.cpu().double()
    W64 = (w.transpose(1, 2) if transposed else w).cpu().double()
    ref = A64 @ W64.transpose(1, 2)
    norm = A64.abs() @ W64.abs().transpose(1, 2)
    got = out.cpu().double()
    live = norm > 0
    err = float(((got - ref).abs()[live] / norm[live]).max())
    assert err < 5e-7 * max(1.0, (K / 96) ** 0.5), f"rowgemm on gradient-sized rows: {err:.2e} of sum |a||w|"
    assert bool((out[:, 5] == 0).all())
    a[0, 7, 3] = float("inf")
    a[batch - 1, 9, K - 1] = float("nan")
    be.rowgemm(a, w, out, K, n_out, batch=batch, transposed=transposed)
    bad = torch.isnan(out).all(-1)
    assert bool(bad[0, 7]) and bool(bad[batch - 1, 9]) and int(bad.sum()) == 2
    assert bool(torch.isfinite(out[~bad]).all())


def _gate_problem(N, Ka, mode, seed):
    """Random operands of one ggnn_lstm_epilogue problem in the workspace layout (gate stride padded
    to 32 floats) + its float64 result."""
    rs = np.random.RandomState(seed)
    G = {_lib.MODE_LSTM: 4, _lib.MODE_LSTM_H0: 3}.get(mode, 1 + 3 * (seed % 2))
    Kg = (Ka + 31) // 32 * 32
    agg = torch.from_numpy(rs.uniform(-1, 1, (N, G * Kg)).astype(np.float32))
    w2 = torch.from_numpy((rs.standard_normal((G, 96, Ka)) * 0.2).astype(np.float32))
    ldp = G * 96 + 32
    pd = torch.from_numpy(rs.uniform(-1, 1, (N, ldp)).astype(np.float32))
    c_in = torch.from_numpy(rs.uniform(-1, 1, (N, 96)).astype(np.float32))
    s_off = 32
    pre = [agg[:, g * Kg:g * Kg + Ka].double() @ w2[g].double().t() + pd[:, s_off + g * 96:s_off + (g + 1) * 96].double()
           for g in range(G)]
    if mode == _lib.MODE_RAW:
        ref = (torch.cat(pre, 1),)
    elif mode == _lib.MODE_LSTM:
        c = torch.sigmoid(pre[1]) * c_in.double() + torch.sigmoid(pre[0]) * torch.tanh(pre[2])
        ref = (torch.sigmoid(pre[3]) * torch.tanh(c), c)
    else:
        c = torch.sigmoid(pre[0]) * torch.tanh(pre[1])
        ref = (torch.sigmoid(pre[2]) * torch.tanh(c), c)
    return dict(agg=agg, w2=w2, pd=pd, c_in=c_in, s_off=s_off, G=G, Kg=Kg, ref=ref)


def _gate_args(P, mode):
    from graingraphnn_amd.packing import bf16_planes
    d = lambda t: t.to(DEV)
    N = P["agg"].size(0)
    out = [torch.full((N, 96 * (P["G"] if mode == _lib.MODE_RAW else 1)), float("nan"), device=DEV),
           torch.full((N, 96), float("nan"), device=DEV)]
    w2 = d(P["w2"])
    args = (d(P["agg"]), w2, d(P["pd"]), P["s_off"], d(P["c_in"]) if mode == _lib.MODE_LSTM else None,
            None if mode == _lib.MODE_RAW else out[0], None if mode == _lib.MODE_RAW else out[1],
            out[0] if mode == _lib.MODE_RAW else None, P["G"], mode, bf16_planes(w2), P["Kg"])
    return args, out


@pytest.mark.parametrize("mode", [_lib.MODE_LSTM, _lib.MODE_LSTM_H0, _lib.MODE_RAW])
@pytest.mark.parametrize("N,Ka", [(1, 100), (15, 196), (16, 100), (17, 196), (95, 196), (96, 100), (1000, 196),
                                  (4099, 100), (20000, 196)])
def test_gate_gemm_and_lstm_against_float64(N, Ka, mode):
    """ggnn_lstm_epilogue alone (every tile count a workgroup can get, ragged last tiles, N < 16).
    Floor: these operands have sum_k |agg||w| ~ 19 per output, and the split GEMM is good to 2.7e-7
    of that (test_gemm_arithmetic_is_fp32_equivalent) = 5e-6 on a pre-activation."""
    P = _gate_problem(N, Ka, mode, N + Ka)
    args, out = _gate_args(P, mode)
    backend().lstm_epilogue(*args)
    for got, ref, what in zip(out, P["ref"], ("h / raw", "c")):
        assert_close(got, ref.float(), f"gates N={N} Ka={Ka} mode={mode} {what}", 1e-5, 5e-6)


@pytest.mark.parametrize("mode", [_lib.MODE_LSTM, _lib.MODE_LSTM_H0])
def test_gate_batch_equals_single_launches(mode):
    """ggnn_lstm_epilogue_batch: four problems of different sizes and widths in one launch give
    the bits of four single launches."""
    shapes = [(20000, 196), (10000, 100), (777, 196), (33, 100)]
    probs = [_gate_problem(N, Ka, mode, 7 * k + 1) for k, (N, Ka) in enumerate(shapes)]
    single, batch = [], []
    for P in probs:
        a, o = _gate_args(P, mode)
        backend().lstm_epilogue(*a)
        single.append(o)
        batch.append(_gate_args(P, mode))
    backend().lstm_epilogue_batch([a for a, _ in batch])
    for (a, o), s, P in zip(batch, single, probs):
        for got, one, ref in zip(o, s, P["ref"]):
            assert torch.equal(got, one)
            assert_close(got, ref.float(), "gate batch", 1e-5, 5e-6)


def test_native_fp32_gemm_mode_in_a_subprocess():
    """GGNN_GEMM=fp32 (native v_mfma_f32_16x16x4_f32 kernels) is fixed per process: run the GEMM,
    cell and forward parity tests once more under it."""
    import os
    import subprocess
    import sys
    if os.environ.get("GGNN_GEMM") == "fp32":
        assert backend().lib.ggnn_gemm_mode() == 0
        return
    assert backend().lib.ggnn_gemm_mode() == 1
    env = dict(os.environ, GGNN_GEMM="fp32")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, "-k",
                        "project or gemm or periodconv or cell_golden or forward_golden"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_three_kernel_decoder_plan_in_a_subprocess():
    """GGNN_DEC=split (projection + sweeps + gate GEMM per decoder cell instead of the fused decoder cell) is fixed per
    process: the cell, forward and rollout goldens and the launch-tape test once more under it."""
    import os
    import subprocess
    import sys
    if os.environ.get("GGNN_DEC") == "split":
        assert backend().fused_decoder is False
        return
    env = dict(os.environ, GGNN_DEC="split")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, "-k",
                        "cell_golden or forward_golden or rollout_golden or tape or three_kernel"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@torch.no_grad()
def test_a_rollout_reports_its_own_range_flag():
    """Two rollouts on one device: an activation beyond fp16's range in one (include/ggnn.h, OPERAND RANGE) is reported by
    THAT rollout's range_exceeded() / state() and by nobody else's -- the fused cells of a rollout's launches write the
    rollout's own flag word, not the device-wide one."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = load_graph("40")
    R, Cm = product_models(10020, 1.0, DEV)
    Xa, Xb = tt(x, DEV), tt(x, DEV)
    ra = GrainRollout(R, Cm, Xa, tt(ei, DEV), tt(ea, DEV), 6)
    rb = GrainRollout(R, Cm, Xb, tt(ei, DEV), tt(ea, DEV), 6)
    backend().range_exceeded(DEV)                   # clear the device-wide word
    Xa["joint"][17, 5] = 7.0e4                      # a feature beyond fp16's range (the encoder cell is fused at every size)
    ra.step()
    rb.step()
    assert not rb.range_exceeded() and not backend().range_exceeded(DEV)
    assert ra.range_exceeded(clear=False)
    with pytest.raises(_lib.GGNNError):
        ra.state()
    rb.state()


def test_fused_decoder_plan_on_the_fixtures_in_a_subprocess():
    """By default the decoder plan follows the graph's size (backend.FUSED_DECODER_MIN_JOINTS: the fixtures of the golden
    tests run the three-kernel plan, the 10k-grain tests the fused cell).  GGNN_DEC=fused forces the fused decoder cell
    whatever the size: the cell, forward and rollout goldens and the launch-tape test once more under it."""
    import os
    import subprocess
    import sys
    if os.environ.get("GGNN_DEC") == "fused":
        assert backend().fused_decoder is True and backend().fused_decoder_min_joints == 0
        return
    if backend().lib.ggnn_gemm_mode() == 1 and "GGNN_DEC" not in os.environ:
        assert backend().fused_decoder is True and backend().fused_decoder_min_joints == backend().FUSED_DECODER_MIN_JOINTS
    env = dict(os.environ, GGNN_DEC="fused")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, "-k",
                        "cell_golden or forward_golden or rollout_golden or tape or fused_decoder_plan or ragged or tiny"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ---------------------------------------------------------------------------------------
# op level: PeriodConv and HeteroPGCLSTM against the reference's golden vectors (cfg1)
# ---------------------------------------------------------------------------------------
@torch.no_grad()
def test_periodconv_golden():
    x, ei, ea = load_graph("40")
    g = golden("cfg1_s1")
    R, _ = product_models(10020, 1.0, DEV)
    n_nodes = {nt: v.shape[0] for nt, v in x.items()}
    h0 = tt(random_state(n_nodes, 7), DEV)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    xh = {nt: torch.cat([X[nt], h0[nt]], 1) for nt in x}
    for et in EDGE_TYPES:
        conv = R.gclstm_decoder.cell_list[0].conv_i.convs[etk(et)]
        arg = xh[et[0]] if et[0] == et[-1] else (xh[et[0]], xh[et[-1]])
        assert_close(conv(arg, EI[et], EA[et]), g["conv_" + etk(et)], f"PeriodConv {et}")


@torch.no_grad()
def test_cell_golden_zero_and_nonzero_state():
    x, ei, ea = load_graph("40")
    g = golden("cfg1_s1")
    R, _ = product_models(10020, 1.0, DEV)
    n_nodes = {nt: v.shape[0] for nt, v in x.items()}
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    h, c = R.gclstm_encoder.cell_list[0](X, EI, EA, None, None)
    for nt in x:
        assert_close(h[nt], g[f"cell0_h_{nt}"], f"cell0 h {nt}")
        assert_close(c[nt], g[f"cell0_c_{nt}"], f"cell0 c {nt}")
    h0, c0 = tt(random_state(n_nodes, 7), DEV), tt(random_state(n_nodes, 8), DEV)
    h, c = R.gclstm_decoder.cell_list[0](X, EI, EA, h0, c0)
    for nt in x:
        assert_close(h[nt], g[f"cell1_h_{nt}"], f"cell1 h {nt}")
        assert_close(c[nt], g[f"cell1_c_{nt}"], f"cell1 c {nt}")
    # generic (k2 = 96) path with an explicit zero state == the h = 0 fast path
    z = {nt: torch.zeros(n_nodes[nt], 96, device=DEV) for nt in x}
    h, c = R.gclstm_encoder.cell_list[0](X, EI, EA, z, z)
    for nt in x:
        assert_close(h[nt], g[f"cell0_h_{nt}"], f"cell0 (explicit zeros) h {nt}")
        assert_close(c[nt], g[f"cell0_c_{nt}"], f"cell0 (explicit zeros) c {nt}")


# ---------------------------------------------------------------------------------------
# model level: regressor deltas and classifier logits against the goldens
# ---------------------------------------------------------------------------------------
def _inputs(tag):
    if tag.startswith("cfg1"):
        return load_graph("40")
    x, ei, ea = load_graph("120")
    x, ea = fold_120(x, ea)
    return x, ei, ea


CASES = [("cfg1_s1", 10020, 1.0), ("cfg1_s3", 10020, 3.0), ("cfg2_s1", 0, 1.0)]


@pytest.mark.parametrize("tag,seed,scale", CASES)
@torch.no_grad()
def test_forward_golden(tag, seed, scale):
    x, ei, ea = _inputs(tag)
    g = golden(tag)
    R, Cm = product_models(seed, scale, DEV)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    yr, yc = R(X, EI, EA), Cm(X, EI, EA)
    for k in ("joint", "grain", "grain_area"):
        assert_close(yr[k], g["R_" + k], f"{tag} regressor {k}")
    for k in ("edge_event", "edge"):
        assert_close(yc[k], g["C_" + k], f"{tag} classifier {k}")
    # drop-in update(): periodic branch of models.py:503-516, in place
    Xo = tt(x)
    oR, _ = oracle_models(seed, scale)
    pred_o = {k: torch.from_numpy(g["R_" + k].copy()) for k in ("joint", "grain")}
    oR.update(Xo, pred_o, {})
    gs = {}
    R.update(X, yr, gs)
    for nt in x:
        assert_close(X[nt], Xo[nt], f"{tag} update x {nt}")
    assert gs["active_grains"].numel() == x["grain"].shape[0]


@pytest.mark.parametrize("use_graph,launches", [(False, "joint"), (True, "joint"), (False, "serial"), (True, "two_streams")])
@pytest.mark.parametrize("tag,seed,scale", CASES)
@torch.no_grad()
def test_rollout_golden(tag, seed, scale, use_graph, launches):
    from graingraphnn_amd import GrainRollout
    x, ei, ea = _inputs(tag)
    g = golden(tag)
    n_steps, span = int(g["meta"][2]), int(g["meta"][3])
    R, Cm = product_models(seed, scale, DEV)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    ro = GrainRollout(R, Cm, X, EI, EA, span, use_graph=use_graph, joint_launches=launches == "joint",
                      concurrent=launches == "two_streams")
    for step in range(1, n_steps + 1):
        pred = ro.step()
        if step == 1:
            for k in ("joint", "grain", "grain_area"):
                assert_close(pred[k], g["R_" + k], f"{tag} step1 R {k}")
            for k in ("edge_event", "edge"):
                assert_close(pred[k], g["C_" + k], f"{tag} step1 C {k}")
        if step in (1, n_steps):
            for nt in x:
                assert_close(X[nt], g[f"step{step}_x_{nt}"], f"{tag} step{step} x {nt}")
            ead = ro.edge_attr_dict()
            for et in EDGE_TYPES:
                assert_close(ead[et], g[f"step{step}_ea_{etk(et)}"], f"{tag} step{step} edge_attr {et}")


@torch.no_grad()
def test_full_120_frame_rollout_against_oracle():
    """BASELINE config 2: seed=0 lxd=120 graph, frames 6..120 in steps of span 6 = 20 steps."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = _inputs("cfg2_s1")
    R, Cm = product_models(0, 1.0, DEV)
    oR, oC = oracle_models(0, 1.0)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    oX, oEI, oEA = tt(x), tt(ei), tt(ea)
    ro = GrainRollout(R, Cm, X, EI, EA, 6)
    for step in range(20):
        ro.step()
        _, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6)
    for nt in x:
        assert_close(X[nt], oX[nt], f"20-step x {nt}")
    assert float(X["grain"][0, 2]) == pytest.approx(120 / 121, abs=1e-6)  # z clamp reached
    for et in EDGE_TYPES:
        assert_close(ro.edge_attr_dict()[et], oEA[et], f"20-step edge_attr {et}")


# ---------------------------------------------------------------------------------------
# edge cases: empty rows, ragged / large degrees (LDS window overflow path), odd sizes
# ---------------------------------------------------------------------------------------
def _ragged_graph(seed, n_g=37, n_j=75, hub_deg=0):
    rs = np.random.RandomState(seed)
    x = {"grain": rs.uniform(0, 1, (n_g, 11)).astype(np.float32),
         "joint": rs.uniform(0, 1, (n_j, 8)).astype(np.float32)}

    def rand_edges(n_s, n_d, E, skip_dst):
        s, d = rs.randint(0, n_s, E), rs.randint(0, n_d, E)
        keep = ~np.isin(d, skip_dst)
        return np.stack([s[keep], d[keep]]).astype(np.int64)

    ei = {GJ: rand_edges(n_g, n_j, 4 * n_j, [0, 5]),      # joints 0 and 5 get no grain edges
          JG: rand_edges(n_j, n_g, 7 * n_g, [n_g - 1]),   # last grain has no in-edges at all
          JJ: rand_edges(n_j, n_j, 3 * n_j, [0])}         # joint 0 has no in-edges of any type
    if hub_deg:
        hub = np.stack([rs.randint(0, n_j, hub_deg), np.full(hub_deg, 3)]).astype(np.int64)
        ei[JG] = np.concatenate([ei[JG], hub], 1)[:, rs.permutation(ei[JG].shape[1] + hub_deg)]
    ea = {et: rs.uniform(0.01, 0.1, (v.shape[1], 1)).astype(np.float32) for et, v in ei.items()}
    return x, ei, ea


@pytest.mark.parametrize("seed,hub_deg", [(0, 0), (1, 37), (2, 900)])
@torch.no_grad()
def test_ragged_graphs_against_oracle(seed, hub_deg):
    """Zero in-degree rows (PyG leaves them at lin_skip only), multi-chunk softmax rows and a
    900-neighbour hub that overflows the 512-slot LDS window of its workgroup."""
    x, ei, ea = _ragged_graph(seed, hub_deg=hub_deg)
    R, Cm = product_models(11 + seed, 1.0, DEV)
    oR, oC = oracle_models(11 + seed, 1.0)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    yr, yc = R(X, EI, EA), Cm(X, EI, EA)
    oyr, oyc = oR(tt(x), tt(ei), tt(ea)), oC(tt(x), tt(ei), tt(ea))
    for k in ("joint", "grain", "grain_area"):
        assert_close(yr[k], oyr[k], f"ragged regressor {k}")
    for k in ("edge_event", "edge"):
        assert_close(yc[k], oyc[k], f"ragged classifier {k}")


@pytest.mark.parametrize("n_g,n_j", [(3, 7), (9, 15), (16, 17), (23, 33)])
@torch.no_grad()
def test_tiny_graphs_against_oracle(n_g, n_j):
    """Fewer nodes than one 16-row tile, exactly one tile, and ragged last tiles: the GEMM kernels'
    sliding / clamped tile addressing."""
    rs = np.random.RandomState(n_g * 100 + n_j)
    x = {"grain": rs.uniform(0, 1, (n_g, 11)).astype(np.float32), "joint": rs.uniform(0, 1, (n_j, 8)).astype(np.float32)}
    ei = {GJ: np.stack([rs.randint(0, n_g, 3 * n_j), np.repeat(np.arange(n_j), 3)]).astype(np.int64),
          JG: np.stack([rs.randint(0, n_j, 5 * n_g), rs.randint(0, n_g, 5 * n_g)]).astype(np.int64),
          JJ: np.stack([rs.randint(0, n_j, 3 * n_j), rs.randint(0, n_j, 3 * n_j)]).astype(np.int64)}
    ea = {et: rs.uniform(0.01, 0.1, (v.shape[1], 1)).astype(np.float32) for et, v in ei.items()}
    R, Cm = product_models(5, 1.0, DEV)
    oR, oC = oracle_models(5, 1.0)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    yr, yc = R(X, EI, EA), Cm(X, EI, EA)
    oyr, oyc = oR(tt(x), tt(ei), tt(ea)), oC(tt(x), tt(ei), tt(ea))
    for k in ("joint", "grain", "grain_area"):
        assert_close(yr[k], oyr[k], f"tiny ({n_g}, {n_j}) regressor {k}")
    for k in ("edge_event", "edge"):
        assert_close(yc[k], oyc[k], f"tiny ({n_g}, {n_j}) classifier {k}")


@pytest.mark.parametrize("layer_size", [64, 32])
@torch.no_grad()
def test_narrow_layer_sizes_forward_and_rollout_against_oracle(layer_size):
    """parameters.py:19: layer_size 64 / 32 on the 96-wide kernels through zero-padded packed weights
    (packing.padded_cell): both forwards on the 40 um fixture, a three-step rollout (two streams, hipGraph) on the folded
    120 um fixture, the module-level cell call with a caller-held hidden state -- each against the oracle of that width."""
    from graingraphnn_amd import GrainRollout
    from test_host_logic import _narrow_models
    (R, Cm), (oR, oC) = _narrow_models(layer_size, 31, DEV)
    x, ei, ea = load_graph("40")
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    yr, yc = R(X, EI, EA), Cm(X, EI, EA)
    oyr, oyc = oR(tt(x), tt(ei), tt(ea)), oC(tt(x), tt(ei), tt(ea))
    for k in ("joint", "grain", "grain_area"):
        assert_close(yr[k], oyr[k], f"layer_size {layer_size} regressor {k}")
    for k in ("edge_event", "edge"):
        assert_close(yc[k], oyc[k], f"layer_size {layer_size} classifier {k}")
    # the cell as a module: (h, c) of the caller's width in and out
    cell, ocell = R.gclstm_decoder.cell_list[0], oR.gclstm_decoder.cell_list[0]
    rs = np.random.RandomState(2)
    h0 = {nt: rs.uniform(-1, 1, (x[nt].shape[0], layer_size)).astype(np.float32) for nt in x}
    c0 = {nt: rs.uniform(-1, 1, (x[nt].shape[0], layer_size)).astype(np.float32) for nt in x}
    h1, c1 = cell(X, EI, EA, tt(h0, DEV), tt(c0, DEV))
    oh1, oc1 = ocell(tt(x), tt(ei), tt(ea), tt(h0), tt(c0))
    for nt in x:
        assert h1[nt].shape == (x[nt].shape[0], layer_size)
        assert_close(h1[nt], oh1[nt], f"layer_size {layer_size} cell h {nt}")
        assert_close(c1[nt], oc1[nt], f"layer_size {layer_size} cell c {nt}")
    # rollout
    x, ei, ea = load_graph("120")
    x, ea = fold_120(x, ea)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    oX, oEI, oEA = tt(x), tt(ei), tt(ea)
    ro = GrainRollout(R, Cm, X, EI, EA, 6, use_graph=True)
    for step in range(3):
        pred = {k: v.clone() for k, v in ro.step().items()}
        opred, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6)
        for k in ("joint", "grain", "grain_area", "edge_event", "edge"):
            assert_close(pred[k], opred[k], f"layer_size {layer_size} step {step} {k}")


@torch.no_grad()
def test_voronoi_graph_rollout_against_oracle():
    """A random Voronoi structure (grain degrees 3..11: rows of up to four units, triangles) through
    three full steps incl. the centre refresh."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = synthetic.voronoi(500, seed=11)
    deg = np.bincount(ei[JG][1])
    assert deg.min() == 3 and deg.max() >= 9
    # small weights: junctions must move by less than a grain per step, or the polygons fold over
    # themselves and the chained min-image of the centre refresh depends on the junction order
    R, Cm = product_models(3, 0.1, DEV)
    oR, oC = oracle_models(3, 0.1)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    oX, oEI, oEA = tt(x), tt(ei), tt(ea)
    ro = GrainRollout(R, Cm, X, EI, EA, 6, refresh_centres=True)
    for step in range(3):
        pred = {k: v.clone() for k, v in ro.step().items()}
        opred, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6, centres=(1.0, None))
        for k in ("joint", "grain", "grain_area", "edge_event", "edge"):
            assert_close(pred[k], opred[k], f"voronoi step {step} {k}")
    for nt in x:
        assert_close(X[nt], oX[nt], f"voronoi x {nt}")


@pytest.mark.parametrize("source", ["reference_generator_fixture", "own_generator"])
@torch.no_grad()
def test_generated_structures_roll_out_like_the_oracle(source):
    """an initial structure from the reference's `--mode=generate` (fixture) and one
    from `synthetic.generate` go through the HIP rollout and match the oracle step by step."""
    import os
    from helpers import GOLDEN
    from graingraphnn_amd import GrainRollout
    if source == "own_generator":
        x, ei, ea = synthetic.generate(lxd=40, seed=5, G=2.0, R=0.4, span=8)
    else:
        g = np.load(os.path.join(GOLDEN, "generated_40_seed1.npz"))
        x = {"grain": g["x_grain"], "joint": g["x_joint"]}
        ei = {et: g["ei_" + etk(et)] for et in EDGE_TYPES}
        ea = {et: g["ea_" + etk(et)] for et in EDGE_TYPES}
    R, Cm = product_models(21, 1.0, DEV)
    oR, oC = oracle_models(21, 1.0)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    oX, oEI, oEA = tt(x), tt(ei), tt(ea)
    ro = GrainRollout(R, Cm, X, EI, EA, 8, refresh_centres=True)
    for step in range(3):
        pred = {k: v.clone() for k, v in ro.step().items()}
        opred, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 8, centres=(1.0, None))
        for k in ("joint", "grain", "grain_area", "edge_event", "edge"):
            assert_close(pred[k], opred[k], f"generated ({source}) step {step} {k}")
    for nt in x:
        assert_close(X[nt], oX[nt], f"generated ({source}) x {nt}")


# ---------------------------------------------------------------------------------------
# BASELINE full size (cfg3): one oracle step + size-independent properties
# ---------------------------------------------------------------------------------------
@torch.no_grad()
def test_cfg3_full_size_step_and_properties():
    from graingraphnn_amd import GrainRollout
    x, ei, ea = synthetic.honeycomb(100, 10, 0)
    R, Cm = product_models(0, 0.3, DEV)
    oR, oC = oracle_models(0, 0.3)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    ro = GrainRollout(R, Cm, X, EI, EA, 6)
    pred = {k: v.clone() for k, v in ro.step().items()}
    oX, oEI, oEA = tt(x), tt(ei), tt(ea)
    opred, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6)
    for k in ("joint", "grain", "grain_area", "edge_event", "edge"):
        assert_close(pred[k], opred[k], f"cfg3 {k}")
    for nt in x:
        assert_close(X[nt], oX[nt], f"cfg3 step1 x {nt}")
    for et in EDGE_TYPES:
        assert_close(ro.edge_attr_dict()[et], oEA[et], f"cfg3 step1 edge_attr {et}")
    # (1) bitwise determinism (atomics-free CSR aggregation): same inputs -> same bits
    X2 = tt(x, DEV)
    ro2 = GrainRollout(R, Cm, X2, tt(ei, DEV), tt(ea, DEV), 6)
    pred2 = ro2.step()
    for k in pred:
        assert torch.equal(pred[k], pred2[k]), f"cfg3 {k} differs between two identical runs"
    # (2) COO order independence: permuting the edge list permutes per-edge outputs only
    perm = {et: np.random.RandomState(9).permutation(ei[et].shape[1]) for et in EDGE_TYPES}
    ei_p = {et: ei[et][:, perm[et]] for et in EDGE_TYPES}
    ea_p = {et: ea[et][perm[et]] for et in EDGE_TYPES}
    X3 = tt(x, DEV)
    ro3 = GrainRollout(R, Cm, X3, tt(ei_p, DEV), tt(ea_p, DEV), 6)
    pred3 = ro3.step()
    for k in ("joint", "grain", "grain_area"):
        assert_close(pred3[k], pred[k], f"cfg3 permuted-COO {k}", 1e-5)
    assert_close(pred3["edge_event"], pred["edge_event"][torch.from_numpy(perm[JJ]).to(DEV)],
                 "cfg3 permuted-COO edge_event", 1e-5)
    # (3) softmax normalisation: sum of alpha is 1 on every row with an in-edge (the fused decoder cell keeps its
    # aggregates on the compute unit: look at them through the split path's sweep, same arithmetic)
    be_ = backend()
    keep_ = be_.fused_decoder
    be_.fused_decoder = False
    try:
        R2s, _ = product_models(0, 0.3, DEV)
        R2s(tt(x, DEV), tt(ei, DEV), tt(ea, DEV))
        sa = R2s._ws.agg_dec["joint"].view(-1, 4, 224)[:, :, 192:196:2]
    finally:
        be_.fused_decoder = keep_
    assert float((sa - 1).abs().max()) < 1e-5


@pytest.mark.parametrize("mode", ["joint_launches_graph", "two_streams_graph"])
@torch.no_grad()
def test_cfg3_ten_step_rollout_as_benched(mode):
    """The exact step bench.py times (BASELINE config 3: 10k-grain honeycomb folded x10, weights
    x0.3, R + C + update + grain-centre refresh through the global frame + edge refresh, hipGraph
    replay), ten consecutive steps against the oracle: predictions of every step, then the state."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea, off = synthetic.honeycomb(100, 10, 0, return_offset=True)
    R, Cm = product_models(0, 0.3, DEV)
    oR, oC = oracle_models(0, 0.3)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    oX, oEI, oEA = tt(x), tt(ei), tt(ea)
    ooff = torch.from_numpy(off)
    kw = dict(joint_launches=True) if mode == "joint_launches_graph" else dict(joint_launches=False, concurrent=True)
    ro = GrainRollout(R, Cm, X, EI, EA, 6, use_graph=True, refresh_centres=True, domain_factor=10.0,
                      domain_offset=ooff, **kw)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))   # the oracle's best setting on a many-core host
    try:
        worst = 0.0
        for step in range(10):
            pred = {k: v.clone() for k, v in ro.step().items()}
            opred, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6, centres=(10.0, ooff))
            for k in ("joint", "grain", "grain_area", "edge_event", "edge"):
                worst = max(worst, assert_close(pred[k], opred[k], f"cfg3 step {step} {k}"))
    finally:
        torch.set_num_threads(threads)
    assert_close(X["joint"], oX["joint"], "cfg3 10 steps x joint")
    assert_close(X["grain"][:, 2:], oX["grain"][:, 2:], "cfg3 10 steps x grain[2:]")
    d = (X["grain"][:, :2].cpu() - oX["grain"][:, :2]).abs()
    assert float(torch.minimum(d, 1 - d).max()) < 1e-4   # frac() may land on either side of 0/1
    for et in EDGE_TYPES:
        assert_close(ro.edge_attr_dict()[et], oEA[et], f"cfg3 10 steps edge_attr {et}")
    print(f"cfg3 10-step rollout ({mode}): worst per-tensor error {worst:.2e}")


@torch.no_grad()
def test_generated_368um_structure_two_steps_against_the_oracle():
    """The headline's size on a node numbering that is NOT the lattice's (bench.py --workload gen368): the
    reference generator's own 368 um sample (graph_trajectory.py:1289-1333 via synthetic.generate: 9 775 grains / 19 550
    junctions / 58 650 edges per type, nodes and edges in Qhull order, in-degrees of grains 3..13), folded x9, as
    bench.py builds it; two steps of the benched rollout (fused cells, two streams, hipGraph) against the oracle."""
    import bench
    from graingraphnn_amd import GrainRollout
    R, Cm, X, EI, EA, (x, ei, ea, fold, off) = bench.build_generated(DEV)
    assert X["grain"].size(0) == 9775 and X["joint"].size(0) == 19550 and EI[JJ].size(1) == 58650
    oR, oC = oracle_models(0, 0.3)
    oX, oEI, oEA = tt(x), tt(ei), tt(ea)
    ooff = torch.from_numpy(off)
    ro = GrainRollout(R, Cm, X, EI, EA, 6, use_graph=True, refresh_centres=True, domain_factor=fold, domain_offset=ooff)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))
    try:
        for step in range(2):
            pred = {k: v.clone() for k, v in ro.step().items()}
            opred, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6, centres=(fold, ooff))
            for k in ("joint", "grain", "grain_area", "edge_event", "edge"):
                assert_close(pred[k], opred[k], f"gen368 step {step} {k}")
    finally:
        torch.set_num_threads(threads)
    assert_close(X["joint"], oX["joint"], "gen368 x joint")
    assert_close(X["grain"][:, 2:], oX["grain"][:, 2:], "gen368 x grain[2:]")
    assert not ro.range_exceeded()


@torch.no_grad()
def test_cfg3_five_hundred_step_rollout_checked_along_its_trajectory():
    """BASELINE config 3 at full length: the 500-step rollout bench.py times (test.py:353-407's loop: both forwards,
    update, grain-centre refresh through the global frame, edge refresh; hipGraph replay, default launch plan).
      * all 500 steps: run twice from the same inputs -> bit-identical states and predictions; every state finite;
      * at 26 points ALONG the HIP trajectory (steps 0-2, every few dozen steps, the last two; 329-331 among them -- the
        oracle's own trajectory puts a grain centre on the fold boundary there) the oracle is given the HIP state and both
        take ONE step from identical inputs: every prediction and the state after the step within the 1e-4 contract.
        Grain centres are compared modulo the fold (frac() may land on either side of 0 / 1); an edge length that
        involves a grain whose centre landed on the other side is excluded, and at most a handful of grains may do so."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea, off = synthetic.honeycomb(100, 10, 0, return_offset=True)
    R, Cm = product_models(0, 0.3, DEV)
    oR, oC = oracle_models(0, 0.3)
    ooff = torch.from_numpy(off)
    GJ, JG, JJ = EDGE_TYPES

    def rollout():
        X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
        return X, GrainRollout(R, Cm, X, EI, EA, 6, use_graph=True, refresh_centres=True, domain_factor=10.0,
                               domain_offset=ooff)

    # (1) the whole trajectory twice: bit-reproducible and finite
    Xa, ra = rollout()
    Xb, rb = rollout()
    for seg in range(10):
        ra.run(50)
        rb.run(50)
        for nt in Xa:
            assert bool(torch.isfinite(Xa[nt]).all()), f"step {50 * (seg + 1)}: x {nt} not finite"
            assert torch.equal(Xa[nt], Xb[nt]), f"step {50 * (seg + 1)}: x {nt} differs between two identical runs"
    for k in ra.pred:
        assert torch.equal(ra.pred[k], rb.pred[k]) and bool(torch.isfinite(ra.pred[k]).all()), k
    for et in EDGE_TYPES:
        assert torch.equal(ra.edge_attr_dict()[et], rb.edge_attr_dict()[et]), et
    assert not ra.range_exceeded()

    # (2) one step from identical inputs at points along the trajectory
    checks = [0, 1, 2, 5, 10, 20, 40, 60, 80, 100, 130, 160, 200, 240, 280, 320, 329, 330, 331, 360, 400, 440, 470, 490,
              498, 499]
    X, ro = rollout()
    oEI = tt(ei)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))
    worst, flipped_total, done = 0.0, 0, 0
    try:
        for step in checks:
            ro.run(step - done)
            torch.cuda.synchronize()
            oX = {nt: X[nt].cpu().clone() for nt in X}
            oEA = {et: ro.edge_attr_dict()[et].cpu().clone() for et in EDGE_TYPES}
            pred = {k: v.clone() for k, v in ro.step().items()}
            done = step + 1
            opred, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6, centres=(10.0, ooff))
            for k in ("joint", "grain", "grain_area", "edge_event", "edge"):
                worst = max(worst, assert_close(pred[k], opred[k], f"cfg3 step {step} {k}"))
            assert_close(X["joint"], oX["joint"], f"cfg3 step {step} x joint")
            assert_close(X["grain"][:, 2:], oX["grain"][:, 2:], f"cfg3 step {step} x grain[2:]")
            d = (X["grain"][:, :2].cpu() - oX["grain"][:, :2]).abs()
            assert float(torch.minimum(d, 1 - d).max()) < 1e-4, f"cfg3 step {step}: grain centres"
            flipped = (d > 0.5).any(1)                      # centres that landed on the other side of the fold
            flipped_total += int(flipped.sum())
            # edge lengths are differences of coordinates: along a 500-step random-weight trajectory some edges shrink to
            # ~1e-3 of the box, where the coordinates' own rounding (1e-7 of a folded coordinate ~1) is more than 1e-4 OF THE
            # LENGTH -- the element-wise floor of these comparisons is therefore 5e-6 absolute (the coordinates themselves
            # are held to the 1e-4 contract above; the per-tensor bound on the lengths stays 1e-4 of the longest)
            hea = ro.edge_attr_dict()
            assert_close(hea[JJ], oEA[JJ], f"cfg3 step {step} edge_attr {JJ}", atol=5e-6)
            for et, grain_row in ((GJ, 0), (JG, 1)):
                keep = ~flipped[oEI[et][grain_row]]
                assert_close(hea[et].cpu().view(-1)[keep], oEA[et].view(-1)[keep], f"cfg3 step {step} edge_attr {et}", atol=5e-6)
    finally:
        torch.set_num_threads(threads)
    assert flipped_total <= 5, f"{flipped_total} grain centres on the other side of the fold over {len(checks)} checked steps"
    print(f"cfg3 500-step rollout: {len(checks)} one-step checks along the trajectory, worst per-tensor error {worst:.2e}, "
          f"{flipped_total} centre(s) across the fold")


# ---------------------------------------------------------------------------------------
# on-device grain-centre refresh (graph.update() + test.py:556-559)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("use_graph", [False, True])
@torch.no_grad()
def test_rollout_with_grain_centres_golden(use_graph):
    """Against vectors the reference's own graph_trajectory.GNN_update produced."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = load_graph("40")
    g = golden("cfg1_centres")
    R, Cm = product_models(10020, 1.0, DEV)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    ro = GrainRollout(R, Cm, X, EI, EA, 6, use_graph=use_graph, refresh_centres=True)
    for step in range(1, 4):
        ro.step()
        if step in (1, 3):
            for nt in x:
                assert_close(X[nt], g[f"step{step}_x_{nt}"], f"centres step{step} x {nt}")
            for et in EDGE_TYPES:
                assert_close(ro.edge_attr_dict()[et], g[f"step{step}_ea_{etk(et)}"],
                             f"centres step{step} edge_attr {et}")


@torch.no_grad()
def test_grain_centres_folded_domain_rollout_against_oracle():
    """cfg2 (x3 folded domain): the centre refresh goes through the global frame
    (test.py:474) and folds back with (c * 3) % 1 (test.py:558-559)."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = load_graph("120")
    oX, oEA, oEI = tt(x), tt(ea), tt(ei)
    off, _ = oracle.scale_feature_patchs(3.0, oX, oEA)
    X = {k: v.clone().to(DEV) for k, v in oX.items()}
    EA = {k: v.clone().to(DEV) for k, v in oEA.items()}
    R, Cm = product_models(0, 1.0, DEV)
    oR, oC = oracle_models(0, 1.0)
    ro = GrainRollout(R, Cm, X, tt(ei, DEV), EA, 6, refresh_centres=True, domain_factor=3.0,
                      domain_offset=off)
    for step in range(5):
        ro.step()
        _, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6, centres=(3.0, off))
    assert_close(X["joint"], oX["joint"], "folded centres x joint")
    assert_close(X["grain"][:, 2:], oX["grain"][:, 2:], "folded centres x grain[2:]")
    d = (X["grain"][:, :2].cpu() - oX["grain"][:, :2]).abs()
    assert float(torch.minimum(d, 1 - d).max()) < 1e-4   # frac() may land on either side of 0/1
    for et in EDGE_TYPES:
        assert_close(ro.edge_attr_dict()[et], oEA[et], f"folded centres edge_attr {et}")


@torch.no_grad()
def test_grain_centres_full_size_and_edge_cases():
    """cfg3 size against the oracle, COO-order independence, straddling grains, rows with
    0 / 1 junctions keep their centre, invalid arguments are refused."""
    be = backend()
    x, ei, _ = synthetic.honeycomb(100, 10, 0)
    rs = np.random.RandomState(4)
    xj = x["joint"].copy()
    xj[:, :2] = (xj[:, :2] + rs.normal(0, 2e-3, (xj.shape[0], 2)) + [0.4031, 0.7717]) % 1
    Xj, Xg = torch.from_numpy(xj).to(DEV), torch.from_numpy(x["grain"].copy()).to(DEV)
    n_g, n_j = Xg.size(0), Xj.size(0)
    csr = be.build_csr(torch.from_numpy(ei[JG]).to(DEV), n_j, n_g)
    be.grain_centres(csr, Xj, Xg)
    ref = oracle.grain_centres(torch.from_numpy(xj[:, :2]), torch.from_numpy(ei[GJ]), n_g)
    assert_close(Xg[:, :2], ref.float(), "cfg3 grain centres", 1e-6)
    assert torch.equal(Xg[:, 2:].cpu(), torch.from_numpy(x["grain"][:, 2:]))
    perm = rs.permutation(ei[JG].shape[1])
    Xg2 = torch.from_numpy(x["grain"].copy()).to(DEV)
    be.grain_centres(be.build_csr(torch.from_numpy(ei[JG][:, perm]).to(DEV), n_j, n_g), Xj, Xg2)
    assert_close(Xg2[:, :2], Xg[:, :2], "cfg3 grain centres, permuted COO", 1e-6)
    # grains with 0 and 1 junctions keep their centre (graph_datastruct.py:685)
    e = torch.tensor([[0, 1, 2, 3], [0, 0, 0, 2]], dtype=torch.int64, device=DEV)
    xj3 = torch.tensor([[0.98, 0.5], [0.02, 0.52], [0.01, 0.47], [0.3, 0.3]], device=DEV)
    xg3 = torch.full((3, 4), 7.0, device=DEV)
    be.grain_centres(be.build_csr(e, 4, 3), xj3, xg3)
    got = xg3.cpu()
    assert torch.equal(got[1:], torch.full((2, 4), 7.0))
    assert_close(got[0, :2], torch.tensor([(0.98 + 1.02 + 1.01) / 3, (0.5 + 0.52 + 0.47) / 3]),
                 "straddling grain", 1e-6)
    with pytest.raises(_lib.GGNNError):
        be.grain_centres(csr, Xj, Xg, 3.0, None)
    with pytest.raises(_lib.GGNNError):
        be.grain_centres(csr, Xj.cpu(), Xg)


# ---------------------------------------------------------------------------------------
# event-driven rollout (device steps + host topology update + CSR rebuild)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("use_graph", [False, True])
@torch.no_grad()
def test_event_rollout_reproduces_reference_trajectory(use_graph):
    """Free-running rollout of the 40 um graph with seeded weights, events ON: steps 1-2 are quiet,
    step 3 eliminates 22 grains, step 4 another 75 (9 of them force-eliminated).  The reference's
    own loop (forward, Rmodel.update, Cmodel.update, GNN_update, centre + edge refresh) produced
    tests/golden/golden_cfg1_events.npz; the edge lists must come out identical, column for
    column, and the refreshed grain centres within the fp32 tolerance."""
    import os
    from helpers import GOLDEN
    from graingraphnn_amd import GrainRollout
    ev = np.load(os.path.join(GOLDEN, "golden_cfg1_events.npz"))
    x, ei, ea = load_graph("40")
    R, Cm = product_models(10020, 1.0, DEV)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    ro = GrainRollout(R, Cm, X, EI, EA, 6, use_graph=use_graph, refresh_centres=True)
    ro.enable_events({"grain": np.ones((118, 1)), "joint": np.ones((236, 1))}, 1e-4, 0.6)
    for step in range(1, 5):
        n_edges_before = ro.edge_index[JJ].size(1)
        pred, events, switches = ro.step_events()
        # the classifier outputs returned are THIS step's predictions on the PRE-event edge list
        # (test.py:383-426 keeps `pred` across Cmodel.update), never fresh buffers of the new size
        assert pred["edge_event"].numel() == n_edges_before and pred["edge"].shape == (n_edges_before, 2)
        assert bool(torch.isfinite(pred["edge_event"]).all()) and float(pred["edge"].abs().max()) <= 1.0
        if step < 3:
            assert len(events) == 0 and len(switches) == 0
            continue
        assert ro.edge_index[JJ].size(1) < n_edges_before
        name = f"mass{step}"
        assert events.tolist() == ev[name + "__out_grain_event"].tolist()
        for et in EDGE_TYPES:
            assert np.array_equal(ro.edge_index[et].cpu().numpy(), ev[name + "__out_ei_" + etk(et)]), (step, et)
        assert np.array_equal(ro.mask["grain"], ev[name + "__out_mask_grain"])
        assert np.array_equal(ro.mask["joint"], ev[name + "__out_mask_joint"])
        assert_close(X["grain"], ev[name + "__next_x_grain"], f"step {step} grains after the centre refresh")
        live = ro.mask["joint"][:, 0] > 0
        assert_close(X["joint"][torch.from_numpy(live).to(DEV)], ev[name + "__out_x_joint"][live],
                     f"step {step} live junctions")
    assert ro.edge_index[JJ].size(1) == 156 and int(ro.mask["grain"].sum()) == 26
    # the rollout keeps running on the shrunken graph
    ro.step_events()
    assert bool(torch.isfinite(X["joint"]).all())


@torch.no_grad()
def test_segment_graphs_survive_topological_events(monkeypatch):
    """SURVEY 8 f-2, "keep the device rollout running between events": with events enabled the edge lists, the CSR tables and
    every per-edge buffer live in allocations of the initial lists' size and are rewritten IN PLACE by an event; the hipGraphs
    of step_events()' two segments are captured once and replayed across events (the per-edge kernels read the number of edges
    from device memory: ggnn_prepare_edge.E_dev).  The reference's own event trajectory (22 + 75 eliminations in steps 3-4): the
    same graph objects before and after, state and lists bit for bit those of a rollout that rebuilds everything per event and
    launches eagerly (GGNN_EVENT_GRAPHS=0), the lists shrinking inside unchanged storage."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = load_graph("40")
    R, Cm = product_models(10020, 1.0, DEV)
    mask = {"grain": np.ones((118, 1)), "joint": np.ones((236, 1))}

    def make(in_place):
        monkeypatch.setenv("GGNN_EVENT_GRAPHS", "1" if in_place else "0")
        X = tt(x, DEV)
        ro = GrainRollout(R, Cm, X, tt(ei, DEV), tt(ea, DEV), 6, use_graph=in_place, refresh_centres=True)
        ro.enable_events(mask, 1e-4, 0.6)
        return ro, X
    ra, Xa = make(True)
    rb, Xb = make(False)
    assert ra._cap is not None and rb._cap is None
    base = {et: ra.edge_index[et].data_ptr() for et in EDGE_TYPES}
    graphs, sizes = None, []
    for step in range(1, 6):
        if step == 5:
            # a caller's own topology ends the arrangement (buffers of its own again, the graphs go); the loop goes on
            ra._set_topology({et: v.clone() for et, v in ra.edge_index.items()},
                             {et: v.clone().view(-1, 1) for et, v in ra.edge_attr.items()})
            assert ra._cap is None and ra._graph_fwd is None and ra._graph_ref is None
        pa, ea_, sa = ra.step_events()
        pb, eb_, sb = rb.step_events()
        assert ea_.tolist() == eb_.tolist() and np.array_equal(sa, sb), step
        for k in ("joint", "grain", "grain_area", "edge_event", "edge"):
            assert torch.equal(pa[k], pb[k]), (step, k)
        for nt in Xa:
            assert torch.equal(Xa[nt], Xb[nt]), (step, nt)
        for et in EDGE_TYPES:
            assert torch.equal(ra.edge_index[et], rb.edge_index[et]), (step, et)
            assert torch.equal(ra.edge_attr[et], rb.edge_attr[et]), (step, et)
            assert step == 5 or ra.edge_index[et].data_ptr() == base[et], (step, et)   # in place: the storage never moves
        if step == 1:
            graphs = (ra._graph_fwd, ra._graph_ref)
            assert graphs[0] is not None and graphs[1] is not None
        if step < 5:
            assert (ra._graph_fwd, ra._graph_ref) == graphs, step       # captured once, never dropped
        sizes.append(ra.edge_index[JJ].size(1))
    assert sizes == [708, 708, 576, 156, 36]                            # steps 3-5 removed edges


@torch.no_grad()
def test_step_events_and_run_events_mix_on_one_rollout():
    """step_events()' segment graphs hold the addresses of the buffers that were current at their capture; run_events() leaves
    its own slots' predictions and the other set of edge lengths / records current.  Alternating the two on ONE rollout
    (quiet steps, the reference trajectory's two eventful steps, a step behind them) gives, bit for bit, what step_events()
    alone gives."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = load_graph("40")
    R, Cm = product_models(10020, 1.0, DEV)
    mask = {"grain": np.ones((118, 1)), "joint": np.ones((236, 1))}

    def make():
        X = tt(x, DEV)
        ro = GrainRollout(R, Cm, X, tt(ei, DEV), tt(ea, DEV), 6, use_graph=True, refresh_centres=True, joint_launches=False,
                          concurrent=True)
        ro.enable_events(mask, 1e-4, 0.6)
        return ro, X
    ra, Xa = make()
    rb, Xb = make()
    plan = [("step", 1), ("run", 1), ("step", 1), ("run", 1), ("step", 1)]   # steps 3 and 4 are the eventful ones
    for how, n in plan:
        if how == "run":
            ev_a, sw_a = ra.run_events(n)
        else:
            out = [ra.step_events()[1:] for _ in range(n)]
            ev_a, sw_a = [o[0] for o in out], [o[1] for o in out]
        out = [rb.step_events()[1:] for _ in range(n)]
        for k in range(n):
            assert ev_a[k].tolist() == out[k][0].tolist() and np.array_equal(sw_a[k], out[k][1]), (how, k)
        for nt in Xa:
            assert torch.equal(Xa[nt], Xb[nt]), (how, nt)
        for et in EDGE_TYPES:
            assert torch.equal(ra.edge_index[et], rb.edge_index[et]) and torch.equal(ra.edge_attr[et], rb.edge_attr[et]), (how, et)
    assert ra.edge_index[JJ].size(1) == 36


@torch.no_grad()
def test_step_events_loop_follows_new_weights_between_its_steps():
    """A step_events() loop checks a SAMPLE of the parameter tensors per step (the full walk is a seventh of an eventful
    step: every 16th step only): load_state_dict between two steps must still be seen at once -- the loop continues bit for
    bit like a rollout built on the new weights from the same state."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = load_graph("40")
    R, Cm = product_models(5, 0.3, DEV)
    R2, Cm2 = product_models(9, 0.3, DEV)
    mask = {"grain": np.ones((118, 1)), "joint": np.ones((236, 1))}
    X = tt(x, DEV)
    ro = GrainRollout(R, Cm, X, tt(ei, DEV), tt(ea, DEV), 6, use_graph=True, refresh_centres=True)
    ro.enable_events(mask, -1.0, 0.999999)
    for _ in range(3):                       # (steps_done 1..3: the next check is a sampled one)
        ro.step_events()
    X2 = {k: v.clone() for k, v in X.items()}
    ref = GrainRollout(R2, Cm2, X2, {et: v.clone() for et, v in ro.edge_index.items()},
                       {et: v.clone().view(-1, 1) for et, v in ro.edge_attr.items()}, 6, use_graph=False, refresh_centres=True)
    ref.enable_events(mask, -1.0, 0.999999)
    R.load_state_dict(R2.state_dict())
    Cm.load_state_dict(Cm2.state_dict())
    for _ in range(2):
        ro.step_events()
        ref.step_events()
    for nt in X:
        assert torch.equal(X[nt], X2[nt]), nt


@torch.no_grad()
def test_speculative_blocks_keep_their_graphs_across_events():
    """run_events() on a topology that changes in place: the ring of slots and the hipGraphs of its blocks are the SAME objects
    before and after the trajectory's eventful steps (round 5 and the first half of round 6 rebuilt the ring and re-captured
    every block after an event), blocks captured before an event are replayed behind it on the shrunken lists, and the
    result is step_events()' bit for bit.  Two passes of 12 steps (events at 3, 4, 5 of the first), weights x0.5 so that the
    second pass is quiet and long enough to reuse 2-, 4- and 8-step blocks."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = load_graph("40")
    R, Cm = product_models(10020, 1.0, DEV)
    mask = {"grain": np.ones((118, 1)), "joint": np.ones((236, 1))}

    def make():
        X = tt(x, DEV)
        ro = GrainRollout(R, Cm, X, tt(ei, DEV), tt(ea, DEV), 6, use_graph=True, refresh_centres=True, joint_launches=False,
                          concurrent=True)
        ro.enable_events(mask, 1e-4, 0.6)
        return ro, X
    ra, Xa = make()
    rb, Xb = make()
    ra.run_events(2)                                   # quiet: a 1-step and (eagerly) ... blocks; the ring exists now
    S, graphs_before = ra._spec, dict(ra._spec["graphs"])
    assert S.get("in_place") and ra._cap is not None
    ev_a, sw_a = ra.run_events(3)                      # steps 3-5: 22, 75 and 27 grains
    assert [len(e) for e in ev_a] == [22, 75, 27]
    assert ra._spec is S and all(S["graphs"].get(k) is g for k, g in graphs_before.items())
    for _ in range(5):
        rb.step_events()
    for nt in Xa:
        assert torch.equal(Xa[nt], Xb[nt]), nt
    ra.area_threshold = rb.area_threshold = -1.0      # the rest of the trajectory kept quiet: blocks of 1, 2, 4, 8 steps
    ra.edge_threshold = rb.edge_threshold = 0.999999
    ra._logit_trigger = rb._logit_trigger = float(np.log(0.999999 / (1.0 - 0.999999)) - 1e-4)
    ra.run_events(15)
    captured = dict(S["graphs"])
    assert ra._spec is S and len(captured) >= 2
    ra.run_events(16)                                  # the graphs captured so far keep serving (new block shapes may join them)
    assert all(S["graphs"].get(k) is g for k, g in captured.items())
    for _ in range(31):
        rb.step_events()
    for nt in Xa:
        assert torch.equal(Xa[nt], Xb[nt]), nt
    for et in EDGE_TYPES:
        assert torch.equal(ra.edge_index[et], rb.edge_index[et]) and torch.equal(ra.edge_attr[et], rb.edge_attr[et]), et


@pytest.mark.parametrize("use_graph,chunks", [(True, (5,)), (True, (3, 2)), (False, (1, 4))])
@torch.no_grad()
def test_speculative_event_loop_equals_step_events(use_graph, chunks):
    """GrainRollout.run_events (every step enqueued as if it had no events, the counts of step k - 1 read after step k is
    enqueued, the step behind an eventful one discarded and x / centres / predictions taken back) against step_events,
    which stops at every step: the seeded 40 um trajectory whose steps 3 and 4 eliminate 22 and 75 grains -- an eventful
    step with a discarded step behind it, one that is the last of its call, quiet steps before and after -- must give
    the same events, edge lists, masks and the same state bit for bit."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = load_graph("40")
    R, Cm = product_models(10020, 1.0, DEV)
    mask = {"grain": np.ones((118, 1)), "joint": np.ones((236, 1))}
    kw = dict(use_graph=use_graph, refresh_centres=True, joint_launches=False, concurrent=True)
    Xa, Xb = tt(x, DEV), tt(x, DEV)
    ra = GrainRollout(R, Cm, Xa, tt(ei, DEV), tt(ea, DEV), 6, **kw)
    rb = GrainRollout(R, Cm, Xb, tt(ei, DEV), tt(ea, DEV), 6, **kw)
    ra.enable_events(mask, 1e-4, 0.6)
    rb.enable_events(mask, 1e-4, 0.6)
    ev_a, sw_a = [], []
    for _ in range(sum(chunks)):
        _, e, sw = ra.step_events()
        ev_a.append(e)
        sw_a.append(sw)
    ev_b, sw_b = [], []
    for n in chunks:
        e, sw = rb.run_events(n)
        ev_b += e
        sw_b += sw
    torch.cuda.synchronize()
    assert [len(e) for e in ev_a][:4] == [0, 0, 22, 75]
    for k in range(sum(chunks)):
        assert np.array_equal(ev_a[k], ev_b[k]) and np.array_equal(sw_a[k], sw_b[k]), k
    for et in EDGE_TYPES:
        assert torch.equal(ra.edge_index[et], rb.edge_index[et]), et
        assert torch.equal(ra.edge_attr_dict()[et], rb.edge_attr_dict()[et]), et
    assert np.array_equal(ra.mask["grain"], rb.mask["grain"]) and np.array_equal(ra.mask["joint"], rb.mask["joint"])
    for nt in Xa:
        assert torch.equal(Xa[nt], Xb[nt]), nt
    last_quiet = len(ev_a[-1]) == 0 and len(sw_a[-1]) == 0
    for k in ra.pred:   # (after an eventful step the per-edge buffers are fresh ones of the new size: nothing to compare)
        if last_quiet or k not in ("edge_event", "edge"):
            assert torch.equal(ra.pred[k], rb.pred[k]), (k, [len(e) for e in ev_a])
    assert ra.steps_done == rb.steps_done == sum(chunks)


@torch.no_grad()
def test_speculative_event_loop_restores_the_z_clamp_flag_of_the_eventful_step():
    """test.py:405's flag is written by every step's Rmodel.update and read by its refresh.  Grain 0 is put
    3.5 steps below zmax, so the flag is raised from the fourth step on; the third step (22 grains vanish on the seeded 40 um
    trajectory) has speculative steps enqueued behind it that raise the flag before the host sees the events.  The
    refresh that follows the rewiring must read the THIRD step's flag (not raised): otherwise every node's z is clamped one
    step early and the trajectory leaves step_events'."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = load_graph("40")
    R, Cm = product_models(10020, 1.0, DEV)
    mask = {"grain": np.ones((118, 1)), "joint": np.ones((236, 1))}
    kw = dict(use_graph=True, refresh_centres=True, joint_launches=False, concurrent=True)
    dz, zmax = np.float32(6 / 121), np.float32(120 / 121)
    x = {k: v.copy() for k, v in x.items()}
    x["grain"][0, 2] = zmax - np.float32(3.5) * dz
    Xa, Xb = tt(x, DEV), tt(x, DEV)
    ra = GrainRollout(R, Cm, Xa, tt(ei, DEV), tt(ea, DEV), 6, **kw)
    rb = GrainRollout(R, Cm, Xb, tt(ei, DEV), tt(ea, DEV), 6, **kw)
    ra.enable_events(mask, 1e-4, 0.6)
    rb.enable_events(mask, 1e-4, 0.6)
    ev_a, z_a = [], []
    for _ in range(4):   # (the clamp moves every node's z to 0.99 at once: random weights then eliminate most grains)
        ev_a.append(ra.step_events()[1])
        z_a.append(float(Xa["joint"][:, 2].max()))
    ev_b, _ = rb.run_events(4)
    torch.cuda.synchronize()
    assert any(len(e) for e in ev_a[:3]), [len(e) for e in ev_a]     # an eventful step before the crossing ...
    assert z_a[2] < float(zmax) and z_a[3] == float(zmax), z_a       # ... and the clamp from the fourth step on
    for k in range(4):
        assert np.array_equal(ev_a[k], ev_b[k]), k
    for nt in Xa:
        assert torch.equal(Xa[nt], Xb[nt]), nt
    for et in EDGE_TYPES:
        assert torch.equal(ra.edge_index[et], rb.edge_index[et]) and torch.equal(ra.edge_attr_dict()[et], rb.edge_attr_dict()[et])


@torch.no_grad()
def test_detect_events_counts():
    be = backend()
    rs = np.random.RandomState(2)
    area = torch.from_numpy(rs.uniform(0, 2e-4, 5000).astype(np.float32)).to(DEV)
    live = torch.from_numpy((rs.uniform(size=5000) > 0.3).astype(np.int32)).to(DEV)
    logit = torch.from_numpy(rs.normal(0, 1, 7001).astype(np.float32)).to(DEV)
    ei = torch.from_numpy(rs.randint(0, 3000, (2, 7001))).to(DEV)
    flags = torch.full((2,), 77, dtype=torch.int32, device=DEV)
    be.detect_events(area, live, 1e-4, logit, ei, 0.405, flags)
    assert int(flags[0]) == int(((live > 0) & (area < 1e-4)).sum())
    assert int(flags[1]) == int(((logit > 0.405) & (ei[0] < ei[1])).sum())


# ---------------------------------------------------------------------------------------
# BASELINE config 4: independent trajectories batched as one disjoint-union graph
# ---------------------------------------------------------------------------------------
@torch.no_grad()
def test_cfg4_batched_trajectories_match_individual_rollouts():
    from graingraphnn_amd import GrainRollout
    from graingraphnn_amd.dist import rollout_trajectories
    x, ei, ea = load_graph("40")
    graphs = [(synthetic.perturbed_copy(x, 1e-3, 1000 + t), ei, ea) for t in range(8)]
    R, Cm = product_models(10020, 1.0, DEV)
    got = rollout_trajectories(R, Cm, graphs, 6, 3, 0, 1, DEV)
    assert got["joint_xy"].shape == (8, 236, 2) and got["grain_area_v"].shape == (8, 118, 2)
    oR, oC = oracle_models(10020, 1.0)
    for t in (0, 3, 7):
        # (1) same bits as rolling the trajectory out alone: rows are independent, CSR order and
        # the MFMA k-order do not depend on where a row sits in the batch
        X, EI, EA = tt(graphs[t][0], DEV), tt(ei, DEV), tt(ea, DEV)
        GrainRollout(R, Cm, X, EI, EA, 6).run(3)
        assert torch.equal(got["joint_xy"][t], X["joint"][:, :2]), f"trajectory {t} joints"
        assert torch.equal(got["grain_area_v"][t], X["grain"][:, 3:5]), f"trajectory {t} grains"
        # (2) and the oracle within tolerance
        oX, oEI, oEA = tt(graphs[t][0]), tt(ei), tt(ea)
        for _ in range(3):
            _, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6)
        assert_close(got["joint_xy"][t], oX["joint"][:, :2], f"cfg4 trajectory {t} joint xy")
        assert_close(got["grain_area_v"][t], oX["grain"][:, 3:5], f"cfg4 trajectory {t} grain area/extraV")


@torch.no_grad()
def test_cfg4_as_baseline_states_it_64_trajectories_20_steps():
    """BASELINE config 4 at its stated size: 64 perturbed 40 um trajectories x 20 steps.
      * one GPU holding all 64 as ONE disjoint-union graph (15 104 junctions: the FUSED decoder cell is the plan) -- every
        trajectory bit-equal to its own rollout under the same plan, seven of them within 1e-4 of the oracle;
      * the shard of one of eight ranks (trajectories t = r mod 8: 1 888 junctions, the THREE-KERNEL decoder plan) at 20
        steps -- bit-equal to the individual rollouts under that plan and within tolerance of the fused results."""
    from graingraphnn_amd import GrainRollout
    from graingraphnn_amd.dist import rollout_trajectories, shard_trajectories
    be = backend()
    x, ei, ea = load_graph("40")
    graphs = [(synthetic.perturbed_copy(x, 1e-3, 1000 + t), ei, ea) for t in range(64)]
    R, Cm = product_models(10020, 1.0, DEV)
    if be.fused_decoder is not True or not 8 * 236 < be.fused_decoder_min_joints <= 64 * 236:
        pytest.skip("the decoder plan is forced by the environment (GGNN_DEC / GGNN_GEMM): the size rule is what this test is about")
    got = rollout_trajectories(R, Cm, graphs, 6, 20, 0, 1, DEV)
    assert got["joint_xy"].shape == (64, 236, 2) and got["grain_area_v"].shape == (64, 118, 2)
    assert bool(torch.isfinite(got["joint_xy"]).all()) and bool(torch.isfinite(got["grain_area_v"]).all())
    mine = shard_trajectories(64, 3, 8)
    assert mine == list(range(3, 64, 8))
    shard = rollout_trajectories(R, Cm, [graphs[t] for t in mine], 6, 20, 0, 1, DEV)

    def alone(t, min_joints):
        keep = be.fused_decoder_min_joints
        be.fused_decoder_min_joints = min_joints
        try:
            X = tt(graphs[t][0], DEV)
            GrainRollout(R, Cm, X, tt(ei, DEV), tt(ea, DEV), 6).run(20)
            return X
        finally:
            be.fused_decoder_min_joints = keep
    for t in range(64):
        X = alone(t, 0)                                     # the fused decoder cell, as in the 64-trajectory union
        assert torch.equal(got["joint_xy"][t], X["joint"][:, :2]), f"trajectory {t} joints (fused plan)"
        assert torch.equal(got["grain_area_v"][t], X["grain"][:, 3:5]), f"trajectory {t} grains (fused plan)"
    for i, t in enumerate(mine):
        X = alone(t, 10 ** 9)                               # the three-kernel plan, as in a rank's 8-trajectory union
        assert torch.equal(shard["joint_xy"][i], X["joint"][:, :2]), f"trajectory {t} joints (split plan)"
        assert torch.equal(shard["grain_area_v"][i], X["grain"][:, 3:5]), f"trajectory {t} grains (split plan)"
        assert_close(shard["joint_xy"][i], got["joint_xy"][t], f"trajectory {t}: split plan vs fused plan, 20 steps")
    oR, oC = oracle_models(10020, 1.0)
    for t in (0, 9, 18, 27, 36, 45, 63):
        oX, oEI, oEA = tt(graphs[t][0]), tt(ei), tt(ea)
        for _ in range(20):
            _, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6)
        assert_close(got["joint_xy"][t], oX["joint"][:, :2], f"cfg4 x 20 steps, trajectory {t} joint xy")
        assert_close(got["grain_area_v"][t], oX["grain"][:, 3:5], f"cfg4 x 20 steps, trajectory {t} grain area/extraV")


# ---------------------------------------------------------------------------------------
# boundary behaviour: checkpoints, cache invalidation, topology changes, strided inputs
# ---------------------------------------------------------------------------------------
@torch.no_grad()
def test_reference_layout_checkpoint_round_trip(tmp_path):
    """test.py:177-184: construct, load_state_dict(torch.load(.pt, map_location='cpu')), .cuda()."""
    from graingraphnn_amd.models import GrainNN_classifier, GrainNN_regressor
    x, ei, ea = load_graph("40")
    oR, oC = oracle_models(77, 1.0)
    torch.save(oR.state_dict(), tmp_path / "regressor0.pt")
    torch.save(oC.state_dict(), tmp_path / "classifier1.pt")
    hp = synthetic.default_hyper("cuda")
    R = GrainNN_regressor(hp)
    R.load_state_dict(torch.load(tmp_path / "regressor0.pt", map_location=torch.device("cpu")))
    R.eval()
    Cm = GrainNN_classifier(hp, R)
    Cm.load_state_dict(torch.load(tmp_path / "classifier1.pt", map_location="cpu"))
    Cm.eval()
    R.cuda()
    Cm.cuda()
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    yr, yc = R(X, EI, EA), Cm(X, EI, EA)
    oyr, oyc = oR(tt(x), tt(ei), tt(ea)), oC(tt(x), tt(ei), tt(ea))
    for k in ("joint", "grain", "grain_area"):
        assert_close(yr[k], oyr[k], f"checkpoint regressor {k}")
    for k in ("edge_event", "edge"):
        assert_close(yc[k], oyc[k], f"checkpoint classifier {k}")


@torch.no_grad()
def test_packed_weights_follow_parameter_updates():
    x, ei, ea = load_graph("40")
    R, _ = product_models(5, 1.0, DEV)
    oR, _ = oracle_models(5, 1.0)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    y0 = R(X, EI, EA)["joint"].clone()
    name = "gclstm_decoder.cell_list.0.conv_c.convs.joint__connect__joint.lin_value.weight"
    dict(R.named_parameters())[name].mul_(1.5)       # in-place update (optimizer-style)
    dict(oR.named_parameters())[name].mul_(1.5)
    R.linear["joint"].bias.add_(0.25)
    oR.linear["joint"].bias.add_(0.25)
    y1 = R(X, EI, EA)
    assert rel_err(y1["joint"], y0) > 1e-3           # the cache did not serve stale weights
    oy = oR(tt(x), tt(ei), tt(ea))
    for k in ("joint", "grain", "grain_area"):
        assert_close(y1[k], oy[k], f"updated-weights regressor {k}")


@pytest.mark.parametrize("use_graph", [False, True])
@torch.no_grad()
def test_rollout_follows_parameter_updates(use_graph):
    """A GrainRollout packs the weights once and bakes their addresses into its hipGraphs; after
    load_state_dict / an optimizer-style in-place update it must re-pack (checked by run() and by
    refresh_weights()), not keep rolling out with the stale copy."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = load_graph("40")
    R, Cm = product_models(5, 1.0, DEV)
    R2, Cm2 = product_models(9, 1.0, DEV)
    ro = GrainRollout(R, Cm, tt(x, DEV), tt(ei, DEV), tt(ea, DEV), 6, use_graph=use_graph)
    ro.run(4)
    R.load_state_dict(R2.state_dict())
    Cm.load_state_dict(Cm2.state_dict())
    X = tt(x, DEV)
    ro.x["joint"].copy_(X["joint"]), ro.x["grain"].copy_(X["grain"])
    for et in EDGE_TYPES:
        ro.edge_attr[et].copy_(torch.from_numpy(ea[et]).view(-1))
    ro.run(4)
    ref = GrainRollout(R2, Cm2, X, tt(ei, DEV), tt(ea, DEV), 6, use_graph=False)
    ref.run(4)
    for nt in x:
        assert torch.equal(ro.x[nt], ref.x[nt]), nt
    name = "gclstm_decoder.cell_list.0.conv_c.convs.joint__connect__joint.lin_value.weight"
    dict(R.named_parameters())[name].mul_(1.5)
    before = ro.x["joint"].clone()
    ro.refresh_weights()
    ro.step()
    dict(R2.named_parameters())[name].mul_(1.5)
    ref.refresh_weights()
    ref.step()
    assert torch.equal(ro.x["joint"], ref.x["joint"]) and not torch.equal(ro.x["joint"], before)


@torch.no_grad()
def test_topology_change_rebuilds_csr():
    """Cmodel.update replaces / edits edge_index after an event (models.py:841-845)."""
    x, ei, ea = load_graph("40")
    R, Cm = product_models(6, 1.0, DEV)
    oR, oC = oracle_models(6, 1.0)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    R(X, EI, EA)
    # (1) new tensors with 40 edges dropped per type
    keep = {et: np.sort(np.random.RandomState(3).permutation(ei[et].shape[1])[40:]) for et in EDGE_TYPES}
    ei2 = {et: np.ascontiguousarray(ei[et][:, keep[et]]) for et in EDGE_TYPES}
    ea2 = {et: ea[et][keep[et]] for et in EDGE_TYPES}
    EI2, EA2 = tt(ei2, DEV), tt(ea2, DEV)
    yr, yc = R(X, EI2, EA2), Cm(X, EI2, EA2)
    oyr, oyc = oR(tt(x), tt(ei2), tt(ea2)), oC(tt(x), tt(ei2), tt(ea2))
    for k in ("joint", "grain", "grain_area"):
        assert_close(yr[k], oyr[k], f"pruned-topology regressor {k}")
    assert yc["edge_event"].shape[0] == ei2[JJ].shape[1]
    assert_close(yc["edge_event"], oyc["edge_event"], "pruned-topology classifier edge_event")
    # (2) in-place edit of an edge_index tensor (same storage, bumped version): swap two sources
    EI2[JJ][0, :2] = EI2[JJ][0, :2].flip(0)
    ei3 = {et: EI2[et].cpu().numpy() for et in EDGE_TYPES}
    yr = R(X, EI2, EA2)
    oyr = oR(tt(x), tt(ei3), tt(ea2))
    for k in ("joint", "grain", "grain_area"):
        assert_close(yr[k], oyr[k], f"edited-topology regressor {k}")
    # (3) an out-of-range index is rejected, not dereferenced
    bad = {et: v.clone() for et, v in EI2.items()}
    bad[GJ][0, 0] = 10 ** 6
    with pytest.raises(IndexError):
        R(X, bad, EA2)


@torch.no_grad()
def test_row_strided_inputs():
    """x_dict tensors that are column slices of wider buffers (row stride > F) are accepted."""
    x, ei, ea = load_graph("40")
    R, Cm = product_models(8, 1.0, DEV)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    wide = {nt: torch.full((X[nt].size(0), X[nt].size(1) + 5), float("nan"), device=DEV) for nt in X}
    Xs = {}
    for nt in X:
        wide[nt][:, : X[nt].size(1)] = X[nt]
        Xs[nt] = wide[nt][:, : X[nt].size(1)]
        assert not Xs[nt].is_contiguous()
    ya, yb = R(X, EI, EA), R(Xs, EI, EA)
    for k in ("joint", "grain", "grain_area"):
        assert torch.equal(ya[k], yb[k])
    ca, cb = Cm(X, EI, EA), Cm(Xs, EI, EA)
    assert torch.equal(ca["edge_event"], cb["edge_event"])


@pytest.mark.parametrize("n_src,n_dst,E,F,hub", [(40, 30, 0, 8, 0), (1, 1, 1, 11, 0), (236, 118, 708, 8, 0),
                                                 (118, 236, 708, 11, 0), (70, 50, 400, 8, 37), (70, 50, 1300, 11, 900),
                                                 (10000, 20000, 60000, 11, 0), (20000, 10000, 60000, 8, 0)])
@torch.no_grad()
def test_encoder_sweep_on_matrix_cores_equals_the_gathering_sweep(n_src, n_dst, E, F, hub):
    """ggnn_period_gat_aggregate_enc_batch (values formed from the edge records by MFMA) against the
    h_src == NULL form of ggnn_period_gat_aggregate (values gathered from a projected source row):
    empty graph, single edge, fixture sizes, a hub of degree `hub`, rows without edges, cfg3 sizes."""
    from graingraphnn_amd.packing import value_fragments
    be = backend()
    rs = np.random.RandomState(E + F)
    src = rs.randint(0, max(n_src - 5, 1), size=E)
    dst = rs.randint(1 if n_dst > 1 else 0, n_dst, size=E)     # destination 0 has no in-edge
    dst[:hub] = 7
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64)).to(DEV)
    xs = torch.from_numpy(rs.uniform(0, 1, (n_src, F)).astype(np.float32)).to(DEV)
    xd = torch.from_numpy(rs.uniform(0, 1, (n_dst, 8)).astype(np.float32)).to(DEV)
    ea = torch.from_numpy(rs.uniform(0.01, 0.1, E).astype(np.float32)).to(DEV)
    G = 3
    Wv = [torch.from_numpy(rs.uniform(-1, 1, (96, F)).astype(np.float32)).to(DEV) for _ in range(G)]
    bv = [torch.from_numpy(rs.uniform(-1, 1, 96).astype(np.float32)).to(DEV) for _ in range(G)]
    p_dst = torch.from_numpy(rs.uniform(-2, 2, (n_dst, G * 16 + 16)).astype(np.float32)).to(DEV)
    p_dst.view(n_dst, -1)[:, 16:].view(n_dst, G, 16)[:, :, 11] = 0      # the score tail is 0 in the bias slot
    csr = be.build_csr(ei, n_src, n_dst)
    einfo = torch.zeros(E + 3, 20, device=DEV)
    be.edge_prepare([(csr, ea, xs, xd, einfo)])
    # gathering sweep: V0 per source node without the three reloc columns, which go through edge_params
    v0 = torch.cat([xs[:, 3:] @ w[:, 3:].t() + b for w, b in zip(Wv, bv)], 1).contiguous()
    ep = torch.stack([w[:, :3].t() for w in Wv]).contiguous()
    ref = torch.zeros(n_dst, G * 128, device=DEV)
    be.aggregate(csr, einfo, v0, p_dst, None, ep, ref, 0, 0, 16, 0, 128, 96, G)
    got = torch.full((n_dst, G * 128), 0.0, device=DEV)
    be.aggregate_enc_batch([(csr, einfo, p_dst, value_fragments(Wv, bv, F), got, 16, 0, 128, 96, G)])
    for g in range(G):
        assert_close(got[:, g * 128:g * 128 + 98], ref[:, g * 128:g * 128 + 98], f"encoder sweep gate {g}", 1e-5, 2e-6)
    again = torch.zeros_like(got)
    be.aggregate_enc_batch([(csr, einfo, p_dst, value_fragments(Wv, bv, F), again, 16, 0, 128, 96, G)])
    assert torch.equal(got, again)                                  # no atomics: bit-reproducible


def _enc_cell_problem(be, rs, n_dst, ins, hub=0, regular=False, F_dst=8, edges=None):
    """Random encoder-cell problem (ggnn_encoder_cell_batch): destination type with `F_dst` features, `ins` =
    [(n_src, F_src, E)] incoming edge types, random plain weights.  Returns the fused-call tuple (weight stream packed
    with packing's slice image), the split path's sweep tuples (ggnn_period_gat_aggregate_enc_batch, where the source
    has <= 11 features) and its gate-epilogue tuple, all from the same weights."""
    from graingraphnn_amd.packing import CELL_P3_CHANNEL, _plane_slices, _spread16, bf16_planes, value_fragments
    G, n_in = 3, len(ins)
    Ka = 96 * n_in + 4
    Kg = (Ka + 31) // 32 * 32
    ncols = 16 * G * n_in + 96 * G
    f = lambda *shape, lo=-1.0, hi=1.0: torch.from_numpy(rs.uniform(lo, hi, shape).astype(np.float32)).to(DEV)
    xd = f(n_dst, F_dst, lo=0.0)
    xs16 = torch.zeros(n_dst, 16, device=DEV)
    xs16[:, :F_dst], xs16[:, 12] = xd, 1.0
    S = torch.zeros(G, 96, 16, device=DEV)                          # summed skip + gate bias on the 16 feature slots
    S[:, :, :F_dst], S[:, :, 12] = f(G, 96, F_dst, lo=-0.5, hi=0.5), f(G, 96, lo=-0.5, hi=0.5)
    w2 = f(G, 96, Ka, lo=-0.2, hi=0.2)
    w2[:, :, 96 * n_in + 2 * n_in:] = 0
    p3 = torch.tensor(CELL_P3_CHANNEL, device=DEV)
    p_dst = torch.zeros(n_dst, ncols, device=DEV)                   # what the split path's projection would hold
    s_off = 16 * G * n_in
    p_dst[:, s_off:] = (xs16 @ S.reshape(G * 96, 16).t())
    fused_sweeps, split_sweeps, blocks = [], [], {}
    agg = torch.zeros(n_dst, G * Kg, device=DEV)
    for d, (n_src, F, E) in enumerate(ins):
        src = rs.randint(0, max(n_src - 5, 1), size=E)
        dst = rs.randint(1 if n_dst > 1 else 0, n_dst, size=E)     # destination 0 has no in-edge
        dst[:hub] = min(7, n_dst - 1)
        if regular:                                                 # every destination the same in-degree (timing tools)
            dst = rs.permutation(np.repeat(np.arange(n_dst), E // n_dst))
        if edges is not None:                                       # a given structure (timing tools): [2, E] per edge type
            src, dst = edges[d][0], edges[d][1]
        ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64)).to(DEV)
        xs = f(n_src, F, lo=0.0)
        ea = f(E, lo=0.01, hi=0.1)
        Wv = [f(96, F) for _ in range(G)]
        bv = [f(96) for _ in range(G)]
        csr = be.build_csr(ei, n_src, n_dst)
        einfo = torch.zeros(E + 3, 20, device=DEV)
        be.edge_prepare([(csr, ea, xs, xd, einfo)])
        fused_sweeps.append((csr, einfo))
        for g in range(G):
            T = torch.zeros(16, 16, device=DEV)                     # u4 = T [x | 1]: rows = record slots the tail meets
            T[:, :F_dst], T[:, 12] = f(16, F_dst, lo=-2.0, hi=2.0), f(16, lo=-2.0, hi=2.0)
            T[14:] = 0                                              # tail slots 14, 15 are always zero
            if F <= 11:
                T[11] = 0                                           # ... and so is the slot that holds 1 for the value bias
            V = torch.zeros(96, 16, device=DEV)
            V[:, :F], V[:, 12] = Wv[g], bv[g]
            blocks[(g, d)] = (_plane_slices(_spread16(torch.cat([V, T]))),
                              _plane_slices(w2[g][:, p3 + 96 * d].contiguous()))
            p_dst[:, 16 * (G * d + g): 16 * (G * d + g + 1)] = xs16 @ T.t()
        if F <= 11:
            split_sweeps.append((csr, einfo, p_dst, value_fragments(Wv, bv, F), agg, 16 * G * d, 96 * d, Kg,
                                 96 * n_in + 2 * d, G))
    slices = []
    for g in range(G):
        for d in range(n_in):
            slices += list(blocks[(g, d)])
        slices.append(_plane_slices(_spread16(S[g])))
    wstream = torch.cat(slices).contiguous().view(-1)
    tail = torch.zeros(G, n_in, 6, 4, 16, device=DEV)
    for d in range(n_in):
        tail[:, d, :, 0] = w2[:, :, 96 * n_in + 2 * d].view(G, 6, 16)
        tail[:, d, :, 3] = w2[:, :, 96 * n_in + 2 * d + 1].view(G, 6, 16)
    out = [torch.empty(n_dst, 96, device=DEV), torch.empty(n_dst, 96, device=DEV)]
    fused = (fused_sweeps, xd, wstream, tail.view(G, n_in, 6, 64).contiguous(), *out)
    h_s, c_s = torch.empty(n_dst, 96, device=DEV), torch.empty(n_dst, 96, device=DEV)
    gate = (agg, w2, p_dst, s_off, None, h_s, c_s, None, G, 1, bf16_planes(w2), Kg)
    return fused, split_sweeps, gate


@pytest.mark.parametrize("n_dst,ins,hub", [
    (236, [(118, 11, 708), (236, 8, 708)], 0),      # junctions of the 40 um fixture: two incoming edge types
    (118, [(236, 8, 708)], 0),                      # grains: one
    (1, [(1, 11, 1)], 0), (5, [(9, 8, 11), (5, 8, 0)], 0), (17, [(30, 12, 60)], 0), (33, [(40, 8, 0), (33, 8, 0)], 0),
    (50, [(70, 8, 400), (70, 11, 1300)], 37), (50, [(70, 11, 1300)], 900), (67, [(30, 8, 500)], 0),
    (20000, [(10000, 11, 60000), (20000, 8, 60000)], 0), (10000, [(20000, 8, 60000)], 0)])
@torch.no_grad()
def test_fused_encoder_cell_against_its_contract(n_dst, ins, hub):
    """ggnn_encoder_cell_batch (round 4: score tails, sweeps, lin_l2, skip and LSTM update of a 16-node tile in ONE
    kernel, nothing but x, the edge records and the weight stream read) against the torch emulation of its contract
    evaluated on the DECODED weight stream, and against ggnn_period_gat_aggregate_enc_batch + ggnn_lstm_epilogue on
    the same weights: the fixture sizes, fewer than 16 rows, ragged last tiles and surplus waves, edge types without
    edges, 12 source features, a hub row of degree `hub` (300 units), rows without edges, cfg3 sizes;
    bit-reproducible (no atomics)."""
    from emulator import TorchEmulatorBackend
    from graingraphnn_amd.backend import CSR
    be = backend()
    rs = np.random.RandomState(n_dst + 7 * len(ins) + hub)
    fused, split_sweeps, gate = _enc_cell_problem(be, rs, n_dst, ins, hub)
    be.encoder_cell_batch([fused])
    h, c = fused[4:]
    cpu = lambda t: t.cpu() if torch.is_tensor(t) else t
    sw_cpu = [(CSR(cs.rowptr.cpu(), cs.col.cpu(), cs.perm.cpu(), cs.row.cpu(), None, None, cs.E), ei.cpu())
              for cs, ei in fused[0]]
    ref = [torch.empty(n_dst, 96), torch.empty(n_dst, 96)]
    TorchEmulatorBackend().encoder_cell_batch([(sw_cpu, *[cpu(t) for t in fused[1:4]], *ref)])
    assert_close(h, ref[0], "fused encoder cell h", 2e-5, 2e-6)
    assert_close(c, ref[1], "fused encoder cell c", 2e-5, 2e-6)
    if len(split_sweeps) == len(ins):
        be.aggregate_enc_batch(split_sweeps)
        be.lstm_epilogue(*gate)
        assert_close(h, gate[5], "fused vs split h", 2e-5, 2e-6)
        assert_close(c, gate[6], "fused vs split c", 2e-5, 2e-6)
    keep = [h.clone(), c.clone()]
    h.fill_(float("nan")), c.fill_(float("nan"))
    be.encoder_cell_batch([fused])
    assert torch.equal(keep[0], h) and torch.equal(keep[1], c)      # no atomics: bit-reproducible
    assert not be.range_exceeded(DEV)                               # O(1) operands: nothing was clamped


@torch.no_grad()
def test_fused_encoder_cell_batch_of_four_equals_single_calls():
    """Four problems (two node types x two models) in one ggnn_encoder_cell_batch = four single calls."""
    be = backend()
    rs = np.random.RandomState(5)
    shapes = [(2086, [(1043, 11, 6258), (2086, 8, 6258)], 8), (1043, [(2086, 8, 6258)], 11),
              (2086, [(1043, 11, 6258), (2086, 8, 6258)], 8), (1043, [(2086, 8, 6258)], 11)]
    probs = [_enc_cell_problem(be, rs, n, ins, F_dst=F)[0] for n, ins, F in shapes]
    be.encoder_cell_batch(probs)
    batched = [[t.clone() for t in p[4:]] for p in probs]
    for p, want in zip(probs, batched):
        for t in p[4:]:
            t.fill_(float("nan"))
        be.encoder_cell_batch([p])
        for a, b in zip(p[4:], want):
            assert torch.equal(a, b)
    with pytest.raises(_lib.GGNNError):
        be.encoder_cell_batch(probs + probs[:1])                    # at most four problems


def _wide(rs, shape, lo, hi, signed=True):
    """Magnitudes 10^U(lo, hi), element by element (a dynamic range no single scale has), random signs."""
    v = 10.0 ** rs.uniform(lo, hi, shape)
    if signed:
        v = v * rs.choice([-1.0, 1.0], shape)
    return torch.from_numpy(v.astype(np.float32))


def _cell_reference(kind, P, dtype):
    """One HeteroPGCLSTM cell of the contract (include/ggnn.h), evaluated on the CPU in `dtype` from the ORIGINAL fp32
    operands and weights of `P` -- nothing decoded from a weight stream: what the reference formulation computes.
    kind = "dec" (h, c given; four gates) or "enc" (zero state; three gates)."""
    t = lambda v: v.cpu().to(dtype)
    x, n = t(P["x_dst"]), P["x_dst"].size(0)
    G = 4 if kind == "dec" else 3
    if kind == "dec":
        xin = torch.cat([t(P["h_dst"]), x, torch.ones(n, 1, dtype=dtype)], 1)                  # [h | x | 1]
    else:
        xin = torch.cat([x, torch.ones(n, 1, dtype=dtype)], 1)                                # [x | 1]
    pre = []
    for g in range(G):
        z = xin @ t(P["skip"][g]).t()
        for d, sw in enumerate(P["sweeps"]):
            rowptr = sw["rowptr"].cpu().long()
            E = int(rowptr[-1])
            dst = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1])
            src = sw["col"].cpu().long()[:E]
            einfo = t(sw["einfo"])
            x4, reloc, a = einfo[:E, :16], einfo[:E, 16:19], einfo[:E, 19]
            u = xin @ t(sw["score"][g]).t()                                                    # dec: [n, 96 + 16]; enc: [n, 16]
            if kind == "dec":
                sc = (u[dst, :96] * t(sw["h_src"])[src]).sum(-1) + (u[dst, 96:] * x4).sum(-1)
                val = torch.relu(t(sw["v_src"])[src][:, sw["v_off"] + g * 96: sw["v_off"] + (g + 1) * 96] + reloc @ t(sw["ep"])[g])
            else:
                sc = (u[dst] * x4).sum(-1)
                val = torch.relu(x4 @ t(sw["value"][g]).t())
            smax = torch.full((n,), float("-inf"), dtype=dtype).scatter_reduce(0, dst, sc, "amax")
            p = (sc - smax[dst]).exp()
            den = torch.zeros(n, dtype=dtype).index_add(0, dst, p)
            alpha = p / (den[dst] + 1e-16)
            A = torch.zeros(n, 96, dtype=dtype).index_add(0, dst, alpha[:, None] * val)
            sa = torch.zeros(n, dtype=dtype).index_add(0, dst, alpha)
            sae = torch.zeros(n, dtype=dtype).index_add(0, dst, alpha * a)
            z = z + A @ t(sw["l2"][g]).t() + sa[:, None] * t(sw["b_l2"][g])[None] + sae[:, None] * t(sw["w_edge"][g])[None]
        pre.append(z)
    if kind == "dec":
        c = torch.sigmoid(pre[1]) * t(P["c_in"]) + torch.sigmoid(pre[0]) * torch.tanh(pre[2])
        return torch.sigmoid(pre[3]) * torch.tanh(c), c
    c = torch.sigmoid(pre[0]) * torch.tanh(pre[1])
    return torch.sigmoid(pre[2]) * torch.tanh(c), c


def _wide_cell_problem(be, kind, rs, n_dst, ins, F_dst, with_edges=True):
    """A decoder / encoder cell problem whose operands span 1e-4 .. 1e2 element by element, with weights scaled so that
    the pre-activations stay O(1).  Returns (the C-ABI call tuple, the dict of ORIGINAL fp32 operands and weights)."""
    G = 4 if kind == "dec" else 3
    d_ = lambda v: v.to(DEV)
    xd = _wide(rs, (n_dst, F_dst), -4, 2, signed=False)
    P = {"x_dst": d_(xd), "sweeps": []}
    K = (96 if kind == "dec" else 0) + F_dst + 1
    if kind == "dec":
        P["h_dst"] = d_(torch.tanh(_wide(rs, (n_dst, 96), -4, 0.5)))
        P["c_in"] = d_(_wide(rs, (n_dst, 96), -4, 0.3))
    # every weight row is scaled by what it multiplies, so that sum |x||w| ~ 1 .. 10 per output
    feat_scale = 1.0 / (float(xd.abs().mean()) * F_dst + (30.0 if kind == "dec" else 0.0) + 1.0)
    wrow = lambda rows, cols, s: _wide(rs, (rows, cols), -2, 0) * s
    P["skip"] = [d_(wrow(96, K, feat_scale)) for _ in range(G)]
    for d, (n_src, F, E) in enumerate(ins):
        E = E if with_edges else 0
        src = rs.randint(0, max(n_src - 5, 1), size=E)
        dst = rs.randint(1 if n_dst > 1 else 0, n_dst, size=E)
        ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64)).to(DEV)
        xs = _wide(rs, (n_src, F), -4, 2, signed=False)
        xs[:, :3] = torch.from_numpy(rs.uniform(0, 1, (n_src, 3)).astype(np.float32))   # coordinates stay in the unit box
        ea = torch.from_numpy(rs.uniform(0.01, 0.1, E).astype(np.float32)).to(DEV)
        csr = be.build_csr(ei, n_src, n_dst)
        einfo = torch.zeros(E + 3, 20, device=DEV)
        be.edge_prepare([(csr, ea, d_(xs), P["x_dst"], einfo)])
        rec_scale = 1.0 / (float(xs[:, 3:].abs().mean()) * max(F - 3, 1) + 2.0)
        sw = {"rowptr": csr.rowptr, "col": csr.col, "einfo": einfo, "csr": csr,
              "l2": [d_(wrow(96, 96, 0.05)) for _ in range(G)], "b_l2": [d_(wrow(96, 1, 0.3)[:, 0]) for _ in range(G)],
              "w_edge": [d_(wrow(96, 1, 0.3)[:, 0]) for _ in range(G)]}
        if kind == "dec":
            sw["h_src"] = d_(torch.tanh(_wide(rs, (n_src, 96), -4, 0.5)))
            sw["v_src"] = d_(_wide(rs, (n_src, 384 * (d + 1) + 96), -4, 1))
            sw["v_off"] = 384 * d
            sw["ep"] = d_(_wide(rs, (4, 3, 96), -2, 0))
            score = []
            for g in range(G):
                W1 = torch.zeros(112, K)
                W1[:96] = wrow(96, K, 0.02 * feat_scale)            # u_h rows (they meet h_src in (-1, 1))
                W1[96:110] = wrow(14, K, rec_scale * feat_scale)     # u4 rows (they meet the edge record)
                score.append(d_(W1))
            sw["score"] = score
        else:
            score, value = [], []
            for g in range(G):
                T = torch.zeros(16, K)
                T[:14] = wrow(14, K, rec_scale * feat_scale)
                if F <= 11:
                    T[11] = 0
                V = torch.zeros(96, 16)
                V[:, :F], V[:, 12] = wrow(96, F, rec_scale), wrow(96, 1, 0.3)[:, 0]
                score.append(d_(T))
                value.append(d_(V))
            sw["score"], sw["value"] = score, value
        P["sweeps"].append(sw)
    return _rebuild_wide_call(be, kind, P)


@pytest.mark.parametrize("kind", ["dec", "enc"])
@torch.no_grad()
def test_fused_cells_are_fp32_equivalent_on_wide_range_operands(kind):
    """The two-piece fp16 arithmetic of the fused cells (csrc/common.h: split_f16x2, three MFMA products) pinned ON THE
    GPU: operands spanning 1e-4 .. 1e2 element by element through ggnn_decoder_cell_batch / ggnn_encoder_cell_batch
    against an fp64 evaluation of the same cell from the ORIGINAL fp32 weights (not the decoded stream, so the weight
    split is inside the comparison).  (i) Whole cells: the error of (h, c) must not exceed that of a plain fp32
    evaluation of the same formulas (torch CPU float32) by more than a factor two.  (ii) One GEMM in isolation -- a
    cell without in-edges is its skip product, and with the other gates' weights zero c' = tanh(pre) / 2 can be
    inverted: normalised by sum |x||w| the product must be within 2e-7 of fp64 (a plain fp32 fma chain: 2e-7 .. 7e-7).
    Nothing is clamped on the way (range flag)."""
    be = backend()
    rs = np.random.RandomState(11 if kind == "dec" else 12)
    ins = [(3000, 11, 18000), (6000, 8, 18000)]
    call, P = _wide_cell_problem(be, kind, rs, 6000, ins, 8)
    run = be.decoder_cell_batch if kind == "dec" else be.encoder_cell_batch
    run([call])
    h, c = call[-2].cpu().double(), call[-1].cpu().double()
    r64 = _cell_reference(kind, P, torch.float64)
    r32 = _cell_reference(kind, P, torch.float32)
    for name, got, ref, f32 in (("h", h, r64[0], r32[0]), ("c", c, r64[1], r32[1])):
        scale = float(ref.abs().max())
        e_hip = float((got - ref).abs().max()) / scale
        e_f32 = float((f32.double() - ref).abs().max()) / scale
        assert e_hip <= 2.0 * e_f32 + 5e-7, f"{kind} cell {name}: HIP {e_hip:.2e} vs plain fp32 {e_f32:.2e} of max|ref|"
        assert e_hip < 1e-5, f"{kind} cell {name}: {e_hip:.2e}"
    # (ii) the skip GEMM alone: no edges; gates i and o with zero weights -> c' = tanh(pre_c) / 2
    call, P = _wide_cell_problem(be, kind, rs, 4096, ins, 8, with_edges=False)
    gc = 2 if kind == "dec" else 1                                          # the cell gate's index (i, f, c, o | i, c, o)
    for g in range(len(P["skip"])):
        if g != gc:
            P["skip"][g].zero_()
    if kind == "dec":
        P["c_in"].zero_()
    n = P["x_dst"].size(0)
    xin = torch.cat(([P["h_dst"].cpu().double()] if kind == "dec" else []) + [P["x_dst"].cpu().double(), torch.ones(n, 1, dtype=torch.float64)], 1)
    P["skip"][gc].mul_(1.5 / float((xin.abs() @ P["skip"][gc].cpu().double().abs().t()).median()))   # sum |x||w| ~ 1.5
    call, _ = _rebuild_wide_call(be, kind, P)
    run([call])
    c = call[-1].cpu().double()
    W = P["skip"][gc].cpu().double()
    ref, norm = xin @ W.t(), xin.abs() @ W.abs().t()
    ok = ref.abs() < 1.0                                                    # where atanh(2 c') is well conditioned
    assert float(ok.double().mean()) > 0.25
    pre = torch.atanh((2.0 * c).clamp(-0.999, 0.999))
    # the read-out itself -- c' = tanh(pre) / 2 formed on the hardware exp / rcp units (absolute error <= ~3e-7,
    # common.h) and stored as fp32 -- has a floor of 3e-7 d pre / d c' = 6e-7 / (1 - tanh(pre)^2)
    floor = 6e-7 / (1.0 - torch.tanh(ref) ** 2)
    err = float((((pre - ref).abs() - floor).clamp(min=0) / norm)[ok].max())
    assert err < 2e-7, f"{kind} cell, skip product alone: {err:.2e} of sum |x||w|"
    assert not be.range_exceeded(DEV)


def _rebuild_wide_call(be, kind, P):
    """(call tuple, P): the operands and ORIGINAL weights of `P` in the kernels' weight-stream image; again after an
    in-place edit of the weights."""
    from graingraphnn_amd.packing import CELL_P3_CHANNEL, DC_GATE_ORDER, _plane_slices, _spread16
    G = 4 if kind == "dec" else 3
    n_dst, F_dst = P["x_dst"].shape
    n_in = len(P["sweeps"])
    K = (96 if kind == "dec" else 0) + F_dst + 1
    p3 = torch.tensor(CELL_P3_CHANNEL, device=DEV)

    def in128(W):
        out = torch.zeros(W.size(0), 128, device=DEV)
        out[:, :K] = W
        return out

    def slots16(W):
        out = torch.zeros(W.size(0), 16, device=DEV)
        out[:, :F_dst], out[:, 12] = W[:, :F_dst], W[:, F_dst]
        return out

    slices = []
    for gi, g in enumerate(DC_GATE_ORDER if kind == "dec" else range(3)):
        for sw in (P["sweeps"][::-1] if kind == "dec" and gi & 1 else P["sweeps"]):   # decoder: backwards for the 2nd / 4th gate
            if kind == "dec":
                slices += [_plane_slices(in128(sw["score"][g])), _plane_slices(sw["l2"][g])]
            else:
                slices += [_plane_slices(_spread16(torch.cat([sw["value"][g], slots16(sw["score"][g])]))),
                           _plane_slices(sw["l2"][g][:, p3].contiguous())]
        slices.append(_plane_slices(in128(P["skip"][g]) if kind == "dec" else _spread16(slots16(P["skip"][g]))))
    wstream = torch.cat(slices).contiguous().view(-1)
    tail = torch.zeros(G, n_in, 6, 4, 16, device=DEV)
    for d, sw in enumerate(P["sweeps"]):
        for g in range(G):
            tail[g, d, :, 0] = sw["b_l2"][g].view(6, 16)
            tail[g, d, :, 1 if kind == "dec" else 3] = sw["w_edge"][g].view(6, 16)
    tail = tail.view(G, n_in, 6, 64).contiguous()
    out = [torch.empty(n_dst, 96, device=DEV), torch.empty(n_dst, 96, device=DEV)]
    if kind == "dec":
        return ([(sw["csr"], sw["einfo"], sw["h_src"], sw["v_src"], sw["v_off"], sw["ep"]) for sw in P["sweeps"]],
                P["x_dst"], P["h_dst"], P["c_in"], wstream, tail, *out), P
    return ([(sw["csr"], sw["einfo"]) for sw in P["sweeps"]], P["x_dst"], wstream, tail, *out), P


@pytest.mark.parametrize("kind", ["dec", "enc"])
@torch.no_grad()
def test_fused_cells_report_an_activation_beyond_fp16_range(kind):
    """include/ggnn.h, OPERAND RANGE: an activation at or beyond 65504 is clamped by the two-piece split AND reported
    through the flag word (backend.range_exceeded); in-range operands leave the flag alone."""
    be = backend()
    rs = np.random.RandomState(3)
    ins = [(118, 11, 708), (236, 8, 708)]
    call, P = _wide_cell_problem(be, kind, rs, 236, ins, 8)
    run = be.decoder_cell_batch if kind == "dec" else be.encoder_cell_batch
    be.range_exceeded(DEV)                     # clear
    run([call])
    assert not be.range_exceeded(DEV)
    P["x_dst"][17, 5] = 7.0e4                  # one feature beyond fp16's range
    run([call])
    assert be.range_exceeded(DEV)              # reported (and cleared by the read)
    assert not be.range_exceeded(DEV)
    assert bool(torch.isfinite(call[-2]).all())   # clamped, not poisoned
    # a NaN is not "in range" either (an fp max would drop it: the kernels track magnitudes as bit patterns)
    P["x_dst"][17, 5] = float("nan")
    run([call])
    assert be.range_exceeded(DEV)
    P["x_dst"][17, 5] = 0.5
    run([call])
    assert not be.range_exceeded(DEV)
    if kind == "dec":                           # ... nor is one among the gathered hidden rows (it reaches the aggregates)
        src = int(P["sweeps"][0]["col"][5])
        P["sweeps"][0]["h_src"][src, 40] = float("inf")
        run([call])
        assert be.range_exceeded(DEV)


def _dec_cell_problem(be, rs, n_dst, ins, hub=0, F_dst=8, edges=None):
    """Random decoder-cell problem (ggnn_decoder_cell_batch): destination type with `F_dst` features, `ins` =
    [(n_src, F_src, E)] incoming edge types, random weight blocks packed by packing._plane_slices in the stream's
    order.  Returns the fused-call tuple."""
    from graingraphnn_amd.packing import DC_GATE_ORDER, _plane_slices
    n_in = len(ins)
    f = lambda *shape, lo=-1.0, hi=1.0: torch.from_numpy(rs.uniform(lo, hi, shape).astype(np.float32)).to(DEV)
    xd, h_dst, c_in = f(n_dst, F_dst, lo=0.0), f(n_dst, 96), f(n_dst, 96)
    slices = []
    for gi, g in enumerate(DC_GATE_ORDER):
        for d in (range(n_in - 1, -1, -1) if gi & 1 else range(n_in)):   # (random blocks: only their count matters here)
            W1 = f(112, 128, lo=-0.15, hi=0.15)
            W1[:, 96 + F_dst + 1:] = 0                       # reduction index: h | x | 1 | zeros
            W1[96 + 14:, :] = 0                              # tail slots 14, 15 of u4 are always zero
            slices += [_plane_slices(W1), _plane_slices(f(96, 96, lo=-0.2, hi=0.2))]
        W4 = f(96, 128, lo=-0.15, hi=0.15)
        W4[:, 96 + F_dst + 1:] = 0
        slices.append(_plane_slices(W4))
    wstream = torch.cat(slices).contiguous().view(-1)
    tail = torch.zeros(4, n_in, 6, 4, 16, device=DEV)
    tail[:, :, :, :2] = f(4, n_in, 6, 2, 16, lo=-0.2, hi=0.2)
    sweeps = []
    for d, (n_src, F, E) in enumerate(ins):
        src = rs.randint(0, max(n_src - 5, 1), size=E)
        dst = rs.randint(1 if n_dst > 1 else 0, n_dst, size=E)     # destination 0 has no in-edge
        dst[:hub] = min(7, n_dst - 1)
        if edges is not None:                                       # a given structure (timing tools): [2, E] per edge type
            src, dst = edges[d][0], edges[d][1]
        ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64)).to(DEV)
        xs = f(n_src, F, lo=0.0)
        ea = f(E, lo=0.01, hi=0.1)
        csr = be.build_csr(ei, n_src, n_dst)
        einfo = torch.zeros(E + 3, 20, device=DEV)
        be.edge_prepare([(csr, ea, xs, xd, einfo)])
        v_src = f(n_src, 384 * (d + 1) + 96)                         # value rows at a column offset, padded rows
        sweeps.append((csr, einfo, f(n_src, 96), v_src, 384 * d, f(4, 3, 96)))
    return (sweeps, xd, h_dst, c_in, wstream, tail.view(4, n_in, 6, 64).contiguous(),
            torch.empty(n_dst, 96, device=DEV), torch.empty(n_dst, 96, device=DEV))


@torch.no_grad()
def test_block_major_value_rows_equal_the_row_major_layout():
    """GGNN_OUT_BLOCK_MAJOR (ABI 24): the projection writes every 96 columns as a contiguous [M, 96] block and the decoder
    cell gathers them from there (v_block_major) -- same bits as the [M, ncols] layout, for a ragged node count, under both
    arithmetic modes of the projection."""
    be = backend()
    rs = np.random.RandomState(5)
    f = lambda *shape, lo=-1.0, hi=1.0: torch.from_numpy(rs.uniform(lo, hi, shape).astype(np.float32)).to(DEV)
    M, F, ncols = 1003, 8, 768
    x, h, wp, bp = f(M, F, lo=0.0), f(M, 96), f(ncols, 8 + 96, lo=-0.2, hi=0.2), f(ncols)
    for prec in (_lib.GGNN_PRECISION_F16X2, 0):
        rows = torch.empty(M, ncols, device=DEV)
        blocks = torch.full((M, ncols + 96), 7.0, device=DEV)   # (a wider buffer, as the workspace's: the blocks fill its front)
        be.project_batch([(x, F, h, wp, bp, rows, prec)])
        be.project_batch([(x, F, h, wp, bp, blocks[:, :ncols], prec | _lib.GGNN_OUT_BLOCK_MAJOR)])
        got = blocks.view(-1)[:M * ncols].view(ncols // 96, M, 96)
        assert torch.equal(got.permute(1, 0, 2).reshape(M, ncols), rows), prec
    # the decoder cell on the same value rows in both layouts
    prob = _dec_cell_problem(be, np.random.RandomState(11), 236, [(118, 11, 708), (236, 8, 708)])
    be.decoder_cell_batch([prob])
    h_rows, c_rows = prob[6].clone(), prob[7].clone()
    sweeps = []
    for csr, einfo, h_src, v_src, v_off, ep in prob[0]:
        n, w = v_src.shape
        w96 = w // 96 * 96
        bm = torch.zeros(n, w, device=DEV)
        bm.view(-1)[:n * w96] = v_src[:, :w96].reshape(n, w96 // 96, 96).permute(1, 0, 2).reshape(-1)
        sweeps.append((csr, einfo, h_src, bm, v_off, ep, True))
    prob2 = (sweeps,) + tuple(prob[1:6]) + (torch.empty_like(prob[6]), torch.empty_like(prob[7]))
    be.decoder_cell_batch([prob2])
    assert torch.equal(prob2[6], h_rows) and torch.equal(prob2[7], c_rows)


@pytest.mark.parametrize("n_dst,ins,hub", [
    (236, [(118, 11, 708), (236, 8, 708)], 0),      # junctions of the 40 um fixture: two incoming edge types
    (118, [(236, 8, 708)], 0),                      # grains: one
    (1, [(1, 11, 1)], 0), (5, [(9, 8, 11), (5, 8, 0)], 0), (17, [(30, 12, 60)], 0), (33, [(40, 8, 0), (33, 8, 0)], 0),
    (50, [(70, 8, 400), (70, 11, 1300)], 37), (50, [(70, 11, 1300)], 900), (67, [(30, 8, 500)], 0),
    (20000, [(10000, 11, 60000), (20000, 8, 60000)], 0), (10000, [(20000, 8, 60000)], 0)])
@torch.no_grad()
def test_fused_decoder_cell_against_its_contract(n_dst, ins, hub):
    """ggnn_decoder_cell_batch (destination-side projections, sweep, lin_l2, skip and LSTM update of a 16-node tile
    in one kernel, weights streamed as two fp16 planes) against the torch emulation of its contract evaluated on the
    DECODED weight stream: fixture sizes, fewer than 16 rows, ragged last tiles and surplus waves, edge types
    without edges, 12 source features, a hub row of degree `hub` (more in-edges than the tile's LDS index window),
    rows without edges, cfg3 sizes; bit-reproducible (no atomics)."""
    from emulator import TorchEmulatorBackend
    from graingraphnn_amd.backend import CSR
    be = backend()
    rs = np.random.RandomState(n_dst + 7 * len(ins) + hub)
    prob = _dec_cell_problem(be, rs, n_dst, ins, hub)
    be.decoder_cell_batch([prob])
    h, c = prob[6], prob[7]
    cpu = lambda t: t.cpu() if torch.is_tensor(t) else t
    sw_cpu = [(CSR(cs.rowptr.cpu(), cs.col.cpu(), cs.perm.cpu(), cs.row.cpu(), None, None, cs.E), ei.cpu(), hs.cpu(),
               vs.cpu(), vo, ep.cpu()) for cs, ei, hs, vs, vo, ep in prob[0]]
    ref = [torch.empty(n_dst, 96), torch.empty(n_dst, 96)]
    TorchEmulatorBackend().decoder_cell_batch([(sw_cpu, *[cpu(t) for t in prob[1:6]], *ref)])
    assert_close(h, ref[0], "fused decoder cell h", 2e-5, 2e-6)
    assert_close(c, ref[1], "fused decoder cell c", 2e-5, 2e-6)
    keep = [h.clone(), c.clone()]
    h.fill_(float("nan")), c.fill_(float("nan"))
    be.decoder_cell_batch([prob])
    assert torch.equal(keep[0], h) and torch.equal(keep[1], c)      # no atomics: bit-reproducible


@torch.no_grad()
def test_fused_decoder_cell_batch_of_four_equals_single_calls():
    """Four problems (two node types x two models) in one ggnn_decoder_cell_batch = four single calls."""
    be = backend()
    rs = np.random.RandomState(6)
    shapes = [(2086, [(1043, 11, 6258), (2086, 8, 6258)]), (1043, [(2086, 8, 6258)]),
              (2086, [(1043, 11, 6258), (2086, 8, 6258)]), (1043, [(2086, 8, 6258)])]
    probs = [_dec_cell_problem(be, rs, n, ins, F_dst=8 if len(ins) == 2 else 11) for n, ins in shapes]
    be.decoder_cell_batch(probs)
    batched = [[t.clone() for t in p[6:]] for p in probs]
    for p, want in zip(probs, batched):
        for t in p[6:]:
            t.fill_(float("nan"))
        be.decoder_cell_batch([p])
        for a, b in zip(p[6:], want):
            assert torch.equal(a, b)
    with pytest.raises(_lib.GGNNError):
        be.decoder_cell_batch(probs + probs[:1])                    # at most four problems


@torch.no_grad()
def test_fused_decoder_cell_equals_the_split_path_end_to_end():
    """Both model forwards with the decoder cell as one kernel (default) and as projection + sweeps + gate GEMM
    (GGNN_DEC=split: the round-2 path) on the 40 um fixture and on a ragged Voronoi structure: same outputs up
    to fp32 re-association."""
    be = backend()
    default = be.fused_decoder
    for tag in ("40", "voronoi"):
        x, ei, ea = load_graph("40") if tag == "40" else synthetic.voronoi(150, seed=4)
        outs = []
        for fused in (True, False):
            be.fused_decoder = fused
            try:
                R, Cm = product_models(21, 1.0, DEV)
                X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
                outs.append({**R(X, EI, EA), **Cm(X, EI, EA)})
            finally:
                be.fused_decoder = default
        for k in ("joint", "grain", "grain_area", "edge_event", "edge"):
            assert_close(outs[0][k], outs[1][k], f"fused vs split decoder, {tag} {k}", 2e-5, 2e-6)


class _OneSweepPerLaunch:
    """The HIP backend with ggnn_period_gat_aggregate_batch replaced by single-sweep launches."""

    def __init__(self, inner):
        self.inner = inner

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def aggregate_batch(self, sweeps):
        for sweep in sweeps:
            self.inner.aggregate(*sweep)


@torch.no_grad()
@pytest.mark.parametrize("which", ["40", "cfg3"])
def test_batched_sweeps_equal_single_launches(which):
    """ggnn_period_gat_aggregate_batch == its sweeps launched one by one, bit for bit (row order
    inside a sweep does not enter any sum), and it refuses sweeps that disagree on n_gates / h."""
    from graingraphnn_amd import engine
    from graingraphnn_amd.packing import NODE_TYPES
    if which == "40":
        x, ei, ea = load_graph("40")
    else:
        x, ei, ea = synthetic.honeycomb(100, 10, 0)
    R, _ = product_models(3, 0.5, DEV)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    be = backend()
    n_nodes = {nt: X[nt].size(0) for nt in NODE_TYPES}
    graph = engine.graph_for(be, EI, n_nodes)
    enc = R.gclstm_encoder.cell_list[0].packed(True)
    dec = R.gclstm_decoder.cell_list[0].packed(False)
    out = []
    for b in (be, _OneSweepPerLaunch(be)):
        ws = engine.Workspace(enc, dec, n_nodes, DEV)
        h, c = engine.run_encoder_decoder(b, enc, dec, graph, ws, X, EA)
        torch.cuda.synchronize()
        out.append((ws, {nt: h[nt].clone() for nt in NODE_TYPES}))
    for nt in NODE_TYPES:
        assert torch.equal(out[0][0].agg_enc[nt], out[1][0].agg_enc[nt])
        assert torch.equal(out[0][0].agg_dec[nt], out[1][0].agg_dec[nt])
        assert torch.equal(out[0][1][nt], out[1][1][nt])
    # argument checks of the batch entry
    ws = out[0][0]
    lay = dec.layout
    sw = [(graph.csr[et], ws.einfo[et], ws.proj[et[0]], ws.proj[et[-1]], ws.h1[et[0]], dec.ep[et],
           ws.agg_dec[et[-1]], lay[et[0]].v_off[et], lay[et[-1]].u_off.get(et, 0), lay[et[-1]].u4_off[et],
           lay[et[-1]].a_off[et], lay[et[-1]].Kg, lay[et[-1]].sc_off[et], dec.G) for et in EDGE_TYPES]
    with pytest.raises(_lib.GGNNError):
        be.aggregate_batch([sw[0], sw[1][:4] + (None,) + sw[1][5:]])   # h_src given / absent
    with pytest.raises(_lib.GGNNError):
        be.aggregate_batch([sw[0], sw[1][:-1] + (3,)])                  # n_gates differ
    with pytest.raises(_lib.GGNNError):
        be.aggregate_batch(sw + sw + sw[:1])                            # more than six sweeps
    with pytest.raises(_lib.GGNNError):
        be.aggregate_batch([])


@torch.no_grad()
def test_forward_launch_tape_tracks_inputs_weights_and_topology():
    """The drop-in forward() re-issues its recorded launches while topology, weights, workspace,
    stream and x tensors stay the same (models.py:_run_cells); anything else re-records."""
    import copy
    x, ei, ea = load_graph("40")
    R, Cm = product_models(12, 1.0, DEV)
    R2, Cm2 = product_models(12, 1.0, DEV)           # never replays: fresh tensors every call
    X, EI = tt(x, DEV), tt(ei, DEV)
    be = backend()
    replays = []
    orig = be.replay
    be.replay = lambda tape: (replays.append(len(tape)), orig(tape))[1]
    try:
        for step in range(4):
            EA = tt(ea, DEV)                         # new edge_attr tensors every step (test.py:562-575)
            for et in EDGE_TYPES:
                EA[et] *= 1.0 + 0.01 * step
            ya, ca = R(X, EI, EA), Cm(X, EI, EA)
            yb, cb = R2(tt({k: v.cpu().numpy() for k, v in X.items()}, DEV), tt(ei, DEV), EA), \
                Cm2(tt({k: v.cpu().numpy() for k, v in X.items()}, DEV), tt(ei, DEV), EA)
            for k in ("joint", "grain", "grain_area"):
                assert torch.equal(ya[k], yb[k]), (step, k)
            assert torch.equal(ca["edge_event"], cb["edge_event"]) and torch.equal(ca["edge"], cb["edge"])
            R.update(X, ya, None)                    # x_dict changes in place, same tensors
        # steps 1..3 replay: edge records + 2 cells x (projections, sweeps, gate GEMMs) = 7 launches per model.
        # One re-record among them: the topology's exact block balance (engine.GraphCSR.balance) arrives with its
        # second forward -- the classifier's first call -- and is part of the tape key, so the regressor's tape of
        # step 0 (recorded with the estimate) is recorded again at step 1; results are the same either way.
        # (the fused decoder cell -- the default plan -- is projection + one kernel instead of projection + sweeps + gates)
        # (... when it runs: by default only from backend.fused_decoder_min_joints junctions on)
        fused_dec = bool(be.fused_decoder) and X["joint"].size(0) >= be.fused_decoder_min_joints
        n_launches = 7 - (2 if be.fused_encoder else 0) - (1 if fused_dec else 0)
        assert replays == [n_launches] * 6, replays
        # new weights -> the tape is dropped and re-recorded
        R.linear["joint"].bias.add_(0.5)
        R.gclstm_decoder.cell_list[0].b_i["joint"].add_(0.1)
        R2.load_state_dict(R.state_dict())
        n = len(replays)
        EA = tt(ea, DEV)
        ya = R(X, EI, EA)
        assert len(replays) == n
        yb = R2(tt({k: v.cpu().numpy() for k, v in X.items()}, DEV), tt(ei, DEV), EA)
        assert torch.equal(ya["joint"], yb["joint"])
        # (the fresh-tensor forwards above pushed this topology out of the CSR cache: the call that re-recorded built
        # a new one, whose exact block balance arrives with ITS second forward and re-records once more)
        assert torch.equal(R(X, EI, EA)["joint"], ya["joint"])
        n = len(replays)
        assert torch.equal(R(X, EI, EA)["joint"], ya["joint"]) and len(replays) == n + 1
        # another stream -> re-record (the stream handle is part of every recorded call)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ys = R(X, EI, EA)
        side.synchronize()
        assert torch.equal(ys["joint"], ya["joint"]) and len(replays) == n + 1
        # deepcopy / pickling keep parameters only
        R3 = copy.deepcopy(R)
        assert R3._tape is None and R3._ws is None
        assert torch.equal(R3(X, EI, EA)["joint"], ya["joint"])
    finally:
        be.replay = orig


@pytest.mark.parametrize("use_graph", [False, True])
@torch.no_grad()
def test_temporal_process_parameters_between_steps(use_graph):
    """test.py:376-378 (`--temporal`): G and R of the step to come are written into the junction features between
    steps.  `GrainRollout.set_process_parameters` + step() against the oracle stepping the same schedule; the edge
    records (which carry the sources' features) are rebuilt because the write is detected, also between replays of
    a captured step."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = load_graph("40")
    R, Cm = product_models(31, 1.0, DEV)
    oR, oC = oracle_models(31, 1.0)
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    oX, oEI, oEA = tt(x), tt(ei), tt(ea)
    ro = GrainRollout(R, Cm, X, EI, EA, 6, use_graph=use_graph, concurrent=True, joint_launches=False)
    for G, Rp in ((2.0, 0.4), (7.5, 1.6), (7.5, 1.6), (0.9, 0.25)):
        ro.set_process_parameters(G, Rp)
        oX["joint"][:, 3], oX["joint"][:, 4] = 1.0 - G / 10.0, Rp / 2.0
        pred = {k: v.clone() for k, v in ro.step().items()}
        opred, oEA = oracle.rollout_step(oR, oC, oX, oEI, oEA, 6)
        for k in ("joint", "grain", "grain_area", "edge_event", "edge"):
            assert_close(pred[k], opred[k], f"temporal G={G} R={Rp} {k}")
    for nt in x:
        assert_close(X[nt], oX[nt], f"temporal x {nt}")


@torch.no_grad()
def test_run_with_multi_step_graph_equals_single_steps():
    """GrainRollout.run(n) replays graphs of RUN_UNROLL steps: bit-identical to n x step()."""
    from graingraphnn_amd.rollout import GrainRollout
    x, ei, ea = load_graph("40")
    out = []
    for mode in ("run", "step", "eager"):
        R, Cm = product_models(5, 1.0, DEV)
        ro = GrainRollout(R, Cm, tt(x, DEV), tt(ei, DEV), tt(ea, DEV), 6, use_graph=mode != "eager",
                          refresh_centres=True)
        if mode == "run":
            ro.run(2 * GrainRollout.RUN_UNROLL + 3)
        else:
            for _ in range(2 * GrainRollout.RUN_UNROLL + 3):
                ro.step()
        assert ro.steps_done == 2 * GrainRollout.RUN_UNROLL + 3
        torch.cuda.synchronize()
        out.append(({k: v.clone() for k, v in ro.x.items()}, {k: v.clone() for k, v in ro.pred.items()}))
    for other in out[1:]:
        for k in ("joint", "grain"):
            assert torch.equal(out[0][0][k], other[0][k])
        for k in out[0][1]:
            assert torch.equal(out[0][1][k], other[1][k])


def test_edge_records_at_wrap_boundaries():
    """periodGATconv.py:209-210: strict comparisons (a difference of exactly +-0.5 is NOT wrapped),
    one shift only (1.25 -> 0.25), all three coordinates; bit-exact against numpy."""
    be = backend()
    vals = np.array([0.0, 0.5, -0.5, 0.5000001, -0.5000001, 0.75, -0.75, 1.25, -1.25, 0.49999997, 1.5, -1.5],
                    dtype=np.float32)
    n = len(vals)
    xs = np.zeros((n, 8), np.float32)
    xs[:, 0], xs[:, 1], xs[:, 2] = vals, vals[::-1], -vals
    xs[:, 3:] = np.arange(n * 5, dtype=np.float32).reshape(n, 5) / 7
    xd = np.zeros((3, 8), np.float32)
    xd[1, :3] = (0.25, -0.25, 0.125)
    xd[2, :3] = (-1.0, 1.0, 0.5)
    src = np.repeat(np.arange(n), 3)
    dst = np.tile(np.arange(3), n)
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64)).to(DEV)
    ea = torch.arange(len(src), dtype=torch.float32, device=DEV) / 100
    csr = be.build_csr(ei, n, 3)
    einfo = torch.zeros(len(src) + 3, 20, device=DEV)
    be.edge_prepare([(csr, ea, torch.from_numpy(xs).to(DEV), torch.from_numpy(xd).to(DEV), einfo)])
    col, row, perm = (t.cpu().numpy()[:len(src)] for t in (csr.col, csr.row, csr.perm))
    rel = xs[col, :3] - xd[row, :3]
    reloc = (-1 * (rel > 0.5) + 1 * (rel < -0.5) + rel).astype(np.float32)
    got = einfo.cpu().numpy()[:len(src)]
    assert np.array_equal(got[:, 0:3], reloc) and np.array_equal(got[:, 16:19], reloc)
    assert np.array_equal(got[:, 3:8], xs[col, 3:8]) and (got[:, 8:11] == 0).all()
    assert (got[:, 11] == 1).all()            # bias row of the encoder sweep's value product (f_src <= 11)
    assert (got[:, 12] == 1).all() and np.array_equal(got[:, 13], ea.cpu().numpy()[perm])
    assert np.array_equal(got[:, 19], got[:, 13]) and (got[:, 14:16] == 0).all()


@torch.no_grad()
@pytest.mark.parametrize("clamp", [False, True])
def test_fused_glue_launches_equal_the_separate_calls(clamp):
    """ggnn_heads_regressor_update == ggnn_heads_regressor + ggnn_step_update and ggnn_step_refresh_prepare ==
    ggnn_step_refresh + ggnn_edge_prepare, bit for bit (same expressions on the same operands), with and
    without the z clamp of test.py:405-407 firing."""
    from graingraphnn_amd.engine import GraphCSR, alloc_einfo, prepare_edges
    be = backend()
    x, ei, ea = load_graph("40")
    X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
    if clamp:
        X["joint"][:, 2] = 0.99
        X["grain"][:, 2] = 0.99
    nj, ng = X["joint"].size(0), X["grain"].size(0)
    g = torch.Generator(device="cpu").manual_seed(3)
    h = {"joint": torch.randn(nj, 96, generator=g).to(DEV), "grain": torch.randn(ng, 96, generator=g).to(DEV)}
    w, b = (torch.randn(2, 2, 96, generator=g) * 0.2).to(DEV), (torch.randn(4, generator=g) * 0.1).to(DEV)
    dz, zmax = float(np.float32(6 / 121)), float(np.float32(120 / 121))
    graph = GraphCSR(be, EI, {"joint": nj, "grain": ng})

    def fresh():
        return ({k: v.clone() for k, v in X.items()}, {et: EA[et].view(-1).clone() for et in EDGE_TYPES},
                {"joint": torch.empty(nj, 2, device=DEV), "grain": torch.empty(ng, 2, device=DEV),
                 "area": torch.empty(ng, device=DEV)}, torch.zeros(2, dtype=torch.int32, device=DEV),
                alloc_einfo(graph, DEV))
    xa, eaa, pa, fa, eia = fresh()
    be.heads_regressor(h["joint"], h["grain"], xa["grain"], w, b, pa["joint"], pa["grain"], pa["area"])
    be.step_update(xa["joint"], xa["grain"], pa["joint"], pa["grain"], dz, zmax, fa)
    be.step_refresh(xa["joint"], xa["grain"], zmax, fa,
                    [(graph.edge_index[et], xa[et[0]], xa[et[-1]], eaa[et]) for et in EDGE_TYPES])
    prepare_edges(be, graph, xa, eaa, eia)
    xb, eab, pb, fb, eib = fresh()
    be.heads_regressor_update(h["joint"], h["grain"], xb["joint"], xb["grain"], w, b, pb["joint"], pb["grain"],
                              pb["area"], dz, zmax, fb)
    mirror = (torch.full_like(xb["joint"], 7.0), torch.full_like(xb["grain"], 7.0))
    be.step_refresh_prepare(xb["joint"], xb["grain"], zmax, fb,
                            [(graph.csr[et], eab[et], xb[et[0]], xb[et[-1]], eib[et]) for et in EDGE_TYPES], mirror=mirror)
    assert int(fa[1]) == int(fb[1]) == int(clamp)
    assert torch.equal(mirror[0], xb["joint"]) and torch.equal(mirror[1], xb["grain"])   # (ABI 24: the classifier's copy)
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k
    for nt in xa:
        assert torch.equal(xa[nt], xb[nt]), nt
    for et in EDGE_TYPES:
        assert torch.equal(eaa[et], eab[et]) and torch.equal(eia[et], eib[et]), et


@torch.no_grad()
@pytest.mark.parametrize("use_graph", [False, True])
def test_pipelined_two_stream_rollout_equals_the_single_stream_plan(use_graph):
    """The pipelined two-stream step (update, grain centres, refresh and the next edge records behind the
    regressor's heads, second edge_attr buffer, no join between the steps of a graph) is a re-ordering of the same
    launches: 23 steps on the folded 120 um fixture -- odd, so both edge_attr buffers end up current once, through
    4-step and single-step graphs -- must reproduce the single-stream plan bit for bit (x, edge_attr and every
    prediction), also after the caller wrote into x between two runs (stale edge records)."""
    from graingraphnn_amd import GrainRollout
    x, ei, ea = load_graph("120")
    x, ea = {k: v.copy() for k, v in x.items()}, {k: v.copy() for k, v in ea.items()}
    off = synthetic.scale_feature_patchs(3.0, x, ea)
    R, Cm = product_models(77, 0.5, DEV)
    ros = []
    for two_streams in (False, True):
        X, EI, EA = tt(x, DEV), tt(ei, DEV), tt(ea, DEV)
        ro = GrainRollout(R, Cm, X, EI, EA, 5, use_graph=use_graph and two_streams, joint_launches=False,
                          concurrent=two_streams, refresh_centres=True, domain_factor=3.0,
                          domain_offset=torch.from_numpy(off))
        assert ro._pipelined() == two_streams
        ro.run(10)
        ro.x["joint"][:, 0] += 0.001              # the caller moves the junctions: the prepared records are stale
        ro.run(13)
        ros.append(ro)
    a, b = ros
    for nt in x:
        assert torch.equal(a.x[nt], b.x[nt]), nt
    for et in EDGE_TYPES:
        assert torch.equal(a.edge_attr[et], b.edge_attr[et]), et
    for k in a.pred:
        assert torch.equal(a.pred[k], b.pred[k]), k
