"""Graph inputs: the committed reference fixtures (`tests/golden/graph_*.npz`) and the
synthetic periodic honeycomb of SURVEY.md section 8(d) (cfg3: 100 x 100 hexagonal grains on
the unit torus = 10 000 grains / 20 000 junctions / 60 000 edges per type).

Feature schema (graph_datastruct.py:827-831 + gradients :1000-1010):
  grain: x, y, z, area, extraV, cosx, sinx, cosz, sinz, span, darea       (11)
  joint: x, y, z, G, R, span, dx, dy                                      (8)
"""
from types import SimpleNamespace

import numpy as np
import torch

GJ = ("grain", "push", "joint")
JG = ("joint", "pull", "grain")
JJ = ("joint", "connect", "joint")
EDGE_TYPES = (GJ, JG, JJ)
FEATURES = {
    "grain": ["x", "y", "z", "area", "extraV", "cosx", "sinx", "cosz", "sinz", "span", "darea"],
    "joint": ["x", "y", "z", "G", "R", "span", "dx", "dy"],
}
TARGETS = {"grain": ["darea", "extraV"], "joint": ["dx", "dy"]}


def default_hyper(device="cuda"):
    """The `hyper` attribute bag of the shipped models (parameters.py:18-50, 97-134;
    test.py:162-175): layer_size 96, layers 1, window 1."""
    return SimpleNamespace(features=FEATURES, targets=TARGETS, layer_size=96, layers=1,
                           metadata=(["grain", "joint"], [GJ, JG, JJ]), out_win=1, window=1,
                           device=device)


def _minimg(d):
    return d - np.round(d)


def edge_lengths(x_dict, edge_index_dict):
    """numpy restatement of test.py:562-575 for building inputs (float64 in, float32 out)."""
    out = {}
    for et, ei in edge_index_dict.items():
        rel = x_dict[et[0]][ei[0], :2] - x_dict[et[-1]][ei[1], :2]
        rel = np.where(rel > 0.5, rel - 1.0, np.where(rel < -0.5, rel + 1.0, rel))
        out[et] = np.sqrt((rel ** 2).sum(1)).astype(np.float32).reshape(-1, 1)
    return out


def scale_feature_patchs(factor: float, x, ea):
    """Host-side data preparation of test.py:29-55 (periodic BC) on numpy fp32 dicts, in place:
    fold a `factor`-times larger domain onto unit training-size patches.  Returns the
    junctions' `domain_offset` [n_joint, 2] that the grain-centre refresh needs (test.py:474)."""
    f = np.float32(factor)
    for et in ea:
        ea[et] *= f
    x["grain"][:, :2] *= f
    x["joint"][:, :2] *= f
    domain_offset = np.floor(x["joint"][:, :2])
    x["joint"][:, :2] -= domain_offset
    x["grain"][:, :2] -= x["grain"][:, :2] - np.mod(x["grain"][:, :2], np.float32(1))
    return domain_offset.astype(np.float32)


def honeycomb(n: int = 100, fold: int = 10, seed: int = 0, shuffle_edges: bool = True,
              return_offset: bool = False):
    """Periodic honeycomb with n x n grains (n even).  Returns numpy dicts
    (x_dict, edge_index_dict, edge_attr_dict) after integer patch folding by `fold`
    (test.py:29-55: xy <- (xy * fold) mod 1), ready for torch.from_numpy; with
    `return_offset` also the junctions' domain_offset = floor(xy * fold) [n_joint, 2]."""
    if n % 2 or n < 4:
        raise ValueError("n must be even and >= 4")
    rs = np.random.RandomState(seed)
    dx = dy = 1.0 / n
    r, c = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    r, c = r.ravel(), c.ravel()
    gid = r * n + c
    odd = r & 1

    def g(rr, cc):
        return (rr % n) * n + (cc % n)

    UR, UL = g(r + 1, c + odd), g(r + 1, c + odd - 1)
    DR, DL = g(r - 1, c + odd), g(r - 1, c + odd - 1)
    T, B = 2 * gid, 2 * gid + 1
    cx, cy = (c + 0.5 * odd) * dx, r * dy
    n_g, n_j = n * n, 2 * n * n
    xj = np.zeros((n_j, 2))
    xj[T] = np.stack([cx, cy + 2 * dy / 3], 1)
    xj[B] = np.stack([cx, cy - 2 * dy / 3], 1)
    xj += rs.normal(0.0, 0.1 * (2 * dy / 3), size=xj.shape)
    xj %= 1.0
    # the six junctions of every grain: N, NE, SE, S, SW, NW
    hexj = np.stack([T, 2 * UR + 1, 2 * DR, B, 2 * DL, 2 * UL + 1], 1)  # [n_g, 6]
    ref = xj[hexj[:, 0]]
    xg = (ref + _minimg(xj[hexj] - ref[:, None, :]).mean(1)) % 1.0
    # edges: joint -> grain (6 per grain), grain -> joint = flip, joint <-> joint (3 per joint)
    jg = np.stack([hexj.ravel(), np.repeat(gid, 6)])
    gj = jg[::-1].copy()
    up2, dn2 = g(r + 2, c), g(r - 2, c)
    jj_src = np.concatenate([T, T, T, B, B, B])
    jj_dst = np.concatenate([2 * UR + 1, 2 * UL + 1, 2 * up2 + 1, 2 * DR, 2 * DL, 2 * dn2])
    jj = np.stack([jj_src, jj_dst])
    ei = {GJ: gj, JG: jg, JJ: jj}
    if shuffle_edges:
        for k, et in enumerate(EDGE_TYPES):
            p = np.random.RandomState(seed + 1 + k).permutation(ei[et].shape[1])
            ei[et] = ei[et][:, p]
    ei = {et: np.ascontiguousarray(v.astype(np.int64)) for et, v in ei.items()}
    # features
    theta_x, theta_z = rs.uniform(0, np.pi / 2, n_g), rs.uniform(0, np.pi / 2, n_g)
    area = np.clip(0.0087 * (1 + 0.1 * rs.normal(size=n_g)), 0.002, None)
    fg = np.zeros((n_g, 11))
    fg[:, :2] = (xg * fold) % 1.0
    fg[:, 3] = area
    fg[:, 5], fg[:, 6], fg[:, 7], fg[:, 8] = np.cos(theta_x), np.sin(theta_x), np.cos(theta_z), np.sin(theta_z)
    fg[:, 9] = 6 / 120
    fj = np.zeros((n_j, 8))
    fj[:, :2] = (xj * fold) % 1.0
    fj[:, 3], fj[:, 4], fj[:, 5] = 0.0, 1.0, 6 / 120
    x = {"grain": fg.astype(np.float32), "joint": fj.astype(np.float32)}
    ea = edge_lengths({k: v.astype(np.float64) for k, v in x.items()}, ei)
    if return_offset:
        return x, ei, ea, np.floor(xj * fold).astype(np.float32)
    return x, ei, ea


def voronoi(n_grains: int = 400, seed: int = 0, fold: int = 1, lattice_noise: float = None,
            shuffle_edges: bool = True, return_offset: bool = False):
    """Random periodic grain structure: the Voronoi tessellation of `n_grains` seed points on the
    unit torus (what graph_datastruct.py:350-464 builds from its 3 x 3 mirrored seeds with
    scipy.spatial.Voronoi: grains = cells, junctions = Voronoi vertices, every junction touches
    exactly three grains and three junctions).  `lattice_noise=None`: uniformly random seeds (grain
    degrees 3..11); a float: a hexagonal lattice with that relative jitter, the reference's default
    initial condition (graph_datastruct.py:118-160).  Returns numpy dicts like `honeycomb`, after
    integer patch folding by `fold` (test.py:29-55)."""
    from scipy.spatial import Voronoi
    rs = np.random.RandomState(seed)
    if lattice_noise is None:
        pts = rs.uniform(0, 1, (n_grains, 2))
    else:
        n = int(round(np.sqrt(n_grains)))
        n += n % 2
        r, c = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        pts = np.stack([(c.ravel() + 0.5 * (r.ravel() & 1)) / n, r.ravel() / n], 1)
        pts = (pts + rs.normal(0, lattice_noise / n, pts.shape)) % 1.0
    return _from_seeds(pts, rs, seed, fold, shuffle_edges, return_offset)


_span_grid = None


def span_for(G: float, R: float) -> int:
    """Frame span of a run at thermal gradient G and pulling speed R: the reference's generator does not take it
    as an argument, it looks it up by nearest neighbour in normalised (G, R) among the 1 441 points of its
    training grid (graph_trajectory.py:1308-1316: `griddata(..., method='nearest')` on GR_train_grid.pkl, i.e. a
    k-d tree query).  The table ships as data (graingraphnn_amd/data/gr_span_grid.npz, extracted by
    tests/golden/make_gr_span_grid.py); pinned to the reference's own expression on 47 (G, R) pairs."""
    global _span_grid
    if _span_grid is None:
        import os
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "gr_span_grid.npz"))
        from scipy.spatial import cKDTree
        _span_grid = (cKDTree(np.stack([z["G"], z["R"]], 1)), z["span"], z["bounds"])
    tree, span, (g0, g1, r0, r1) = _span_grid
    _, k = tree.query(np.array([(G - g0) / (g1 - g0), (R - r0) / (r1 - r0)]))
    return int(span[k])


def generate(lxd: float = 40.0, seed: int = 0, G: float = 2.0, R: float = 0.4, span: int = None,
             grain_size: float = 4.0, noise: float = 0.01):
    """The reference's `graph_trajectory.py --mode=generate --lxd --seed --G --R` (:1289-1333): an `lxd` x `lxd` um
    periodic initial grain structure, and for a given seed THE SAME SAMPLE the reference pickles -- junction
    numbering, the three edge lists column for column, every feature -- see `generator.reference_sample`, which this
    is.  `span`: None = the reference's (G, R) lookup (`span_for`).  Needs scipy and Pillow (as the reference does).
    For structures with exact polygon areas, other seed distributions or shuffled edge lists see `voronoi`."""
    from .generator import reference_sample
    return reference_sample(lxd, seed, G, R, span, grain_size, noise)


def lattice_structure(lxd: float = 40.0, seed: int = 0, G: float = 2.0, R: float = 0.4, span: int = None,
                      grain_size: float = 4.0, noise: float = 0.01, shuffle_edges: bool = False):
    """The reference's lattice (graph_datastruct.py:118-160: spacing grain_size / lxd, jitter variance noise / lxd /
    (lxd / 40), seeds inside the unit box) tessellated by this package's own periodic Voronoi construction
    (`_from_seeds`): exact polygon areas instead of the reference's 0.08 um raster, an own random stream, junctions
    numbered by position.  Same distribution as `generate`, not the same sample (round 2's generator, kept for
    structures whose areas must tile the domain exactly)."""
    if span is None:
        span = span_for(G, R)
    rs = np.random.RandomState(seed)
    dx = grain_size / lxd
    rows, cols = int(1 / dx) + 1, int(1 / dx)
    r, c = np.meshgrid(np.arange(2 * rows), np.arange(cols), indexing="ij")
    pts = np.stack([(c.ravel() + 0.5 * (r.ravel() % 2)) * np.sqrt(3.0) * dx + 0.1 * dx,
                    r.ravel() * 0.5 * dx + 0.25 * dx], 1)
    pts = pts + rs.normal(0.0, np.sqrt(noise / lxd / (lxd / 40.0)), pts.shape)
    pts = pts[np.all((pts > 0) & (pts < 1), axis=1)]
    x, ei, ea = _from_seeds(pts, rs, seed, 1, shuffle_edges, False)
    patches = (lxd / 40.0) ** 2                     # areas are fractions of one 40 um patch
    x["grain"][:, 3] *= np.float32(patches)
    x["grain"][:, 9] = x["joint"][:, 5] = np.float32(span / 120)
    x["joint"][:, 3], x["joint"][:, 4] = np.float32(1 - G / 10), np.float32(R / 2)
    return x, ei, ea


def _from_seeds(pts, rs, seed, fold, shuffle_edges, return_offset):
    """Periodic Voronoi tessellation of the seed points `pts` (unit torus) -> (x, ei, ea) numpy dicts."""
    from scipy.spatial import Voronoi
    n_g = len(pts)
    shifts = np.array([[dx, dy] for dx in (-1, 0, 1) for dy in (-1, 0, 1)], dtype=np.float64)
    tiled = (pts[None, :, :] + shifts[:, None, :]).reshape(-1, 2)      # copy k of seed g = row k * n_g + g
    vor = Voronoi(tiled)
    # junctions = Voronoi vertices inside the unit cell; vertex -> canonical junction id by position
    V = vor.vertices
    inside = np.all((V >= 0) & (V < 1), axis=1)
    jid = -np.ones(len(V), dtype=np.int64)
    jid[inside] = np.arange(int(inside.sum()))
    xj = V[inside]
    n_j = len(xj)
    key = {tuple(np.round(p, 9)): k for k, p in enumerate(xj)}

    def canon(v):  # any copy of a vertex -> its junction id (or -1 far outside the 3 x 3 tiling's core)
        if jid[v] >= 0:
            return int(jid[v])
        return key.get(tuple(np.round(V[v] % 1.0, 9)), -1)

    jj, jg = set(), set()
    for (a, b), (p, q) in zip(vor.ridge_vertices, vor.ridge_points):
        if a < 0 or b < 0 or not (inside[a] or inside[b]):
            continue
        ca, cb = canon(a), canon(b)
        if ca < 0 or cb < 0:
            raise RuntimeError("Voronoi vertex without a periodic image: increase n_grains")
        jj.add((ca, cb))
        jj.add((cb, ca))
        for v, cv in ((a, ca), (b, cb)):
            if inside[v]:
                jg.add((cv, int(p % n_g)))
                jg.add((cv, int(q % n_g)))
    jj = np.array(sorted(jj), dtype=np.int64).T
    jg = np.array(sorted(jg), dtype=np.int64).T
    if n_j != 2 * n_g or jj.shape[1] != 3 * n_j or jg.shape[1] != 3 * n_j:
        raise RuntimeError(f"degenerate tessellation (N_j={n_j}, N_g={n_g}, E_jj={jj.shape[1]}, E_jg={jg.shape[1]})")
    gj = jg[::-1].copy()
    ei = {GJ: gj, JG: jg, JJ: jj}
    if shuffle_edges:
        for k, et in enumerate(EDGE_TYPES):
            p = np.random.RandomState(seed + 1 + k).permutation(ei[et].shape[1])
            ei[et] = ei[et][:, p]
        ei[GJ] = ei[JG][::-1].copy()      # grain->joint stays the flip of joint->grain (models.py:837)
    ei = {et: np.ascontiguousarray(v.astype(np.int64)) for et, v in ei.items()}
    # grain centre = mean of its (min-imaged) junctions; area = shoelace of the polygon around it
    xg, area = np.zeros((n_g, 2)), np.zeros(n_g)
    order = np.argsort(jg[1], kind="stable")
    bounds = np.searchsorted(jg[1][order], np.arange(n_g + 1))
    for g in range(n_g):
        js = jg[0][order[bounds[g]:bounds[g + 1]]]
        rel = _minimg(xj[js] - xj[js[0]])
        ctr = rel.mean(0)
        ang = np.argsort(np.arctan2(rel[:, 1] - ctr[1], rel[:, 0] - ctr[0]))
        poly = rel[ang]
        area[g] = 0.5 * abs(np.dot(poly[:, 0], np.roll(poly[:, 1], -1)) - np.dot(poly[:, 1], np.roll(poly[:, 0], -1)))
        xg[g] = (xj[js[0]] + ctr) % 1.0
    theta_x, theta_z = rs.uniform(0, np.pi / 2, n_g), rs.uniform(0, np.pi / 2, n_g)
    fg = np.zeros((n_g, 11))
    fg[:, :2] = (xg * fold) % 1.0
    fg[:, 3] = area * fold * fold          # area as a fraction of one training-size patch
    fg[:, 5], fg[:, 6], fg[:, 7], fg[:, 8] = np.cos(theta_x), np.sin(theta_x), np.cos(theta_z), np.sin(theta_z)
    fg[:, 9] = 6 / 120
    fj = np.zeros((n_j, 8))
    fj[:, :2] = (xj * fold) % 1.0
    fj[:, 3], fj[:, 4], fj[:, 5] = 0.0, 1.0, 6 / 120
    x = {"grain": fg.astype(np.float32), "joint": fj.astype(np.float32)}
    ea = edge_lengths({k: v.astype(np.float64) for k, v in x.items()}, ei)
    if return_offset:
        return x, ei, ea, np.floor(xj * fold).astype(np.float32)
    return x, ei, ea


def load_fixture(path):
    """tests/golden/graph_*.npz -> (x_dict, edge_index_dict, edge_attr_dict) of numpy arrays."""
    z = np.load(path)
    x = {"grain": z["x_grain"], "joint": z["x_joint"]}
    ei = {et: z["ei_" + "__".join(et)] for et in EDGE_TYPES}
    ea = {et: z["ea_" + "__".join(et)] for et in EDGE_TYPES}
    return x, ei, ea


def to_torch(x, ei, ea, device="cpu"):
    def conv(d):  # private copies: rollouts mutate x_dict in place
        return {k: torch.from_numpy(np.array(v, copy=True, order="C")).to(device) for k, v in d.items()}
    return conv(x), conv(ei), conv(ea)


def perturbed_copy(x, sigma: float, seed: int):
    """cfg4: weight-independent per-trajectory perturbation of the joint xy."""
    rs = np.random.RandomState(seed)
    out = {k: v.copy() for k, v in x.items()}
    out["joint"][:, :2] += rs.normal(0.0, sigma, size=out["joint"][:, :2].shape).astype(np.float32)
    return out


def disjoint_union(graphs):
    """cfg4: batch independent graphs into one disjoint-union graph (what PyG DataLoader
    collation does, train.py:365-366): node rows concatenated, edge indices offset.  Returns
    (x, ei, ea, slices) with slices[t] = {'grain': (lo, hi), 'joint': (lo, hi)}."""
    off = {"grain": 0, "joint": 0}
    xs = {"grain": [], "joint": []}
    eis = {et: [] for et in EDGE_TYPES}
    eas = {et: [] for et in EDGE_TYPES}
    slices = []
    for x, ei, ea in graphs:
        slices.append({nt: (off[nt], off[nt] + x[nt].shape[0]) for nt in xs})
        for nt in xs:
            xs[nt].append(x[nt])
        for et in EDGE_TYPES:
            shift = np.array([[off[et[0]]], [off[et[-1]]]], dtype=np.int64)
            eis[et].append(ei[et] + shift)
            eas[et].append(ea[et])
        for nt in xs:
            off[nt] += x[nt].shape[0]
    return ({nt: np.concatenate(v, 0) for nt, v in xs.items()},
            {et: np.ascontiguousarray(np.concatenate(v, 1)) for et, v in eis.items()},
            {et: np.concatenate(v, 0) for et, v in eas.items()}, slices)
