"""Initial grain structures exactly as the reference's generator makes them.

`reference_sample(lxd, seed, G, R)` restates, step for step, what
`python graph_trajectory.py --mode=generate --lxd=L --seed=S --G=g --R=r` pickles
(graph_trajectory.py:1289-1333), so that seed S here IS the reference's graph for seed S -- same junction
numbering, same three edge lists column for column, same features:

  1. seeds   (graph_datastruct.py:118-160, 207-281)  hexagonal lattice, jitter drawn in ONE
             multivariate_normal(size = rows * cols * 5) call from the legacy global stream seeded with `seed`
             and indexed by a running counter that starts at 1; the seeds inside the unit box are mirrored
             eight times (periodic images, in the reference's order), all handed to scipy.spatial.Voronoi;
  2. cells   (:350-464)  Voronoi regions in scipy's order; a region is a grain when none of its vertices is at
             infinity or outside (-0.5, 1.5)^2; vertices are identified by their coordinates modulo 1 ROUNDED TO
             FOUR DECIMALS (junction ids in order of first appearance), periodic copies of a cell (same vertex
             set) are dropped, grains are numbered from 1 in order of appearance; vertices shared by four cells
             after the rounding ("quadruples") are split in two as the reference does;
  3. graph   (:654-800, init branch)  junction -> sorted grain triple (a dict: a triple met twice keeps the LAST
             junction); per grain the polygon (junctions chained by periodic_move, shifted into the box, sorted
             counter-clockwise around their mean = the grain centre), and the junction-junction edge list = the
             polygons' sides in that order (every side once per adjacent grain: both directions appear);
  4. areas   (:553-610)  the polygons rasterised with PIL on a 2s x 2s image (s = int(lxd / 0.08) + 1), the four
             s x s quadrants folded with max; a grain's area = its pixel count / 501^2 (one 40 um patch);
  5. angles  (:290-306)  three randn(n_grains) draws from the same stream, behind the jitter;
  6. tensors (graph_trajectory.py:901-1005, graph_datastruct.py:980-1010)  features, the three edge lists and the
             edge lengths; span from the (G, R) lookup (synthetic.span_for).

Where the ORDER of a Python container decides the result (iteration over a set of three grain ids, dict
insertion order) the same container type is used here: the order is part of what is being reproduced.
Pinned by tests/golden/generated_40_seed{1,2}.npz (made by the unmodified reference,
tests/golden/make_golden_generated.py): edge lists bit for bit, features and edge lengths to fp32.
Needs scipy (Voronoi) and Pillow (the raster), as the reference does; the same scipy / qhull build gives the same
vertex order -- on another build the tessellation is the same but the numbering may differ."""
import math
from collections import defaultdict

import numpy as np

EPS = 1e-12
MESH = 0.08       # um per pixel of the reference's raster
PATCH = 40.0      # um: areas are fractions of one training-size patch


def _lattice(dx, noise, rs):
    """graph_datastruct.py:118-160 (periodic): seeds inside the unit box + their eight periodic images."""
    rows, cols = int(1 / dx) + 1, int(1 / dx)
    shiftx, shifty = 0.1 * dx, 0.25 * dx
    jitter = rs.multivariate_normal(mean=np.zeros(2), cov=np.eye(2) * noise, size=rows * cols * 5)
    pts, count = [], 0
    for row in range(rows * 2):
        for col in range(cols):
            count += 1
            x = ((col + (0.5 * (row % 2))) * np.sqrt(3)) * dx + shiftx + jitter[count, 0]
            y = row * 0.5 * dx + shifty + jitter[count, 1]
            if -EPS <= x <= 1 + EPS and -EPS <= y <= 1 + EPS:
                pts += [[x, y], [x + 1, y], [x - 1, y], [x, y + 1], [x, y - 1], [x + 1, y + 1], [x - 1, y - 1],
                        [x - 1, y + 1], [x + 1, y - 1]]
    return pts


def _min_image_onto(p, pc):
    """periodic_move (graph_datastruct.py:55-72): p shifted by whole periods next to pc."""
    x, y = p
    rx, ry = x - pc[0], y - pc[1]
    return [x + (-1 * (rx > 0.5) + 1 * (rx < -0.5)), y + (-1 * (ry > 0.5) + 1 * (ry < -0.5))]


def _periodic_dist(p, pc):
    """periodic_dist_ (graph_datastruct.py:75-86)."""
    x, y = p
    xc, yc = pc
    if x < xc - 0.5 - EPS:
        x += 1
    if x > xc + 0.5 + EPS:
        x -= 1
    if y < yc - 0.5 - EPS:
        y += 1
    if y > yc + 0.5 + EPS:
        y -= 1
    return math.sqrt((x - xc) ** 2 + (y - yc) ** 2)


def _ccw_key(point, center):
    """counterclock (graph_datastruct.py:100-116): (angle in [0, 2 pi), distance)."""
    vx, vy = point[0] - center[0], point[1] - center[1]
    ln = math.hypot(vx, vy)
    if ln == 0:
        return -math.pi, 0
    ang = math.atan2(vy, vx)
    return (2 * math.pi + ang, ln) if ang < 0 else (ang, ln)


def tessellate(seeds):
    """Steps 2 and 3: -> (vertices {junction: (x, y)}, joint2vertex {sorted grain triple: junction},
    region_center {grain: [x, y]}, region_coors {grain: polygon}, edges [[src, dst]])."""
    from scipy.spatial import Voronoi
    vor = Voronoi(seeds)
    vertices, vert_map, vertex2joint = {}, {}, defaultdict(set)
    cells, alpha = [], 0
    for region in vor.regions:
        ok = bool(region)
        for index in region:
            if index == -1:
                ok = False
                break
            x, y = vor.vertices[index]
            if x <= -0.5 - EPS or y <= -0.5 - EPS or x >= 1.5 + EPS or y >= 1.5 + EPS:
                ok = False
                break
        if not ok:
            continue
        cell = []
        for index in region:
            point = (round(vor.vertices[index][0] % 1, 4), round(vor.vertices[index][1] % 1, 4))
            if point not in vert_map:
                vert_map[point] = len(vert_map)
                vertices[vert_map[point]] = point
            cell.append(vert_map[point])
        key = tuple(sorted(cell))
        if key in cells:
            continue               # a periodic copy of a cell already taken
        cells.append(key)
        alpha += 1
        for v in cell:
            vertex2joint[v].add(alpha)

    # vertices shared by four cells after the rounding: split (graph_datastruct.py:430-461)
    quadruples = {}
    for k, v in vertex2joint.copy().items():
        if len(v) > 3:
            grains = list(v)
            new = len(vertex2joint)
            first = grains[0]
            v.remove(first)
            vertex2joint[new] = v.copy()
            v.add(first)
            vertices[new] = vertices[k]
            n1 = cells[first - 1]
            for other in grains[1:]:
                if len(set(n1).intersection(set(cells[other - 1]))) == 1:
                    remove = other
                    break
            v.remove(remove)
            vertex2joint[k] = v.copy()
            v.remove(first)
            v = list(v)
            quadruples.update({v[0]: (k, new), v[1]: (k, new)})

    joint2vertex = dict((tuple(sorted(v)), k) for k, v in vertex2joint.items())

    regions, region_coors = defaultdict(list), defaultdict(list)
    for k, v in joint2vertex.items():
        for region in set(k):          # (a set on purpose: its iteration order fixes the grains' order below)
            regions[region].append(v)
            region_coors[region].append(vertices[v])
    region_center, edges = {}, []
    for region, verts in region_coors.items():
        if len(verts) <= 1:
            continue
        in_region = regions[region]
        for i in range(1, len(in_region)):
            verts[i] = _min_image_onto(verts[i], verts[i - 1])
        inbound = [True, True]
        for vert in verts:
            inbound = [i and (j > -EPS) for i, j in zip(inbound, vert)]
        moved = [[i + 1 * (not j) for i, j in zip(vert, inbound)] for vert in verts]
        xs, ys = zip(*moved)
        region_center[region] = [np.mean(xs), np.mean(ys)]
        order = sorted(range(len(moved)), key=lambda t: _ccw_key(moved[t], region_center[region]))
        region_coors[region] = [moved[i] for i in order]
        ring = [in_region[i] for i in order]
        regions[region] = ring
        sides, keep = [], True
        for i in range(len(ring)):
            cur, nxt = ring[i], ring[i + 1] if i < len(ring) - 1 else ring[0]
            if region in quadruples and (cur in quadruples[region] or nxt in quadruples[region]):
                if len(set(vertex2joint[cur]).intersection(set(vertex2joint[nxt]))) != 2:
                    keep = False
            sides.append([cur, nxt])
        if not keep:
            v1, v2 = quadruples[region]
            for e in sides:
                for c in (0, 1):
                    if e[c] == v1:
                        e[c] = v2
                    elif e[c] == v2:
                        e[c] = v1
        edges.extend(sides)
    return vertices, joint2vertex, region_center, region_coors, edges


def raster_areas(region_coors, s):
    """plot_polygons (graph_datastruct.py:553-610, periodic): pixel count of every grain on the folded raster."""
    import PIL.Image as Image
    import PIL.ImageDraw as ImageDraw
    image = Image.new("RGB", (2 * s, 2 * s))
    draw = ImageDraw.Draw(image)
    for gid, poly in region_coors.items():
        r = gid // (255 * 255)
        g = (gid - r * 255 * 255) // 255
        b = gid - r * 255 * 255 - g * 255
        p = [tuple(np.asarray(np.array(pt) * s, dtype=int)) for pt in poly]
        if len(p) > 1:
            draw.polygon(p, fill=(r, g, b))
    img = np.array(image, dtype=int)
    img = img[:, :, 0] * 255 * 255 + img[:, :, 1] * 255 + img[:, :, 2]
    field = np.max(np.stack([img[:s, :s], img[s:, :s], img[:s, s:], img[s:, s:]]), axis=0)
    if not np.all(field > 0):
        raise RuntimeError("the rasterised grains do not cover the domain (the reference asserts here too)")
    ids, counts = np.unique(field, return_counts=True)
    return dict(zip(ids.tolist(), counts.tolist()))


def reference_sample(lxd: float = 40.0, seed: int = 0, G: float = 2.0, R: float = 0.4, span: int = None,
                     grain_size: float = 4.0, noise: float = 0.01):
    """-> (x_dict, edge_index_dict, edge_attr) numpy dicts: the reference's `--mode=generate` sample of `seed`
    (fp32 features / int64 edges, as data_loader.py:65-86 casts them)."""
    from .synthetic import EDGE_TYPES, span_for
    rs = np.random.RandomState(seed)       # == np.random.seed(seed) + the global legacy stream
    seeds = _lattice(grain_size / lxd, noise / lxd / (lxd / PATCH), rs)
    vertices, joint2vertex, centre, coors, edges = tessellate(seeds)
    n_g, n_j = len(coors), len(vertices)
    s_img = int(lxd / MESH) + 1
    counts = raster_areas(coors, s_img)
    ux, uy, uz = rs.randn(n_g), rs.randn(n_g), rs.randn(n_g)
    theta_x = np.arctan2(uy, ux) % (math.pi / 2)
    theta_z = np.arctan2(np.sqrt(ux ** 2 + uy ** 2), uz) % (math.pi / 2)
    if span is None:
        span = span_for(G, R)

    s_patch = int(np.round(PATCH / MESH)) + 1
    fg = np.zeros((n_g, 11))
    for grain, c in centre.items():
        fg[grain - 1, 0:2] = c
        fg[grain - 1, 3] = counts.get(grain, 0) / s_patch ** 2
    fg[:, 5], fg[:, 6], fg[:, 7], fg[:, 8] = np.cos(theta_x), np.sin(theta_x), np.cos(theta_z), np.sin(theta_z)
    fg[:, 9] = span / 120
    fj = np.zeros((n_j, 8))
    for joint, c in vertices.items():
        fj[joint, 0:2] = c
    fj[:, 3], fj[:, 4], fj[:, 5] = 1 - G / 10, R / 2, span / 120

    gj, gj_len = [], []
    for grains, joint in joint2vertex.items():
        for grain in grains:
            gj.append([grain - 1, joint])
            gj_len.append(_periodic_dist(vertices[joint], centre[grain]))
    jj = [[a, b] for a, b in edges if a > -1 and b > -1]
    jj_len = [_periodic_dist(vertices[a], vertices[b]) for a, b in jj]
    gj = np.array(gj, dtype=np.int64).T
    ei = {EDGE_TYPES[0]: np.ascontiguousarray(gj), EDGE_TYPES[1]: np.ascontiguousarray(gj[::-1]),
          EDGE_TYPES[2]: np.ascontiguousarray(np.array(jj, dtype=np.int64).T)}
    gl = np.array(gj_len, dtype=np.float32)[:, None]
    ea = {EDGE_TYPES[0]: gl, EDGE_TYPES[1]: gl.copy(), EDGE_TYPES[2]: np.array(jj_len, dtype=np.float32)[:, None]}
    x = {"grain": fg.astype(np.float32), "joint": fj.astype(np.float32)}
    return x, ei, ea
