#!/usr/bin/env python3
"""Fixtures from the reference's own initial-structure generator.

Runs, in THIS container only (needs /root/reference), the unmodified
    python graph_trajectory.py --mode=generate --lxd=40 --seed=S --save_dir=<tmp>
(graph_trajectory.py:1289-1333 -> graph_datastruct.py:118-160, 350-465 for the tessellation,
graph_trajectory.py:901-1005 for the features) with the import stubs of tools/oracle_stub (h5py,
termcolor) on the path, and stores what it pickled -- node features, the three edge lists, the edge
lengths -- as tests/golden/generated_40_seed<S>.npz.  Data only; no reference source is stored.

    python tests/golden/make_golden_generated.py [seed ...]        (default: seeds 1, 2 at 40 um and
                                                                    seed 7 at 80 um with G = 10, R = 0.2)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
STUB = os.path.join(ROOT, "tools", "oracle_stub")


def run(seed, lxd=40, G=None, R=None):
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, PYTHONPATH=STUB, MPLBACKEND="Agg")
        extra = ([f"--G={G}"] if G is not None else []) + ([f"--R={R}"] if R is not None else [])
        subprocess.run([sys.executable, "graph_trajectory.py", "--mode=generate", f"--lxd={lxd}", f"--seed={seed}",
                        f"--save_dir={tmp}/"] + extra, cwd=REF, env=env, check=True, stdout=subprocess.DEVNULL)
        pkl = [f for f in os.listdir(tmp) if f.startswith(f"seed{seed}_")][0]
        sys.path[:0] = [STUB, REF]
        os.environ.setdefault("MPLBACKEND", "Agg")
        import dill
        with open(os.path.join(tmp, pkl), "rb") as f:
            g = dill.load(f)[0]
    d = {"x_grain": np.asarray(g.feature_dicts["grain"]).astype(np.float32),
         "x_joint": np.asarray(g.feature_dicts["joint"]).astype(np.float32),
         "span": np.float32(g.span), "G": np.float32(g.physical_params["G"]), "R": np.float32(g.physical_params["R"])}
    for et, v in g.edge_index_dicts.items():
        d["ei_" + "__".join(et)] = np.asarray(v).astype(np.int64)
        d["ea_" + "__".join(et)] = np.asarray(g.edge_weight_dicts[et]).astype(np.float32)
    out = os.path.join(HERE, f"generated_{lxd}_seed{seed}.npz")
    np.savez_compressed(out, **d)
    print(out, pkl, {k: getattr(v, "shape", v) for k, v in d.items()})


if __name__ == "__main__":
    if sys.argv[1:]:
        for s in (int(a) for a in sys.argv[1:]):
            run(s)
    else:
        run(1)
        run(2)
        run(7, lxd=80, G=10.0, R=0.2)   # four patches, another span of the (G, R) table (60)
