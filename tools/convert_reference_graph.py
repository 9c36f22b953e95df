#!/usr/bin/env python3
"""Convert a reference graph pickle (graphs/*/seed*_G*_R*_span*.pkl, a dill dump of
[GrainHeterograph, ...], graph_datastruct.py:825-849) to the neutral `.npz` the product reads
(`synthetic.load_fixture`): float32 features, int64 edge lists, float32 edge lengths, exactly as
data_loader.py:65,79,86 casts them.  Needs the reference checkout (for the pickled classes) and
therefore only runs where it is mounted:

    python tools/convert_reference_graph.py /root/reference graphs/40_40/seed10020_G1.904_R0.558_span6.pkl out.npz
"""
import os
import sys

import numpy as np


def main(ref, pkl, out):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "tools", "oracle_stub"), ref]
    os.environ.setdefault("MPLBACKEND", "Agg")
    import dill
    with open(pkl if os.path.isabs(pkl) else os.path.join(ref, pkl), "rb") as f:
        g = dill.load(f)[0]
    d = {"x_grain": np.asarray(g.feature_dicts["grain"]).astype(np.float32),
         "x_joint": np.asarray(g.feature_dicts["joint"]).astype(np.float32)}
    for et, v in g.edge_index_dicts.items():
        d["ei_" + "__".join(et)] = np.asarray(v).astype(np.int64)
        d["ea_" + "__".join(et)] = np.asarray(g.edge_weight_dicts[et]).astype(np.float32)
    if hasattr(g, "mask"):
        for k, v in g.mask.items():
            d["mask_" + k] = np.asarray(v).astype(np.int64)
    np.savez_compressed(out, **d)
    print(out, {k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    if len(sys.argv) != 4:
        sys.exit(__doc__)
    main(*sys.argv[1:])
