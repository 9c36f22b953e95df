#!/usr/bin/env python3
"""HBM-side traffic per launch of the step's heavy kernels, from the per-kernel PMC summaries that
tools/profile_decoder.sh writes (separate rocprofv3 --pmc passes over `bench.py --steps 5 --warmup 2 --profile
--no-graph --serial`; bytes = 128 RDREQ_128B + 64 RDREQ_64B + 32 RDREQ_32B (+ 64 x the rest) read, 64 WRREQ_64B
+ 32 x the rest written: the request counters of the L2's memory side, MI355X_MICROARCH.md section HBM) ->
profiles/r3_pmc_kernels.json, stamped with the ABI version and a hash of the kernels' sources.  bench.py quotes
a kernel's traffic only from a record whose stamp matches the library it runs.

    python tools/pmc_kernels.py gpurun_out/TAG/pmc_split_summary.csv [gpurun_out/TAG/pmc_fused_summary.csv]
"""
import csv
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(paths):
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    doc = {"kernel_source_hash": bench.kernel_source_hash(),
           "how": "rocprofv3 --kernel-trace --pmc <one counter group per pass> -- python3 bench.py --steps 5 --warmup 2 "
                  "--profile --no-graph --serial (tools/profile_decoder.sh); per launch, averaged over the regressor's "
                  "and the classifier's launches as bench.py's event timing is",
           "plans": {}}
    for p in paths:
        plan = "fused" if "fused" in os.path.basename(p) else "split"
        rows = {}
        for r in csv.DictReader(open(p)):
            if float(r["read_MB_per_launch"]) + float(r["write_MB_per_launch"]) < 5:
                continue
            rows[r["kernel"]] = {"read_bytes": int(float(r["read_MB_per_launch"]) * 1e6),
                                 "written_bytes": int(float(r["write_MB_per_launch"]) * 1e6),
                                 "valu_per_mfma": float(r["valu_per_mfma"]) if float(r["mfma_busy_Mcycles"]) > 0 else None,
                                 "mfma_busy_cycles": int(float(r["mfma_busy_Mcycles"]) * 1e6),
                                 "wait_any_frac": float(r["wait_any_frac"]), "wait_inst_frac": float(r["wait_inst_frac"])}
        doc["plans"][plan] = rows
    out = os.path.join(ROOT, "profiles", "r3_pmc_kernels.json")
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(out, {k: list(v) for k, v in doc["plans"].items()})


if __name__ == "__main__":
    main(sys.argv[1:])
