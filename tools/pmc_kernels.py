#!/usr/bin/env python3
"""HBM-side traffic per launch of the step's heavy kernels, from the per-kernel PMC summaries that
tools/profile_decoder.sh writes (separate rocprofv3 --pmc passes over `bench.py --steps 5 --warmup 2 --profile
--no-graph --serial`; bytes = 128 RDREQ_128B + 64 RDREQ_64B + 32 RDREQ_32B (+ 64 x the rest) read, 64 WRREQ_64B
+ 32 x the rest written: the request counters of the L2's memory side, MI355X_MICROARCH.md section HBM) ->
profiles/r6_pmc_kernels.json, stamped with the ABI version and a hash of the kernels' sources.  bench.py quotes
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
        raw = list(csv.DictReader(open(p)))
        if raw and "counter" in raw[0]:   # tools/profile_round.sh's long format: kernel, counter, mean_per_launch, launches
            per = {}
            for r in raw:
                per.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean_per_launch"])
            raw = []
            for kn, c in per.items():
                g = lambda k: c.get(k, 0.0)
                rd = 32 * g("TCC_EA0_RDREQ_32B_sum") + 64 * g("TCC_EA0_RDREQ_64B_sum") + 128 * g("TCC_EA0_RDREQ_128B_sum")
                rd += 64 * max(g("TCC_EA0_RDREQ_sum") - g("TCC_EA0_RDREQ_32B_sum") - g("TCC_EA0_RDREQ_64B_sum") - g("TCC_EA0_RDREQ_128B_sum"), 0)
                wr = 64 * g("TCC_EA0_WRREQ_64B_sum") + 32 * max(g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum"), 0)
                wc = max(g("SQ_WAVE_CYCLES"), 1.0)
                raw.append({"kernel": kn, "read_MB_per_launch": rd / 1e6, "write_MB_per_launch": wr / 1e6,
                            "valu_per_mfma": g("SQ_INSTS_VALU") / max(g("SQ_INSTS_MFMA"), 1.0),
                            "mfma_busy_Mcycles": g("SQ_VALU_MFMA_BUSY_CYCLES") / 1e6,
                            "wait_any_frac": g("SQ_WAIT_ANY") / wc, "wait_inst_frac": g("SQ_WAIT_INST_ANY") / wc})
        for r in raw:
            if float(r["read_MB_per_launch"]) + float(r["write_MB_per_launch"]) < 5:
                continue
            rows[r["kernel"]] = {"read_bytes": int(float(r["read_MB_per_launch"]) * 1e6),
                                 "written_bytes": int(float(r["write_MB_per_launch"]) * 1e6),
                                 "valu_per_mfma": float(r["valu_per_mfma"]) if float(r["mfma_busy_Mcycles"]) > 0 else None,
                                 "mfma_busy_cycles": int(float(r["mfma_busy_Mcycles"]) * 1e6),
                                 "wait_any_frac": float(r["wait_any_frac"]), "wait_inst_frac": float(r["wait_inst_frac"])}
        doc["plans"][plan] = rows
    out = os.path.join(ROOT, "profiles", "r6_pmc_kernels.json")
    if os.path.exists(out) and len(doc["plans"]) < 2:   # keep the other plan's record of an earlier run (its own stamp)
        try:
            old = json.load(open(out))
            for plan, rows in old.get("plans", {}).items():
                if plan not in doc["plans"]:
                    doc["plans"][plan] = rows
                    doc.setdefault("other_plan_stamp", {})[plan] = old.get("kernel_source_hash")
        except (OSError, ValueError):
            pass
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(out, {k: list(v) for k, v in doc["plans"].items()})


if __name__ == "__main__":
    main(sys.argv[1:])
