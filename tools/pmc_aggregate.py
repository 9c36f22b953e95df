#!/usr/bin/env python3
"""Derive the memory-side traffic of the aggregation sweeps from rocprofv3 PMC passes.

On the GPU box (one counter group per pass; rocprofv3 must be followed directly by the program):
    bash tools/profile_round.sh TAG          # runs the passes below, then this script
    for g in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
        rocprofv3 --kernel-trace --pmc $g --output-format csv -d gpurun_out/TAG/pmc/$i -- \
            python3 bench.py --steps 5 --warmup 2 --profile --no-graph --serial
    done
then:  python tools/pmc_aggregate.py gpurun_out/TAG/pmc gpurun_out/TAG/r2_pmc_aggregate.json
The record is stamped with bench.kernel_source_hash() (ABI version + hash of the sweep's sources) and
the sweeps per launch of the traced command; bench.py only quotes it for the kernel it describes.
"""
import csv
import glob
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"decoder": "aggregate_kernel<4, true>", "encoder": "aggregate_enc_kernel<3>"}


def collect(root, kernel):
    sums, counts = {}, {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel not in row["Kernel_Name"]:
                    continue
                k = row["Counter_Name"]
                sums[k] = sums.get(k, 0.0) + float(row["Counter_Value"])
                counts[k] = counts.get(k, 0) + 1
    return {k: {"mean_per_launch": sums[k] / counts[k], "launches": counts[k]} for k in sorted(sums)}


def traffic(c):
    m = lambda k: c[k]["mean_per_launch"]
    reads = 128 * m("TCC_EA0_RDREQ_128B_sum") + 64 * m("TCC_EA0_RDREQ_64B_sum") + 32 * m("TCC_EA0_RDREQ_32B_sum")
    writes = 64 * m("TCC_EA0_WRREQ_64B_sum") + 32 * (m("TCC_EA0_WRREQ_sum") - m("TCC_EA0_WRREQ_64B_sum"))
    return reads, writes, {"FETCH_SIZE_KB_x2_gfx950_rule": 2 * 1024 * m("FETCH_SIZE") if "FETCH_SIZE" in c else None,
                           "WRITE_SIZE_KB": 1024 * m("WRITE_SIZE") if "WRITE_SIZE" in c else None}


def main(root, out):
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    c = collect(root, KERNELS["decoder"])
    reads, writes, cross = traffic(c)
    doc = {
        "kernel": "ggnn::" + KERNELS["decoder"],
        "kernel_source_hash": bench.kernel_source_hash(),
        "sweeps_per_launch": 2.5,
        "workload": "cfg3 decoder-cell sweep launches of bench.py, one set of launches per model (regressor: g->j, "
                    "j->g, j->j in one launch; classifier: g->j, j->j; averaged over the launches as bench.py's "
                    "roofline does): bench.py --steps 5 --warmup 2 --profile --no-graph --serial under rocprofv3 "
                    "--kernel-trace --pmc <one group per pass>",
        "counters": c,
        "read_bytes_per_launch": reads, "write_bytes_per_launch": writes,
        "traffic_bytes_per_launch": reads + writes,
        "cross_check": cross,
        "method": "HBM-side bytes = 128*TCC_EA0_RDREQ_128B + 64*TCC_EA0_RDREQ_64B + 32*TCC_EA0_RDREQ_32B (reads) + "
                  "64*TCC_EA0_WRREQ_64B + 32*(TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B) (writes); FETCH_SIZE doubled per "
                  "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B) is the cross-check",
        "algorithmic_bytes_per_launch": ((93720004 + 93360004 + 124560004) + (93720004 + 124560004)) / 2,
    }
    ce = collect(root, KERNELS["encoder"])
    if ce:
        r, w, x = traffic(ce)
        doc["encoder_sweep"] = {"kernel": "ggnn::" + KERNELS["encoder"], "counters": ce, "read_bytes_per_launch": r,
                                "write_bytes_per_launch": w, "traffic_bytes_per_launch": r + w, "cross_check": x,
                                "note": "three sweeps (one model's encoder cell) per launch"}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: doc[k] for k in ("kernel_source_hash", "read_bytes_per_launch", "write_bytes_per_launch",
                                           "traffic_bytes_per_launch", "cross_check")}))
    if ce:
        print(json.dumps({k: doc["encoder_sweep"][k] for k in ("read_bytes_per_launch", "write_bytes_per_launch",
                                                                "traffic_bytes_per_launch")}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
