#!/bin/bash
# GPU box: in-situ kernel durations of a serialised rollout step + SQ counters of the gate kernel.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-pg}
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 50 --warmup 5 --profile --no-graph --serial > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
i=0
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/pmc$i -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --serial --profile > $OUT/pmc$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, sys, collections
out = os.environ.get("OUT_DIR") or sorted(glob.glob("gpurun_out/*/pmc1"))[-1].rsplit("/", 1)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = (row["Kernel_Name"][:90], row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
with open(os.path.join(out, "pmc_summary.csv"), "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for (kn, cn), (s, n) in sorted(acc.items()):
        f.write(f"\"{kn}\",{cn},{s / n:.1f},{n}\n")
PY
head -25 $OUT/kernel_stats.csv | cut -c1-160
