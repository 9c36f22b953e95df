# Development aid: LDS / vector-memory / texture-path counters of the two cell kernels (rocprofv3 --pmc, one pass per group,
# --kernel-trace only).   gpurun -- bash tools/probes/pmc_cells.sh
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repository copy on the GPU box)}" || exit 1
OUT=gpurun_out/pmccells
mkdir -p $OUT
i=0
for g in "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
         "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/p$i -- python3 bench.py --steps 5 --warmup 2 --profile --no-graph --serial > $OUT/pmc_$i.log 2>&1
  tail -1 $OUT/pmc_$i.log | cut -c1-120
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
with open(os.path.join(out, "cells_summary.csv"), "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for (kn, cn), (s, n) in sorted(acc.items()):
        if "dec_cell" in kn or "enc_cell" in kn or "project_x6" in kn:
            f.write(f"{kn},{cn},{s / n:.1f},{n}\n")
print(open(os.path.join(out, "cells_summary.csv")).read())
PY
rm -rf $OUT/p[0-9]*
