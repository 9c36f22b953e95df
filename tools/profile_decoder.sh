#!/bin/bash
# Decoder cell, fused vs split (gpurun -- bash tools/profile_decoder.sh TAG): per-kernel rocprofv3 stats and
# HBM-side request counters (separate --pmc passes) of `bench.py --profile --no-graph --serial` with the decoder
# as one fused kernel (GGNN_DEC=fused) and as projection + sweeps + gate GEMM (default).  -> gpurun_out/TAG/
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repository copy on the GPU box)}" || exit 1
OUT=gpurun_out/${1:-decoder}
mkdir -p $OUT
for mode in fused split; do
  export GGNN_DEC=$mode
  i=0
  for g in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/pmc_$mode/$i -- python3 bench.py --steps 5 --warmup 2 --profile --no-graph --serial > $OUT/pmc_${mode}_$i.log 2>&1
  done
  python3 - $OUT $mode <<'PY'
import csv, glob, os, sys, collections
out, mode = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(out, "pmc_" + mode, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
rows = {}
for (kn, cn), (s, n) in acc.items():
    if "ggnn" in kn:
        rows.setdefault(kn, {})[cn] = s / n
with open(os.path.join(out, f"pmc_{mode}_summary.csv"), "w") as f:
    f.write("kernel,read_MB_per_launch,write_MB_per_launch,valu_per_mfma,mfma_busy_Mcycles,wait_any_frac,wait_inst_frac\n")
    for kn, c in sorted(rows.items()):
        rd = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0)
        rest = c.get("TCC_EA0_RDREQ_sum", 0) - c.get("TCC_EA0_RDREQ_32B_sum", 0) - c.get("TCC_EA0_RDREQ_64B_sum", 0) - c.get("TCC_EA0_RDREQ_128B_sum", 0)
        rd += 64 * max(rest, 0)
        wr = 64 * c.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * max(c.get("TCC_EA0_WRREQ_sum", 0) - c.get("TCC_EA0_WRREQ_64B_sum", 0), 0)
        wc = max(c.get("SQ_WAVE_CYCLES", 0), 1)
        f.write(f"\"{kn}\",{rd / 1e6:.1f},{wr / 1e6:.1f},{c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_INSTS_MFMA', 0), 1):.2f},"
                f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1e6:.2f},{c.get('SQ_WAIT_ANY', 0) / wc:.3f},{c.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}\n")
PY
  rm -rf $OUT/pmc_$mode
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$mode -- python3 bench.py --steps 48 --warmup 4 --profile --no-graph --serial > $OUT/stats_$mode.log 2>&1
  find $OUT/stats_$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${mode}_serial_kernel_stats.csv
  rm -rf $OUT/stats_$mode
  python3 tools/kernel_table.py $OUT/${mode}_serial_kernel_stats.csv > $OUT/${mode}_kernel_table.txt
done
unset GGNN_DEC
for i in 1 2 3; do
  for mode in fused split; do
    GGNN_DEC=$mode python3 bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', l['value'], 'steps/s', l['ms_per_step'], 'ms/step')" >> $OUT/ab_steps_per_s.txt
  done
done
cat $OUT/pmc_fused_summary.csv $OUT/pmc_split_summary.csv $OUT/ab_steps_per_s.txt
