#!/bin/bash
# Round evidence on the GPU box (gpurun -- bash tools/profile_round.sh TAG): per-kernel PMC passes under both decoder
# plans (default = fused decoder cell; GGNN_DEC=split = projection + sweeps + gate GEMM, with tools/pmc_aggregate.py
# for its sweep), per-kernel rocprofv3 stats of the step in its launch modes, and the bench lines.
# Everything lands in gpurun_out/TAG/.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repository copy on the GPU box)}" || exit 1
OUT=gpurun_out/${1:-round}
mkdir -p $OUT
for plan in fused split; do
  export GGNN_DEC=$plan
  mkdir -p $OUT/pmc_$plan
  i=0
  for g in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/pmc_$plan/$i -- python3 bench.py --steps 5 --warmup 2 --profile --no-graph --serial > $OUT/pmc_${plan}_$i.log 2>&1
  done
  if [ $plan = split ]; then python3 tools/pmc_aggregate.py $OUT/pmc_split $OUT/r6_pmc_aggregate_sweep.json > $OUT/pmc_aggregate.log 2>&1; fi
  python3 - $OUT $plan <<'PY'
import csv, glob, os, sys, collections
out, plan = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(out, "pmc_" + plan, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
with open(os.path.join(out, f"pmc_{plan}_summary.csv"), "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for (kn, cn), (s, n) in sorted(acc.items()):
        if "ggnn" in kn:
            f.write(f"\"{kn}\",{cn},{s / n:.1f},{n}\n")
PY
  rm -rf $OUT/pmc_$plan
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial_$plan -- python3 bench.py --steps 48 --warmup 4 --profile --no-graph --serial > $OUT/serial_$plan.log 2>&1
  find $OUT/serial_$plan -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/serial_${plan}_kernel_stats.csv
  rm -rf $OUT/serial_$plan
done
unset GGNN_DEC
for mode in joint two-streams; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -- python3 bench.py --steps 48 --warmup 4 --profile --no-graph --$mode > $OUT/$mode.log 2>&1
  find $OUT/$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${mode}_kernel_stats.csv
  rm -rf $OUT/$mode
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -- python3 bench.py --profile > $OUT/default.log 2>&1
find $OUT/default -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/default_command_kernel_stats.csv
rm -rf $OUT/default
python3 tools/kernel_table.py $OUT/serial_fused_kernel_stats.csv $OUT/serial_split_kernel_stats.csv $OUT/joint_kernel_stats.csv > $OUT/kernel_table.txt
cp $OUT/r6_pmc_aggregate_sweep.json profiles/r6_pmc_aggregate_sweep.json 2>/dev/null   # so that the bench lines below quote them
python3 tools/pmc_kernels.py $OUT/pmc_split_summary.csv $OUT/pmc_fused_summary.csv > $OUT/pmc_kernels.log 2>&1
cp profiles/r6_pmc_kernels.json $OUT/r6_pmc_kernels.json
timeout 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
GGNN_DEC=split timeout 600 python3 bench.py --no-cpu-baseline > $OUT/bench_split_plan.json 2> $OUT/bench_split_plan.err
tail -1 $OUT/bench_default.json | cut -c1-300
cat $OUT/pmc_aggregate.log
