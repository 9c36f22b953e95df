#!/bin/bash
# Round evidence on the GPU box (gpurun -- bash tools/profile_round.sh TAG [pmc|stats|bench ...]): per-kernel PMC passes
# under both decoder plans (default = fused decoder cell; GGNN_DEC=split = projection + sweeps + gate GEMM, with
# tools/pmc_aggregate.py for its sweep), per-kernel rocprofv3 stats of the step in its launch modes, and the bench lines.
# Everything lands in gpurun_out/TAG/.  Sections (default: all three, in this order; a GPU call is limited to 20 minutes:
# `pmc` and `stats bench` fit one call each).
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repository copy on the GPU box)}" || exit 1
OUT=gpurun_out/${1:-round}
mkdir -p $OUT
shift || true
SECTIONS="${*:-pmc stats bench}"
want() { case " $SECTIONS " in *" $1 "*) return 0;; *) return 1;; esac; }
if want pmc; then
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
done
unset GGNN_DEC
# the fused plan's PMC passes on the reference generator's own structure (bench.py --workload gen368: nodes in Qhull order)
mkdir -p $OUT/pmc_gen368
i=0
for g in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/pmc_gen368/$i -- python3 bench.py --workload gen368 --steps 5 --warmup 2 --profile --no-graph --serial > $OUT/pmc_gen368_$i.log 2>&1
done
python3 - $OUT gen368 <<'PY'
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
rm -rf $OUT/pmc_gen368
cp $OUT/r6_pmc_aggregate_sweep.json profiles/r6_pmc_aggregate_sweep.json 2>/dev/null
python3 tools/pmc_kernels.py $OUT/pmc_split_summary.csv $OUT/pmc_fused_summary.csv > $OUT/pmc_kernels.log 2>&1
cp profiles/r6_pmc_kernels.json $OUT/r6_pmc_kernels.json
cat $OUT/pmc_aggregate.log; tail -3 $OUT/pmc_kernels.log
fi
if want stats; then
for plan in fused split; do
  export GGNN_DEC=$plan
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial_$plan -- python3 bench.py --steps 48 --warmup 4 --profile --no-graph --serial > $OUT/serial_$plan.log 2>&1
  find $OUT/serial_$plan -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/serial_${plan}_kernel_stats.csv
  rm -rf $OUT/serial_$plan
done
unset GGNN_DEC
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gen368 -- python3 bench.py --workload gen368 --steps 48 --warmup 4 --profile --no-graph --serial > $OUT/gen368.log 2>&1
find $OUT/gen368 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/serial_fused_gen368_kernel_stats.csv
rm -rf $OUT/gen368
for mode in joint two-streams; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -- python3 bench.py --steps 48 --warmup 4 --profile --no-graph --$mode > $OUT/$mode.log 2>&1
  find $OUT/$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${mode}_kernel_stats.csv
  rm -rf $OUT/$mode
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -- python3 bench.py --profile > $OUT/default.log 2>&1
find $OUT/default -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/default_command_kernel_stats.csv
rm -rf $OUT/default
python3 tools/kernel_table.py $OUT/serial_fused_kernel_stats.csv $OUT/serial_split_kernel_stats.csv $OUT/joint_kernel_stats.csv > $OUT/kernel_table.txt
cat $OUT/kernel_table.txt
fi
if want bench; then
timeout 700 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
GGNN_DEC=split timeout 400 python3 bench.py --no-cpu-baseline > $OUT/bench_split_plan.json 2> $OUT/bench_split_plan.err
timeout 400 python3 bench.py --workload gen368 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_gen368.json 2> $OUT/bench_gen368.err
timeout 400 python3 bench.py --workload cfg2 --steps 120 --warmup 10 --no-cpu-baseline > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
timeout 400 python3 bench.py --workload cfg4 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
for f in default driver_command split_plan gen368 cfg2 cfg4; do tail -1 $OUT/bench_$f.json | cut -c1-200; done
fi
