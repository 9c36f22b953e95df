#!/bin/bash
# Round-end evidence run on the GPU box (gpurun -- bash tools/profile_round.sh): PMC passes for
# tools/pmc_aggregate.py, the rocprofv3 kernel-stats profiles kept under profiles/, and a default bench line.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests -m gpu -x -q -k "batched_sweeps or golden" 2>&1 | tail -2
i=0
for g in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d gpurun_out/pmc3/$i -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --serial > gpurun_out/pmc3_$i.log 2>&1
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof15 -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-graph --serial > gpurun_out/prof15.log 2>&1
tail -1 gpurun_out/prof15.log > gpurun_out/prof15_bench.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof15d -- python3 bench.py --no-cpu-baseline > gpurun_out/prof15d.log 2>&1
python3 bench.py > gpurun_out/bench15.json 2>gpurun_out/bench15.err; tail -1 gpurun_out/bench15.json
