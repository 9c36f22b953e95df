#!/bin/bash
# A/B of the decoder plans in the rollout (gpurun -- bash tools/ab_decoder_plans.sh) -> gpurun_out/ab/ab.txt
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
mkdir -p gpurun_out/ab
for rep in 1 2; do
for mode in split fused fused-classifier fused-regressor; do
  export GGNN_DEC=$mode
  python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', '$rep', d['value'])"
done
done | tee gpurun_out/ab/ab.txt
