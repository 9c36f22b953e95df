mkdir -p gpurun_out/ab
for rep in 1 2; do
for mode in split fused fused-classifier fused-regressor; do
  if [ $mode = split ]; then unset GGNN_DEC; else export GGNN_DEC=$mode; fi
  python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$mode', '$rep', d['value'])"
done
done | tee gpurun_out/ab/ab.txt
export GGNN_DEC=fused-classifier GGNN_DC_KERNEL=ws
python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fused-classifier ws', d['value'])" | tee -a gpurun_out/ab/ab.txt
