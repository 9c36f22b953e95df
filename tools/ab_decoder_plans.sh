mkdir -p gpurun_out/ab
for rep in 1 2; do
for mode in split fused fused-classifier fused-regressor; do
  export GGNN_DEC=$mode
  python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$mode', '$rep', d['value'])"
done
done | tee gpurun_out/ab/ab.txt
