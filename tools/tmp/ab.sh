V=graingraphnn_amd/csrc/build/variants
for v in b24 b36 b35 b24 b36; do
  if [ -n "$v" ]; then export GGNN_LIB_PATH=$PWD/$V/libggnn_$v.so; else unset GGNN_LIB_PATH; fi
  echo "== variant '$v'"; python3 tools/kbench.py --reps 30 | grep aggregate
done
for v in "" b36 "" b36; do
  if [ -n "$v" ]; then export GGNN_LIB_PATH=$PWD/$V/libggnn_$v.so; else unset GGNN_LIB_PATH; fi
  echo "== bench '$v'"; python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c60-110
done
