#!/bin/bash
# Development aid: build an ablation variant of libggnn.so.
#   tools/build_variant.sh NAME "-DMACRO1 -DMACRO2"   ->  graingraphnn_amd/csrc/build/variants/libggnn_NAME.so
# Select it at run time with GGNN_LIB_PATH=<that file> (see graingraphnn_amd/_lib.py).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/graingraphnn_amd/csrc
OUT=$SRC/build/variants
mkdir -p "$OUT/obj_$1"
for f in abi csr project project_x6 aggregate aggregate_bwd gates gates_x6 heads step; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" $2 -c "$SRC/$f.hip" -o "$OUT/obj_$1/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libggnn_$1.so" "$OUT"/obj_$1/*.o
rm -rf "$OUT/obj_$1"
echo "$OUT/libggnn_$1.so"
