#!/bin/bash
# Development aid: libggnn_<name>.so = the product objects with ONE source recompiled under extra flags.
#   tools/mkvariant.sh name source.hip -DFLAG ...     (run after the normal build; not part of the product)
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/graingraphnn_amd/csrc"
name=$1; src=$2; shift 2
base=${src%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -Wall -Wno-unused-function -fno-slp-vectorize "$@" -c "$src" -o "build/${name}_${base}.o"
objs=$(ls build/*.o | grep -v -e "build/[a-z0-9]*_${base}\.o" -e "^build/${base}\.o" | grep -E "^build/(abi|csr|project|project_x6|aggregate|aggregate_enc|enc_cell|dec_cell|aggregate_bwd|gates|gates_x6|heads|step|lstm_train|wgrad|rowgemm|train_misc|pack|topology)\.o$" || true)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/graingraphnn_amd/libggnn_${name}.so" $objs "build/${name}_${base}.o"
echo "built libggnn_${name}.so"
