#!/bin/bash
# Development aid: device ISA + resource usage of one source (tools/isa.sh dec_cell.hip [-DFLAGS]) -> /tmp/<name>.s
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
src=$1; shift
out=/tmp/$(basename "${src%.hip}").s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I"$ROOT/include" -Wno-unused-function -fno-slp-vectorize --cuda-device-only -S "$ROOT/graingraphnn_amd/csrc/$src" -o "$out" "$@" 2>&1 | grep -v "hip-link" || true
grep -E "\.(name|vgpr_count|agpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):" "$out" | paste - - - - - - | sed 's/  */ /g'
