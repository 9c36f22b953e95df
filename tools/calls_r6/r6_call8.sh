#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6h
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
rc=$?
echo "pytest rc $rc"; tail -4 $OUT/pytest.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED" $OUT/pytest.log | head; exit $rc; fi
GGNN_VLAYOUT=rows timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "golden or cfg3_ten or cfg4 or pipelined or fused_decoder" > $OUT/pytest_rows.log 2>&1
echo "pytest (rows layout) rc $?"; tail -2 $OUT/pytest_rows.log
bash tools/profile_round.sh r6_v2 pmc
