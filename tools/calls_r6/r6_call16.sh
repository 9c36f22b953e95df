#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6p
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
rc=$?
echo "pytest rc $rc"; tail -4 $OUT/pytest.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED" $OUT/pytest.log | head; exit $rc; fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python tests/fuzz_forward.py --n 12 --seed 6 > $OUT/fuzz_forward.log 2>&1; echo "fuzz_forward rc $?"; tail -2 $OUT/fuzz_forward.log
bash tools/profile_round.sh r6_v3 pmc
