#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
timeout -k 10 300 python tools/probes/fuzz63.py 2>&1 | grep -v amdgpu.ids
