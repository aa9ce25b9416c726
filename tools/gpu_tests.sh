#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x -n 4 2>&1 | tail -15 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_gpu_tests.txt
