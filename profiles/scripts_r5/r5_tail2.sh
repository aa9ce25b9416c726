#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
tools/r5_ab.sh "default tail16 tail32 tail64 tail96 tail128" "natural camera" 2>&1 | grep -v amdgpu.ids | grep "==\|k_encode\|four" | tee gpurun_out/r5_tail2.txt
