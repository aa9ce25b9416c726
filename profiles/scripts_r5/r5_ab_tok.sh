#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
tools/r5_ab.sh "prev default prev default" "natural" 2>&1 | grep -v amdgpu.ids | grep "==\|decode_tok\|four" | tee gpurun_out/r5_ab_tok.txt
