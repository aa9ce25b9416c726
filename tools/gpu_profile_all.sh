#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
( time tools/profile_all.sh ) 2>&1 | grep -v amdgpu.ids | tail -60
ls gpurun_out | grep r4_ | head -40
