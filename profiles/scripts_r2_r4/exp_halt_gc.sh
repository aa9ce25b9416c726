#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r4_halt_gc.txt; : > $O
for m in default off freeze; do EXP_GC=$m timeout 120 python tools/exp_ramp.py hd 1200 0 4 2>&1 | grep -v amdgpu.ids >> $O; done
EXP_GC=default timeout 120 python tools/exp_ramp.py hd 1200 0 1 2>&1 | grep -v amdgpu.ids >> $O
EXP_GC=off timeout 120 python tools/exp_ramp.py 8k 600 0 4 2>&1 | grep -v amdgpu.ids >> $O
cat $O
