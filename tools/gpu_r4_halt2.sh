#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r4_halt_log.txt; : > $O
AMD_LOG_LEVEL=4 AMD_LOG_LEVEL_FILE=/tmp/amdlog timeout 300 python tools/exp_ramp.py hd 330 0 4 >> $O 2>&1
for f in /tmp/amdlog*; do python tools/halt_gap.py $f 15 70 >> $O 2>&1; done
tail -c 3000 $O
