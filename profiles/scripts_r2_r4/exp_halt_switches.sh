#!/bin/bash
# the one-off ~60 ms all-queue stall a few hundred frames into a new stream: which runtime resource is it?
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r4_halt.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout 120 python tools/exp_ramp.py hd 1200 ${ST:-1} ${SN:-4} >> $O 2>&1; }
run X=1
ST=0 run X=1
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=8
run ROC_AQL_QUEUE_SIZE=4096
run ROC_AQL_QUEUE_SIZE=65536
run ROC_SIGNAL_POOL_SIZE=16384
run HSA_KERNARG_POOL_SIZE=16777216
run HIP_FORCE_DEV_KERNARG=0
run HSA_NO_SCRATCH_RECLAIM=1
run DEBUG_HIP_DYNAMIC_QUEUES=0
SN=1 run X=1
cat $O
