#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d /tmp/valu -- python $R/tools/encoder_valu_budget.py --run > /tmp/valu.log 2>&1
tail -3 /tmp/valu.log
cd $R
python tools/encoder_valu_budget.py --report /tmp/valu | tee gpurun_out/r4_encoder_valu_budget.txt
