#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
{ for p in 2 4 5 6; do for run in 1 2 3; do echo "pattern $p, process $run:"; gpujpeg_amd/lib/ubench_hlp $p 2>&1 | tr '\n' ' ' | sed 's/GB\/s each way//g; s/thread(s)://g'; echo; done; done; } | tee gpurun_out/r5_bimodal3.txt
