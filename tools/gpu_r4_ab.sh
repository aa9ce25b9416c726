#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
run() { echo -n "$1 $2 $3 "; env $1 timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/$2 $3 2>&1 | tail -1 | sed 's/np.float64(\([0-9.]*\))/\1/g'; }
{
for l in libgpujpeg_r3.so libgpujpeg.so libgpujpeg_noticket.so libgpujpeg_nogroup.so libgpujpeg_noticket_pf0.so libgpujpeg_r3.so libgpujpeg.so; do run X=0 $l 8k; done
run GJ_ENC_RESIDENT=2058 libgpujpeg_noticket.so 8k
run GJ_ENC_RESIDENT=2058 libgpujpeg_noticket_pf0.so 8k
} > gpurun_out/r4_ab.txt 2>&1
cat gpurun_out/r4_ab.txt
