#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
T=${TAG:-r4h}
timeout 600 python -m pytest tests -m gpu -x -q -k "tiles_and_gather or full_size" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -3 gpurun_out/${T}_pytest.log
run() { echo -n "$1 $2 $3 "; env $1 timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/$2 $3 2>&1 | tail -1 | sed 's/np.float64(\([0-9.]*\))/\1/g'; }
{
for w in 8k 4k hd; do run X=0 libgpujpeg.so $w; done
for st in 1 2 3 4 6 8; do run GJ_ENC_STAGGER=$st libgpujpeg.so 8k; done
for st in 2 4; do run GJ_ENC_STAGGER=$st libgpujpeg.so 4k; done
} > gpurun_out/${T}_solo.txt 2>&1
cat gpurun_out/${T}_solo.txt
GJ_ENC_STAGGER=3 timeout 200 python tools/encoder_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_phases_8k.txt; tail -22 gpurun_out/${T}_phases_8k.txt
