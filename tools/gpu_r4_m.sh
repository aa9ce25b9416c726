#!/bin/bash
# round 4: the one-launch marker scan: decoder parity (all decoder tests, damaged streams, fuzz), per-kernel times, phase trace
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
T=${TAG:-r4m}
timeout 900 python -m pytest tests -m gpu -x -q -k "${GJ_K:-not exhaustive}" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -3 gpurun_out/${T}_pytest.log
run() { echo -n "$1 $2 $3 "; env $1 timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/$2 $3 2>&1 | tail -1 | sed 's/np.float64(\([0-9.]*\))/\1/g'; }
{
for w in 8k 4k hd 16k 16k422; do run X=0 libgpujpeg.so $w; done
} > gpurun_out/${T}_solo.txt 2>&1
cat gpurun_out/${T}_solo.txt
timeout 200 python tools/decoder_phases.py 2>&1 | grep -v amdgpu.ids | tail -9 > gpurun_out/${T}_phases_markers.txt; cat gpurun_out/${T}_phases_markers.txt
