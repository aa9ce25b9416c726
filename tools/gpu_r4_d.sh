#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
T=${TAG:-r4d}
timeout 600 python -m pytest tests -m gpu -x -q -k "one_launch or full_size" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -3 gpurun_out/${T}_pytest.log
{
for t in ${TAILS:-0 64 128 512 -64 -128 -256 -512}; do
  for w in 8k hd 4k; do
    echo -n "GJ_ENC_TAIL=$t $w "; GJ_ENC_TAIL=$t timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so $w 2>&1 | tail -1 | sed 's/np.float64(\([0-9.]*\))/\1/g'
  done
done
echo -n "16k422 "; timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so 16k422 2>&1 | tail -1 | sed 's/np.float64(\([0-9.]*\))/\1/g'
echo -n "16k422 own gather "; GJ_ENC_TAIL=-256 timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so 16k422 2>&1 | tail -1 | sed 's/np.float64(\([0-9.]*\))/\1/g'
} > gpurun_out/${T}_solo.txt 2>&1
cat gpurun_out/${T}_solo.txt
timeout 200 python tools/encoder_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_phases_8k.txt; tail -5 gpurun_out/${T}_phases_8k.txt
timeout 200 python tools/encoder_phases.py --workload hd 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_phases_hd.txt; tail -4 gpurun_out/${T}_phases_hd.txt
for t in 0 -256; do
GJ_ENC_TAIL=$t timeout 300 python bench.py --lean > gpurun_out/${T}_bench_lean$t.json 2> gpurun_out/${T}_bench_lean.err; tail -2 gpurun_out/${T}_bench_lean.err
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_lean$t.json')); print('headline', d['value'], [ (k['kernel'],k['ms']) for k in d['roofline']['by_kernel']])"
done
