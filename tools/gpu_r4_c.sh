#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
T=${TAG:-r4c}
timeout 600 python -m pytest tests -m gpu -x -q -k "one_launch or full_size" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -3 gpurun_out/${T}_pytest.log
{
for w in 8k hd 4k 16k422; do
  timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so $w 2>&1 | tail -1
done
for w in 8k hd 4k; do
  echo "GJ_ENC_BLOCKS=1 $w"; GJ_ENC_BLOCKS=1 timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so $w 2>&1 | tail -1
done
for t in ${TAILS:-16 64 128}; do
  echo "GJ_ENC_TAIL=$t"; GJ_ENC_TAIL=$t timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so 8k 2>&1 | tail -1
  GJ_ENC_TAIL=$t timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so hd 2>&1 | tail -1
done
} > gpurun_out/${T}_solo.txt 2>&1
cat gpurun_out/${T}_solo.txt
timeout 200 python tools/encoder_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_phases_8k.txt; tail -22 gpurun_out/${T}_phases_8k.txt
timeout 200 python tools/encoder_phases.py --workload hd 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_phases_hd.txt; tail -5 gpurun_out/${T}_phases_hd.txt
timeout 300 python bench.py --lean > gpurun_out/${T}_bench_lean.json 2> gpurun_out/${T}_bench_lean.err; tail -2 gpurun_out/${T}_bench_lean.err
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_lean.json')); print('headline', d['value'], [ (k['kernel'],k['ms']) for k in d['roofline']['by_kernel']])"
