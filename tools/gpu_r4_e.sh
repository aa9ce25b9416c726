#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
T=${TAG:-r4f}
timeout 600 python -m pytest tests -m gpu -x -q -k "one_launch or full_size" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -3 gpurun_out/${T}_pytest.log
run() { echo -n "$1 $2 $3 "; env $1 timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/$2 $3 2>&1 | tail -1 | sed 's/np.float64(\([0-9.]*\))/\1/g'; }
{
for w in 8k 4k hd 16k; do
  run GJ_ENC_TAIL=0 libgpujpeg.so $w
  run GJ_ENC_TAIL=-512 libgpujpeg.so $w
done
for l in libgpujpeg_pf0.so libgpujpeg_pf8.so; do run GJ_ENC_TAIL=-512 $l 8k; run GJ_ENC_TAIL=-512 $l 4k; done
for r in 512 768 896 1280 2058; do run "GJ_ENC_TAIL=-512 GJ_ENC_RESIDENT=$r" libgpujpeg.so 8k; done
for t in 64 128 512; do run GJ_ENC_TAIL=$t libgpujpeg.so 8k; done
run GJ_ENC_TAIL=0 libgpujpeg.so 16k422; run GJ_ENC_TAIL=-1024 libgpujpeg.so 16k422
} > gpurun_out/${T}_solo.txt 2>&1
cat gpurun_out/${T}_solo.txt
GJ_ENC_TAIL=-512 timeout 200 python tools/encoder_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_phases_8k.txt; tail -22 gpurun_out/${T}_phases_8k.txt
for t in 0 -512; do
GJ_ENC_TAIL=$t timeout 300 python bench.py --lean > gpurun_out/${T}_bench_lean$t.json 2> gpurun_out/${T}_bench_lean.err; tail -2 gpurun_out/${T}_bench_lean.err
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_lean$t.json')); print('headline', d['value'], [ (k['kernel'],k['ms']) for k in d['roofline']['by_kernel']])"
done
