#!/bin/bash
# round 4, first look at the one-launch encoder on the GPU: parity suite, per-kernel times against the round-3 library, tail shares sweep,
# phase trace, lean headline. Outputs under gpurun_out/r4a_*.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4a_pytest.log
tail -4 gpurun_out/r4a_pytest.log
{
for w in 8k hd 4k 16k422; do
  for l in libgpujpeg_r3.so libgpujpeg.so; do
    timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/$l $w 2>&1 | tail -1
  done
done
for t in 1 16 64 128 512 2048; do
  echo "GJ_ENC_TAIL=$t"; GJ_ENC_TAIL=$t timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so 8k 2>&1 | tail -1
  GJ_ENC_TAIL=$t timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so hd 2>&1 | tail -1
done
} > gpurun_out/r4a_solo.txt 2>&1
cat gpurun_out/r4a_solo.txt
timeout 200 python tools/encoder_phases.py > gpurun_out/r4a_phases_8k.txt 2>&1; tail -22 gpurun_out/r4a_phases_8k.txt
timeout 200 python tools/encoder_phases.py --workload hd > gpurun_out/r4a_phases_hd.txt 2>&1; tail -6 gpurun_out/r4a_phases_hd.txt
timeout 300 python bench.py --lean > gpurun_out/r4a_bench_lean.json 2> gpurun_out/r4a_bench_lean.err; tail -2 gpurun_out/r4a_bench_lean.err
python -c "
import json; d=json.load(open('gpurun_out/r4a_bench_lean.json')); print('headline', d['value'], [ (k['kernel'],k['ms']) for k in d['roofline']['by_kernel']])"
timeout 300 python bench.py --lean --lib gpujpeg_amd/lib/libgpujpeg_r3.so > gpurun_out/r4a_bench_lean_r3.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r4a_bench_lean_r3.json')); print('headline r3', d['value'], [ (k['kernel'],k['ms']) for k in d['roofline']['by_kernel']])"
