cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -6 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_18_bench.json 2> gpurun_out/r4_18_bench.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('gpurun_out/r4_18_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['traffic']); print({k:(v.get('mpix_s'), v.get('frames_s')) for k,v in d['workloads'].items()})"
