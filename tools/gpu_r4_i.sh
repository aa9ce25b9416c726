#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python bench.py > gpurun_out/r4_bench_i.json 2> gpurun_out/r4_bench_i.err ) 2>&1 | tail -3
tail -c 600 gpurun_out/r4_bench_i.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_i.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel'], d['roofline']['ms'], d['roofline']['frac'], d['config'].get('cpu_affinity_rank0'))
print({k:(v.get('mpix_s'), v.get('frames_s'), (v.get('roofline') or {}).get('frac')) for k,v in d.get('workloads',{}).items()})
print(d['roofline']['contended'])
PY
timeout 600 tools/gpu_r4_halt.sh 2>&1 | tail -60
timeout 600 python -m pytest tests/test_gpu_multiproc.py -q -x 2>&1 | tail -3
