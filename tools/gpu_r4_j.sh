#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python bench.py > gpurun_out/r4_bench_j.json 2> gpurun_out/r4_bench_j.err ) 2>&1 | tail -3
grep -v amdgpu.ids gpurun_out/r4_bench_j.err | tail -c 1500
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_j.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel'], d['roofline']['ms'], d['roofline']['frac'], d['config']['timed_with'][-60:])
print({k:(v.get('mpix_s'), v.get('frames_s'), (v.get('roofline') or {}).get('frac')) for k,v in d.get('workloads',{}).items()})
print(d['encode_only'], d['decode_only'], d['full_api'])
PY
for w in 8k hd 4k; do timeout 300 python bench.py --lean --python-loop --workload $w 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('python loop', d['metric'], d['value'])"; done
timeout 300 python bench.py --batch 256 --workload 4k --steps 3 --warmup 1 --python-loop 2>/dev/null | cut -c1-200
timeout 300 python bench.py --batch 256 --workload 4k --steps 3 --warmup 1 --streams 6 2>/dev/null | cut -c1-200
timeout 300 python bench.py --lean --workload hd --streams 6 2>/dev/null | cut -c1-120
