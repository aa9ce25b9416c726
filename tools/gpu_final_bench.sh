#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_11_bench.json 2> gpurun_out/r4_11_bench.err ) 2>&1 | tail -3
for i in 1 2 3; do python bench.py --lean 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8k headline run $i', d['value'])"; done
for i in 1 2; do python bench.py --lean --workload 16k422 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('16k422 run $i', d['value'], d['roofline']['by_kernel'])"; done
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_11_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel'], d['roofline']['ms'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('valu_issue_frac'))
for k in d['roofline']['by_kernel']: print('  ', k)
print(d['roofline']['by_direction'])
print({k:(v.get('mpix_s'), v.get('frames_s'), (v.get('roofline') or {}).get('frac'), v.get('solo_gpu_ms')) for k,v in d.get('workloads',{}).items()})
print(d['encode_only']['mpix_s'], d['decode_only']['mpix_s'], {k:v['mpix_s'] for k,v in d['full_api'].items() if k!='note'})
print(d['cpu_baseline']['value'], d['cpu_baseline_all_cores']['value'])
PY
