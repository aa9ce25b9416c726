#!/bin/bash
# what the round ends with: every -m gpu test, smoke, the profile set of every BASELINE configuration (stamps profiles/r5_traffic.json with the hash of
# the device sources), then the driver's bench command, whose line quotes that traffic
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -6 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
tools/profile_all.sh 2>&1 | grep -E "^==|^k_encode|^k_huffman|^k_idct|merged"
cp gpurun_out/r5_traffic.json profiles/r5_traffic.json
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel'], d['roofline']['ms'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('valu_issue_frac'))
print({k:(v.get('mpix_s'), v.get('frames_s')) for k,v in d.get('workloads',{}).items()})
print(d['encode_only']['mpix_s'], d['decode_only']['mpix_s'], {k:v['mpix_s'] for k,v in d['full_api'].items() if k!='note'})
PY
