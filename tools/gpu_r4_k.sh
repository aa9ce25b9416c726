#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for a in "" "--no-pin" "--python-loop" "--no-pin --python-loop"; do
timeout 300 python bench.py --no-workloads --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', d['value'], {k:v['mpix_s'] for k,v in d['full_api'].items() if k!='note'}, d['encode_only']['mpix_s'], d['decode_only']['mpix_s'])"; done
lscpu | grep -i "numa\|model name\|socket"
