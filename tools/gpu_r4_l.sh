#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for m in 0 15 1 2 4 8 5 10; do
GJ_COPY_MARKS=$m timeout 300 python bench.py --no-workloads --no-cpu-baseline --python-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('marks $m', d['value'], {k:v['mpix_s'] for k,v in d['full_api'].items() if k!='note'})"; done
