#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
run() { echo "== $*"; for w in hd 4k 8k; do env "$@" timeout 300 python bench.py --lean --workload $w 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', d['metric'], d['value'])"; done; env "$@" timeout 300 python bench.py --batch 256 --workload 4k --steps 3 --warmup 1 2>/dev/null | cut -c1-110; }
run X=1
run ROC_ACTIVE_WAIT_TIMEOUT=200
run HSA_ENABLE_INTERRUPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=200 HSA_ENABLE_INTERRUPT=0
