#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for a in "" "--no-pin" "" "--no-pin"; do for w in 8k hd; do timeout 300 python bench.py --lean --workload $w $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$a]', d['metric'], d['value'], 'solo k_encode', d['roofline']['by_kernel'][0]['ms'])"; done; done
uptime; nproc; ps -eo pcpu,psr,comm --sort=-pcpu | head -8
