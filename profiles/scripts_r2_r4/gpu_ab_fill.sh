#!/bin/bash
# batch plan of the sub-sequence decoders in a saturated launch: how full should the LDS stage be planned? (GJ_DEC_FILL, 32nds)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 6 --warmup 2 --batch 256 --batch-api batch --batch-streams 1 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s')"; }
for w in 4k hd; do for f in 17 20 23 26 29; do echo -n "$w fill=$f: "; GJ_DEC_FILL=$f run --workload $w; done; done
