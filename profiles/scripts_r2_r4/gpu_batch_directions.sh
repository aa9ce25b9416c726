#!/bin/bash
# each direction alone through the batch calls (the reference's tables quote encode and decode separately)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for w in hd 4k hd422; do for m in encode decode; do timeout 300 python bench.py --steps 6 --warmup 2 --batch 256 --batch-api batch --batch-streams 2 --workload $w --mode $m 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $m only:', d['value'], 'frames/s', d['mpix_s'], 'Mpix/s')"; done; done
