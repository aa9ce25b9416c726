#!/bin/bash
# is the slow mode of the host-staged batch a state of the process or does it come and go? passes' times per pipeline, no warm-up pass
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
{
for i in 1 2 3 4 5 6; do
  echo -n "bench run $i: "; python bench.py --batch 256 --workload 4k --batch-io host --streams 4 --steps 5 --warmup 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['config'].get('pass_ms_per_pipeline')[0], d['config'].get('pass_ms_per_pipeline')[3])"
done
} 2>&1 | tee gpurun_out/r5_bimodal2.txt
