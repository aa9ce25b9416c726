#!/bin/bash
# how often does the host-staged 256 x 4K batch (frame at a time, four pipelines, copy lanes) fall into its slow mode, and does the bare pattern do it too?
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
{
for i in 1 2 3 4 5 6; do echo -n "ubench run $i: "; gpujpeg_amd/lib/ubench_hlp 2>&1 | grep "P2  4"; done
for i in 1 2 3 4 5 6; do
  echo -n "bench run $i: "; python bench.py --batch 256 --workload 4k --batch-io host --streams 4 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d.get('pass_ms'))"
done
} 2>&1 | tee gpurun_out/r5_bimodal.txt
