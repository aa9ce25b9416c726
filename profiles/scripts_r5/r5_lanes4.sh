#!/bin/bash
# the batch calls with pinned host memory in and out, with and without the copy lanes
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
one() { python bench.py "$@" 2>&1 | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t); print('%10.1f %s' % (d['value'], d['unit']))
except Exception: print('FAILED', t[-300:])"; }
{
for rep in 1 2; do
for lanes in 0 1; do
  export GJ_COPY_LANES=$lanes
  for bs in 1 2 4; do
    echo -n "rep $rep lanes $lanes  256 x 4K through the batch calls, host in and out, $bs pipeline(s): "; one --batch 256 --workload 4k --batch-io host --batch-api batch --batch-streams $bs --steps 2 --warmup 1
  done
  echo -n "rep $rep lanes $lanes  256 x HD through the batch calls, host in and out, 2 pipelines: "; one --batch 256 --workload hd --batch-io host --batch-api batch --batch-streams 2 --steps 2 --warmup 1
done
done
} 2>&1 | tee gpurun_out/r5_lanes4.txt
