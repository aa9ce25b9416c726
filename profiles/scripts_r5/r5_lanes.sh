#!/bin/bash
# the host link with and without the copy lanes (GJ_COPY_LANES=0: every coder copies on its own stream, as before): pinned host buffers in and out
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
one() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%10.1f %s' % (d['value'], d['unit']))"; }
{
for lanes in 0 1; do
  export GJ_COPY_LANES=$lanes
  for wl in 8k 4k hd; do
    for st in 1 2 4; do
      for mode in encode decode both; do
        echo -n "lanes $lanes  $wl  $st pipeline(s)  $mode: "; one --workload $wl --lean --host-io --streams $st --mode $mode --steps 3 --warmup 1 --min-seconds 0.4
      done
    done
  done
  echo -n "lanes $lanes  256 x 4K frame at a time, host in and out, 4 pipelines: "; one --batch 256 --workload 4k --batch-io host --streams 4 --steps 2 --warmup 1
done
} 2>&1 | tee gpurun_out/r5_lanes.txt
