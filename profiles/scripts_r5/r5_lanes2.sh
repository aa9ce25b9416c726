#!/bin/bash
# copy lanes: from which size on? (GJ_COPY_LANES=<MiB>; 0 = off)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
one() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%10.1f %s' % (d['value'], d['unit']))"; }
{
for lanes in 0 1 4 16; do
  export GJ_COPY_LANES=$lanes
  for wl in 8k 4k hd; do
    for st in 2 4; do
      echo -n "lanes from $lanes MiB  $wl  $st pipeline(s)  both: "; one --workload $wl --lean --host-io --streams $st --mode both --steps 3 --warmup 1 --min-seconds 0.4
    done
  done
  echo -n "lanes from $lanes MiB  256 x 4K frame at a time, host in and out, 4 pipelines: "; one --batch 256 --workload 4k --batch-io host --streams 4 --steps 2 --warmup 1
  echo -n "lanes from $lanes MiB  256 x HD frame at a time, host in and out, 4 pipelines: "; one --batch 256 --workload hd --batch-io host --streams 4 --steps 2 --warmup 1
done
} 2>&1 | tee gpurun_out/r5_lanes2.txt
