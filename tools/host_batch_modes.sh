#!/bin/bash
# the host-staged 256 x 4K batch (BASELINE config 5 from pinned host memory, four pipelines) N times per environment: is a process in the fast mode
# (both directions of the host link at once, ~1 700 frames/s) or in the slow one (~1 017)? usage: tools/host_batch_modes.sh N "ENV=VAL ..." ["ENV=VAL ..." ...]  ("-" = default)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
N=$1; shift
one() { env "$@" python bench.py --batch 256 --workload 4k --batch-io host --streams 4 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], end=' ')"; }
for envs in "$@"; do
  echo -n "[$envs] frames/s: "
  for i in $(seq $N); do if [ "$envs" = "-" ]; then one A=1; else one $envs; fi; done
  echo
done
