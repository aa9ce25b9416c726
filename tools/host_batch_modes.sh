#!/bin/bash
# the host-staged 256 x 4K batch (BASELINE config 5 from pinned host memory, four pipelines) N times per environment: which mode of the host link is a process in?
# usage: tools/host_batch_modes.sh N "ENV=VAL" ["ENV=VAL" ...]   (developer settings reach the library through bench.py; A=1 for "no setting")
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
one() { env "$@" python bench.py --batch 256 --workload 4k --batch-io host --streams 4 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], end=' ')"; }
N=$1; shift
for envs in "$@"; do
  echo -n "[$envs] frames/s: "
  for i in $(seq $N); do one $envs; done
  echo
done
