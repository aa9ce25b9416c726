#!/bin/bash
# the slow mode of the host-staged 4K batch and where the process's threads and pinned pages sit: launch threads bound to the GPU's NUMA node (--pin on), and
# the whole process started on that node / on the other one (numactl, when present)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
one() { "$@" python bench.py --batch 256 --workload 4k --batch-io host --streams 4 --steps 2 --warmup 1 $PIN 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['config'].get('cpu_affinity_rank0'))"; }
{
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo " <- numa nodes of the drm cards"; lscpu | grep -i "numa" | head -6
for i in 1 2 3 4; do PIN="--pin on"; echo -n "pin on, run $i: "; one; done
for i in 1 2 3 4; do PIN=""; echo -n "no pin, run $i: "; one; done
if command -v taskset > /dev/null; then
  N0=$(cat /sys/devices/system/node/node0/cpulist 2>/dev/null); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
  for i in 1 2 3; do PIN=""; echo -n "taskset node0 ($N0), run $i: "; one taskset -c $N0; done
  [ -n "$N1" ] && for i in 1 2 3; do PIN=""; echo -n "taskset node1 ($N1), run $i: "; one taskset -c $N1; done
fi
} 2>&1 | tee gpurun_out/r5_bimodal4.txt
