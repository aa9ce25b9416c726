#!/bin/bash
# round-end profiles of every BASELINE configuration: per workload a kernel trace (+ stats) and the FETCH_SIZE / WRITE_SIZE /
# SQ counter passes (tools/profile.sh), then one stamped profiles-ready JSON with the traffic and instruction counts of all workloads.
# Results under gpurun_out/ (copy r6_<cfg>_* and r6_traffic.json to profiles/).
cd "${GRAFT_REPO_ROOT:-.}"
for w in ${WORKLOADS:-8k hd 4k 16k 16k422}; do
  echo "== $w"
  WORKLOAD=$w TAG=r6_$w STEPS=${STEPS:-20} tools/profile.sh 2>&1 | tail -12
done
echo "== batch256 (256 x 4K frames, four pipelines)"
BATCH=256 TAG=r6_batch256_4k tools/profile.sh 2>&1 | tail -10
python tools/merge_traffic.py gpurun_out r6 ${WORKLOADS:-8k hd 4k 16k 16k422} batch256_4k
