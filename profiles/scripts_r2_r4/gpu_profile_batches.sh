#!/bin/bash
# kernel trace of a batched pass (one pipeline): per-kernel time per chunk of frames -> profiles/r4_batch256_{4k,hd,hd422}_batched_kernel_stats.txt
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; OUT=$PWD/gpurun_out
for w in 4k hd hd422; do
  rm -rf $OUT/prof_stats; cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $OUT/../bench.py --batch 256 --workload $w --batch-api batch --batch-streams 1 --steps 6 --warmup 2 > $OUT/prof_batch_$w.log 2>&1
  cd $OUT/..
  python tools/rocprof_summary.py $OUT r4_batch256_${w}_batched "cmd: rocprofv3 --kernel-trace --stats -- python bench.py --batch 256 --workload $w --batch-api batch --batch-streams 1 --steps 6 --warmup 2 (one launch = a chunk of frames: 256 HD frames, 74 4K frames)" 2>/dev/null | grep -E "^k_|^void k_" | head -8
  tail -1 $OUT/prof_batch_$w.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], 'frames/s under rocprofv3')"
done
rm -rf $OUT/prof_stats
