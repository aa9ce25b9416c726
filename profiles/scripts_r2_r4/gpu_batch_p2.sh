#!/bin/bash
# frame batches, second look: the four-pipeline anomaly at 4K (pass times, free HBM), HD batches through tokens against planes, 8K batches,
# and the kernel trace of a batched pass (per-kernel time per chunk of 64 frames)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; OUT=$PWD/gpurun_out
run() { # label, then bench arguments
  local label=$1; shift
  timeout 300 python bench.py --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$label:', d['value'], 'frames/s', d['mpix_s'], 'Mpix/s | pass ms', c['pass_ms_per_pipeline'], '| HBM free', c['hbm_free_gb_at_end'], 'of', c['hbm_total_gb'], '|', c['api'][-34:])"
}
run "4k batch S=4" --batch 256 --workload 4k --batch-api batch --batch-streams 4
run "4k batch S=3" --batch 256 --workload 4k --batch-api batch --batch-streams 3
run "4k batch S=2" --batch 256 --workload 4k --batch-api batch --batch-streams 2
run "hd batch S=2 (tokens)" --batch 256 --workload hd --batch-api batch --batch-streams 2
GJ_DEC_NO_TOKENS=1 run "hd batch S=2 (planes)" --batch 256 --workload hd --batch-api batch --batch-streams 2
run "8k batch S=2" --batch 32 --workload 8k --batch-api batch --batch-streams 2
run "8k batch S=1" --batch 32 --workload 8k --batch-api batch --batch-streams 1
run "8k frame S=4" --batch 32 --workload 8k --batch-api frame --streams 4
for w in 4k hd; do
  rm -rf $OUT/prof_stats; cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $OUT/../bench.py --batch 256 --workload $w --batch-api batch --batch-streams 1 --steps 3 --warmup 1 > $OUT/prof_batch_$w.log 2>&1
  cd $OUT/..
  python tools/rocprof_summary.py $OUT r4_batch256_${w}_batched "cmd: rocprofv3 --kernel-trace --stats -- python bench.py --batch 256 --workload $w --batch-api batch --batch-streams 1 --steps 3 --warmup 1 (one launch = a chunk of 64 frames)" 2>/dev/null | grep -E "^k_|^void k_" | head -8
done
rm -rf $OUT/prof_stats
