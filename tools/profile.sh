#!/bin/bash
# rocprofv3 passes on the GPU box for ONE workload of bench.py (WORKLOAD=hd|4k|8k|16k|16k422, default 8k; BATCH=256 for the 256 x 4K batch):
# kernel trace + stats with one pipeline (the kernels alone: what bench.py's roofline.ms is compared with), then FETCH_SIZE, WRITE_SIZE
# and the SQ instruction counters, each in its own PMC pass (never combined with sys / hip traces). Results: gpurun_out/<TAG>_*;
# tools/rocprof_summary.py condenses them (and stamps the traffic JSON with the hash of the device sources).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
W=${WORKLOAD:-8k}
TAG=${TAG:-r6_$W}
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq
ARGS="--workload $W --streams ${STREAMS:-1} --lean ${BENCH_ARGS}"
if [ -n "$BATCH" ]; then ARGS="--workload 4k --batch $BATCH --streams ${STREAMS:-4} ${BENCH_ARGS}"; fi
TRACE_ARGS="$ARGS" # (the counter passes of a batch run another shape: the header of the summary names both commands, each with its own arguments)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python bench.py --steps ${STEPS:-20} --warmup 3 $ARGS > $OUT/prof_stats.log 2>&1
tail -1 $OUT/prof_stats.log | cut -c1-300
PM="--steps 3 --warmup 1 --min-seconds 0 --calibrate"
if [ -n "$BATCH" ]; then # (the counter passes crash rocprofv3 with four host threads launching: one pipeline, a batch of 32 -- the traffic per frame is the same)
  PM="--steps 1 --warmup 1"
  ARGS="--workload 4k --batch 32 --streams 1 ${BENCH_ARGS}"
fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python bench.py $PM $ARGS > $OUT/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python bench.py $PM $ARGS > $OUT/prof_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/prof_sq -- python bench.py $PM $ARGS > $OUT/prof_sq.log 2>&1
python tools/rocprof_summary.py $OUT $TAG "kernel trace + stats: rocprofv3 --kernel-trace --stats -- python bench.py --steps ${STEPS:-20} --warmup 3 $TRACE_ARGS | PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_*; each its own run): rocprofv3 --kernel-trace --pmc ... -- python bench.py $PM $ARGS" > $OUT/${TAG}_summary.log 2>&1 || true
grep -E "^k_|^void k_" $OUT/${TAG}_kernel_stats.txt | head -14
