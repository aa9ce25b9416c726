#!/bin/bash
# rocprofv3 passes on the GPU box: kernel trace + stats, then the HBM traffic counters in separate PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with sys/hip traces) and the SQ instruction counters in a fourth. Results: gpurun_out/prof_*/
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq
ARGS="--steps ${STEPS:-20} --warmup 3 --lean ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python bench.py $ARGS > $OUT/prof_stats.log 2>&1
tail -1 $OUT/prof_stats.log | cut -c1-400
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python bench.py --steps 3 --warmup 1 --lean --min-seconds 0 --calibrate ${BENCH_ARGS} > $OUT/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python bench.py --steps 3 --warmup 1 --lean --min-seconds 0 --calibrate ${BENCH_ARGS} > $OUT/prof_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/prof_sq -- python bench.py --steps 3 --warmup 1 --lean --min-seconds 0 --calibrate ${BENCH_ARGS} > $OUT/prof_sq.log 2>&1
find $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write -name "*.csv" | head -20
python tools/rocprof_summary.py $OUT ${TAG:-r1_xx} || true
