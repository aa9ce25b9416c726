#!/bin/bash
# Round-2 profile of one direction alone (default: the encoder, one pipeline, GPU otherwise idle): rocprofv3 kernel trace + stats,
# then the SQ instruction / wait counters in their own PMC pass (never combined with sys/hip traces). Results: gpurun_out/
# usage: TAG=r2_xx MODE=encode|decode|both STREAMS=1 tools/profile_r2.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
TAG=${TAG:-r2_xx}
ARGS="--streams ${STREAMS:-1} --mode ${MODE:-encode} --lean ${BENCH_ARGS}"
rm -rf $OUT/prof_stats $OUT/prof_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python bench.py --steps ${STEPS:-100} --warmup 5 $ARGS > $OUT/prof_stats.log 2>&1
tail -1 $OUT/prof_stats.log | cut -c1-300
python tools/rocprof_summary.py $OUT $TAG "cmd: rocprofv3 --kernel-trace --stats -- python bench.py --steps ${STEPS:-100} --warmup 5 $ARGS" | head -30
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/prof_sq -- python bench.py --steps 3 --warmup 1 $ARGS > $OUT/prof_sq.log 2>&1
python - <<PY | tee $OUT/${TAG}_sq_counters.txt
import csv, glob, collections
f = glob.glob("gpurun_out/prof_sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    if not k.startswith("k_"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
print("# rocprofv3 --pmc SQ_* (one pass), per-dispatch averages; cmd: python bench.py --steps 3 --warmup 1 $ARGS")
for k, v in acc.items():
    n = max(cnt[k], 1)
    print(k, "dispatches", n, " ".join(f"{c}={x / n:.4g}" for c, x in sorted(v.items())), f"VALU_per_wave={v['SQ_INSTS_VALU'] / max(v['SQ_WAVES'], 1):.0f}")
PY
