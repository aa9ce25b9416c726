#!/bin/bash
# SQ counters per kernel (one PMC pass): where do the wave cycles go?
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
rm -rf $OUT/prof_sq
rocprofv3 --kernel-trace --pmc ${COUNTERS:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY} --output-format csv -d $OUT/prof_sq -- python bench.py --steps 3 --warmup 1 --lean --min-seconds 0 ${BENCH_ARGS} > $OUT/prof_sq.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    if not k.startswith("k_"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, v in acc.items():
    n = max(cnt[k], 1)
    print(k, "dispatches", n, " ".join(f"{c}={x / n:.3g}" for c, x in sorted(v.items())))
PY
