#!/bin/bash
# rocprofv3 kernel stats of one workload (W, default 8k) with one pipeline
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for W in ${WL:-8k}; do
rm -rf gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 20 --warmup 3 --workload $W --streams 1 --lean > gpurun_out/prof_stats_$W.log 2>&1
echo "== $W"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_stats/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:9]:
    if r["Name"].startswith("void at::") or r["Name"].startswith("at::"): continue
    print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1000, 2))
PY
done
