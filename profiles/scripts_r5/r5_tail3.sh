#!/bin/bash
# the product with the last 32 tiles split (default) against GJ_ENC_TAIL=0: tests, then 8K / 16K solo kernel times and four-pipeline rates
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -3 ) | tee gpurun_out/r5_gpu_tests.txt
summ() { python - "$1" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "k_encode" in n and int(r["Calls"]) > 5:
            print("   %-60s calls %5s avg %9.2f us  min %9.2f  max %9.2f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
{
for wl in 8k 16k; do
for tail in 0 32; do
  export GJ_ENC_TAIL=$tail
  rm -rf /tmp/abx
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abx -- python bench.py --workload $wl --streams 1 --lean --steps 10 --warmup 3 > /tmp/abx.log 2>&1
  echo "== $wl GJ_ENC_TAIL=$tail"; summ /tmp/abx
  for m in both encode; do
    python bench.py --workload $wl --lean --steps 20 --warmup 3 --mode $m 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   four pipelines $wl tail $tail $m', d['value'])"
  done
done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_tail3.txt
