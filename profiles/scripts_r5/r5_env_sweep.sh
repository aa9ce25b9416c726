#!/bin/bash
# usage: tools/r5_env_sweep.sh VAR "v1 v2 ..." [bench args]: kernel trace of the 8K frame (one pipeline) per value of a developer switch, then the
# four-pipeline headline for each; prints the product kernels' average durations
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=$PWD/gpurun_out
VAR=$1; VALS=$2; shift 2
summ() { python - "$1" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("k_encode", "k_gather", "k_huffman_decode", "k_idct", "k_marker")) and int(r["Calls"]) > 5:
            print("   %-60s calls %5s avg %9.2f us  min %9.2f  max %9.2f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
for v in $VALS; do
  rm -rf $OUT/sw_$v
  env $VAR=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sw_$v -- python bench.py --workload 8k --streams 1 --lean --steps 20 --warmup 3 "$@" > $OUT/sw_$v.log 2>&1
  echo "== $VAR=$v (one pipeline, kernels alone)"; summ $OUT/sw_$v
  rm -rf $OUT/sw_$v
  for m in both encode; do
    env $VAR=$v python bench.py --workload 8k --lean --steps 20 --warmup 3 --mode $m "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   four pipelines, $m:', d['value'], 'Mpix/s')"
  done
done
