#!/bin/bash
# usage: tools/r5_ab.sh "libA libB ..." [patterns]: kernel traces of the 8K frame (one pipeline) for variant builds gpujpeg_amd/lib/libgpujpeg_<name>.so
# (`default` = the product library), then their four-pipeline rates
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=$PWD/gpurun_out
LIBS=$1; PATS=${2:-natural}
summ() { python - "$1" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("k_encode", "k_gather", "k_huffman_decode", "k_idct", "k_marker")) and int(r["Calls"]) > 5:
            print("   %-60s calls %5s avg %9.2f us  min %9.2f  max %9.2f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
for pat in $PATS; do
  for v in $LIBS; do
    L=""; [ $v != default ] && L="--lib gpujpeg_amd/lib/libgpujpeg_$v.so"
    rm -rf $OUT/ab_${v}_$pat
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_${v}_$pat -- python bench.py --workload 8k --pattern $pat --streams 1 --lean --steps 20 --warmup 3 $L > $OUT/ab_${v}_$pat.log 2>&1
    echo "== $v $pat"; summ $OUT/ab_${v}_$pat
    rm -rf $OUT/ab_${v}_$pat
  done
done
for v in $LIBS; do
  L=""; [ $v != default ] && L="--lib gpujpeg_amd/lib/libgpujpeg_$v.so"
  for m in both encode decode; do
    python bench.py --workload 8k --lean --steps 20 --warmup 3 --python-loop --mode $m $L 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   four pipelines $v $m', d['value'])"
  done
done
