#!/bin/bash
# usage: tools/r5_ab_wl.sh "libs" "workloads": kernel traces (one pipeline) and four-pipeline enc+dec rates per variant build and workload
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=$PWD/gpurun_out
LIBS=$1; WLS=${2:-hd}
summ() { python - "$1" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    tot = 0.0
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("k_encode", "k_gather", "k_huffman_decode", "k_idct", "k_marker")) and int(r["Calls"]) > 5:
            print("   %-60s calls %5s avg %9.2f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
}
for w in $WLS; do
  for v in $LIBS; do
    L=""; [ $v != default ] && L="--lib gpujpeg_amd/lib/libgpujpeg_$v.so"
    rm -rf $OUT/ab_${v}_$w
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_${v}_$w -- python bench.py --workload $w --streams 1 --lean --steps 20 --warmup 3 $L > $OUT/ab_${v}_$w.log 2>&1
    echo "== $v $w"; summ $OUT/ab_${v}_$w
    rm -rf $OUT/ab_${v}_$w
    for m in both decode; do
      python bench.py --workload $w --lean --steps 20 --warmup 3 --python-loop --mode $m $L 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   four pipelines $v $w $m', d['value'])"
    done
  done
done
