#!/bin/bash
# round 5, GPU session 2: the -m gpu suite on the library whose colour transforms run on the matrix pipe (encoder + token-fed IDCT), then A/B kernel
# traces of the 8K frame (natural and the reference's camera sample) against round 4's device code, four-pipeline rates, instruction counters
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=$PWD/gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -25 ) 2>&1 | grep -v amdgpu.ids | tee $OUT/r5_s2_tests.txt
summ() { python - "$1" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("k_encode", "k_gather", "k_huffman_decode", "k_idct", "k_marker")) and int(r["Calls"]) > 5:
            print("   %-60s calls %5s avg %9.2f us  min %9.2f  max %9.2f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
for pat in natural camera; do
  for v in r4base default; do
    L=""; [ $v != default ] && L="--lib gpujpeg_amd/lib/libgpujpeg_$v.so"
    rm -rf $OUT/ab_${v}_$pat
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_${v}_$pat -- python bench.py --workload 8k --pattern $pat --streams 1 --lean --steps 20 --warmup 3 $L > $OUT/ab_${v}_$pat.log 2>&1
    echo "== $v $pat"; summ $OUT/ab_${v}_$pat
    rm -rf $OUT/ab_${v}_$pat
  done
done 2>&1 | tee $OUT/r5_s2_ab_kernels.txt
for v in r4base default; do
  L=""; [ $v != default ] && L="--lib gpujpeg_amd/lib/libgpujpeg_$v.so"
  for m in both encode decode; do
    python bench.py --workload 8k --lean --steps 20 --warmup 3 --python-loop --mode $m $L 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v $m', d['value'], d.get('ms_per_step'))"
  done
done 2>&1 | tee $OUT/r5_s2_ab_rates.txt
for v in r4base default; do
  L=""; [ $v != default ] && L="--lib gpujpeg_amd/lib/libgpujpeg_$v.so"
  rm -rf $OUT/sq_$v
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq_$v -- python bench.py --workload 8k --streams 1 --lean --steps 3 --warmup 1 --min-seconds 0 $L > $OUT/sq_$v.log 2>&1
  python - $OUT/sq_$v $v <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_encode" not in k and "k_idct_tok" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(sys.argv[2], k[:50], {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
  rm -rf $OUT/sq_$v
done 2>&1 | tee $OUT/r5_s2_sq.txt
