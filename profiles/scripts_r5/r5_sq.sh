#!/bin/bash
# usage: tools/r5_sq.sh "libA libB ..." [kernel substrings, |-separated]: SQ instruction counters of the 8K frame's kernels (one pipeline) per variant build
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=$PWD/gpurun_out
LIBS=$1; KS=${2:-k_encode|k_gather|k_huffman_decode_tok|k_idct_tok|k_marker}
for v in $LIBS; do
  L=""; [ $v != default ] && L="--lib gpujpeg_amd/lib/libgpujpeg_$v.so"
  rm -rf $OUT/sq_$v
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq_$v -- python bench.py --workload 8k --streams 1 --lean --steps 3 --warmup 1 --min-seconds 0 $L > $OUT/sq_$v.log 2>&1
  python - $OUT/sq_$v $v "$KS" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
ks = sys.argv[3].split("|")
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(x in k for x in ks): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    d = {c: v / n[(k, c)] for c, v in acc[k].items()}
    w = d.get("SQ_WAVES", 1)
    print("%-8s %-44s waves %6d  VALU/wave %7.1f  SALU/wave %6.1f  LDS/wave %6.1f  wait_any %4.2f  active_valu %4.2f of wave cycles" % (
        sys.argv[2], k[:44], w, d.get("SQ_INSTS_VALU", 0) / w, d.get("SQ_INSTS_SALU", 0) / w, d.get("SQ_INSTS_LDS", 0) / w,
        d.get("SQ_WAIT_ANY", 0) / max(1, d.get("SQ_WAVE_CYCLES", 1)), d.get("SQ_ACTIVE_INST_VALU", 0) / max(1, d.get("SQ_WAVE_CYCLES", 1))))
PY
  rm -rf $OUT/sq_$v
done
