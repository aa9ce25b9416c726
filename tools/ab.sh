#!/bin/bash
# A/B of library builds on the GPU box: tools/ab.sh TAG [bench args with commas] -- lib1 lib2 ...   (names under gpujpeg_amd/lib, e.g. libgpujpeg.so libgpujpeg_win1.so)
# every build runs the same lean bench.py line through --lib (frame loop in Python for all of them), twice, results under gpurun_out/<TAG>_<lib>_<n>.json
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
ARGS=""
while [ "$1" != "--" ] && [ -n "$1" ]; do ARGS="$ARGS ${1//,/ }"; shift; done
shift
for rep in 0 1; do
  for lib in "$@"; do
    f=gpurun_out/${TAG}_${lib%.so}_$rep.json
    timeout 600 python bench.py --lean --verify $ARGS --lib gpujpeg_amd/lib/$lib > $f 2> gpurun_out/${TAG}_${lib%.so}_$rep.err || tail -3 gpurun_out/${TAG}_${lib%.so}_$rep.err
    echo "== $lib rep $rep"
    python tools/bench_brief.py $f
    python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('verified', d.get('verified_bit_exact_encode'), d.get('verified_bit_exact_decode'))"
  done
done
