#!/bin/bash
# the lean bench.py line of the product library under several developer settings: tools/tune_sweep.sh TAG [bench args with commas] -- SETTING[,SETTING...] ...
# ("-" = no setting). Results gpurun_out/<TAG>_<n>.json
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
ARGS=""
while [ "$1" != "--" ] && [ -n "$1" ]; do ARGS="$ARGS ${1//,/ }"; shift; done
shift
n=0
for set in "$@"; do
  T=""
  if [ "$set" != "-" ]; then for s in ${set//,/ }; do T="$T --tune $s"; done; fi
  f=gpurun_out/${TAG}_$n.json
  timeout 600 python bench.py --lean $ARGS $T > $f 2> gpurun_out/${TAG}_$n.err || tail -3 gpurun_out/${TAG}_$n.err
  echo "== $set"
  python tools/bench_brief.py $f | head -3
  n=$((n+1))
done
