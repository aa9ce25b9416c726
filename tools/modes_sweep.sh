#!/bin/bash
# one workload in the three modes of bench.py (encode + decode, encode only, decode only), lean line each: tools/modes_sweep.sh WORKLOAD [more bench args]
cd "${GRAFT_REPO_ROOT:-.}"
W=$1; shift
for m in both encode decode; do echo "== $W $m"; timeout 300 python bench.py --lean --workload $W --mode $m "$@" 2>/dev/null | python tools/bench_brief.py /dev/stdin | head -1; done
