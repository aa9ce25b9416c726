#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; R=$PWD
for w in hd 4k 8k; do
rm -rf /tmp/kt; cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --lean --streams 1 --workload $w > /tmp/kt.log 2>&1; cd $R
echo "== $w, one pipeline"; python tools/kernel_gaps.py /tmp/kt
done
