#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for w in hd 8k; do for s in 1 4; do echo "== $w streams $s"; GJ_HOST_TIMING=1 timeout 300 python bench.py --lean --workload $w --streams $s 2>&1 | grep -E "host timing|\"value\"" | cut -c1-200 | sort | uniq -c | sort -rn | head -6; done; done
