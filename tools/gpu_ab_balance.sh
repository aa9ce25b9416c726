#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; R=$PWD
for e in 0 1; do for w in 8k 4k; do for pat in natural camera; do
rm -rf /tmp/kt; cd /tmp; GJ_DEC_NO_BALANCE=$e timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --lean --streams 1 --workload $w --pattern $pat > /tmp/kt.log 2>&1; cd $R
echo "no_balance=$e $w $pat: $(python - <<'PY'
import csv,glob
for f in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_huffman_decode_tok' in r['Name']: print('tok avg us', round(float(r['AverageNs'])/1e3,2), 'calls', r['Calls'])
PY
)"; done; done; done
