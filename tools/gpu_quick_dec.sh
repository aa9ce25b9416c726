#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; R=$PWD
for tk in "" 1; do
rm -rf /tmp/kt; cd /tmp; GJ_DEC_TOKENS=$tk timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --lean --streams 1 --workload hd > /tmp/kt.log 2>&1; cd $R
echo "GJ_DEC_TOKENS=$tk hd: $(python - <<'PY'
import csv,glob
for f in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Name'].startswith(('k_','void k_')): print(r['Name'].replace('void ','')[:22], round(float(r['AverageNs'])/1e3,2), end='; ')
PY
)"; GJ_DEC_TOKENS=$tk timeout 300 python bench.py --lean --workload hd 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   throughput', d['value'])"; done
