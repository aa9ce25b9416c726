#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "tiles_and_gather or bit_exact" 2>&1 | tail -1
for sp in 0 100000; do for w in hd 4k 8k; do
rm -rf /tmp/kt; cd /tmp; GJ_ENC_SPLIT=$sp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --lean --streams 1 --workload $w > /tmp/kt.log 2>&1; cd $R
echo "split<=$sp $w: $(python - <<'PY'
import csv,glob
for f in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_gather' in r['Name'] or 'k_encode' in r['Name']: print(r['Name'][:16], round(float(r['AverageNs'])/1e3,2), end='; ')
PY
)"; done; done
for sp in 0 100000; do for w in hd 4k; do GJ_ENC_SPLIT=$sp timeout 300 python bench.py --lean --workload $w 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split<=$sp', d['metric'], d['value'])"; done; done
