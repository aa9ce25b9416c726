#!/bin/bash
# the marker scan on long streams: parity at 16K (RGB and config 4), every forced shape, then the decoder's kernels alone at 16K / config 4 / 8K
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "16k or marker_scan or frame_batch" 2>&1 | tail -2
for w in 16k422 16k 8k; do
rm -rf /tmp/kt; cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --lean --streams 1 --workload $w > /tmp/kt.log 2>&1; cd $R
echo "$w alone: $(python - <<'PY'
import csv,glob
for f in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Name'].startswith('k_') or ' k_' in r['Name'][:8]: print(r['Name'].split('(')[0].replace('void ','')[:22], round(float(r['AverageNs'])/1e3,2), end='; ')
PY
)"; done
for w in 16k422 16k; do timeout 300 python bench.py --lean --workload $w 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w four pipelines', d['value'])"; done
