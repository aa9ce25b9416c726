#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; R=$PWD
for l in libgpujpeg_ahead0.so libgpujpeg_ahead1.so libgpujpeg.so; do for w in 8k 4k; do
rm -rf /tmp/kt; cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --lean --streams 1 --workload $w --lib $R/gpujpeg_amd/lib/$l > /tmp/kt.log 2>&1; cd $R
echo "$l $w: $(python - <<'PY'
import csv,glob
for f in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_gather' in r['Name']: print(r['Name'][:16], round(float(r['AverageNs'])/1e3,2), end='; ')
PY
)"; done; done
