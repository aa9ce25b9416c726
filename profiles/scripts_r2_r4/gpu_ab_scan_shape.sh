cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; R=$PWD
for sh in 0 401 1601; do for w in 8k 4k; do
rm -rf /tmp/kt; cd /tmp; GJ_SCAN_SHAPE=$sh timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --lean --streams 1 --workload $w > /tmp/kt.log 2>&1; cd $R
echo "shape $sh $w: $(python - <<'PY'
import csv,glob
for f in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_marker' in r['Name']: print(r['Name'].split('(')[0].replace('void ','')[:22], round(float(r['AverageNs'])/1e3,2), end='; ')
PY
)"; done; done
