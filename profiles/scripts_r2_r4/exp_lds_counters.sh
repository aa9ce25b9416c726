#!/bin/bash
# is the entropy decoder bound by LDS? instruction counts, bank-conflict cycles and LDS-busy cycles of the kernels of one 8K frame
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; R=$PWD
rm -rf /tmp/ldsc; cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/ldsc -- python $R/bench.py --lean --streams 1 --steps 3 --warmup 1 --min-seconds 0 > /tmp/ldsc.log 2>&1
tail -2 /tmp/ldsc.log | cut -c1-200
cd $R
python - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob('/tmp/ldsc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if not k.startswith('k_'): continue
        a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, c in acc.items():
    print(k[:40], {n: round(v[1] / v[0]) for n, v in c.items()})
PY
