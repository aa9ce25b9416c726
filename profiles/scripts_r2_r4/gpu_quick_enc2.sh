#!/bin/bash
# one look at the encoder on the GPU: its parity tests, its kernels alone at 8K / 4K / HD (rocprofv3), the headline and the batched workloads
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -q -x -n 4 -k "${GJ_K:-bit_exact or tiles_and_gather or frame_batch or random_configurations or packed_422 or tst_patterns}" 2>&1 | tail -2
for w in 8k 4k hd; do
rm -rf /tmp/kt; cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --lean --streams 1 --workload $w > /tmp/kt.log 2>&1; cd $R
echo "$w alone: $(python - <<'PY'
import csv,glob
for f in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Name'].startswith('k_') or ' k_' in r['Name'][:8]: print(r['Name'].split('(')[0][:24], round(float(r['AverageNs'])/1e3,2), end='; ')
PY
)"; done
timeout 300 python bench.py --lean 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8k headline', d['value'], d['roofline']['kernel'], d['roofline']['ms'])"
timeout 300 python bench.py --lean --mode encode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8k encode only', d['value'])"
for w in 4k hd; do timeout 300 python bench.py --steps 6 --warmup 2 --batch 256 --batch-api batch --batch-streams 1 --workload $w 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w batched', d['value'], 'frames/s')"; done
