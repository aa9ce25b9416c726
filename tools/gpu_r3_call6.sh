#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[GPUJPEG\]\|Using slower\|Skipping\|No marker\|Expected marker" | tail -6 > gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
WORKLOAD=8k TAG=r3_07_8k timeout 900 tools/profile.sh
timeout 300 python bench.py --lean > gpurun_out/r3_07_head.json 2> gpurun_out/r3_07_head.err
python -c "import json; d=json.load(open('gpurun_out/r3_07_head.json')); print('headline', d['value'], d['roofline']['contended']['kernel_ms'])"
for w in hd 4k; do timeout 300 python bench.py --lean --workload $w > gpurun_out/r3_07_$w.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r3_07_$w.json')); print('$w', d['value'], [(k['kernel'], k['ms']) for k in d['roofline']['by_kernel']])"; done
