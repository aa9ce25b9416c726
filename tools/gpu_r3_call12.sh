#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for s in 2 3 4 5 6 8; do timeout 200 python bench.py --lean --streams $s 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams', d['config']['streams_per_gpu'], d['value'])"; done
timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_multiproc.py -m gpu -x -q 2>&1 | tail -3
