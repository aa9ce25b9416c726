#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "token or tok or damaged or decoder or dense or baseline or full" 2>&1 | tail -2
for p in natural camera gradient; do timeout 300 python bench.py --lean --pattern $p 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$p', d['value'], [ (k['kernel'],k['ms']) for k in d['roofline']['by_kernel']])"; done
