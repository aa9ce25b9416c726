#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "token or tok or damaged or decoder or dense" 2>&1 | tail -3
python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so 8k 2>/dev/null | tail -1
python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so 4k 2>/dev/null | tail -1
python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so 16k 2>/dev/null | tail -1
for w in 8k 4k; do timeout 300 python bench.py --lean --workload $w 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['metric'], d['value'])"; done
