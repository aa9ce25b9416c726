#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_refhip.py -q -x -n 4 -k "encode or tiles or bit_exact or refhip or stress" 2>&1 | tail -2
for w in 8k 4k hd; do python tools/solo_kernels.py gpujpeg_amd/lib/libgpujpeg.so $w 2>/dev/null | tail -1; done
timeout 300 python bench.py --lean 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['metric'], d['value'])"
