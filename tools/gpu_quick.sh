#!/bin/bash
# one quick look on the GPU box: phase trace of the token decoder, the lean headline line with the per-kernel times, parity tests of the decoder
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 300 python tools/decoder_phases.py 2>/dev/null | head -24
timeout 300 python bench.py --lean 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['value'], d['unit']); print(json.dumps(d.get('kernels_ms_solo') or d.get('solo_kernel_ms') or {k:v for k,v in d.items() if 'kernel' in k}, indent=0)[:1500])"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${GJ_K:-dec or token or tok}" 2>&1 | tail -3
