#!/bin/bash
# the last tiles of an 8K frame coded one component per workgroup (-DGJ_ENC_TAIL_SPLIT=n): parity of one variant, then solo durations and four-pipeline rates
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
cp gpujpeg_amd/lib/libgpujpeg.so /tmp/product.so
cp gpujpeg_amd/lib/libgpujpeg_tail256.so gpujpeg_amd/lib/libgpujpeg.so
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 -k "8k or 4k or 16k or bit_exact" 2>&1 | tail -3 ) | tee gpurun_out/r5_tail_tests.txt
cp /tmp/product.so gpujpeg_amd/lib/libgpujpeg.so
tools/r5_ab.sh "default tail64 tail256 tail512" "natural" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_tail.txt
