#!/bin/bash
# the token-fed IDCT with the zero columns of a wave left out (-DGJ_IDCT_SKIP) at four and at three workgroups per CU
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
cp gpujpeg_amd/lib/libgpujpeg.so /tmp/product.so
cp gpujpeg_amd/lib/libgpujpeg_idct4skip.so gpujpeg_amd/lib/libgpujpeg.so
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 2>&1 | tail -3 ) | tee gpurun_out/r5_idct_tests.txt
cp /tmp/product.so gpujpeg_amd/lib/libgpujpeg.so
tools/r5_ab.sh "default idct4skip idct3 idct3skip" "natural camera" 2>&1 | grep -v amdgpu.ids | grep "==\|k_idct\|four" | tee gpurun_out/r5_idct.txt
