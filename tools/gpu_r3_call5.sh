#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "token or full_size or random_streams" 2>&1 | tail -2
WORKLOAD=8k TAG=r3_05_8k timeout 900 tools/profile.sh
grep -E "k_huffman_decode_tok|k_idct_tok|k_encode_rgb444|k_assemble|k_marker|k_build" gpurun_out/r3_05_8k_sq_counters.txt gpurun_out/r3_05_8k_hbm_traffic.txt
