#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "damaged or refhip_damaged or hostile or header_cache or token or random_streams or full_size" 2>&1 | tail -2
for tb in 0 8 16 32 64; do
  rm -rf gpurun_out/prof_stats
  GJ_SCAN_TB=$tb rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 20 --warmup 3 --streams 1 --mode decode --lean > gpurun_out/prof_stats.log 2>&1
  python tools/rocprof_summary.py gpurun_out r3_08_tb$tb > /dev/null 2>&1
  echo "GJ_SCAN_TB=$tb"; grep -E "k_marker|k_huffman_decode_tok|k_idct_tok" gpurun_out/r3_08_tb${tb}_kernel_stats.txt
done
