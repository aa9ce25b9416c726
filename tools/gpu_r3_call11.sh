#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "token or random_streams or 16k_422 or entropy_decoder_variants" 2>&1 | tail -2
for v in tok notok; do
  if [ $v = notok ]; then export GJ_DEC_NO_TOKENS=1; else unset GJ_DEC_NO_TOKENS; fi
  timeout 300 python bench.py --lean --workload 16k422 --streams 1 --mode decode > gpurun_out/r3_11_16k422_$v.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/r3_11_16k422_$v.json')); print('16k422 $v decode-only 1 pipeline', d['value'], [(k['kernel'], k['ms']) for k in d['roofline']['by_kernel'] if k['kernel'].startswith('dec')])"
  timeout 300 python bench.py --lean --workload 16k422 > gpurun_out/r3_11_16k422_head_$v.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/r3_11_16k422_head_$v.json')); print('16k422 $v enc+dec 4 pipelines', d['value'])"
done
unset GJ_DEC_NO_TOKENS
WORKLOAD=16k422 TAG=r3_11_16k422 BENCH_ARGS="--mode decode" timeout 600 tools/profile.sh
grep -E "k_huffman|k_idct|k_marker" gpurun_out/r3_11_16k422_hbm_traffic.txt gpurun_out/r3_11_16k422_sq_counters.txt | cut -c1-220
