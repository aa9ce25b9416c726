#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[GPUJPEG\]\|Using slower\|Skipping\|No marker\|Expected marker" | tail -3
rm -rf gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 20 --warmup 3 --streams 1 --mode decode --lean > gpurun_out/prof_stats.log 2>&1
python tools/rocprof_summary.py gpurun_out r3_10_dec > /dev/null 2>&1; grep -E "k_marker|k_huffman_decode_tok|k_idct_tok|Buffer" gpurun_out/r3_10_dec_kernel_stats.txt
timeout 300 python bench.py --lean > gpurun_out/r3_10_head.json 2> gpurun_out/r3_10_head.err
python -c "import json; d=json.load(open('gpurun_out/r3_10_head.json')); print('headline', d['value'], d['roofline']['contended']['kernel_ms'])"
timeout 300 python bench.py --lean --mode decode > gpurun_out/r3_10_deconly.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r3_10_deconly.json')); print('decode only 4 pipelines', d['value'])"
timeout 300 python bench.py --lean --mode encode > gpurun_out/r3_10_enconly.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r3_10_enconly.json')); print('encode only 4 pipelines', d['value'])"
timeout 300 python bench.py --workload 4k --batch 256 --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch256 4k', d['value'], 'frames/s')"
