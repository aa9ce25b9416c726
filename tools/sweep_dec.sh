#!/bin/bash
# decoder tuning sweep: prints the entropy decoder slot time for (G, SUB) pairs
cd "${GRAFT_REPO_ROOT:-.}"
for sub in 8 16 32; do for G in 8 16 24 32 48 64; do
  r=$(GJ_DEC_SUB=$sub GJ_DEC_G=$G python bench.py --steps 10 --warmup 2 --lean ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms']['dec:k_huffman_decode'], d['value'])")
  echo "sub=$sub G=$G huffman_decode_ms,value = $r"
done; done
