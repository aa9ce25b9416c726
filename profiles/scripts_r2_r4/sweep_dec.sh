#!/bin/bash
# sweep of the sub-sequence decoder's batch size (segments per workgroup) and sub-sequence length: solo kernel time + 4-pipeline rate
cd "${GRAFT_REPO_ROOT:-.}"
for G in ${GS:-16 21 28 33 43}; do
  for sub in ${SUBS:-16}; do
    r=$(GJ_DEC_SUB=$sub GJ_DEC_G=$G python bench.py --steps 5 --warmup 2 --lean ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k={x['kernel']:x['ms'] for x in d['roofline']['by_kernel']}; print(k.get('dec:k_huffman_decode_par'), k.get('dec:k_idct_tok_rgb444'), d['value'])")
    echo "G=$G sub=$sub -> $r"
  done
done
