#!/bin/bash
# k_idct_tok_rgb444 without its two spilled registers (count and DC term kept packed) against the tree before
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 -k "token or tok or 8k or 4k or 16k or batch or fuzz" 2>&1 | tail -3 ) | tee gpurun_out/r5_idct2_tests.txt
{
tools/r5_ab.sh "base default" "natural camera" 2>&1 | grep -v amdgpu.ids | grep "==\|k_idct\|four"
for rep in 1 2; do for v in base default; do
  L=""; [ $v != default ] && L="--lib gpujpeg_amd/lib/libgpujpeg_$v.so"
  for m in both decode; do
    python bench.py --workload 8k --lean --steps 20 --warmup 3 --python-loop --mode $m $L 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   rep $rep four pipelines $v $m', d['value'])"
  done
  python bench.py --workload 4k --lean --steps 20 --warmup 3 --python-loop --mode both $L 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   rep $rep four pipelines 4k $v both', d['value'])"
done; done
} | tee gpurun_out/r5_idct2.txt
