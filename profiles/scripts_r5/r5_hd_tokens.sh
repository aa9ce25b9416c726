#!/bin/bash
# HD frames one call per frame: plane mode (default below 300 K blocks) against forced token mode
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
one() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%10.1f %s' % (d['value'], d['unit']), {k: v for k, v in d['roofline'].get('contended', {}).get('kernel_ms', {}).items()} if '--lean' in sys.argv else '', [(k['kernel'], k['ms']) for k in d['roofline']['by_kernel']])"; }
{
for rep in 1 2; do
for tok in "" 1; do
  export GJ_DEC_TOKENS=$tok; [ -z "$tok" ] && unset GJ_DEC_TOKENS
  for m in both decode; do
    echo -n "rep $rep tokens '$tok' hd $m: "; one --workload hd --lean --mode $m --steps 10 --warmup 3
  done
done
done
} 2>&1 | tee gpurun_out/r5_hd_tokens.txt
