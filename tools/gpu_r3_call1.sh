#!/bin/bash
# round 3, first GPU call: parity suite on the new token-mode entropy decoder, A/B of its two copy variants, decoder profile
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[GPUJPEG\]\|Using slower\|Skipping\|No marker\|Expected marker" | tail -8 > gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
for v in 0 1; do
  GJ_DEC_TOK_NOCOOP=$v timeout 300 python bench.py --lean --streams 1 --mode decode --steps 20 > gpurun_out/ab_dec_nocoop$v.json 2> gpurun_out/ab_dec_nocoop$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_dec_nocoop$v.json"))
print("nocoop=$v decode-only 1 pipeline:", d["value"], "Mpix/s", [(k["kernel"], k["ms"]) for k in d["roofline"]["by_kernel"]])
PY
done
for v in 0 1; do
  GJ_DEC_TOK_NOCOOP=$v timeout 300 python bench.py --lean > gpurun_out/ab_head_nocoop$v.json 2> gpurun_out/ab_head_nocoop$v.err
  python -c "import json; d=json.load(open('gpurun_out/ab_head_nocoop$v.json')); print('nocoop=$v headline', d['value'], d['roofline']['contended']['kernel_ms'])"
done
TAG=r3_01_dec_solo BENCH_ARGS="--streams 1 --mode decode" timeout 900 tools/profile.sh > gpurun_out/profile.log 2>&1; tail -30 gpurun_out/profile.log
