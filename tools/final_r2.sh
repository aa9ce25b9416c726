#!/bin/bash
# round-end refresh on the GPU box: GPU test suite, profile passes (kernel trace + HBM traffic + SQ counters), the default bench line
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[GPUJPEG\]\|Using slower\|Skipping\|No marker\|Expected marker" | tail -4
TAG=${TAG:-r2_12} tools/profile.sh > gpurun_out/profile.log 2>&1; tail -3 gpurun_out/profile.log
cp gpurun_out/${TAG:-r2_12}_traffic.json profiles/r2_traffic.json
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json
for s in 3 5 6; do python bench.py --lean --streams $s 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams', d['config']['streams_per_gpu'], d['value'])"; done
