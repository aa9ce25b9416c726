#!/bin/bash
# standard GPU-box check: parity tests, then the headline bench. Outputs land in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout ${PYTEST_TIMEOUT:-600} python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python bench.py ${BENCH_ARGS} > gpurun_out/bench.log 2> gpurun_out/bench.err
tail -3 gpurun_out/bench.err
cat gpurun_out/bench.log
