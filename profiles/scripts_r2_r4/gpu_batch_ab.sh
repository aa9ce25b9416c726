#!/bin/bash
# frame batches on the GPU: the new parity tests, oracle verification through device pointers, then frames/s of the batch calls against
# one libgpujpeg call per frame for 256 x 4K (BASELINE config 5) and 256 x HD, and the 8K headline as a regression check
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "frame_batch or tiles_and_gather or token_mode_decoder or marker_scan" 2>&1 | tail -3 ) | grep -v amdgpu.ids
for w in hd 4k; do
  timeout 300 python bench.py --batch 24 --workload $w --batch-api batch --verify --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('verify $w batch api:', d.get('verified_bit_exact'), d['config']['api'][-40:])"
done
run() { # workload api streams
  timeout 300 python bench.py --batch 256 --workload $1 --batch-api $2 --batch-streams $3 --streams $3 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 streams=$3:', d['value'], 'frames/s', d['mpix_s'], 'Mpix/s')"
}
run 4k frame 4
for s in 1 2 4; do run 4k batch $s; done
run hd frame 4
for s in 1 2 4; do run hd batch $s; done
timeout 300 python bench.py --lean 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8k headline', d['value'], d['roofline']['kernel'], d['roofline']['ms'])"
