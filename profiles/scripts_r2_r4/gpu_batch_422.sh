#!/bin/bash
# frame batches of packed 4:2:2 frames (interleaved scan: k_encode_uyvy422, k_huffman_decode_par<il> into planes, k_idct_fused_uyvy422): oracle check, then
# frames/s against one libgpujpeg call per frame
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 300 python bench.py --batch 12 --workload hd422 --batch-api batch --batch-streams 1 --verify --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('verify hd422 batch api:', d.get('verified_bit_exact'), d['config']['api'][-34:])"
run() { local label=$1; shift; timeout 300 python bench.py --steps 4 --warmup 2 --batch 256 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label:', d['value'], 'frames/s', d['mpix_s'], 'Mpix/s')"; }
run "hd422 frame S=4" --workload hd422 --batch-api frame --streams 4
run "hd422 batch S=1" --workload hd422 --batch-api batch --batch-streams 1
run "hd422 batch S=2" --workload hd422 --batch-api batch --batch-streams 2
