#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for l in libgpujpeg.so libgpujpeg_five.so; do for w in 8k 4k 16k; do python tools/solo_kernels.py gpujpeg_amd/lib/$l $w 2>/dev/null | tail -1; done; done
for l in libgpujpeg.so libgpujpeg_five.so; do for w in 8k 4k; do timeout 300 python bench.py --lean --workload $w --lib gpujpeg_amd/lib/$l 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['metric'], d['value'])"; done; done
GJ_TRACE_LIB=x true
