#!/bin/bash
# round 3, third GPU call: phase timeline of the token-mode entropy decoder, A/B of sub-sequence lengths
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "token or full_size or random_streams or batch_frames" 2>&1 | tail -3
timeout 300 python tools/decoder_phases.py > gpurun_out/r3_03_phases_8k.txt 2>&1; cat gpurun_out/r3_02_phases_8k.txt | grep -v "^\[GPUJPEG\]" | tail -34
for v in "" _sync48 _sync96; do
  timeout 300 python bench.py --lib gpujpeg_amd/lib/libgpujpeg$v.so --lean --streams 1 --mode decode --steps 20 > gpurun_out/ab2_dec$v.json 2> gpurun_out/ab2_dec$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/ab2_dec$v.json"))
print("variant '$v' decode-only 1 pipeline:", d["value"], "Mpix/s", [(k["kernel"], k["ms"]) for k in d["roofline"]["by_kernel"] if k["kernel"].startswith("dec")])
PY
done
for v in "" _sync48 _sync96; do
  timeout 300 python bench.py --lib gpujpeg_amd/lib/libgpujpeg$v.so --lean > gpurun_out/ab2_head$v.json 2> gpurun_out/ab2_head$v.err
  python -c "import json; d=json.load(open('gpurun_out/ab2_head$v.json')); print('variant \'$v\' headline', d['value'], d['roofline']['contended']['kernel_ms'])"
done
