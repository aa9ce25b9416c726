#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 300 python tools/decoder_phases.py > gpurun_out/r3_04_phases_8k_stagger4.txt 2>&1; grep -v "^\[GPUJPEG\]" gpurun_out/r3_04_phases_8k_stagger4.txt | tail -31 | head -22
for v in "" _stag2 _stag4 _stag6; do
  timeout 300 python bench.py --lib gpujpeg_amd/lib/libgpujpeg$v.so --lean --streams 1 --mode decode --steps 20 > gpurun_out/ab4_dec$v.json 2> gpurun_out/ab4_dec$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/ab4_dec$v.json"))
print("variant '$v' decode-only 1 pipeline:", d["value"], "Mpix/s", [(k["kernel"], k["ms"]) for k in d["roofline"]["by_kernel"] if k["kernel"].startswith("dec")])
PY
done
for v in "" _stag2 _stag4; do
  timeout 300 python bench.py --lib gpujpeg_amd/lib/libgpujpeg$v.so --lean > gpurun_out/ab4_head$v.json 2> gpurun_out/ab4_head$v.err
  python -c "import json; d=json.load(open('gpurun_out/ab4_head$v.json')); print('variant \'$v\' headline', d['value'], d['roofline']['contended']['kernel_ms'])"
done
