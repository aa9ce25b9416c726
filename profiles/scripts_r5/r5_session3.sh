#!/bin/bash
# round 5, session 3: host link in both directions; the token decoder's stage with a padding dword per 32 (GJ_TOK_PAD) against the product
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/ubench/pcie_duplex.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_pcie_duplex.txt
# parity of the variant: the token tests with the variant in the product's place (this copy of the tree only)
cp gpujpeg_amd/lib/libgpujpeg.so /tmp/product.so
cp gpujpeg_amd/lib/libgpujpeg_tokpad.so gpujpeg_amd/lib/libgpujpeg.so
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 4 -k "token or tok or 8k or 4k or 16k" 2>&1 | tail -3 ) | tee gpurun_out/r5_tokpad_tests.txt
cp /tmp/product.so gpujpeg_amd/lib/libgpujpeg.so
{
tools/r5_ab.sh "default tokpad" "natural camera"
for v in default tokpad; do
  L=""; [ $v != default ] && L="--lib gpujpeg_amd/lib/libgpujpeg_$v.so"
  rm -rf /tmp/ldsc
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/ldsc -- python bench.py --workload 8k --lean --streams 1 --steps 3 --warmup 1 --min-seconds 0 $L > /tmp/ldsc.log 2>&1
  python - $v <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob('/tmp/ldsc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if not k.startswith('k_'): continue
        a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, c in acc.items():
    d = {n: v[1] / v[0] for n, v in c.items()}
    print("%-8s %-36s LDS insts %9d  IDX_ACTIVE %10d  BANK_CONFLICT %10d  conflict share %.3f" % (sys.argv[1], k[:36], d.get('SQ_INSTS_LDS', 0), d.get('SQ_LDS_IDX_ACTIVE', 0), d.get('SQ_LDS_BANK_CONFLICT', 0), d.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, d.get('SQ_LDS_IDX_ACTIVE', 1))))
PY
done
tools/r5_sq.sh "default tokpad" "k_huffman_decode_tok"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_tokpad.txt
