cd "${GRAFT_REPO_ROOT:-.}"
for pat in noise; do for tk in "" "GJ_DEC_TOKENS=1"; do
  env $tk python bench.py --workload 8k --pattern $pat --lean --steps 5 --warmup 2 --mode decode 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$pat [$tk] decode-only', d['value'], 'Mpix/s; solo kernels', r['contended']['kernel_ms'], {k['kernel']:k['ms'] for k in r['by_kernel']})"
done; done
for q in 90 100; do for tk in "" "GJ_DEC_TOKENS=1"; do
  env $tk python bench.py --workload 8k --pattern camera --quality $q --lean --steps 5 --warmup 2 --mode decode 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('camera q$q [$tk] decode-only', d['value'], 'Mpix/s; solo', {k['kernel']:k['ms'] for k in r['by_kernel']})"
done; done
