cd "${GRAFT_REPO_ROOT:-.}"
for s in 4 6 8 12; do
  python bench.py --batch 256 --workload 4k --batch-io host --streams $s --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('host-staged 256 x 4K, $s pipelines:', d['value'], d['unit'])"
done
