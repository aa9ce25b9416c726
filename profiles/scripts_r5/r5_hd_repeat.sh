cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2 3; do for v in r4base default; do
  L=""; [ $v != default ] && L="--lib gpujpeg_amd/lib/libgpujpeg_$v.so"
  for m in both decode; do
  python bench.py --workload hd --lean --steps 20 --warmup 3 --python-loop --mode $m $L 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$rep $v hd $m', d['value'])"
  done
done; done
