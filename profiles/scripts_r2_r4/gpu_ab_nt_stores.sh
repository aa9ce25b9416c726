cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for i in 1 2 3; do for v in nt nont; do
  L=""; [ $v = nont ] && L="--lib gpujpeg_amd/lib/libgpujpeg_nont.so"
  timeout 200 python bench.py --lean --workload 16k422 $L 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'dec solo', d['roofline']['by_direction']['decode']['ms'])"
done; done
for v in nt nont; do L=""; [ $v = nont ] && L="--lib gpujpeg_amd/lib/libgpujpeg_nont.so"
  timeout 200 python bench.py --lean --workload 16k422 --mode decode $L 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v decode only', d['value'])"; done
