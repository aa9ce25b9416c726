cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
run() { local label=$1; shift; timeout 300 python bench.py --steps 6 --warmup 2 --batch 256 --batch-api batch "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label:', d['value'], 'frames/s', d['config']['api'][-36:])"; }
run "hd S=1" --workload hd --batch-streams 1
run "hd S=2" --workload hd --batch-streams 2
run "hd422 S=2" --workload hd422 --batch-streams 2
run "4k S=1" --workload 4k --batch-streams 1
run "4k S=2" --workload 4k --batch-streams 2
