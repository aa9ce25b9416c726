#!/bin/bash
# One gpurun call = one session on the GPU box: the steps named on the command line run in order, everything lands under gpurun_out/<TAG>_*.
#   usage: tools/gpu_session.sh TAG step [step ...]
#   steps: tests            pytest -m gpu (all)                         tests:<expr>     pytest -m gpu -k <expr>
#          bench            python bench.py (the driver's command)      bench:<args>     python bench.py <args> (spaces as commas)
#          prof:<workload>  tools/profile.sh for hd|4k|8k|16k|16k422    profall          tools/profile_all.sh
#          sh:<script>      another script of tools/ (arguments after commas)
# (round 5 kept one script per experiment, profiles/scripts_r5/; this replaces them)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out
mkdir -p $OUT
for step in "$@"; do
  name=${step%%:*}; arg=${step#*:}; [ "$arg" = "$step" ] && arg=""
  arg=${arg//,/ }
  echo "=== $step"
  case $name in
    tests) if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$arg" > $OUT/${TAG}_gpu_tests.txt 2>&1; else timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.txt 2>&1; fi
           tail -3 $OUT/${TAG}_gpu_tests.txt ;;
    bench) n=$(ls $OUT/${TAG}_bench*.json 2>/dev/null | wc -l)
           timeout 900 python bench.py $arg > $OUT/${TAG}_bench$n.json 2> $OUT/${TAG}_bench$n.err; tail -c 400 $OUT/${TAG}_bench$n.err
           python tools/bench_brief.py $OUT/${TAG}_bench$n.json ;;
    prof)  WORKLOAD=$arg TAG=${TAG}_$arg tools/profile.sh 2>&1 | tail -14 ;;
    profall) tools/profile_all.sh 2>&1 | tail -60 ;;
    sh)    set -- $arg; s=$1; shift; bash tools/$s "$@" 2>&1 | tail -40 ;;
    *) echo "unknown step $step" ;;
  esac
done
