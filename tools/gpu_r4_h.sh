#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
T=${TAG:-r4i}
timeout 600 python -m pytest tests -m gpu -x -q -k "tiles_and_gather or full_size" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -3 gpurun_out/${T}_pytest.log
run() { echo -n "$1 $2 $3 "; env $1 timeout 120 python tools/solo_kernels.py gpujpeg_amd/lib/$2 $3 2>&1 | tail -1 | sed 's/np.float64(\([0-9.]*\))/\1/g'; }
{
for w in 8k 4k hd 16k; do run X=0 libgpujpeg.so $w; done
for r in ${RES:-896 1280 2058}; do run GJ_ENC_RESIDENT=$r libgpujpeg.so 8k; done
for l in ${LIBS}; do run X=0 $l 8k; run X=0 $l 4k; done
} > gpurun_out/${T}_solo.txt 2>&1
cat gpurun_out/${T}_solo.txt
timeout 200 python tools/encoder_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_phases_8k.txt; tail -4 gpurun_out/${T}_phases_8k.txt
export TMPDIR=/tmp
rm -rf gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 20 --warmup 3 --workload 8k --streams 1 --lean > gpurun_out/${T}_prof_stats.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_stats/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:6]:
    print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
rm -rf gpurun_out/prof_sq
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/prof_sq -- python bench.py --steps 3 --warmup 1 --min-seconds 0 --workload 8k --streams 1 --lean > gpurun_out/${T}_prof_sq.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    if not k.startswith("k_"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, v in acc.items():
    n = max(cnt[k], 1)
    print(k, "dispatches", n, " ".join(f"{c}={x / n:.4g}" for c, x in sorted(v.items())))
PY
