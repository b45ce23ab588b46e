#!/bin/bash
# usage: ab_libs.sh libA.so libB.so ... (prebuilt under scratch/libs): alternate them on this box, rocprof kernel averages
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
n=0
for rep in 1 2; do for lib in "$@"; do
  n=$((n+1)); cp scratch/libs/$lib velocycle_amd/libvelocycle_hip.so
  for mode in ${MODES:-vjoint vcond phase}; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_${n}_$mode -- python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-modes --mode $mode ${BENCH_ARGS} > gpurun_out/abl_${n}_$mode.log 2>&1
    python - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/abl_${n}_$mode/**/*kernel_stats.csv", recursive=True)[0]
val = [json.loads(l)["value"] for l in open("gpurun_out/abl_${n}_$mode.log") if l.startswith('{"metric"')]
out = [f"{float(r['AverageNs'])/1e3:7.2f}" for r in csv.DictReader(open(f)) if "vc_main" in r["Name"] and int(r["Calls"]) > 10]
print("[$lib] $mode steps/s", val, "K_main", out)
PY
  done
done; done
