#!/bin/bash
# usage: ab_multi.sh "flags|ENV=.. ENV2=.." ... : per entry rebuild (if flags changed) and print rocprof kernel averages
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
n=0; last="__none__"
for ent in "$@"; do
  fl="${ent%%|*}"; ev="${ent#*|}"
  n=$((n+1))
  if [ "$fl" != "$last" ]; then
    make -C velocycle_amd/csrc clean >/dev/null 2>&1
    make -C velocycle_amd/csrc -j32 EXTRA="$fl" 2>&1 | grep -E "error" | head -3
    last="$fl"
  fi
  for mode in ${MODES:-vjoint vcond}; do
    env $ev rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abm_${n}_$mode -- python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-modes --mode $mode ${BENCH_ARGS} > gpurun_out/abm_${n}_$mode.log 2>&1
    python - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/abm_${n}_$mode/**/*kernel_stats.csv", recursive=True)[0]
val = [json.loads(l)["value"] for l in open("gpurun_out/abm_${n}_$mode.log") if l.startswith('{"metric"')]
out = []
for r in csv.DictReader(open(f)):
    if "vc_main" in r["Name"]:
        out.append(f"{r['Name'].split('<')[1][:22]} {float(r['AverageNs'])/1e3:7.2f}")
print("[$fl | $ev] $mode steps/s", val, " | ".join(out))
PY
  done
done
