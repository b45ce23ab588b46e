#!/bin/bash
# usage: ab_env.sh OUTDIR "mode:NAME=VAL[+NAME=VAL...],..." ...     ("-" = no override)
# A/B of run-time knobs (VC_GPL, VC_BLOCKS_PER_CU, VC_CELLS_PER_WAVE, VC_COUNT_STORAGE ...) of the in-tree library on ONE
# box: for every mode the settings are alternated twice.  Prints steps/s, the hipEvent kernel average and the kernel name.
out=$1; shift
mkdir -p $out
for ent in "$@"; do
  mode="${ent%%:*}"; sets="${ent#*:}"
  for rep in 1 2; do for s in ${sets//,/ }; do
    tag="${s//[^A-Za-z0-9_]/_}"
    ( if [ "$s" != "-" ]; then for kv in ${s//+/ }; do export "$kv"; done; fi
      python bench.py --steps ${STEPS:-100} --warmup 20 --repeats 5 --no-cpu-baseline --no-extra-modes --mode $mode ${BENCH_ARGS} > $out/${mode}_${tag}_$rep.json 2> $out/${mode}_${tag}_$rep.err )
    python - <<PY
import json
try:
    j = json.load(open("$out/${mode}_${tag}_$rep.json"))
    r = j["roofline"]
    print(f"[$mode] %-28s rep $rep  steps/s %8.1f  ms/step %.4f  K_main %7.2f us  %s" % ("$s", j["value"], j["ms_per_step"], r["kernel_avg_us"], r["kernel"]))
except Exception as e:
    print("[$mode] $s rep $rep FAILED", e)
PY
  done; done
done
