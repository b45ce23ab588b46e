#!/bin/bash
# usage: ab_variants.sh OUTDIR "mode:lib1,lib2,..." ...
# A/B of prebuilt library VARIANTS (scratch/libs/<name>.so, built with `make BUILD=build_x OUT=../../scratch/libs/x.so
# EXTRA=-D...`) on ONE box: for every mode the variants are alternated twice; bench.py loads the variant through
# VC_LIB_PATH (the in-tree product library is never touched).  Prints steps/s and the hipEvent kernel average.
out=$1; shift
mkdir -p $out
for ent in "$@"; do
  mode="${ent%%:*}"; libs="${ent#*:}"
  for rep in 1 2; do for lib in ${libs//,/ }; do
    VC_LIB_PATH=$PWD/scratch/libs/$lib.so python bench.py --steps ${STEPS:-100} --warmup 20 --repeats 5 --no-cpu-baseline --no-extra-modes --mode $mode ${BENCH_ARGS} > $out/${mode}_${lib}_$rep.json 2> $out/${mode}_${lib}_$rep.err
    python - <<PY
import json
try:
    j = json.load(open("$out/${mode}_${lib}_$rep.json"))
    print(f"[$mode] %-8s rep $rep  steps/s %8.1f  ms/step %.4f  K_main %7.2f us  hbm %.3f  clock %s" % ("$lib", j["value"], j["ms_per_step"], j["roofline"]["kernel_avg_us"], j["roofline"]["frac"], j["device_clock_mhz"]["after_timed_region"]))
except Exception as e:
    print("[$mode] $lib rep $rep FAILED", e)
PY
  done; done
done
