#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline for one round: profiles/tools/collect_profiles.sh r02_x
# (run on the GPU box through gpurun; raw output under gpurun_out/<tag>_*, summaries under gpurun_out/<tag>/ to be
# copied into profiles/).  The profiled command is the driver's own: python3 bench.py --gpus 1 --steps 20 --warmup 5
# (+ --no-cpu-baseline: the CPU leg is not GPU work).  Counter passes are separate runs (kernel-trace / stats never
# together with --pmc), one counter group per pass, as MI355X_MICROARCH.md prescribes.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
CMD="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- $CMD > $out/bench_under_rocprof.json 2> $out/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_eager -- $CMD --no-graph > $out/bench_under_rocprof_eager.json 2> $out/stats_eager.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -- $CMD --no-graph > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write -- $CMD --no-graph > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc_sq -- $CMD --no-graph > /dev/null 2>&1
# (round 6, VERDICT r5 item 4: the wave-cycle side of the VALU floor -- own pass: a counter the part does not have fails this pass only)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/${tag}_pmc_sq2 -- $CMD --no-graph > $out/pmc_sq2.err 2>&1
python3 profiles/summarize.py gpurun_out/${tag}_stats $out/kernel_stats_graph.txt "$tag: $CMD (hipGraph replay)" > /dev/null
python3 profiles/summarize.py gpurun_out/${tag}_stats_eager $out/kernel_stats_eager.txt "$tag: $CMD --no-graph" > /dev/null
for p in fetch write sq sq2; do python3 profiles/summarize.py gpurun_out/${tag}_pmc_$p $out/pmc_$p.txt "$tag: --pmc pass ($p) of $CMD --no-graph" > /dev/null; done
cat $out/pmc_fetch.txt $out/pmc_write.txt $out/pmc_sq.txt $out/pmc_sq2.txt > $out/pmc.txt
# the roofline of bench_under_rocprof_eager.json reproduced from the kernel trace of THAT process (one collection: stats + JSON of one run on one board)
python3 profiles/tools/roofline_from_trace.py gpurun_out/${tag}_stats_eager $out/bench_under_rocprof_eager.json $out/roofline_check.txt > /dev/null
python3 profiles/tools/roofline_from_trace.py gpurun_out/${tag}_stats $out/bench_under_rocprof.json $out/roofline_check_graph.txt > /dev/null
python3 profiles/tools/pmc_to_traffic.py $out/pmc.txt $out/latest_traffic.json $tag
# gpurun copies at most 64 MiB back: the raw traces (7 passes x 10-15 MB) go, their summaries stay; the small per-kernel stats csv of the
# two trace passes is kept beside them
for p in stats stats_eager; do f=$(ls gpurun_out/${tag}_$p/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_$p.csv; done
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_stats_eager gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_sq gpurun_out/${tag}_pmc_sq2
# the plain run last, with the traffic file of THIS library in place (bench.py takes `roofline.traffic` from
# profiles/latest_traffic.json only if its csrc hash matches the library it runs)
cp $out/latest_traffic.json profiles/latest_traffic.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
head -30 $out/kernel_stats_eager.txt
tail -c 600 $out/bench.json
