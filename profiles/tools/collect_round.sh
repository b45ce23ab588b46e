#!/bin/bash
# Everything profiles/README.md lists for a round, in one gpurun call:  gpurun -- 'bash profiles/tools/collect_round.sh r06'
# (collect_profiles.sh = the driver's command under rocprofv3 + the PMC passes + the plain bench lines; then the shard sweep as a rank of an
# N-rank run sees it and as a single rank, the K-particle step, the tutorial-shaped flow, two more bench lines, the 400 000-cell line).
# Output: gpurun_out/<tag>/ and gpurun_out/<tag>_*; copy what is to be judged into profiles/.
tag=${1:-r06}
bash profiles/tools/collect_profiles.sh $tag > gpurun_out/${tag}_collect.log 2>&1
VC_PW_INLINE=0 python profiles/tools/step_time_vs_shard.py vjoint --nccl > gpurun_out/${tag}_step_vs_shard_vjoint_as_multirank.txt 2>&1
python profiles/tools/step_time_vs_shard.py vjoint > gpurun_out/${tag}_step_vs_shard_vjoint_1rank.txt 2>&1
python profiles/tools/particles_step.py vjoint 3 > gpurun_out/${tag}_particles.txt 2>&1
python profiles/tools/fit_wall_time.py 50000 2000 1000 500 sparse > gpurun_out/${tag}_fit_wall_time_50k_sparse.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_second_run.json 2> /dev/null
python bench.py > gpurun_out/${tag}_bench_default_steps.json 2> /dev/null
python bench.py --cells 400000 --no-cpu-baseline --no-extra-modes --steps 20 --warmup 5 > gpurun_out/${tag}_bench_400k_cells.json 2>/dev/null
tail -3 gpurun_out/${tag}_collect.log | cut -c1-400
