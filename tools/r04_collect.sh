#!/bin/bash
# Round-4 evidence in one go (gpurun_out/r04_*): bench lines, rocprof stats + PMC, rank sweep, data sets, wave timeline, iteration trace
cd "$GRAFT_REPO_ROOT" || exit 1
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
python bench.py --steps 20 > gpurun_out/r04_bench_steps20.json 2>/dev/null
bash tools/refresh_profiles.sh > gpurun_out/r04_refresh.log 2>&1
python tools/rank_sweep.py > gpurun_out/r04_rank_sweep.txt 2>&1
bash tools/datasets_all.sh > gpurun_out/r04_datasets.txt 2>&1
bash tools/stpcg_trace.sh 100000 5 20 > gpurun_out/r04_stpcg_trace.txt 2>&1
bash tools/variant.sh "-DCORA_SPMM_TIMES" python tools/spmm_timeline.py > gpurun_out/r04_timeline.txt 2>&1
CORA_REBUILD_UNITS=kernels_spmm_g0 python cora_amd/build.py > /dev/null 2>&1
tools/bin/launch_lab > gpurun_out/r04_launch_lab.txt 2>&1
echo done
