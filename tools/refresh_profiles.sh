#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ (copy the summaries into profiles/ afterwards).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/r04_stats gpurun_out/r04_pmc
mkdir -p gpurun_out/r04_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_stats -o bench -- python bench.py --steps 1000 --warmup 100 --cpu-seconds 2 > gpurun_out/r04_stats/bench.log 2>&1
echo "stats rc=$?"; grep '^{"metric' gpurun_out/r04_stats/bench.log > gpurun_out/r04_stats/bench_under_rocprof.json; cut -c1-300 gpurun_out/r04_stats/bench_under_rocprof.json
python tools/trace_runs.py gpurun_out/r04_stats/bench_kernel_trace.csv "k_spmm<5, 3, 2>" 50 > gpurun_out/r04_stats/hvp_runs.txt; cat gpurun_out/r04_stats/hvp_runs.txt
bash tools/pmc_passes.sh gpurun_out/r04_pmc "k_spmm|k_subblock|k_rowop" tools/pmc_hbm.txt -- python bench.py --steps 50 --warmup 5 --cpu-seconds 0.5
python tools/pmc_summary.py gpurun_out/r04_pmc > gpurun_out/r04_pmc/summary.txt; cat gpurun_out/r04_pmc/summary.txt
find gpurun_out/r04_stats -name "*kernel_stats.csv" | head; find gpurun_out/r04_stats -name "*.csv" -size +2M -delete
