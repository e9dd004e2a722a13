#!/bin/bash
# Regenerates the round's rocprofv3 evidence under gpurun_out/r06_* (copy the summaries into profiles/ afterwards):
#   kernel trace + stats of the bench command (the in-loop Hvp k_spmm<5, 3, 3> and the back-to-back k_spmm<5, 3, 2> are
#   different kernels, so the stats file's averages ARE the two populations), PMC passes (separate, no trace domains),
#   the trace of one STPCG iteration, the rank sweep and the reference's data sets.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=gpurun_out/r06_stats
rm -rf $R gpurun_out/r06_pmc; mkdir -p $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R -o bench -- python bench.py --steps 1000 --warmup 100 --cpu-seconds 2 --pmc-traffic off > $R/bench.log 2>&1
echo "stats rc=$?"; grep '^{"metric' $R/bench.log > $R/bench_under_rocprof.json; cut -c1-300 $R/bench_under_rocprof.json
python tools/trace_runs.py $R/bench_kernel_trace.csv "k_spmm<5, 3, 2>" 50 > $R/hvp_runs.txt; cat $R/hvp_runs.txt
python tools/trace_runs.py $R/bench_kernel_trace.csv "k_spmm<5, 3, 3>" 1 > $R/hvp_in_loop_runs.txt; tail -3 $R/hvp_in_loop_runs.txt
bash tools/pmc_passes.sh gpurun_out/r06_pmc "k_spmm|k_subblock|k_rowop" tools/pmc_hbm.txt -- python bench.py --steps 50 --warmup 5 --cpu-seconds 0.5 --pmc-traffic off
python tools/pmc_summary.py gpurun_out/r06_pmc > gpurun_out/r06_pmc/summary.txt; cat gpurun_out/r06_pmc/summary.txt
bash tools/stpcg_trace.sh 100000 5 40 > gpurun_out/r06_stpcg_iteration_trace.txt 2>&1; tail -9 gpurun_out/r06_stpcg_iteration_trace.txt
python tools/rank_sweep.py > gpurun_out/r06_rank_sweep.md 2>gpurun_out/r06_rank_sweep.err; tail -12 gpurun_out/r06_rank_sweep.md
bash tools/datasets_all.sh > gpurun_out/r06_datasets.txt 2>&1; cat gpurun_out/r06_datasets.txt
python bench.py 2>/dev/null | grep '^{"metric' > gpurun_out/r06_bench.json                                  # -> profiles/r06_bench.json
python bench.py --op cert --rank 10 2>/dev/null | grep '^{"metric' > gpurun_out/r06_bench_cert.json      # -> profiles/r06_bench_cert.json (BASELINE config 5: 10 columns)
find $R -name "*kernel_stats.csv" | head; find $R gpurun_out/r06_pmc gpurun_out/stpcg_trace -name "*.csv" -size +2M -delete
