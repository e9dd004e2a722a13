"""Splits a rocprofv3 kernel trace (…_kernel_trace.csv) into runs of consecutive launches of one kernel and prints
the average duration of each long run -- so that the bench's timed region (K back-to-back launches of the Hvp
kernel) can be read out of a trace that also holds the rotated (HBM-resident) launches and the solver legs.

    python tools/trace_runs.py gpurun_out/r02_stats/bench_kernel_trace.csv 'k_spmm<5, 3, 2>' [min_run]
"""
import csv
import sys


def main():
    path, needle = sys.argv[1], sys.argv[2]
    min_run = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    runs, cur = [], []
    for s, e, name in rows:
        if needle in name:
            cur.append(e - s)
        else:
            if len(cur) >= min_run:
                runs.append(cur)
            cur = []
    if len(cur) >= min_run:
        runs.append(cur)
    print(f"# runs of >= {min_run} consecutive launches of a kernel matching {needle!r} in {path}")
    for i, run in enumerate(runs):
        run_sorted = sorted(run)
        print(f"run {i}: launches={len(run)} avg_us={sum(run) / len(run) / 1e3:.3f} "
              f"median_us={run_sorted[len(run) // 2] / 1e3:.3f} min_us={run_sorted[0] / 1e3:.3f} max_us={run_sorted[-1] / 1e3:.3f}")


if __name__ == "__main__":
    main()
