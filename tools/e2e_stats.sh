#!/bin/bash
# usage: tools/e2e_stats.sh <plaza2|1e4|...>  -- rocprofv3 kernel stats of the end-to-end staircase (top kernels by total time)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/e2e_stats; mkdir -p gpurun_out/e2e_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e2e_stats -o e -- python tools/e2e_probe.py "$@" 2>&1 | grep -v "^[EW]2026" | tail -3
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/e2e_stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("device busy %.3f s in %d launches" % (tot / 1e9, sum(int(r["Calls"]) for r in rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%6.1f%%  calls %7s  avg %8.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"].replace("cora::", "")[:70]))
PY
find gpurun_out/e2e_stats -name "*.csv" -size +1M -delete
