#!/bin/bash
# per-kernel averages of the bench command under rocprofv3 (kernel trace + stats)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=${1:-gpurun_out/stats_bench}; shift
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b -- python bench.py --steps 300 --warmup 50 --cpu-seconds 0.3 "$@" > $out/bench.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-90s calls %6s avg %9.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
PY
tail -1 $out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d['extras'].items() if 'stpcg' in k or 'apply' in k})"
