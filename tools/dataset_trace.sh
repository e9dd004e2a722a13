#!/bin/bash
# usage: tools/dataset_trace.sh <name>  -- kernel trace of the staircase on a reference data set: per-kernel totals and one STPCG iteration
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/dataset_trace; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python tools/dataset_solve.py tests/golden/datasets/$1.pyfg 2>&1 | tail -1
python - <<'PY'
import csv, glob, re, collections
f = glob.glob("gpurun_out/dataset_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"].replace("cora::", "").split("(")[0][:50]
    tot[n][0] += 1; tot[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernel totals:")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-52s calls %6d total %9.1f us avg %6.2f us" % (n, c, t, t / c))
idx = [i for i, r in enumerate(rows) if re.search(r"k_spmm<\d+, \d, 3>", r["Kernel_Name"])]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[a]["Start_Timestamp"]); prev = t0
print("one STPCG iteration in the middle of the run (%d kernels):" % (b - a))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  %7.1f us dur %5.1f gap %4.1f grid %7s wg %4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r["Kernel_Name"].replace("cora::", "").split("(")[0][:48]))
    prev = e
print("  iteration: %.1f us (next product starts %.1f us after this one)" % ((prev - t0) / 1e3, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
busy = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows) / 1e3
print("whole run: %d kernels, busy %.0f us of %.0f us" % (len(rows), busy, span))
PY
find $out -name "*.csv" -size +1M -delete
