#!/bin/bash
# usage: tools/outer_trace.sh <name>  -- what runs BETWEEN two inner solves of the staircase on a reference data set (kernel trace)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/outer_trace; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o t -- python tools/dataset_solve.py tests/golden/datasets/$1.pyfg 2>&1 | tail -1
python - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/outer_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [dict(t0=int(r["Start_Timestamp"]), t1=int(r["End_Timestamp"]), name=r["Kernel_Name"].replace("cora::", "").split("(")[0][:56]) for r in csv.DictReader(open(f))]
for g in glob.glob("gpurun_out/outer_trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        rows.append(dict(t0=int(r["Start_Timestamp"]), t1=int(r["End_Timestamp"]), name="  copy " + r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?"))))
rows.sort(key=lambda r: r["t0"])
inits = [i for i, r in enumerate(rows) if "k_stpcg_init" in r["name"]]
segs = []
for i in inits[1:]:
    a = i
    while a > 0 and "k_tangent_project_update" not in rows[a]["name"] and "k_subblock" not in rows[a]["name"]: a -= 1
    b = i
    while b < len(rows) - 1 and not re.search(r"k_spmm<\d+, \d, 3>", rows[b]["name"]): b += 1
    segs.append((a, b))
print("%d outer segments" % len(segs))
segs.sort(key=lambda ab: rows[ab[1]]["t0"] - rows[ab[0]]["t0"])
a, b = segs[len(segs) // 2]
t0 = rows[a]["t0"]; prev = rows[a]["t1"]
for r in rows[a:b + 1]:
    print("  %8.1f us dur %6.1f gap %6.1f %s" % ((r["t0"] - t0) / 1e3, (r["t1"] - r["t0"]) / 1e3, (r["t0"] - prev) / 1e3, r["name"]))
    prev = max(prev, r["t1"])
import statistics
lens = [(rows[b]["t0"] - rows[a]["t0"]) / 1e3 for a, b in segs]
print("outer segment (last kernel of an inner solve -> first in-loop product of the next): median %.1f us, mean %.1f us, total %.1f ms" % (statistics.median(lens), statistics.mean(lens), sum(lens) / 1e3))
PY
find $out -name "*.csv" -size +1M -delete
