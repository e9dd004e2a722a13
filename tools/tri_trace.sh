#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/tri
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tri -o tri -- python tools/tri_probe.py 100000 5 5 2>&1 | tail -5
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tri/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last apply: find the last k_tangent_project and walk back to the previous one
idx = [i for i, r in enumerate(rows) if "tangent_project" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
busy = 0
prev_end = t0
print("n kernels in one apply:", b - a)
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    name = r["Kernel_Name"].split("(")[0][:40]
    print("%8.1f us  dur %6.1f  gap %5.1f  grid %8s wg %4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), name))
    prev_end = e
print("total %.1f us busy %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
PY
