#!/bin/bash
# usage: tools/stpcg_trace.sh [n_poses] [p] [iterations] [dim]  -- kernel trace of one STPCG iteration (the last full one)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/stpcg_trace; mkdir -p gpurun_out/stpcg_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/stpcg_trace -o t -- python tools/stpcg_probe.py "$@" 2>&1 | tail -4
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/stpcg_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import re
# an iteration starts with the product that carries kappa (k_spmm<LD, d, 3> = EPI_HVP_K)
idx = [i for i, r in enumerate(rows) if re.search(r"k_spmm<\d+, \d, 3>", r["Kernel_Name"])]
a, b = idx[-3], idx[-2]
t0 = int(rows[a - 1]["End_Timestamp"])
busy = 0
prev_end = t0
print("kernels in one iteration:", b - a)
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    name = r["Kernel_Name"].replace("cora::", "").split("(")[0][:44]
    print("%8.1f us  dur %6.1f  gap %5.1f  grid %8s wg %4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), name))
    prev_end = e
print("total %.1f us busy %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
PY
find gpurun_out/stpcg_trace -name "*.csv" -size +1M -delete
