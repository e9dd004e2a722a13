#!/bin/bash
# usage (GPU box): tools/overlap_trace.sh [poses] [world]  -- kernel trace of partitioned products in the split form: how many of the
# exchange's kernels (k_move_rows: pack / scatter) ran inside the time span of an interior k_spmm launch of the SAME rank (thread)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/overlap_trace; mkdir -p gpurun_out/overlap_trace
CORA_PROBE_ONLY_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/overlap_trace -o t -- python tools/overlap_probe.py "${1:-1000000}" "${2:-2}" 20 2>&1 | tail -2
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/overlap_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by_thread = collections.defaultdict(list)
for r in rows:
    by_thread[r.get("Thread_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
tot = inside = 0
shown = 0
for th, ks in by_thread.items():
    spmm = [(s, e) for s, e, n in ks if "k_spmm" in n and e - s > 20000]   # the interior launches (the long ones)
    for s, e, n in ks:
        if "k_move_rows" not in n:
            continue
        tot += 1
        hit = [(a, b) for a, b in spmm if a <= s and e <= b]
        if hit:
            inside += 1
            if shown < 4:
                a, b = hit[0]
                print("thread %s: k_move_rows %.1f-%.1f us inside k_spmm 0.0-%.1f us" % (th, (s - a) / 1e3, (e - a) / 1e3, (b - a) / 1e3))
                shown += 1
print("k_move_rows launches: %d, of them inside an interior k_spmm of the same rank: %d" % (tot, inside))
PY
find gpurun_out/overlap_trace -name "*.csv" -size +1M -delete
