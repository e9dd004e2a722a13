"""Quick Hvp timing at 10^5 poses: back to back, rotated (HBM-resident) and inside STPCG, for variant builds.
python tools/hvp_quick.py [label]"""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "1000", "--warmup", "100", "--cpu-seconds", "0.2", "--pmc-traffic", "off"],
                     capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(out); e = d["extras"]
print("%-50s hvp %.2f us | hbm-resident %.2f us | spmm %.2f (hbm %.2f) | in-stpcg %.2f | iteration %.1f | parity %.1e" % (
    sys.argv[1] if len(sys.argv) > 1 else "", d.get("roofline_cache", d["roofline"])["kernel_us"], d["roofline_hbm"]["kernel_us"], e["spmm_us"],
    d["roofline_hbm_spmm"]["kernel_us"], e["hvp_in_stpcg_us"], e["stpcg_iteration_us"], d["parity_max_rel_err_vs_cpu"]), flush=True)
