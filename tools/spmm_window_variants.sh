#!/bin/bash
# usage (GPU box): tools/spmm_window_variants.sh  -- the X-window kernel against its switches (rebuilds per flag set)
cd "$GRAFT_REPO_ROOT" || exit 1
line() {
  timeout 300 python bench.py --steps 1000 --warmup 100 --cpu-seconds 0.2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
e = d['extras']
print('%-60s hvp %.2f us  hbm-resident %.2f us  spmm %.2f us  in-stpcg %.2f us  iteration %.1f us  parity %.1e' % (sys.argv[1], d['roofline']['kernel_us'], d['roofline_hbm']['kernel_us'], e['spmm_us'], e['hvp_in_stpcg_us'], e['stpcg_iteration_us'], d['parity_max_rel_err_vs_cpu']))
" "$1"
  timeout 300 python bench.py --op cert --rank 10 --steps 500 --warmup 50 --cpu-seconds 0.2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s certificate operator, 10 columns: %.2f us' % ('', d['roofline']['kernel_us']))
"
}
line "shipped build (window, 3 slots, pose slices first)"
CORA_SLICE_LJF=0 line "shipped build, CORA_SLICE_LJF=0 (chain order)"
for flags in "$@"; do
  CORA_EXTRA_HIPCC_FLAGS="$flags" python cora_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  line "$flags"
done
