#!/bin/bash
# usage: tools/spmm_variants.sh "<flags 1>" "<flags 2>" ...  -- rebuilds on the GPU box per flag set and prints the Hvp figures
cd "$GRAFT_REPO_ROOT" || exit 1
for flags in "$@"; do
  CORA_EXTRA_HIPCC_FLAGS="$flags" CORA_REBUILD_UNITS="${VARIANT_UNITS:-kernels_spmm_g0}" python cora_amd/build.py > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  timeout 300 python bench.py --steps 1000 --warmup 100 --cpu-seconds 0.2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
e = d['extras']
print('%-50s hvp %.2f us  hbm-resident %.2f us  spmm %.2f us  in-stpcg %.2f us  iteration %.1f us  parity %.1e' % (sys.argv[1], d['roofline']['kernel_us'], d['roofline_hbm']['kernel_us'], e['spmm_us'], e['hvp_in_stpcg_us'], e['stpcg_iteration_us'], d['parity_max_rel_err_vs_cpu']))
" "$flags"
done
