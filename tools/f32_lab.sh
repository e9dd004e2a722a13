#!/bin/bash
# lab: substitution sweeps with fp32 coefficient storage (kernel build -DCORA_SUB_F32=1 + CORA_SUB_F32=1 at run time)
cd "$GRAFT_REPO_ROOT" || exit 1
CORA_EXTRA_HIPCC_FLAGS="-DCORA_SUB_F32=1" CORA_REBUILD_UNITS=kernels_tri_g0 python cora_amd/build.py > /dev/null 2>&1 || { echo build failed; exit 1; }
CORA_SUB_F32=1 python tools/hvp_quick.py f32_factor | tail -1
CORA_SUB_F32=1 bash tools/stpcg_trace.sh 100000 5 20 2>&1 | tail -8
CORA_SUB_F32=1 python tools/e2e_1e5.py 100000 120 gt 2>&1 | tail -1 | cut -c1-220
CORA_REBUILD_UNITS=kernels_tri_g0 python cora_amd/build.py > /dev/null 2>&1
python tools/hvp_quick.py f64_factor | tail -1
