#!/bin/bash
# lab: fp32 coefficient storage of the substitution sweeps at 10^6 poses (bandwidth-bound sweeps) -- iteration trace with and without
cd "$GRAFT_REPO_ROOT" || exit 1
CORA_EXTRA_HIPCC_FLAGS="-DCORA_SUB_F32=1" CORA_REBUILD_UNITS=kernels_tri_g0 python cora_amd/build.py > /dev/null 2>&1 || { echo build failed; exit 1; }
echo "== fp32 coefficients"; CORA_SUB_F32=1 bash tools/stpcg_trace.sh 1000000 5 12 2>&1 | tail -8 | grep -v "^W2026\|^E2026"
CORA_REBUILD_UNITS=kernels_tri_g0 python cora_amd/build.py > /dev/null 2>&1
echo "== fp64 coefficients"; bash tools/stpcg_trace.sh 1000000 5 12 2>&1 | tail -8 | grep -v "^W2026\|^E2026"
