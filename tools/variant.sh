#!/bin/bash
# usage: tools/variant.sh "<extra hipcc flags>" <command...>  -- rebuilds the library ON THE GPU BOX with the flags, runs the command
# VARIANT_UNITS=<object-name prefixes> (default kernels_spmm_g0: the row strides 2-5 of the product) limits the rebuild
cd "$GRAFT_REPO_ROOT" || exit 1
flags=$1; shift
CORA_EXTRA_HIPCC_FLAGS="$flags" CORA_REBUILD_UNITS="${VARIANT_UNITS:-kernels_spmm_g0}" python cora_amd/build.py > /dev/null 2>&1 || { echo "build failed"; exit 1; }
echo "== variant: $flags"
"$@"
