#!/bin/bash
# usage: tools/variant.sh "<extra hipcc flags>" <command...>  -- rebuilds the library ON THE GPU BOX with the flags, runs the command
cd "$GRAFT_REPO_ROOT" || exit 1
flags=$1; shift
CORA_EXTRA_HIPCC_FLAGS="$flags" python cora_amd/build.py --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
echo "== variant: $flags"
"$@"
