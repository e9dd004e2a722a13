#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for f in plaza2 single_drone; do
  for mode in "" "--implicit"; do
    echo "== $f $mode"; ./examples/cora_main tests/golden/datasets/$f.pyfg $mode 2>&1 | grep -E "final cost|wall clock|error"
  done
done
