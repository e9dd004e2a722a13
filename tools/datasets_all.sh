#!/bin/bash
# every data set of the reference end to end (tools/dataset_solve.py), one line each
cd "$GRAFT_REPO_ROOT" || exit 1
for f in single_drone plaza2 plaza1 tiers mrclam3b mrclam5a mrclam6; do
  python tools/dataset_solve.py tests/golden/datasets/$f.pyfg 2>&1 | tail -1
done
