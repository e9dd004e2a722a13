#!/bin/bash
# quick GPU check of the implicit formulation + the suites touched by the factor refactor
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_implicit.py tests/test_gpu_solver.py tests/test_gpu_host.py tests/test_gpu_cora.py -x -q 2>&1 | tail -30
