cd "$GRAFT_REPO_ROOT"
for b in 1 2 3; do echo "batch $b:"; CORA_STPCG_BATCH=$b python tools/hvp_quick.py b$b | tail -1; CORA_STPCG_BATCH=$b python tools/e2e_1e5.py 100000 120 gt 2>&1 | tail -2; done
