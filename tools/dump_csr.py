"""Dumps a synthetic problem's CSR to a flat binary for tools/spmm_lab."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from synth import make_problem
n = int(sys.argv[1]); out = sys.argv[2]
A, Q, dm = make_problem(d=3, n=n, n_landmarks=10, n_ranges=n // 2, seed=42)
with open(out, "wb") as f:
    np.array([dm.d, dm.n, dm.r, dm.n_trans, Q.nnz], dtype=np.int64).tofile(f)
    Q.rowptr.tofile(f); Q.col.tofile(f); Q.val.tofile(f)
print("dumped", dm.N, Q.nnz)
