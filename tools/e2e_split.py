"""Time split of the staircase on a data set (solveCORA verbose + CORA_TRI_TIMING).  python tools/e2e_split.py [file.pyfg]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CORA_TRI_TIMING"] = "1"
from cora_amd import capi, host
f = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "datasets", "plaza2.pyfg")
P = host.Problem.from_pyfg(f)
P.update()
x0 = P.op("getRandomInitialGuess")
P.precond_info()
for rep in range(2):
    t = time.perf_counter()
    res = P.solve(x0, max_rank=10, max_seconds=600, verbose=(rep == 1))
    print("run %d: staircase %.3f s f=%.6f hvps=%d levels=%d" % (rep, time.perf_counter() - t, res["f"], res["hvps"], res["levels"]), flush=True)
