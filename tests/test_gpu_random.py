"""A fixed-seed slice of the randomised sweep of tools/stress.py: random graph sizes (1 to 9000 poses, with and
without landmarks, ranges and loop closures), ranks, both formulations, all preconditioners, every operator
against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [3, 11])
def test_random_graphs_match_oracle(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress.py"), "14", str(seed)], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "FAILED" not in out.stdout and "!!" not in out.stdout, out.stdout[-2000:]
    worst = eval(out.stdout.strip().splitlines()[-1])   # the summary dict printed last
    assert all(float(v) < 1e-6 for v in worst.values()), worst
