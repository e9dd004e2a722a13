"""The reference's OWN test translation units -- ALL SEVEN: tests/test_optimizer_helpers.cpp, tests/test_cora.cpp,
tests/test_parse_pyfg.cpp, tests/test_certification.cpp, tests/test_geometry.cpp,
tests/test_construct_problem.cpp, tests/test.cpp -- compiled unmodified, from where they lie in the reference tree, against include/CORA/*.h and
libcora_hip.so (oracle/build_ref_tests.py; Catch2 and the Eigen-based test helper replaced by the stand-ins under
tests/drop_in/shim/), and RUN against the committed golden fixtures (byte-identical copies of the reference's tests/data).

The binaries are built in the build container, where /root/reference is mounted (__graft_entry__.build()), into
oracle/_ref/ (git-ignored) and travel to the GPU box with the snapshot; the reference's sources never do."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import build_ref_tests  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _binary(unit):
    if build_ref_tests.available():
        build_ref_tests.build()
    path = os.path.join(build_ref_tests.OUT, "ref_" + unit)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/ref_%s was not built: the reference tree is only mounted in the build container" % unit)
    return path


def _run(unit, tmp_path):
    """The reference's tests look their data up under <cwd>/bin/data (tests/test_utils.cpp:96-107): give them that."""
    exe = _binary(unit)
    os.makedirs(tmp_path / "bin", exist_ok=True)
    if not os.path.exists(tmp_path / "bin" / "data"):
        os.symlink(GOLDEN, tmp_path / "bin" / "data")
    r = subprocess.run([exe], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout[-3000:])
    return r


def test_reference_parse_pyfg_tests_pass(tmp_path):
    """tests/test_parse_pyfg.cpp:16-37: the three factor graphs parse and all eight sub-matrices of each equal the
    reference's fixtures (Eigen's isApprox) -- host code only, runs without a GPU."""
    r = _run("test_parse_pyfg", tmp_path)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "3 test cases, 0 failed" in r.stdout


def test_reference_construct_problem_tests_pass(tmp_path):
    """tests/test_construct_problem.cpp:22-127: a problem built through the add* API (one odometry edge; one range edge between
    landmarks) has the fixtures' sub-matrices, and the true state -- times any orthogonal matrix -- lies in the null space of
    its data matrix.  Host code only, runs without a GPU."""
    r = _run("test_construct_problem", tmp_path)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "2 test cases, 0 failed, 7 assertions" in r.stdout


def test_reference_block_cholesky_tests_pass(tmp_path):
    """tests/test.cpp:25-214 (round 6: the seventh and last unit): getBlockCholeskyFactorization / blockCholeskySolve of a
    hand-coded 3 x 3 matrix, of a random SPD matrix of odd size below 100 and of two 3 x 3 diagonal blocks of a 6 x 6 matrix,
    each against the dense inverse -- on the identity, on the matrix itself and on a vector (Eigen's isApprox, 1e-12).
    Built with Eigen's element-wise sparse interface (insert / coeff / coeffRef / makeCompressed), the VectorXi comma
    initialiser and Catch2's GENERATE(take(1, filter(.., random(..)))).  Host code only, runs without a GPU."""
    r = _run("test", tmp_path)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "3 test cases, 0 failed, 10 assertions" in r.stdout


@pytest.mark.gpu
def test_reference_optimizer_helper_tests_pass(tmp_path):
    """tests/test_optimizer_helpers.cpp:13-54: cost, Euclidean and Riemannian gradient and the Hessian-vector product of
    the three fixtures against the reference's known answers at its own tolerance (1e-6), through CORA::Problem on the GPU."""
    r = _run("test_optimizer_helpers", tmp_path)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "3 test cases, 0 failed, 12 assertions" in r.stdout


@pytest.mark.gpu
def test_reference_solve_tests_pass(tmp_path):
    """tests/test_cora.cpp:42-86: parse, random start, solveCORA on the three fixtures (the reference asserts that
    nothing throws and the shapes are right)."""
    r = _run("test_cora", tmp_path)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "3 test cases, 0 failed" in r.stdout


@pytest.mark.gpu
def test_reference_certification_tests_pass(tmp_path):
    """tests/test_certification.cpp:45-125: fast_verification on I, I - xx', I - 2xx' (10 and 1 000 rows, both interfaces:
    a start block and a block size) -- certified / not certified with theta = -1 and x up to sign at 1e-6 --, Lambda = 0 and
    a certified certificate at the ground truth of the RA-SLAM fixture, S at the random point equal to S_rand.mm, not
    certified there with theta = x' S x."""
    r = _run("test_certification", tmp_path)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "3 test cases, 0 failed" in r.stdout


@pytest.mark.gpu
def test_reference_geometry_tests_pass(tmp_path):
    """tests/test_geometry.cpp:10-81: ObliqueManifold -- random sample, projection to the manifold and to the tangent space,
    retraction -- on one unit circle and on five unit spheres (the manifold classes run their geometry on the GPU)."""
    r = _run("test_geometry", tmp_path)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "2 test cases, 0 failed" in r.stdout


@pytest.mark.gpu
def test_reference_parse_pyfg_tests_pass_on_the_gpu_box(tmp_path):
    r = _run("test_parse_pyfg", tmp_path)
    assert r.returncode == 0 and "3 test cases, 0 failed" in r.stdout
