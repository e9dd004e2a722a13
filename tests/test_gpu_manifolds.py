"""The reference's manifold classes (StiefelProduct, ObliqueManifold; points stored p x kn / r x n) backed by the
device kernels: the assertions of tests/test_geometry.cpp:11-88, plus the Stiefel product against numpy."""
import numpy as np
import pytest

from cora_amd import host

pytestmark = pytest.mark.gpu


def test_unit_sphere_functions():          # tests/test_geometry.cpp:11-45
    r, n = 2, 1
    Y = host.manifold_op("oblique", "random_sample", r, n, seed=1)
    assert abs(np.linalg.norm(Y) - 1) < 1e-12
    assert abs(host.manifold_op("oblique", "innerProduct", r, n, A=Y, B=Y) - 1) < 1e-12
    V = np.random.default_rng(0).uniform(-1, 1, (r, n))
    Vt = host.manifold_op("oblique", "projectToTangentSpace", r, n, A=Y, B=V)
    assert np.abs(Vt).max() > 0
    assert abs(host.manifold_op("oblique", "innerProduct", r, n, A=Vt, B=Y)) < 1e-6
    Yr = host.manifold_op("oblique", "retract", r, n, A=Y, B=Vt)
    assert abs(np.linalg.norm(Yr) - 1) < 1e-12 and not np.allclose(Yr, Y)


def test_oblique_manifold_functions():     # tests/test_geometry.cpp:47-88
    r, n = 3, 5
    rng = np.random.default_rng(2)
    Y = host.manifold_op("oblique", "projectToManifold", r, n, A=rng.uniform(-1, 1, (r, n)))
    assert np.abs(np.linalg.norm(Y, axis=0) - 1).max() < 1e-12
    V = host.manifold_op("oblique", "projectToTangentSpace", r, n, A=Y, B=rng.uniform(-1, 1, (r, n)))
    assert np.abs(host.manifold_op("oblique", "projectToTangentSpace", r, n, A=Y, B=Y)).max() < 1e-14
    assert np.linalg.norm(V) > 1e-6
    assert abs(host.manifold_op("oblique", "innerProduct", r, n, A=V, B=Y)) < 1e-6
    Yr = host.manifold_op("oblique", "retract", r, n, A=Y, B=V)
    assert np.abs(np.linalg.norm(Yr, axis=0) - 1).max() < 1e-12 and not np.allclose(Yr, Y)


@pytest.mark.parametrize("k,p,n", [(2, 2, 7), (3, 5, 40), (3, 3, 1)])
def test_stiefel_product(k, p, n):
    rng = np.random.default_rng(k * 100 + p)
    A = rng.standard_normal((p, k * n))
    Y = host.manifold_op("stiefel", "projectToManifold", p, n, k=k, A=A)
    for i in range(n):
        B = Y[:, i * k:(i + 1) * k]
        assert np.abs(B.T @ B - np.eye(k)).max() < 1e-12                       # orthonormal k-frame
        U, _, Vt = np.linalg.svd(A[:, i * k:(i + 1) * k], full_matrices=False)
        assert np.abs(B - U @ Vt).max() < 1e-10                                # the polar factor (StiefelProduct.cpp:29-33)
    V = rng.standard_normal((p, k * n))
    T = host.manifold_op("stiefel", "projectToTangentSpace", p, n, k=k, A=Y, B=V)
    S = host.manifold_op("stiefel", "SymBlockDiagProduct", p, n, k=k, A=Y, B=np.hstack([Y, V]))
    assert np.abs(T - (V - S)).max() < 1e-12                                   # StiefelProduct.h:79-81
    for i in range(n):
        Yi, Ti = Y[:, i * k:(i + 1) * k], T[:, i * k:(i + 1) * k]
        G = Yi.T @ Ti
        assert np.abs(G + G.T).max() < 1e-12                                   # tangent: Y^T T skew-symmetric
    # general three-operand product against numpy
    Bm, Cm = rng.standard_normal((p, k * n)), rng.standard_normal((p, k * n))
    got = host.manifold_op("stiefel", "SymBlockDiagProduct", p, n, k=k, A=A, B=np.hstack([Bm, Cm]))
    ref = np.zeros_like(A)
    for i in range(n):
        s = slice(i * k, (i + 1) * k)
        M = Bm[:, s].T @ Cm[:, s]
        ref[:, s] = A[:, s] @ (0.5 * (M + M.T))
    assert np.abs(got - ref).max() < 1e-12
    R = host.manifold_op("stiefel", "random_sample", p, n, k=k, seed=5)
    assert np.abs(R[:, :k].T @ R[:, :k] - np.eye(k)).max() < 1e-12
