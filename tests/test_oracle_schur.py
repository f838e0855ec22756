"""oracle/schur_oracle.py (KKT blocks -> Schur system, lambda -> primal step) pinned by algebra: the reference
tree has no code or vectors for these steps (parity unpinned), so the block formulas are held against dense fp64
constructions -- S against C G^-1 C' assembled from the full matrices, and the whole chain against numpy.linalg.solve of
the KKT system."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import schur_oracle as so

SHAPES = [(14, 7, 8), (14, 7, 1), (2, 1, 5), (3, 3, 2), (5, 2, 9), (12, 4, 6), (4, 6, 3)]


@pytest.mark.parametrize("nx,nu,N", SHAPES)
def test_block_formulas_match_dense_kkt(nx, nu, N):
    d = so.gen(nx, nu, N, seed=nx + 10 * N)
    G, C, g, c = (d[k][0] for k in "GCgc")
    S, gamma, Ginv = so.form_schur(nx, nu, N, G, C, g, c)
    Gd, Cd, gv, cv = so.dense_kkt(nx, nu, N, G, C, g, c)
    Sd = Cd @ np.linalg.inv(Gd) @ Cd.T
    assert np.abs(orc.dense_from_bt(nx, N, S) - Sd).max() < 1e-12 * np.abs(Sd).max()
    assert np.abs(gamma + cv + Cd @ np.linalg.solve(Gd, gv)).max() < 1e-12 * max(1.0, np.abs(gamma).max())
    z, lam = so.dense_kkt_solve(nx, nu, N, G, C, g, c)
    assert np.abs(Sd @ lam - gamma).max() < 1e-10 * max(1.0, np.abs(gamma).max())
    assert np.abs(so.recover_primal(nx, nu, N, G, C, g, lam) - z).max() < 1e-10 * max(1.0, np.abs(z).max())
    # the step satisfies the constraints and G^-1 really is the inverse, block by block
    assert np.abs(Cd @ z - cv).max() < 1e-10
    Q, R, *_ = so.unpack(nx, nu, N, G, C, g, c)
    Qi, Ri, *_ = so.unpack(nx, nu, N, Ginv, C, g, c)
    assert all(np.abs(a @ b - np.eye(nx)).max() < 1e-12 for a, b in zip(Q, Qi))
    assert all(np.abs(a @ b - np.eye(nu)).max() < 1e-12 for a, b in zip(R, Ri))


def test_schur_system_is_symmetric_positive_definite_and_pcg_solves_it():
    nx, nu, N = 14, 7, 16
    d = so.gen(nx, nu, N, seed=5)
    G, C, g, c = (d[k][0] for k in "GCgc")
    S, gamma, _ = so.form_schur(nx, nu, N, G, C, g, c)
    Sd = orc.dense_from_bt(nx, N, S)
    assert np.abs(Sd - Sd.T).max() < 1e-13 * np.abs(Sd).max()
    assert np.linalg.eigvalsh(0.5 * (Sd + Sd.T)).min() > 0
    # block-Jacobi preconditioner, the reference's recurrence (oracle/pcg_oracle.c) on the formed system
    Pinv = np.zeros((N, 3, nx, nx))
    Sb = S.reshape(N, 3, nx, nx)
    for k in range(N):
        Pinv[k, 1] = np.linalg.inv(Sb[k, 1].T).T
    out = orc.pcg(nx, N, S, Pinv.reshape(-1), gamma, tol=1e-26, max_iter=500)  # the exit test is on |eta| = |r . Pinv r|
    z, lam = so.dense_kkt_solve(nx, nu, N, G, C, g, c)
    assert np.linalg.norm(out["lambda_"] - lam) < 1e-8 * np.linalg.norm(lam)
