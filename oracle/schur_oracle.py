"""oracle/schur_oracle.py -- TEST INFRASTRUCTURE ONLY.

The two MPCGPU steps either side of the PCG solve, restated in numpy: forming the Schur system S, gamma from the
KKT blocks of one linearised MPC problem, and recovering the primal step from lambda.  PARITY UNPINNED: the
reference tree holds no code for these steps (/root/reference/README.md:2-11 only states the system that comes out
of them, README.md:66-77 cites the paper that describes them), so there is nothing to be identical to; what pins
this file is algebra -- `dense_kkt_solve` solves the full KKT system in fp64 with numpy.linalg and the tests require
the block formulas, the PCG solution of S lambda = gamma and the recovered step to agree with it.

The problem (N knots, nx states, nu controls; z = (x_0, u_0, x_1, u_1, ..., x_{N-1}), N(nx+nu)-nu entries):

    minimise    sum_k  1/2 x_k' Q_k x_k + q_k' x_k  +  sum_{k<N-1} 1/2 u_k' R_k u_k + r_k' u_k
    subject to  x_0 = c_0,    x_{k+1} - A_k x_k - B_k u_k = c_{k+1}

i.e. 1/2 z'Gz + g'z subject to Cz = c with G = diag(Q_0, R_0, Q_1, ..., Q_{N-1}).  Stationarity Gz + g + C'lambda = 0 gives

    S lambda = gamma,   S = C G^-1 C'  (positive definite, block-tridiagonal),   gamma = -(c + C G^-1 g)
    z = -G^-1 (g + C' lambda)

Blocks, with theta_k / phi_k as in the paper (Adabag et al., MPCGPU, arXiv 2309.08079, section IV):

    D_0 = Q_0^-1                                                     gamma_0 = -(c_0 + Q_0^-1 q_0)
    D_k = A_j Q_j^-1 A_j' + B_j R_j^-1 B_j' + Q_k^-1     (j = k-1)    gamma_k = -(c_k + Q_k^-1 q_k - A_j Q_j^-1 q_j - B_j R_j^-1 r_j)
    L_k = -A_j Q_j^-1,   R_k = L_{k+1}' = -Q_k^-1 A_k'

Packed layouts (what gbdpcg_form_schur_* takes; every block column-major, as everywhere in this library):
    G  [Q_0 R_0 Q_1 R_1 ... Q_{N-1}]      (nx^2+nu^2) N - nu^2
    C  [A_0 B_0 A_1 B_1 ... B_{N-2}]      (nx^2+nx nu)(N-1)
    g  [q_0 r_0 q_1 r_1 ... q_{N-1}]      (nx+nu) N - nu             (z has the same layout)
    c  [c_0 ... c_{N-1}]                  nx N
"""
from __future__ import annotations

import numpy as np


def sizes(nx, nu, N):
    return {"G": (nx * nx + nu * nu) * N - nu * nu, "C": (nx * nx + nx * nu) * (N - 1), "g": (nx + nu) * N - nu, "c": nx * N,
            "S": 3 * nx * nx * N, "gamma": nx * N, "z": (nx + nu) * N - nu, "Ginv": (nx * nx + nu * nu) * N - nu * nu}


def gen(nx, nu, N, seed=0, batch=1, dtype=np.float64, cond=30.0):
    """Random well-posed problems: SPD cost blocks with eigenvalues in [1, cond], dynamics of spectral radius ~1."""
    rng = np.random.default_rng(seed)
    sz = sizes(nx, nu, N)
    out = {k: np.zeros((batch, sz[k]), dtype=np.float64) for k in ("G", "C", "g", "c")}

    def spd(m):
        q, _ = np.linalg.qr(rng.standard_normal((m, m)))
        return (q * np.exp(rng.uniform(0.0, np.log(cond), m))) @ q.T

    for b in range(batch):
        G, C = [], []
        for k in range(N):
            G.append(spd(nx).T.reshape(-1))
            if k < N - 1:
                G.append(spd(nu).T.reshape(-1))
                A = np.eye(nx) + 0.3 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
                B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
                C.append(A.T.reshape(-1))
                C.append(B.T.reshape(-1))
        out["G"][b] = np.concatenate(G)
        out["C"][b] = np.concatenate(C) if C else np.zeros(0)
        out["g"][b] = rng.standard_normal(sz["g"])
        out["c"][b] = 0.1 * rng.standard_normal(sz["c"])
    return {k: v.astype(dtype) for k, v in out.items()}


def unpack(nx, nu, N, G, C, g, c):
    """One problem's packed arrays -> lists of (nx,nx) / (nu,nu) / ... matrices (row-major numpy views of the math)."""
    G = np.asarray(G, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    sg, sc, sv = nx * nx + nu * nu, nx * nx + nx * nu, nx + nu
    Q = [G[k * sg:k * sg + nx * nx].reshape(nx, nx).T for k in range(N)]
    R = [G[k * sg + nx * nx:(k + 1) * sg].reshape(nu, nu).T for k in range(N - 1)]
    A = [C[k * sc:k * sc + nx * nx].reshape(nx, nx).T for k in range(N - 1)]
    B = [C[k * sc + nx * nx:(k + 1) * sc].reshape(nu, nx).T for k in range(N - 1)]
    q = [g[k * sv:k * sv + nx] for k in range(N)]
    r = [g[k * sv + nx:(k + 1) * sv] for k in range(N - 1)]
    cc = [c[k * nx:(k + 1) * nx] for k in range(N)]
    return Q, R, A, B, q, r, cc


def form_schur(nx, nu, N, G, C, g, c):
    """Block formulas of the module docstring, fp64.  Returns S ([L|D|R] column-major, 3 nx^2 N), gamma (nx N),
    Ginv (the layout of G with every block inverted)."""
    Q, R, A, B, q, r, cc = unpack(nx, nu, N, G, C, g, c)
    Qi = [np.linalg.inv(m) for m in Q]
    Ri = [np.linalg.inv(m) for m in R]
    S = np.zeros((N, 3, nx * nx))
    gamma = np.zeros((N, nx))
    for k in range(N):
        D = Qi[k].copy()
        v = cc[k] + Qi[k] @ q[k]
        if k > 0:
            j = k - 1
            W = A[j] @ Qi[j]
            V = B[j] @ Ri[j]
            D += W @ A[j].T + V @ B[j].T
            v -= W @ q[j] + V @ r[j]
            S[k, 0] = (-W).T.reshape(-1)
        if k < N - 1:
            S[k, 2] = (-(Qi[k] @ A[k].T)).T.reshape(-1)
        S[k, 1] = D.T.reshape(-1)
        gamma[k] = -v
    Ginv = []
    for k in range(N):
        Ginv.append(Qi[k].T.reshape(-1))
        if k < N - 1:
            Ginv.append(Ri[k].T.reshape(-1))
    return S.reshape(-1), gamma.reshape(-1), np.concatenate(Ginv)


def recover_primal(nx, nu, N, G, C, g, lam):
    """z = -G^-1 (g + C' lambda), block by block, fp64."""
    Q, R, A, B, q, r, _ = unpack(nx, nu, N, G, C, g, np.zeros(nx * N))
    lam = np.asarray(lam, dtype=np.float64).reshape(N, nx)
    z = []
    for k in range(N):
        t = q[k] + lam[k]
        if k < N - 1:
            t = t - A[k].T @ lam[k + 1]
        z.append(-np.linalg.solve(Q[k], t))
        if k < N - 1:
            z.append(-np.linalg.solve(R[k], r[k] - B[k].T @ lam[k + 1]))
    return np.concatenate(z)


def dense_kkt(nx, nu, N, G, C, g, c):
    """The full matrices: Gd (nz,nz), Cd (nx N, nz), g, c."""
    Q, R, A, B, q, r, cc = unpack(nx, nu, N, G, C, g, c)
    sv = nx + nu
    nz = sv * N - nu
    Gd = np.zeros((nz, nz))
    Cd = np.zeros((nx * N, nz))
    for k in range(N):
        o = k * sv
        Gd[o:o + nx, o:o + nx] = Q[k]
        Cd[k * nx:(k + 1) * nx, o:o + nx] = np.eye(nx)
        if k < N - 1:
            Gd[o + nx:o + sv, o + nx:o + sv] = R[k]
            Cd[(k + 1) * nx:(k + 2) * nx, o:o + nx] = -A[k]
            Cd[(k + 1) * nx:(k + 2) * nx, o + nx:o + sv] = -B[k]
    return Gd, Cd, np.asarray(g, dtype=np.float64), np.asarray(c, dtype=np.float64)


def dense_kkt_solve(nx, nu, N, G, C, g, c):
    """Solve [G C'; C 0] [z; lambda] = [-g; c] with numpy.linalg (fp64).  Returns z, lambda."""
    Gd, Cd, gv, cv = dense_kkt(nx, nu, N, G, C, g, c)
    nz, nl = Gd.shape[0], Cd.shape[0]
    K = np.zeros((nz + nl, nz + nl))
    K[:nz, :nz] = Gd
    K[:nz, nz:] = Cd.T
    K[nz:, :nz] = Cd
    sol = np.linalg.solve(K, np.concatenate([-gv, cv]))
    return sol[:nz], sol[nz:]
