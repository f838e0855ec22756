"""Synthetic block-tridiagonal Schur systems  S lambda = gamma  (Gen(n,N,seed,a)).

    S = G W G^T  (symmetrised),   G = block lower-bidiagonal: I on the diagonal,
                                      -a*Q_k on the sub-diagonal (Q_k orthogonal),
                                  W = blkdiag(I + M_k M_k^T),  M_k ~ N(0, 1/n)
    gamma ~ N(0,1)^{nN},  lambda_0 = 0

which is the MPC Schur-complement structure (block-bidiagonal dynamics times a
block-diagonal cost inverse) and is exactly block-tridiagonal SPD.  a = 0.5 gives
kappa(S) ~ 27 and 9-10 PCG iterations at tol 1e-6 with the symmetric-stair
preconditioner.

Storage is the reference's compressed block-tridiagonal layout
(/root/reference/include/pcg.cuh:104-110, utils.cuh:80): per knot k three
column-major n x n blocks [L_k | D_k | R_k]; batches are problem-major.

Two generators with the same math:
  * gen_numpy : canonical, reproducible anywhere (own counter-based RNG, no
                dependence on numpy's or torch's generator streams); used by the
                parity tests and the golden fixtures.
  * gen_torch : same construction with torch ops on the GPU (torch RNG; Q_k from Householder
                products instead of QR), for bench-sized batches where only the
                shape/conditioning matters.
"""
from __future__ import annotations

import numpy as np

_U64 = np.uint64


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + _U64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = x
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def normals(seed: int, stream: int, count: int) -> np.ndarray:
    """count N(0,1) doubles from a counter-based generator keyed by (seed, stream).

    u64 words come from splitmix64 applied to (key + counter); two 53-bit
    uniforms per normal via Box-Muller.  Deterministic on any platform.
    """
    with np.errstate(over="ignore"):
        key = _splitmix64(np.array([seed], dtype=np.uint64) * _U64(0x632BE59BD9B4E019)
                          + _U64(stream) * _U64(0xD1342543DE82EF95))[0]
        ctr = np.arange(2 * count, dtype=np.uint64)
        w = _splitmix64(key + ctr * _U64(0x9E3779B97F4A7C15))
    u = ((w >> _U64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)  # (0,1)
    u1, u2 = u[:count], u[count:]
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def pack_bt(L: np.ndarray, D: np.ndarray, R: np.ndarray) -> np.ndarray:
    """[..., N, n, n] row/col-indexed blocks -> flat [..., N*3*n*n] column-major [L|D|R]."""
    blk = np.stack([L, D, R], axis=-3)                 # [..., N, 3, r, c]
    blk = np.swapaxes(blk, -1, -2)                     # [..., N, 3, c, r]  (column-major blocks)
    return np.ascontiguousarray(blk).reshape(*blk.shape[:-4], -1)


def unpack_bt(n: int, N: int, M: np.ndarray):
    blk = np.asarray(M).reshape(*np.shape(M)[:-1], N, 3, n, n)
    blk = np.swapaxes(blk, -1, -2)
    return blk[..., 0, :, :], blk[..., 1, :, :], blk[..., 2, :, :]


def stair_pinv_blocks(L, D, R, xp=np):
    """Symmetric-stair preconditioner blocks from the blocks of S ([..., N, n, n] each):
    diag D_k^-1, right -D_k^-1 R_k D_{k+1}^-1, left -D_k^-1 L_k D_{k-1}^-1.
    xp = numpy or torch (both spell linalg.inv / zeros_like / roll the same way)."""
    Dinv = xp.linalg.inv(D)
    Dn = xp.roll(Dinv, -1, -3)      # D_{k+1}^-1 (wraps at k = N-1, where R_{N-1} = 0 is unused)
    Dp = xp.roll(Dinv, 1, -3)       # D_{k-1}^-1 (wraps at k = 0, where L_0 = 0 is unused)
    return -(Dinv @ L @ Dp), Dinv, -(Dinv @ R @ Dn)


def gen_numpy(n: int, N: int, seed: int = 1234, a: float = 0.5, batch: int = 1,
              dtype=np.float64, pinv: str = "stair"):
    """Batch of Gen(n,N,seed+i,a), i < batch.  Returns dict of flat arrays in `dtype`
    (generated in fp64, converted last): S, Pinv [batch, 3n^2N], gamma [batch, nN]."""
    S_all, P_all, g_all = [], [], []
    eye = np.eye(n)
    for i in range(batch):
        s = seed + i
        A = normals(s, 0, (N - 1) * n * n).reshape(N - 1, n, n) if N > 1 else np.zeros((0, n, n))
        Mk = normals(s, 1, N * n * n).reshape(N, n, n) / np.sqrt(n)
        gamma = normals(s, 2, N * n)
        if N > 1:
            Q, Rq = np.linalg.qr(A)
            sg = np.sign(np.diagonal(Rq, axis1=-2, axis2=-1))
            sg[sg == 0] = 1.0
            Q = Q * sg[:, None, :]                      # unique QR: diag(R) > 0
        else:
            Q = A
        W = eye + Mk @ np.swapaxes(Mk, -1, -2)
        D = W.copy()
        L = np.zeros((N, n, n))
        if N > 1:
            QW = Q @ W[:-1]                              # Q_k W_{k-1}, k = 1..N-1
            D[1:] += (a * a) * (QW @ np.swapaxes(Q, -1, -2))
            L[1:] = -a * QW
        D = 0.5 * (D + np.swapaxes(D, -1, -2))
        R = np.zeros((N, n, n))
        R[:-1] = np.swapaxes(L[1:], -1, -2)
        S_all.append(pack_bt(L, D, R))
        if pinv == "stair":
            P_all.append(pack_bt(*stair_pinv_blocks(L, D, R)))
        elif pinv == "jacobi":
            z = np.zeros_like(D)
            P_all.append(pack_bt(z, np.linalg.inv(D), z))
        elif pinv == "identity":
            z = np.zeros_like(D)
            P_all.append(pack_bt(z, np.broadcast_to(eye, D.shape).copy(), z))
        else:
            raise ValueError(pinv)
        g_all.append(gamma)
    return dict(n=n, N=N, batch=batch,
                S=np.stack(S_all).astype(dtype), Pinv=np.stack(P_all).astype(dtype),
                gamma=np.stack(g_all).astype(dtype))


def _torch_splitmix64(x):
    """splitmix64 finaliser on int64 tensors (two's-complement wrap = uint64 arithmetic; logical shifts
    spelled as arithmetic shift + mask)."""
    def s64(c):  # uint64 constant -> the int64 with the same bits
        return c - (1 << 64) if c >= (1 << 63) else c

    def shr(v, k):
        return (v >> k) & ((1 << (64 - k)) - 1)
    x = x + s64(0x9E3779B97F4A7C15)
    z = (x ^ shr(x, 30)) * s64(0xBF58476D1CE4E5B9)
    z = (z ^ shr(z, 27)) * s64(0x94D049BB133111EB)
    return z ^ shr(z, 31)


def normals_torch(seeds, stream: int, count: int, device):
    """normals(seed, stream, count) for every seed of `seeds` at once, as a [len(seeds), count] fp64 tensor on
    `device`: the same counter-based words as the numpy generator (identical 53-bit uniforms; log / cos / sqrt
    of the device's libm, so the normals agree to rounding, not to the bit)."""
    import math
    import torch
    with np.errstate(over="ignore"):
        keys = _splitmix64(np.asarray(seeds, dtype=np.uint64) * _U64(0x632BE59BD9B4E019)
                           + _U64(stream) * _U64(0xD1342543DE82EF95))
    key = torch.from_numpy(keys.view(np.int64).copy()).to(device)[:, None]
    ctr = torch.arange(2 * count, dtype=torch.int64, device=device)[None, :]
    w = _torch_splitmix64(key + ctr * (0x9E3779B97F4A7C15 - (1 << 64)))
    u = (((w >> 11) & ((1 << 53) - 1)).to(torch.float64) + 0.5) * (1.0 / 9007199254740992.0)
    u1, u2 = u[:, :count], u[:, count:]
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * math.pi) * u2)


def _qr_q_positive(A):
    """Q factor of the unique QR (diag(R) > 0) of every n x n matrix of A [..., n, n]: n batched Householder
    reflections (elementwise + bmm work only; torch.linalg.qr on the GPU launches several kernels per matrix)."""
    import torch
    n = A.shape[-1]
    R = A.clone()
    Q = torch.eye(n, dtype=A.dtype, device=A.device).expand(A.shape).clone()
    for j in range(n):
        x = R[..., j:, j]
        nrm = x.norm(dim=-1, keepdim=True)
        sgn = torch.where(x[..., :1] < 0, -torch.ones_like(nrm), torch.ones_like(nrm))
        v = x.clone()
        v[..., :1] += sgn * nrm                     # v = x + sign(x0) |x| e0
        v = v / v.norm(dim=-1, keepdim=True).clamp_min(1e-300)
        v = v.unsqueeze(-1)                         # [..., n-j, 1]
        R[..., j:, :] -= 2.0 * v @ (v.transpose(-1, -2) @ R[..., j:, :])
        Q[..., :, j:] -= 2.0 * (Q[..., :, j:] @ v) @ v.transpose(-1, -2)
    d = torch.diagonal(R, dim1=-2, dim2=-1)
    sg = torch.where(d < 0, -torch.ones_like(d), torch.ones_like(d))
    return Q * sg.unsqueeze(-2)


def gen_torch_seeded(n: int, N: int, lo: int, hi: int, device, dtype, seed: int = 1234, a: float = 0.5,
                     chunk: int = 256):
    """Problems lo .. hi-1 of the batch Gen(n, N, seed + i, a) -- SURVEY.md section 8d: problem i depends on
    seed + i only, whatever the rank or world size that generates it -- built on `device` with torch ops from
    the same counter-based random words as gen_numpy (same construction; equal to gen_numpy to fp64 rounding).
    Returns flat S [hi-lo, 3n^2N] and gamma [hi-lo, nN] in `dtype`, and Pinv (host-formula symmetric stair)."""
    import torch
    cnt = hi - lo
    S = torch.empty((cnt, N * 3 * n * n), device=device, dtype=dtype)
    P = torch.empty_like(S)
    gamma = torch.empty((cnt, N * n), device=device, dtype=dtype)
    eye = torch.eye(n, device=device, dtype=torch.float64)

    def pack(L, D, R):
        blk = torch.stack([L, D, R], dim=-3).transpose(-1, -2)
        return blk.reshape(blk.shape[0], -1)

    for c0 in range(0, cnt, chunk):
        b = min(chunk, cnt - c0)
        seeds = [seed + lo + c0 + i for i in range(b)]
        Mk = normals_torch(seeds, 1, N * n * n, device).reshape(b, N, n, n) / np.sqrt(n)
        gam = normals_torch(seeds, 2, N * n, device)
        W = eye + Mk @ Mk.transpose(-1, -2)
        D = W.clone()
        L = torch.zeros_like(W)
        if N > 1:
            A = normals_torch(seeds, 0, (N - 1) * n * n, device).reshape(b, N - 1, n, n)
            Q = _qr_q_positive(A)
            QW = Q @ W[:, :-1]
            D[:, 1:] += (a * a) * (QW @ Q.transpose(-1, -2))
            L[:, 1:] = -a * QW
        D = 0.5 * (D + D.transpose(-1, -2))
        R = torch.zeros_like(W)
        if N > 1:
            R[:, :-1] = L[:, 1:].transpose(-1, -2)
        S[c0:c0 + b] = pack(L, D, R).to(dtype)
        P[c0:c0 + b] = pack(*stair_pinv_blocks(L, D, R, xp=torch)).to(dtype)
        gamma[c0:c0 + b] = gam.to(dtype)
    return dict(n=n, N=N, batch=cnt, S=S, Pinv=P, gamma=gamma)


def gen_torch(n: int, N: int, batch: int, device, dtype, seed: int = 1234, a: float = 0.5,
              chunk: int = 256):
    """Same construction on `device` with torch ops (torch RNG).  Returns flat tensors
    S, Pinv [batch, 3n^2N] and gamma [batch, nN] in `dtype` (built in fp64, chunked)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    S = torch.empty((batch, N * 3 * n * n), device=device, dtype=dtype)
    P = torch.empty_like(S)
    gamma = torch.empty((batch, N * n), device=device, dtype=dtype)
    eye = torch.eye(n, device=device, dtype=torch.float64)

    def pack(L, D, R):
        blk = torch.stack([L, D, R], dim=-3).transpose(-1, -2)
        return blk.reshape(blk.shape[0], -1)

    for lo in range(0, batch, chunk):
        b = min(chunk, batch - lo)
        # Q_k: product of n Householder reflections of random vectors -- exactly orthogonal and pure
        # elementwise/bmm work (torch.linalg.qr on the GPU launches several kernels per matrix)
        Vh = torch.randn((b, N - 1, n, n), generator=g, device=device, dtype=torch.float64)
        Mk = torch.randn((b, N, n, n), generator=g, device=device, dtype=torch.float64) / n ** 0.5
        gam = torch.randn((b, N * n), generator=g, device=device, dtype=torch.float64)
        Q = eye.expand(b, N - 1, n, n).clone()
        for j in range(n):
            v = Vh[..., j]
            v = v / v.norm(dim=-1, keepdim=True)
            Q = Q - 2.0 * (Q @ v.unsqueeze(-1)) * v.unsqueeze(-2)
        W = eye + Mk @ Mk.transpose(-1, -2)
        D = W.clone()
        L = torch.zeros_like(W)
        QW = Q @ W[:, :-1]
        D[:, 1:] += (a * a) * (QW @ Q.transpose(-1, -2))
        L[:, 1:] = -a * QW
        D = 0.5 * (D + D.transpose(-1, -2))
        R = torch.zeros_like(W)
        R[:, :-1] = L[:, 1:].transpose(-1, -2)
        S[lo:lo + b] = pack(L, D, R).to(dtype)
        P[lo:lo + b] = pack(*stair_pinv_blocks(L, D, R, xp=torch)).to(dtype)
        gamma[lo:lo + b] = gam.to(dtype)
    return dict(n=n, N=N, batch=batch, S=S, Pinv=P, gamma=gamma)


def kkt_torch(nx: int, nu: int, N: int, batch: int, device, dtype, seed: int = 0):
    """Synthetic KKT blocks of `batch` linearised MPC problems in the packed layouts gbdpcg_form_schur_* takes
    (include/gbdpcg.h), drawn on the device: cost blocks M M' + I (symmetric positive definite), dynamics
    A = I + noise, B, gradients and constraint residuals normal.  Returns flat G, C, g, c.  Measurement input (bench.py,
    tools/schur_run.py); the parity tests draw theirs with oracle/schur_oracle.gen."""
    import torch
    gen = torch.Generator(device=device).manual_seed(seed)
    sg, sc, sv = nx * nx + nu * nu, nx * nx + nx * nu, nx + nu

    def spd(m, count):
        a = torch.randn(count, m, m, device=device, dtype=dtype, generator=gen) / m ** 0.5
        return a @ a.transpose(1, 2) + torch.eye(m, device=device, dtype=dtype)

    G = torch.zeros(batch, N, sg, device=device, dtype=dtype)
    G[:, :, :nx * nx] = spd(nx, batch * N).reshape(batch, N, -1)
    G[:, :, nx * nx:] = spd(nu, batch * N).reshape(batch, N, -1)
    G = G.reshape(batch, -1)[:, :sg * N - nu * nu].contiguous()
    C = torch.zeros(batch, max(N - 1, 0), sc, device=device, dtype=dtype)
    if N > 1:
        A = torch.eye(nx, device=device, dtype=dtype) + 0.3 * torch.randn(batch * (N - 1), nx, nx, device=device, dtype=dtype,
                                                                         generator=gen) / nx ** 0.5
        C[:, :, :nx * nx] = A.transpose(1, 2).reshape(batch, N - 1, -1)
        C[:, :, nx * nx:] = torch.randn(batch, N - 1, nx * nu, device=device, dtype=dtype, generator=gen) / nx ** 0.5
    g = torch.randn(batch, sv * N - nu, device=device, dtype=dtype, generator=gen)
    c = 0.1 * torch.randn(batch, nx * N, device=device, dtype=dtype, generator=gen)
    return G.reshape(-1), C.reshape(-1), g.reshape(-1), c.reshape(-1)
