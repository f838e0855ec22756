"""Where does a cluster-path solve first differ from the CPU oracle?  (development aid; the oracle is the checker)
   python gbd-pcg_amd/tools/cluster_diag.py [N]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/gbd-pcg_amd/", 1)[0])
from gbd_pcg_amd import binding, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, B = 14, 1
d = synth.gen_numpy(n, N, seed=500 + N, batch=B, dtype=np.float32)
s = binding.Solver(0)
s.set_symmetric(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dS, dP, dg = t(d["S"]), t(d["Pinv"]), t(d["gamma"])
for mi in (0, 1, 2, 3):
    lam = torch.zeros_like(dg)
    r, p = torch.zeros_like(dg), torch.zeros_like(dg)
    it, fl = s.solve(n, N, B, dS, dP, dg, lam, r, p, tol=0.0, max_iter=mi)
    torch.cuda.synchronize()
    ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], tol=0.0, max_iter=mi)
    for key, got in (("lambda_", lam), ("r", r), ("p", p)):
        g = got.cpu().numpy().reshape(N, n)
        o = np.asarray(ob[key]).reshape(N, n)
        err = np.abs(g - o).max(axis=1)
        worst = np.argsort(err)[-4:][::-1]
        print(f"max_iter {mi} {key:8s} max err {err.max():.3e} (scale {np.abs(o).max():.3e}) worst knots {worst.tolist()} errs {[float('%.2e' % err[k]) for k in worst]}")

# prologue with a non-zero initial guess: r = gamma - S lambda exercises the lambda halos (read from the input vector)
lam0 = (synth.normals(7, 0, n * N) * 0.1).astype(np.float32)
lam = t(lam0.reshape(1, -1).copy())
r, p = torch.zeros_like(dg), torch.zeros_like(dg)
s.solve(n, N, B, dS, dP, dg, lam, r, p, tol=0.0, max_iter=0)
torch.cuda.synchronize()
ob = orc.pcg_batch(n, N, B, d["S"], d["Pinv"], d["gamma"], lambda0=lam0.reshape(1, -1), tol=0.0, max_iter=0)
rg = r.cpu().numpy().reshape(N, n); ro = np.asarray(ob["r"]).reshape(N, n)
pg = p.cpu().numpy().reshape(N, n); po = np.asarray(ob["p"]).reshape(N, n)
print("lambda0 != 0: r err at knots 62..65", np.abs(rg - ro).max(axis=1)[62:66], " p err", np.abs(pg - po).max(axis=1)[62:66])
# what would p_63 be if the halo r_64 were zero / were r_62 / were gamma_64?
Pm = d["Pinv"].reshape(N, 3, n, n).astype(np.float64)   # [k][block][col][row]
blk = lambda k, b: Pm[k, b].T                            # n x n, row-major
r64 = ro[64].astype(np.float64)
for name, halo in (("zero", 0 * r64), ("r_62", ro[62].astype(np.float64)), ("r_63", ro[63].astype(np.float64)), ("r_65", ro[65].astype(np.float64))):
    cand = blk(63, 0) @ ro[62] + blk(63, 1) @ ro[63] + blk(63, 2) @ halo
    print(f"  p_63 with halo = {name}: |gpu - cand| = {np.abs(pg[63] - cand).max():.3e}")
true = blk(63, 0) @ ro[62] + blk(63, 1) @ ro[63] + blk(63, 2) @ r64
print("  sanity: oracle p_63 vs host product", np.abs(po[63] - true).max())
hal = np.linalg.solve(blk(63, 2), pg[63].astype(np.float64) - blk(63, 0) @ ro[62] - blk(63, 1) @ ro[63])
print("  halo that explains the GPU's p_63:", np.round(hal, 4))
print("  true r_64                        :", np.round(r64, 4))
best = min(range(N), key=lambda k: np.abs(ro[k] - hal).max())
print("  nearest knot of r:", best, "max diff", np.abs(ro[best] - hal).max())
