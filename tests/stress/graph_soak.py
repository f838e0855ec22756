#!/usr/bin/env python3
"""Soak of the host side: handles and graphs created and destroyed in a loop (solve, formation + solve, the whole KKT step; three
shapes), every graph replayed a few times, device memory in use compared before and after -- a leak of a graph, a stream or a
workspace shows as growth.  Then a long run of back-to-back replays of the headline solve graph (results checked at the end).
     python tests/stress/graph_soak.py [rounds=60]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gbd_pcg_amd import binding, synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)


def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return total - free


shapes = [(14, 128, 64, torch.float32), (36, 64, 1, torch.float64), (8, 50, 300, torch.float32)]
data = []
for n, N, B, dt in shapes:
    g = synth.gen_torch(n, N, B, dev, dt, seed=5)
    data.append((n, N, B, g["S"], g["gamma"]))
kkt = synth.kkt_torch(14, 7, 32, 40, dev, torch.float32, seed=1)
base = None
for r in range(rounds):
    s = binding.Solver(0)
    for n, N, B, S, gamma in data:
        P = torch.empty_like(S)
        lam, rr, pp = torch.zeros_like(gamma), torch.empty_like(gamma), torch.empty_like(gamma)
        it = torch.zeros(B, dtype=torch.int32, device=dev)
        fl = torch.zeros(B, dtype=torch.uint8, device=dev)
        s.form_pinv(n, N, B, S, Pinv=P)
        g1 = s.graph_solve(n, N, B, S, P, gamma, lam, rr, pp, 1e-6, 50, it, fl)
        g2 = s.graph_form_pinv_solve(n, N, B, S, P, gamma, lam, rr, pp, 1e-6, 50, it, fl)
        for _ in range(3):
            lam.zero_()
            g1.launch()
            lam.zero_()
            g2.launch()
        torch.cuda.synchronize()
        assert int(fl.sum()) == 0 and int(it.min()) >= 1 and bool(torch.isfinite(lam).all())
        g1.close()
        g2.close()
    G, C, gg, cc = kkt
    S, gamma, Ginv = s.form_schur(14, 7, 32, 40, G, C, gg, cc)
    P, lam, z = torch.empty_like(S), torch.zeros_like(gamma), torch.empty_like(gg)
    it = torch.zeros(40, dtype=torch.int32, device=dev)
    fl = torch.zeros(40, dtype=torch.uint8, device=dev)
    g3 = s.graph_kkt_step(14, 7, 32, 40, G, C, gg, cc, S, gamma, Ginv, P, lam, None, None, 1e-6, 100, it, fl, z)
    for _ in range(3):
        g3.launch()
    torch.cuda.synchronize()
    assert int(fl.sum()) == 0 and bool(torch.isfinite(z).all())
    g3.close()
    s.close()
    del P, lam, z, S, gamma, Ginv
    if r == 4:
        torch.cuda.empty_cache()
        base = used()
torch.cuda.empty_cache()
grown = used() - base
print(f"{rounds} rounds of create / replay / destroy: device memory in use grew by {grown / 2**20:.1f} MiB since round 5")
assert grown < 64 * 2**20, "something is not given back"

# long run of replays of the headline graph
n, N, B = 14, 128, 1024
g = synth.gen_torch(n, N, B, dev, torch.float32, seed=9)
s = binding.Solver(0)
P = s.form_pinv(n, N, B, g["S"])
lam, rr, pp = torch.zeros_like(g["gamma"]), torch.empty_like(g["gamma"]), torch.empty_like(g["gamma"])
it = torch.zeros(B, dtype=torch.int32, device=dev)
fl = torch.zeros(B, dtype=torch.uint8, device=dev)
gr = s.graph_solve(n, N, B, g["S"], P, g["gamma"], lam, rr, pp, 1e-6, 50, it, fl)
lam.zero_()
gr.launch()
torch.cuda.synchronize()
want_it, want_lam = it.clone(), lam.clone()
for k in range(3000):
    lam.zero_()
    gr.launch()
torch.cuda.synchronize()
assert torch.equal(it, want_it) and torch.equal(lam, want_lam), "a replay differs from the first one"
print("3000 back-to-back replays of the config-3 solve graph: bit-identical with the first")
gr.close()
s.close()
