"""What the instrumentation of a bench step costs: config-3 default-path graph replays enqueued back to back, with and
without lambda.zero_() and a HIP-event pair per step (python gbd-pcg_amd/tools/step_overhead.py)."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/gbd-pcg_amd/", 1)[0])
from gbd_pcg_amd import binding, synth
n, N, B, iters = 14, 128, 1024, 25
s = binding.Solver(0)
g = synth.gen_torch_seeded(n, N, 0, B, "cuda", torch.float32)
S, gamma = g["S"], g["gamma"]
P = s.form_pinv(n, N, B, S, binding.PINV_STAIR)
lam = torch.zeros_like(gamma); it = torch.zeros(B, dtype=torch.int32, device="cuda"); fl = torch.zeros(B, dtype=torch.uint8, device="cuda")
gr = s.graph_solve(n, N, B, S, P, gamma, lam, None, None, 0.0, iters, it, fl)
def run(K, zero, events):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        if zero: lam.zero_()
        if events: evs[k][0].record()
        gr.launch()
        if events: evs[k][1].record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for _ in range(3): run(50, True, True)
for zero, events in ((True, True), (True, False), (False, False), (False, True)):
    print("zero_", zero, "events", events, "-> ms per step %.4f" % min(run(200, zero, events) for _ in range(3)))
