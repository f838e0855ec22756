"""Back-to-back graph replays in every combination of stream / interleaved fills / events: per-replay time must not
depend on the pattern (it did when the symmetry flags were initialised by a hipGraph memset node)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from gbd_pcg_amd import binding, synth
n,N,B=14,128,1024
solver=binding.Solver(0)
g=synth.gen_torch(n,N,B,"cuda",torch.float32,seed=1234)
S,gamma=g["S"],g["gamma"]
P=solver.form_pinv(n,N,B,S,binding.PINV_STAIR)
lam=torch.zeros_like(gamma); it=torch.zeros(B,dtype=torch.int32,device="cuda"); fl=torch.zeros(B,dtype=torch.uint8,device="cuda")
def run(tag, mode, stream, zero, events):
    solver.set_symmetric(mode)
    gr=solver.graph_solve(n,N,B,S,P,gamma,lam,None,None,0.0,25,it,fl)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        st = torch.cuda.current_stream()
        for _ in range(5):
            lam.zero_(); gr.launch(st)
        torch.cuda.synchronize()
        evs=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        t0=torch.cuda.Event(enable_timing=True); t1=torch.cuda.Event(enable_timing=True)
        t0.record(st)
        for a,b in evs:
            if events: a.record(st)
            if zero: lam.zero_()
            gr.launch(st)
            if events: b.record(st)
        t1.record(st)
        torch.cuda.synchronize()
    gr.close()
    print(f"{tag}: {t0.elapsed_time(t1)/30*1e3:.1f} us per replay")
side=torch.cuda.Stream()
for mode in (2,):
    run(f"mode {mode} null stream, zero+events", mode, None, True, True)
    run(f"mode {mode} null stream, events only", mode, None, False, True)
    run(f"mode {mode} null stream, zero only", mode, None, True, False)
    run(f"mode {mode} null stream, nothing", mode, None, False, False)
    run(f"mode {mode} side stream, zero+events", mode, side, True, True)
    run(f"mode {mode} side stream, nothing", mode, side, False, False)
