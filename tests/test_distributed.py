"""N > 1 host logic on CPU: world_size-2 gloo.  The data path has no collective (independent
problems); what is distributed is the sharding of the batch and the aggregation of the numbers
bench.py reports (max elapsed over ranks, total units)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gbd_pcg_amd import sharding


def test_shard_range_covers_batch_exactly():
    for batch in (1, 7, 8, 1024, 8192, 8193):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gbd_pcg_amd import synth
        from oracle import oracle as orc
        # each rank solves its own shard of an 6-problem batch with the CPU oracle (stand-in for the
        # GPU solve in this CPU-only test) and the ranks aggregate throughput numbers
        n, N, B = 14, 8, 6
        lo, hi = sharding.shard_range(B, rank, world)
        d = synth.gen_numpy(n, N, seed=1234 + lo, batch=hi - lo, dtype=np.float64)
        ob = orc.pcg_batch(n, N, hi - lo, d["S"], d["Pinv"], d["gamma"], tol=1e-6, max_iter=50)
        units = float(ob["iters"].sum())
        elapsed, total = sharding.aggregate(1.0 + rank, units)
        dist.barrier()
        if rank == 0:
            full = synth.gen_numpy(n, N, seed=1234, batch=B, dtype=np.float64)
            of = orc.pcg_batch(n, N, B, full["S"], full["Pinv"], full["gamma"], tol=1e-6, max_iter=50)
            out.put((elapsed, total, float(of["iters"].sum())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_aggregation():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    elapsed, total, want = out.get(timeout=5)
    assert elapsed == 2.0          # max over ranks
    assert total == want           # shards together did exactly the work of the whole batch


def test_seeded_generator_is_rank_independent():
    """bench.py builds rank g's shard with synth.gen_torch_seeded(lo, hi): problem i must be Gen(n, N, 1234 + i)
     whatever slice it is generated in -- equal to the canonical numpy generator to fp64
    rounding, and identical between two different slicings of the same batch."""
    from gbd_pcg_amd import synth
    n, N, B = 14, 9, 6
    ref = synth.gen_numpy(n, N, seed=1234, batch=B, dtype=np.float64)
    whole = synth.gen_torch_seeded(n, N, 0, B, "cpu", torch.float64, seed=1234, chunk=4)
    parts = [synth.gen_torch_seeded(n, N, lo, hi, "cpu", torch.float64, seed=1234)
             for lo, hi in (sharding.shard_range(B, r, 4) for r in range(4)) if hi > lo]
    for key in ("S", "Pinv", "gamma"):
        got = whole[key].numpy()
        assert np.abs(got - ref[key]).max() <= 1e-12 * np.abs(ref[key]).max(), key
        assert np.array_equal(np.concatenate([p[key].numpy() for p in parts]), got), key


def test_bench_launcher_starts_one_rank_per_gpu():
    """`python bench.py --gpus 2` with no torch.distributed environment must itself start two fresh ranks
    (rendezvous over gloo here: --dry-run exercises the launcher, sharding and aggregation without a GPU and
    reports no metric value).  The line rank 0 prints carries the ranks the process group saw."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--dry-run"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["dry_run"] is True and rec["value"] is None
    assert rec["n_gpus"] == 2 and rec["shards"] == [[0, 1024], [1024, 2048]]
    assert len(set(rec["pids"])) == 2 and os.getpid() not in rec["pids"]
    assert rec["max_elapsed"] == 2.0 and rec["total_units"] == 2 * 1024 * 25 * 3
    # a rank count that contradicts the environment is an error, not a silent one-rank run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-run"],
                         capture_output=True, text=True, timeout=60, env=dict(env, WORLD_SIZE="2", RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr
    # a rank that dies takes the job down, and the launcher says which rank died of what (its stderr tail)
    dead = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                          capture_output=True, text=True, timeout=120, env=dict(env, GBDPCG_BENCH_DRY_FAIL_RANK="1"))
    assert dead.returncode != 0
    assert "bench.py rank 1, exit code 1" in dead.stderr and "injected failure on rank 1" in dead.stderr


def test_kkt_generator_shapes_and_determinism():
    """synth.kkt_torch (input of bench.py's mpc_step block and of tools/schur_run.py): packed sizes of include/gbdpcg.h, the same
    blocks for the same seed, symmetric positive definite cost blocks."""
    import numpy as np
    import torch
    from gbd_pcg_amd import synth
    nx, nu, N, B = 6, 3, 5, 2
    a = synth.kkt_torch(nx, nu, N, B, "cpu", torch.float64, seed=3)
    b = synth.kkt_torch(nx, nu, N, B, "cpu", torch.float64, seed=3)
    sizes = [B * ((nx * nx + nu * nu) * N - nu * nu), B * (nx * nx + nx * nu) * (N - 1), B * ((nx + nu) * N - nu), B * nx * N]
    assert [t.numel() for t in a] == sizes and all(torch.equal(x, y) for x, y in zip(a, b))
    G = a[0].reshape(B, -1).numpy()
    for k in range(N):
        Q = G[0, k * (nx * nx + nu * nu):][:nx * nx].reshape(nx, nx)
        assert np.allclose(Q, Q.T) and np.linalg.eigvalsh(Q).min() >= 1.0 - 1e-9
    one = synth.kkt_torch(nx, nu, 1, B, "cpu", torch.float32)
    assert one[1].numel() == 0 and one[0].numel() == B * nx * nx
