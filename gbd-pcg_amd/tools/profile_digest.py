#!/usr/bin/env python3
"""gpurun_out/prof_<round>/ (written by profile_round.sh on the GPU box) -> the files committed under profiles/:

    python gbd-pcg_amd/tools/profile_digest.py r02

  profiles/<round>_bench.jsonl                 the bench line of that run (appended)
  profiles/<round>_bench_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of `bench.py --no-configs` (this library's
                                               kernels; config 3 dispatches only); ..._with_configs.csv: the default command
  profiles/<round>_pmc_traffic.json            FETCH_SIZE / WRITE_SIZE per launch (pmc_traffic.py)
  profiles/<round>_pmc_sq.json                 SQ counters per launch + derived VALU-issue utilisation and LDS
                                               bank-conflict share (read by bench.py into roofline.valu_issue)
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SIMDS = 256 * 4   # MI355X: 256 CUs x 4 SIMDs


def counters(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        names = {}
        for row in csv.DictReader(open(f)):
            per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = row["Kernel_Name"]
        for (d, c), v in per.items():
            acc[names[d]][c].append(v)
    return acc


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}")
    dst = os.path.join(ROOT, "profiles")
    # gpurun MERGES a call's output into gpurun_out/: files of an earlier profile round (other process ids in their names) stay
    # behind next to the new ones.  Everything older than half an hour before the newest file is a leftover: removed.
    files = [os.path.join(d, f) for d, _, fs in os.walk(src) for f in fs]
    newest = max(os.path.getmtime(f) for f in files)
    for f in files:
        if os.path.getmtime(f) < newest - 1800:
            os.remove(f)
    line = [ln for ln in open(os.path.join(src, "bench.json")) if ln.startswith("{")][-1]
    with open(os.path.join(dst, f"{rnd}_bench.jsonl"), "a") as f:
        f.write(line if line.endswith("\n") else line + "\n")
    # kernel stats
    for sub, tag in (("stats", "bench_kernel_stats"), ("stats_configs", "bench_kernel_stats_with_configs")):
        stats = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            # bench.py times the streaming general kernel in a child process, which leaves a stats file of its own: the
            # bench's is the one with the headline kernel in it
            stats.sort(key=lambda f: -sum(1 for ln in open(f) if "pcg_resident_sym_kernel" in ln or "gbdpcg" in ln))
            main = [f for f in stats if "pcg_resident_sym_kernel" in open(f).read()]
            rows = list(csv.reader(open((main or stats)[0])))
            keep = [rows[0]] + [r for r in rows[1:] if "gbdpcg" in r[0]]
            with open(os.path.join(dst, f"{rnd}_{tag}.csv"), "w", newline="") as f:
                csv.writer(f).writerows(keep)
    # traffic
    subprocess.check_call([sys.executable, os.path.join(ROOT, "gbd-pcg_amd", "tools", "pmc_traffic.py"),
                           os.path.join(src, "fetch"), os.path.join(src, "write"), os.path.join(src, "cal"),
                           os.path.join(dst, f"{rnd}_pmc_traffic.json")], stdout=subprocess.DEVNULL)
    # the steps either side of the solve (schur.hip): stage times + kernel stats + traffic of tools/schur_run.py
    st = os.path.join(src, "schur.txt")
    if os.path.exists(st):
        nx, nu, N, B = 14, 7, 128, 1024
        alg = {"schur_form_quad": B * N * (2 * (nx * nx + nu * nu) + nx * nx + nx * nu + 2 * nx + nu + 3 * nx * nx + nx) * 4,
               "schur_recover_quad_kernel": B * N * ((nx * nx + nu * nu) + nx * nx + nx * nu + 2 * (nx + nu) + nx) * 4}
        rec = {"_how": "gbd-pcg_amd/tools/schur_run.py (1024 problems, stateSize 14, controlSize 7, knotPoints 128, fp32) under rocprofv3 "
                       "--kernel-trace --stats, and in separate passes --pmc FETCH_SIZE / --pmc WRITE_SIZE (KiB; reads x2: the gfx950 "
                       "correction of pmc_traffic.json); algorithmic bytes = every input once + every output once (to the knot: the "
                       "last knot of a problem has no R, A, B, r)", "kernels": {}}
        stats = glob.glob(os.path.join(src, "schur_stats", "**", "*kernel_stats.csv"), recursive=True)
        avg = {}
        for f in stats:
            for row in csv.DictReader(open(f)):
                for k in alg:
                    if k in row["Name"]:
                        avg[k] = (float(row["AverageNs"]), int(row["Calls"]))
        import pmc_traffic
        fetch = pmc_traffic.per_kernel(os.path.join(src, "schur_fetch"), "FETCH_SIZE")
        write = pmc_traffic.per_kernel(os.path.join(src, "schur_write"), "WRITE_SIZE")
        for k, a in alg.items():
            f, _ = pmc_traffic.pick(fetch, k)
            w, _ = pmc_traffic.pick(write, k)
            r = {"algorithmic_bytes_per_launch": a}
            if k in avg:
                r.update({"AverageNs": avg[k][0], "calls": avg[k][1], "GBps": a / avg[k][0], "frac_of_8TBps": a / avg[k][0] / 8000.0})
            if f is not None and w is not None:
                r.update({"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes_per_launch": 2 * f * 1024 + w * 1024,
                          "traffic_over_algorithmic": (2 * f * 1024 + w * 1024) / a})
            rec["kernels"][k] = r
        json.dump(rec, open(os.path.join(dst, f"{rnd}_schur_kernels.json"), "w"), indent=1)
        with open(os.path.join(dst, f"{rnd}_schur.txt"), "w") as f:
            f.write("# gbd-pcg_amd/tools/schur_run.py on one MI355X: KKT blocks -> S, gamma, G^-1 (form_schur), stair Pinv, converged solve, lambda -> z\n"
                    "# (recover_primal); event-timed medians, GB/s = every input once + every output once / time.  Shapes: the headline batch,\n"
                    "# a horizon that is not a multiple of 4, fp64, the quadrotor block sizes (12 / 4), block sizes without a four-knot kernel (9 / 3: the LDS kernels), then the phase stamps\n"
                    "# of one step of the four-knot formation kernel (shader cycles; diagnostic build).\n")
            f.write(open(st).read())
    # the kernels of one whole inner step (examples/kkt_step_loop under rocprofv3 --kernel-trace --stats)
    ks = glob.glob(os.path.join(src, "kkt_stats", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        rows = list(csv.reader(open(ks[0])))
        keep = [rows[0]] + [r for r in rows[1:] if "gbdpcg" in r[0]]
        with open(os.path.join(dst, f"{rnd}_kkt_step_kernel_stats.csv"), "w", newline="") as f:
            csv.writer(f).writerows(keep)
        lp = os.path.join(src, "kkt_step_loop.txt")
        if os.path.exists(lp):
            with open(os.path.join(dst, f"{rnd}_kkt_step_loop.txt"), "w") as f:
                f.write("# gbd-pcg_amd/examples/kkt_step_loop 1024 128 20 under rocprofv3 --kernel-trace (one graph replay per step: KKT blocks -> S, gamma,\n"
                        "# G^-1 -> stair Pinv -> PCG to |eta| < 1e-10, warm-started -> primal step; KKT residuals checked in fp64 on the host)\n")
                f.write(open(lp).read())
    for name, dstname, head in (("resident_stamps.txt", "resident_stamps.txt",
                                 "# gbd-pcg_amd/tools/rs_stamps.py (diagnostic build -DGBDPCG_RS_STAMPS: stamps in the loop, iterations ~20 % slower than shipped):\n"
                                 "# median over the 256 workgroups of the time between phase boundaries of each round, us (100 MHz real-time clock)\n"),
                                ("solve_shapes.jsonl", "solve_shapes.jsonl", ""), ("ab_pinv.txt", "pinv_ab_latest.txt", ""),
                                ("pinv_shapes.txt", "pinv_shapes.txt", "# gbd-pcg_amd/tools/pinv_shapes.py: stair Phi^-1 formation by block size, 1024 problems x 128 knots\n"),
                                ("spmv_shapes.txt", "spmv_shapes.txt", "# gbd-pcg_amd/tools/spmv_shapes.py: general-storage SpMV by block size, 1024 problems x 128 knots, >= 1.3 GB rotation\n"),
                                ("single_problem.jsonl", "single_problem.jsonl", "")):
        sp = os.path.join(src, name)
        if os.path.exists(sp) and os.path.getsize(sp) > 0:
            body = "".join(ln for ln in open(sp) if "amdgpu.ids" not in ln)
            with open(os.path.join(dst, f"{rnd}_{dstname}"), "w") as f:
                f.write(head + body)
    # SQ counters
    out = {"_how": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU "
                   "SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE on `bench.py --steps 3 --warmup 1` "
                   "(config 3), one MI355X; mean per dispatch over the long (fixed-25-iteration) dispatches of each kernel. "
                   "valu_issue_utilisation = SQ_INSTS_VALU x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the share of the "
                   "kernel's SIMD-cycles in which a vector instruction of a single wave could have been issued (one wave issues a "
                   "VALU instruction every 4 cycles, MI355X_MICROARCH.md); lds_bank_conflict_share = SQ_LDS_BANK_CONFLICT / "
                   "SQ_LDS_IDX_ACTIVE.  SQ cycle counters are in units of 4 cycles.",
           "kernels": {}}
    for name, cs in counters(os.path.join(src, "sq")).items():
        if "gbdpcg" not in name:
            continue
        rec = {}
        ref = cs.get("SQ_WAVE_CYCLES", [0])
        big = max(ref) if ref else 0
        for c, v in cs.items():
            sel = [x for x, r in zip(v, ref)] if len(v) != len(ref) else [x for x, r in zip(v, ref) if r > 0.5 * big]
            sel = sel or v
            rec[c] = sum(sel) / len(sel)
        rec["dispatches"] = len([r for r in ref if r > 0.5 * big])
        if rec.get("GRBM_GUI_ACTIVE"):
            cyc = rec["GRBM_GUI_ACTIVE"] / 8.0
            rec["kernel_cycles"] = cyc
            rec["valu_issue_utilisation"] = rec.get("SQ_INSTS_VALU", 0.0) * 4.0 / (cyc * SIMDS)
        if rec.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_share"] = rec.get("SQ_LDS_BANK_CONFLICT", 0.0) / rec["SQ_LDS_IDX_ACTIVE"]
        if rec.get("SQ_WAVE_CYCLES"):
            rec["wait_any_share"] = rec.get("SQ_WAIT_ANY", 0.0) / rec["SQ_WAVE_CYCLES"]
        short = name.split("(")[0].replace("void gbdpcg::", "").replace(" ", "")
        out["kernels"][short] = rec
    json.dump(out, open(os.path.join(dst, f"{rnd}_pmc_sq.json"), "w"), indent=1)
    # per-config table
    cj = os.path.join(src, "configs.jsonl")
    if os.path.exists(cj):
        lines = ["# gbd-pcg_amd/tools/bench_configs.py on one MI355X: every single-GPU BASELINE config and every path that can run it,",
                 "# hipGraph replay, median of 50 replays (each replay = lambda.zero_() + graph launch, event-timed, so ~10 us of launch",
                 "# path is inside).  fixed5 / fixed25 = exit_tol 0 with 5 / 25 iterations, tol1e-6 = converge; us/iter = (fixed25 - fixed5) / 20",
                 "# = the cost of one more real iteration.  C2 = n14 N64 fp32 x1, C3 = n14 N128 fp32 x1024, C4 = n36 N256 fp64 x1,",
                 "# C2x64 / C4x16 = small batches, C5on1 = config 5's 8192 problems on one GPU; persist = one persistent launch,",
                 "# persist1r = its opt-in single-reduction form,\n# general = the fused path with gbdpcg_set_symmetric(0): what storage that is not bit-symmetric gets (cluster kernel where the shape has it).  GB/s algorithmic = SURVEY 8d full-storage bytes / time."]
        recs = [json.loads(ln) for ln in open(cj) if ln.startswith("{")]
        by = collections.OrderedDict()
        for r in recs:
            by.setdefault((r["config"], r.get("path", "spmv")), {})[r["run"]] = r
        for (cfg, path), runs in by.items():
            if path == "spmv":
                r = runs["spmv"]
                lines.append(f"{cfg:6s} spmv                 {r['ms_median'] * 1e3:9.1f} us   {r['algorithmic_GBps']:8.0f} GB/s algorithmic")
                continue
            t5, t25, tc = runs.get("fixed5"), runs.get("fixed25"), runs.get("tol1e-6")
            if not (t5 and t25 and tc):
                continue
            per = (t25["ms_median"] - t5["ms_median"]) * 1e3 / 20.0
            lines.append(f"{cfg:6s} {path:10s} fixed25 {t25['ms_median'] * 1e3:9.1f} us  converged {tc['ms_median'] * 1e3:9.1f} us ({tc['iters_mean']:.1f} it)  "
                         f"{per:8.2f} us/iter  {t25['problem_iters_per_s']:.3e} problem-iter/s  {t25['algorithmic_GBps']:7.0f} GB/s algorithmic")
        open(os.path.join(dst, f"{rnd}_configs.txt"), "w").write("\n".join(lines) + "\n")
    for name in ("counter_fit.txt", "persist_stamps.txt", "cluster.txt", "hop_probe.txt"):
        if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)) > 0:
            open(os.path.join(dst, f"{rnd}_{name}"), "w").write(open(os.path.join(src, name)).read())
    if os.path.exists(os.path.join(src, "bw_probe.txt")):
        open(os.path.join(dst, f"{rnd}_bw_probe.txt"), "w").write(open(os.path.join(src, "bw_probe.txt")).read())
    print("digested", src, "->", dst)


if __name__ == "__main__":
    main()
