#!/bin/bash
# Everything profiles/rNN_* is derived from, in one GPU-box call (run from the repo root):
#   bash gbd-pcg_amd/tools/profile_round.sh r03
# rocprofv3 passes of bench.py (kernel stats; FETCH_SIZE, WRITE_SIZE and SQ counters in separate --pmc passes, each with
# --kernel-trace only) and of the bandwidth probe that calibrates FETCH_SIZE on reads of known size.  Raw output goes to
# gpurun_out/prof_<round>/ ; gbd-pcg_amd/tools/profile_digest.py turns it into the files committed under profiles/.
set -o pipefail
R=${1:-r03}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$R
rm -rf "$OUT"; mkdir -p "$OUT"
[ -x gbd-pcg_amd/tools/bw_probe ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Iinclude gbd-pcg_amd/tools/bw_probe.hip -o gbd-pcg_amd/tools/bw_probe -Lgbd-pcg_amd/csrc -lgbdpcg -Wl,-rpath,'$ORIGIN/../csrc' || exit 1
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
echo "[1/6] bench line"
python3 $B > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "[2/6] kernel stats (config 3 only: the dispatches roofline.kernel_ms is measured on; then with the configs block)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B --no-cpu-baseline --no-configs --no-converged > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_configs -- python3 $B --no-cpu-baseline > $OUT/stats_configs.log 2>&1 || { tail -5 $OUT/stats_configs.log; exit 1; }
echo "[3/6] FETCH_SIZE"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -- python3 $B --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-converged > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
echo "[4/6] WRITE_SIZE"
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -- python3 $B --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-converged > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
echo "[5/6] SQ counters"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    -d $OUT/sq -- python3 $B --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-converged > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
echo "[6/6] FETCH_SIZE calibration (bw_probe)"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/cal -- $ROOT/gbd-pcg_amd/tools/bw_probe > $OUT/bw_probe.txt 2> $OUT/cal.log || { tail -5 $OUT/cal.log; exit 1; }
echo "[7] every single-GPU config, all paths (bench_configs.py)"
python3 $ROOT/gbd-pcg_amd/tools/bench_configs.py --reps 50 > $OUT/configs.jsonl 2> $OUT/configs.err || { tail -5 $OUT/configs.err; exit 1; }
echo "[8] per-iteration SQ counters of the resident kernel (counter_fit.py)"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    -d $OUT/cfit -- python3 $ROOT/gbd-pcg_amd/tools/counter_fit.py run > $OUT/cfit.log 2>&1 || { tail -5 $OUT/cfit.log; exit 1; }
python3 $ROOT/gbd-pcg_amd/tools/counter_fit.py digest $OUT/cfit > $OUT/counter_fit.txt
echo "[9] phase stamps of the persistent kernel (diagnostic build, if present)"
if [ -f $ROOT/gbd-pcg_amd/csrc/variants/libgbdpcg_stamps.so ]; then
  GBDPCG_LIB=$ROOT/gbd-pcg_amd/csrc/variants/libgbdpcg_stamps.so python3 $ROOT/gbd-pcg_amd/tools/persist_stamps.py > $OUT/persist_stamps.txt 2>/dev/null || true
fi
echo "[10] the cluster kernel (general storage): time per iteration and fixed cost, phase stamps, and what one hand-off costs"
{
  python3 $ROOT/gbd-pcg_amd/tools/ab_cluster.py 128 1024 base
  python3 $ROOT/gbd-pcg_amd/tools/ab_cluster.py 128 128 base
  python3 $ROOT/gbd-pcg_amd/tools/ab_cluster.py 128 1 base
  python3 $ROOT/gbd-pcg_amd/tools/ab_cluster.py 256 1 base
  if [ -f $ROOT/gbd-pcg_amd/csrc/variants/libgbdpcg_clstamps.so ]; then
    echo "--- diagnostic build with stamps (its iterations are slower than the shipped build's: the stamps sit in the loop)"
    GBDPCG_LIB=$ROOT/gbd-pcg_amd/csrc/variants/libgbdpcg_clstamps.so python3 $ROOT/gbd-pcg_amd/tools/cluster_stamps.py 128 1024 | tail -8
  fi
} > $OUT/cluster.txt 2>/dev/null || true
echo "[11] the steps either side of the solve (schur.hip): stage times, kernel stats, traffic, phase stamps"
{
  python3 $ROOT/gbd-pcg_amd/tools/schur_run.py --reps 40
  python3 $ROOT/gbd-pcg_amd/tools/schur_run.py --reps 40 --N 50 --batch 2048
  python3 $ROOT/gbd-pcg_amd/tools/schur_run.py --reps 20 --dtype f64
  python3 $ROOT/gbd-pcg_amd/tools/schur_run.py --reps 20 --nx 12 --nu 4
  python3 $ROOT/gbd-pcg_amd/tools/schur_run.py --reps 20 --nx 9 --nu 3
  if [ -f $ROOT/gbd-pcg_amd/csrc/variants/libgbdpcg_schurstamps.so ]; then
    GBDPCG_LIB=$ROOT/gbd-pcg_amd/csrc/variants/libgbdpcg_schurstamps.so python3 $ROOT/gbd-pcg_amd/tools/schur_run.py --reps 5 --stamps
  fi
} > $OUT/schur.txt 2>/dev/null || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/schur_stats -- python3 $ROOT/gbd-pcg_amd/tools/schur_run.py --reps 40 > $OUT/schur_stats.log 2>&1 || { tail -5 $OUT/schur_stats.log; exit 1; }
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/schur_fetch -- python3 $ROOT/gbd-pcg_amd/tools/schur_run.py --reps 3 > $OUT/schur_fetch.log 2>&1 || { tail -5 $OUT/schur_fetch.log; exit 1; }
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/schur_write -- python3 $ROOT/gbd-pcg_amd/tools/schur_run.py --reps 3 > $OUT/schur_write.log 2>&1 || { tail -5 $OUT/schur_write.log; exit 1; }
echo "[12] the whole inner step as one graph (examples/kkt_step_loop): which kernels a step is made of"
if [ -x $ROOT/gbd-pcg_amd/examples/kkt_step_loop ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kkt_stats -- $ROOT/gbd-pcg_amd/examples/kkt_step_loop 1024 128 20 > $OUT/kkt_step_loop.txt 2> $OUT/kkt_stats.log || { tail -5 $OUT/kkt_stats.log; exit 1; }
fi
echo "[13] phase stamps of a round of the resident symmetric kernel (diagnostic build, if present)"
if [ -f $ROOT/gbd-pcg_amd/csrc/variants/libgbdpcg_rsstamps.so ]; then
  GBDPCG_LIB=$ROOT/gbd-pcg_amd/csrc/variants/libgbdpcg_rsstamps.so python3 $ROOT/gbd-pcg_amd/tools/rs_stamps.py > $OUT/resident_stamps.txt 2>/dev/null || true
fi
echo "[14] converged / fixed-count solves by shape on the default path (solve_shapes.py)"
python3 $ROOT/gbd-pcg_amd/tools/solve_shapes.py 14,128,1024,f32 12,128,1024,f32 16,128,1024,f32 10,128,1024,f32 8,256,1024,f32 13,128,1024,f32 3,128,1024,f32 5,128,1024,f32 7,128,1024,f32 7,64,1024,f32 9,128,1024,f32 11,128,1024,f32 15,128,1024,f32 18,128,1024,f32 20,128,1024,f32 14,128,1024,f64 13,128,1024,f64 7,128,1024,f64 16,128,1024,f64 12,128,1024,f64 10,128,1024,f64 8,128,1024,f64 14,256,64,f32 14,300,1024,f32 14,512,256,f32 8,600,512,f32 14,64,1,f32 36,256,1,f64 > $OUT/solve_shapes.jsonl 2>/dev/null || true
echo "[14b] one problem by shape (which path takes it), stair formation and SpMV by block size"
python3 $ROOT/gbd-pcg_amd/tools/solve_shapes.py 14,64,1,f32 14,128,1,f32 14,300,1,f32 16,300,1,f32 18,128,1,f32 20,64,1,f32 24,128,1,f32 36,256,1,f32 13,32,1,f32 14,32,1,f64 14,256,1,f64 24,256,1,f64 36,256,1,f64 36,256,2,f64 36,256,4,f64 36,256,8,f64 24,128,8,f32 20,64,12,f32 14,128,16,f32 14,128,128,f32 > $OUT/single_problem.jsonl 2>/dev/null || true
python3 $ROOT/gbd-pcg_amd/tools/pinv_shapes.py > $OUT/pinv_shapes.txt 2>/dev/null || true
python3 $ROOT/gbd-pcg_amd/tools/spmv_shapes.py > $OUT/spmv_shapes.txt 2>/dev/null || true
echo "[15] the stair kernel: fp32 MFMA + one LDS-DMA request per workgroup against the VALU kernel (ab_pinv.py)"
python3 $ROOT/gbd-pcg_amd/tools/ab_pinv.py > $OUT/ab_pinv.txt 2>/dev/null || true
[ -x $ROOT/gbd-pcg_amd/tools/bin/hop_probe ] && $ROOT/gbd-pcg_amd/tools/bin/hop_probe > $OUT/hop_probe.txt 2>/dev/null || true
# keep what the digest needs, drop the bulky traces
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT
