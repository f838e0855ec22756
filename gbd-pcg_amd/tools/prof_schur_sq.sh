#!/bin/bash
# SQ counters of the formation / recovery / stair kernels (one rocprofv3 --pmc pass over schur_run.py); run from the repo root on the GPU box:
#   bash gbd-pcg_amd/tools/prof_schur_sq.sh   -> gpurun_out/schur_sq.txt
ROOT=$PWD
OUT=$ROOT/gpurun_out/schur_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    -d $OUT -- python3 $ROOT/gbd-pcg_amd/tools/schur_run.py --reps 3 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - <<PY > $ROOT/gpurun_out/schur_sq.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    if "schur" not in k and "pinv" not in k and "pcg_" not in k: continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    print(k)
    print("   dispatches %d, cycles %.0f, VALU instructions %.3g, issue utilisation (x4 / 1024 SIMDs) %.3f, SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES %.3f, LDS instructions %.3g, WAIT_INST_ANY / WAVE_CYCLES %.3f"
          % (len(c.get("SQ_INSTS_VALU", [])), cyc, m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_VALU", 0) * 4 / (cyc * 1024) if cyc else 0,
             m.get("SQ_ACTIVE_INST_VALU", 0) / m.get("SQ_BUSY_CYCLES", 1), m.get("SQ_INSTS_LDS", 0), m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
PY
cat $ROOT/gpurun_out/schur_sq.txt
