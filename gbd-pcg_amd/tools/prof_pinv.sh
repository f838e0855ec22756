set -o pipefail
ROOT=$PWD; OUT=$ROOT/gpurun_out/prof_pinv; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
for arm in mfma valu; do
  if [ $arm = valu ]; then export GBDPCG_PINV_NO_MFMA=1; else unset GBDPCG_PINV_NO_MFMA; fi
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/${arm}_a -- python3 $ROOT/gbd-pcg_amd/tools/pinv_one.py > $OUT/${arm}_a.log 2>&1 || tail -3 $OUT/${arm}_a.log
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA -d $OUT/${arm}_b -- python3 $ROOT/gbd-pcg_amd/tools/pinv_one.py > $OUT/${arm}_b.log 2>&1 || tail -3 $OUT/${arm}_b.log
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${arm}_s -- python3 $ROOT/gbd-pcg_amd/tools/pinv_one.py 14 20 > $OUT/${arm}_s.log 2>&1 || tail -3 $OUT/${arm}_s.log
  echo "== $arm"; python3 $ROOT/gbd-pcg_amd/tools/pmc_summary.py $OUT/${arm}_a pinv; python3 $ROOT/gbd-pcg_amd/tools/pmc_summary.py $OUT/${arm}_b pinv
  grep -h pinv $OUT/${arm}_s/*/*kernel_stats.csv | cut -c1-200
done
find $OUT -name "*.db" -delete
