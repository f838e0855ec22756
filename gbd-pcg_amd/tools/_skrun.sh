for rep in 1 2; do
for v in "" old; do
  if [ -z "$v" ]; then unset GBDPCG_LIB; else export GBDPCG_LIB=$PWD/gbd-pcg_amd/csrc/variants/libgbdpcg_$v.so; fi
  echo "== ${v:-base}"; timeout -k 10 120 python gbd-pcg_amd/tools/schur_run.py --reps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['form_schur_us'], d['recover_primal_us'])" || true
done
done
