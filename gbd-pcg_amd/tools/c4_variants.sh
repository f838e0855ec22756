#!/bin/bash
# C4 (n=36, N=256, fp64, one problem) on the persistent path: variant libraries x knots per workgroup.
#   bash gbd-pcg_amd/tools/c4_variants.sh "base p16" "1 2 3"
for V in ${1:-base}; do for K in ${2:-2 3}; do
  L=gbd-pcg_amd/csrc/libgbdpcg.so; [ "$V" != base ] && L=gbd-pcg_amd/csrc/variants/libgbdpcg_$V.so
  echo "variant=$V K=$K"
  GBDPCG_LIB=$L GBDPCG_PERSIST_K=$K timeout -k 10 300 python gbd-pcg_amd/tools/bench_configs.py --only C4 --reps 50 2>&1 | grep persist | python -c "
import sys,json
t={}
for l in sys.stdin:
    r=json.loads(l); t[(r['path'],r['run'])]=r['ms_median']*1e3
for p in ('persist','persist1r'):
    print('   %-9s fixed5 %.1f fixed25 %.1f conv %.1f  => %.2f us/iter' % (p,t[(p,'fixed5')],t[(p,'fixed25')],t[(p,'tol1e-6')],(t[(p,'fixed25')]-t[(p,'fixed5')])/20))
"; done; done
