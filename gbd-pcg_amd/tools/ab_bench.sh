#!/bin/bash
# A/B on ONE device: interleave bench.py runs of the default library and of each variant library.
# usage: ab_bench.sh <rounds> <variant.so> [<variant.so> ...]   (run from the repo root)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in default "$@"; do
    if [ "$lib" = default ]; then unset GBDPCG_LIB; else export GBDPCG_LIB=$PWD/$lib; fi
    python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib round $r: %.4e iter/s  pcg %.0f GB/s %.3f ms  spmv %.0f GB/s'%(d['value'],d['roofline']['achieved'],d['roofline']['kernel_ms'],d['spmv']['achieved']))"
  done
done
