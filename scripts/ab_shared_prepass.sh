#!/bin/bash
# A/B of the shared-matrix pre-pass (csrc/ridge_gram.hpp ridge_gram_prepass_kernel): the main library (chunked loads)
# against cppnumericalsolvers_amd/variants/lib_prepass_direct.so (-DMI355_GRAM_PREPASS_DIRECT, only dispatch_ridge_gram.o
# swapped), bench.py --workload cfg4big / cfg4.
for wl in cfg4big cfg4; do
  for v in "" prepass_direct; do
    if [ -z "$v" ]; then LIB=""; NAME="chunked (main library)"; else LIB="$PWD/cppnumericalsolvers_amd/variants/lib_$v.so"; NAME=$v; fi
    for i in 1 2; do
      MI355_LBFGS_LIBRARY=$LIB python bench.py --workload $wl --no-secondary --no-cpu-baseline --no-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$wl', '$NAME', '%.4e solves/s kernel %.3f ms' % (d['value'], r['kernel_ms']))"
    done
  done
done
