#!/bin/bash
# A/B of the own-matrix pre-pass (csrc/ridge_gram.hpp ridge_gram_own_prepass_kernel) on ONE GPU box: the main library
# (blocked form, MI355_GRAM_OWN_CHUNK = 16) against variants that swap only dispatch_ridge_gram.o
# (cppnumericalsolvers_amd/variants/lib_own_{kc8,kc4,direct}.so), bench.py --workload cfg4own, then the own-matrix parity
# tests under every library.
for v in "" own_kc8 own_kc4 own_direct; do
  if [ -z "$v" ]; then LIB=""; NAME="kc16 (main library)"; else LIB="$PWD/cppnumericalsolvers_amd/variants/lib_$v.so"; NAME=$v; fi
  for i in 1 2; do
    MI355_LBFGS_LIBRARY=$LIB python bench.py --workload cfg4own --no-secondary --no-cpu-baseline --no-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$NAME', '%.4e solves/s kernel %.3f ms' % (d['value'], r['kernel_ms']), {k: r[k] for k in r if 'prepass' in k})"
  done
done
for v in "" own_kc8 own_kc4; do
  if [ -z "$v" ]; then LIB=""; else LIB="$PWD/cppnumericalsolvers_amd/variants/lib_$v.so"; fi
  echo "parity under ${v:-main}:"; MI355_LBFGS_LIBRARY=$LIB python -m pytest tests/test_gpu_ridge_gram.py -m gpu -x -q -k "own or per_problem" 2>&1 | tail -2
done
