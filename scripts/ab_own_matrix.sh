for v in "" own_kb8_mr0 own_kb16_mr0 own_kb32_mr0; do
  if [ -z "$v" ]; then LIB=""; NAME="kb8_mr10 (main library)"; else LIB="$PWD/cppnumericalsolvers_amd/variants/lib_$v.so"; NAME=$v; fi
  for i in 1 2; do
    MI355_LBFGS_LIBRARY=$LIB python bench.py --workload cfg4own --no-secondary --no-cpu-baseline --no-counters 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$NAME', '%.4e solves/s kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms']), d['roofline']['kernel'])"
  done
done
