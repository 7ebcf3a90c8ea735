#!/bin/bash
# The primal-SVM constraint-family solve against the number of problems in flight (scripts/auglag_bench.py --svm-primal);
# MI355_DEBUG_SOLVE_WAVES / MI355_DEBUG_SOLVE_BLOCKS (csrc/engine_internal.hpp) reshape the resident grid.
run() { python scripts/auglag_bench.py --svm-primal --batch $1 --steps 2 --cpu-sample 1 --outer-limit 60 --loop fused 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves=${MI355_DEBUG_SOLVE_WAVES:-default} blocks=${MI355_DEBUG_SOLVE_BLOCKS:-default} B=$1: %.1f ms  %.0f solves/s' % (d['ms_per_step'], d['value']))"; }
for b in 1 16 128 512 1024 4096 16384; do run $b; done
export MI355_DEBUG_SOLVE_WAVES=1 MI355_DEBUG_SOLVE_BLOCKS=256
for b in 256 1024; do run $b; done
