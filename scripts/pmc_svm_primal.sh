#!/bin/bash
# Counter passes over the primal-SVM constraint-family solve (scripts/auglag_bench.py --svm-primal): where do a wavefront's
# cycles go?  usage (GPU box): bash scripts/pmc_svm_primal.sh [batch]
set -u
B=${1:-512}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_svm_primal
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/scripts/auglag_bench.py --svm-primal --batch $B --steps 1 --warmup 1 --cpu-sample 8 --outer-limit 60 --loop fused"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/a -o p -- $BENCH > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -o p -- $BENCH > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/c -o p -- $BENCH > $OUT/c.log 2>&1
python - <<PY
import csv,collections,glob
acc=collections.defaultdict(list); dur=[]
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lbfgs_solve_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
m={k:sum(v)/len(v) for k,v in acc.items()}
print("svm_primal batch $B: kernel_ms=%.2f" % (sum(dur)/max(1,len(dur))))
for k in sorted(m): print("  %-32s %.5g"%(k,m[k]))
w=m.get('SQ_WAVES',1)
for k in ('SQ_INSTS_VALU','SQ_INSTS_LDS','SQ_INSTS_VMEM_RD','SQ_INSTS_SALU','SQ_WAVE_CYCLES','SQ_WAIT_INST_ANY'):
    if k in m: print("  per wavefront %-22s %.4g" % (k, m[k]/w))
PY
