#!/bin/bash
# SQ counter pass for one bench configuration.  usage: pmc_sq.sh <tag> <bench args...>
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-counters $*"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/a -o p -- $BENCH > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -o p -- $BENCH > $OUT/b.log 2>&1
python - <<PY
import csv,collections,glob
acc=collections.defaultdict(list); dur=[]
for f in glob.glob("$OUT/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'lbfgs' in r['Kernel_Name'] or 'ridge_mfma' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
m={k:sum(v)/len(v) for k,v in acc.items()}
print("$TAG", "kernel_ms=%.2f"%(sum(dur)/len(dur)))
for k in sorted(m): print("  %-24s %.4g"%(k,m[k]))
if 'SQ_ACTIVE_INST_VALU' in m and 'GRBM_GUI_ACTIVE' in m:
    print("  VALU busy frac (ACTIVE_INST_VALU*4/1024 / (GUI_ACTIVE/8)) = %.3f"%(m['SQ_ACTIVE_INST_VALU']*4/1024/(m['GRBM_GUI_ACTIVE']/8)))
    print("  VALU insts per wave = %.0f, cycles per VALU inst = %.2f"%(m['SQ_INSTS_VALU']/m['SQ_WAVES'], m['SQ_ACTIVE_INST_VALU']*4/m['SQ_INSTS_VALU']))
PY
