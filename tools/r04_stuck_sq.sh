#!/bin/bash
# r04: SQ counters of ONE stuck arena's step (one wavefront, one working arena): instruction mix and issue / wait cycles of the serial chain
# usage: tools/r04_stuck_sq.sh <preset> <index>
P=${1:-G}; I=${2:-0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04_stuck_sq_${P}$I; mkdir -p $OUT
export RR_NO_MEMO=1 HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 $ROOT/tools/stuck_arena_step.py $P $I > $OUT/a.txt 2>&1 || { tail -5 $OUT/a.txt; exit 1; }
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/b -- python3 $ROOT/tools/stuck_arena_step.py $P $I > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $OUT/c -- python3 $ROOT/tools/stuck_arena_step.py $P $I > $OUT/c.txt 2>&1 || { tail -5 $OUT/c.txt; exit 1; }
cd $ROOT
tail -1 $OUT/a.txt
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
print("# one stuck arena ($P #$I) stepped alone, RR_NO_MEMO=1: SQ counters per k_step dispatch (one wavefront; means over the dispatches with Grid_Size 64)")
for sub in "abc":
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub)
    if not fs: print("no csv for", sub); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_step" in r["Kernel_Name"] and int(r["Grid_Size"]) == 64:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[2:] if len(v) > 4 else v
        print("%-24s %12.0f  (%d dispatches)" % (k, sum(v) / len(v), len(v)))
PY
