#!/bin/bash
# SQ issue/stall counters of k_step over a bench run: tools/pmc_sq2.sh <tag> [bench args]   (RR_LIB_PATH selects the build)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; shift || true
OUT=$ROOT/gpurun_out/pmc2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/a.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/b.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/c -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/c.txt 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
for sub in ("a","b","c"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub)
    if not fs: print("no csv for", sub); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_step" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, "%.0f" % (sum(v)/len(v)), "per wave %.0f" % (sum(v)/len(v)/8192))
PY
