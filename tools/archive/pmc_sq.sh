#!/bin/bash
# SQ instruction/cycle counters for the step kernel (own pass, no tracing besides kernel-trace)
set -e
TAG=${1:-sq}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || true
python3 - <<PY
import csv, glob, collections
for sub in ("a","b"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub)
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_step" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, sum(v)/len(v))
PY
