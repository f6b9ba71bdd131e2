#!/bin/bash
# SQ counters of the k_step launches of tools/slow_arena_bench.py (one wavefront running the squeezed arena)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_slow_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 $ROOT/tools/slow_arena_bench.py 1 > $OUT/a.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/b -- python3 $ROOT/tools/slow_arena_bench.py 1 > $OUT/b.txt 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
for sub in ("a","b"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub)
    if not fs: print("no csv for", sub); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_step" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, sorted(v)[len(v)//2])
PY
