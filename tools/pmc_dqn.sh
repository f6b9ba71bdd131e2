#!/bin/bash
# SQ counters of the fused DQN learn kernel (k_dqn_fwd_bwd) over tools/dqn_update_bench.py: tools/pmc_dqn.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}
OUT=$ROOT/gpurun_out/pmc_dqn_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 $ROOT/tools/dqn_update_bench.py 32768 20 > $OUT/a.txt 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/b -- python3 $ROOT/tools/dqn_update_bench.py 32768 20 > $OUT/b.txt 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
for sub in ("a","b"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub)
    if not fs: print("no csv for", sub); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_dqn_fwd_bwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, "%.0f" % (sum(v)/len(v)), "per wave %.0f" % (sum(v)/len(v)/1024))
PY
