#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/a.json 2> $OUT/a.err
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/a/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(list); dur=[]
for r in csv.DictReader(open(f)):
    if "k_step" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
d = sum(dur)/len(dur)
m = {k: sum(v)/len(v) for k,v in acc.items()}
print("kernel ns", d)
for k,v in m.items(): print(k, v)
clk = m["GRBM_GUI_ACTIVE"]/8/ (d*1e-9)
print("clock GHz", clk/1e9)
cycles = d*1e-9*clk
print("kernel cycles", cycles, "VALU active cycles per SIMD", m["SQ_ACTIVE_INST_VALU"]*4/1024, "util", m["SQ_ACTIVE_INST_VALU"]*4/1024/cycles)
PY
