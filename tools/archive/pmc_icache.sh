#!/bin/bash
# instruction-cache counters of k_step over a bench run: tools/pmc_icache.sh <tag> [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; shift || true
OUT=$ROOT/gpurun_out/pmci_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/a.txt 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/a/*/*_counter_collection.csv")
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "k_step" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "%.0f" % (sum(v)/len(v)))
PY
