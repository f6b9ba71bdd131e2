#!/bin/bash
# per-kernel GPU time of the batched DQN loop (config 5): rocprofv3 --kernel-trace --stats over a short run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/dqn_stats; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT${PYTHONPATH:+:$PYTHONPATH}  # cwd is /tmp: the package is found through the path, python3 stays right behind `--`
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 -m roborugby_amd.dqn --num-envs 65536 --steps 120 --log-every 0 > $OUT/run.txt 2>&1
cd $ROOT
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%6.2f%% %8d calls avg %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
