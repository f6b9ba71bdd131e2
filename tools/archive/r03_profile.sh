#!/bin/bash
# r03 rocprofv3 evidence for bench.py's default (driver) command: kernel trace + stats, then FETCH_SIZE / WRITE_SIZE in two separate
# --pmc passes.  bench.py pre-rolls one whole episode outside its timed region (arenas at random episode phases), so the trace holds
# thousands of k_step launches: tools/summarize_profile.py reports the LAST K of them (the timed region) next to the all-launch stats.
# usage: tools/archive/r03_profile.sh <tag> [bench args...]
set -e
TAG=${1:-r03}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench_stats.json 2> $OUT/bench_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_write.json 2> $OUT/bench_write.err
ls $OUT/stats/*/ | head
