#!/bin/bash
# Collects the rocprofv3 evidence for bench.py (run on the GPU box through gpurun).
# usage: tools/profile_r01.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench_stats.json 2> $OUT/bench_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_write.json 2> $OUT/bench_write.err
find $OUT -name "*.csv" | head -50
