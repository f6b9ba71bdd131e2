#!/bin/bash
# A/B of fused-DQN builds: tools/archive/r03_dqn_ab.sh <tag> lib1 lib2 ...  (update / act timing of each, then the gradient tests with the LAST one)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; TAG=$1; shift
mkdir -p gpurun_out/$TAG
for rep in 1 2; do
  for lib in "$@"; do
    RR_LIB_PATH=$(readlink -f $lib) timeout -k 10 120 python3 tools/dqn_update_bench.py 2>/dev/null | grep -v amdgpu.ids
  done
done | tee gpurun_out/$TAG/ab.txt
for last; do :; done
RR_LIB_PATH=$(readlink -f $last) timeout -k 10 300 python3 -m pytest tests/test_gpu_dqn_fused.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee -a gpurun_out/$TAG/ab.txt
