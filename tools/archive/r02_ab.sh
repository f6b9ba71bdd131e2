#!/bin/bash
# generic A/B of variant libraries: tools/r02_ab.sh <tag> "<bench args>;<bench args>;..." lib1 lib2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; TAG=$1; ARGSETS=$2; shift 2
mkdir -p gpurun_out/$TAG
IFS=';' read -ra SETS <<< "$ARGSETS"
for args in "${SETS[@]}"; do
  echo "== $args"
  bash tools/ab.sh "$args" "$@"
done 2>&1 | tee gpurun_out/$TAG/ab.txt
