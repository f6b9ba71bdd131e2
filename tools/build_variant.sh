#!/bin/bash
# Tuning build: tools/build_variant.sh <name> [extra hipcc flags]  ->  roborugby_amd/variants/lib_<name>.so
# (only the two default configurations, G/VW8 and T/VW2 in fp64, so it compiles in ~20 s; never the product library)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/roborugby_amd/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -shared -DRR_CFG_SUBSET=${RR_SUBSET:-1} "$@" \
  -o $ROOT/roborugby_amd/variants/lib_$NAME.so $ROOT/roborugby_amd/csrc/rr_kernels.hip $ROOT/roborugby_amd/csrc/rr_dqn.hip
echo built lib_$NAME.so
