#!/bin/bash
# occupancy probe on G: 5 arenas per wavefront (24 idle lanes) so that LDS admits 3 waves per SIMD, 168 VGPRs forced (52 spilled)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out/r02_apw
for args in "--steps 100 --warmup 20" "--policy chase --steps 100 --warmup 150"; do
  echo "== $args"
  bash tools/ab.sh "$args" roborugby_amd/variants/lib_apw8hint.so roborugby_amd/variants/lib_apw5.so
done 2>&1 | tee gpurun_out/r02_apw/ab.txt
