#!/bin/bash
# r02 profiling call: phase split (random / chase), SQ counters with and without the LDS bank padding, chase bench lines
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_prof1
mkdir -p $OUT
cd $ROOT
for p in G T; do for m in random chase; do
  RR_LIB_PATH=$ROOT/roborugby_amd/variants/lib_prof.so timeout -k 10 200 python3 tools/phase_profile.py $p $m > $OUT/phase_${p}_${m}.txt 2>&1 || { echo "phase $p $m failed"; tail -3 $OUT/phase_${p}_${m}.txt; exit 1; }
  cat $OUT/phase_${p}_${m}.txt
done; done
timeout -k 10 400 bash tools/pmc_sq2.sh G_pad > $OUT/sq_G_pad.txt 2>&1 || { echo "pmc pad failed"; tail $OUT/sq_G_pad.txt; exit 1; }
cat $OUT/sq_G_pad.txt
RR_LIB_PATH=$ROOT/roborugby_amd/variants/lib_nopad.so timeout -k 10 400 bash tools/pmc_sq2.sh G_nopad > $OUT/sq_G_nopad.txt 2>&1 || { echo "pmc nopad failed"; exit 1; }
cat $OUT/sq_G_nopad.txt
for p in G T; do
  timeout -k 10 300 python3 bench.py --preset $p --policy chase --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_${p}_chase.json 2> $OUT/bench_${p}_chase.err || { echo "chase $p failed"; exit 1; }
  cat $OUT/bench_${p}_chase.json
done
