#!/bin/bash
# CPU sanitizers over the kernel's phase source (the host-emulated wave of tests/emu/rr_emu.cpp; GPU sanitizers are not available on the
# pool): builds the emulation with UBSan, then ASan, as the default and as the parity build, and runs the emulation test files against it.
# usage: tools/sanitize_emulation.sh   (restores the normal libraries afterwards)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT/tests/emu
python3 -c "import sys; sys.path.insert(0, '$ROOT/tests'); import emu_lib as el; el.build(False); el.build(True)"
cp librr_emu.so /tmp/librr_emu.so.bak; cp librr_emu_exact.so /tmp/librr_emu_exact.so.bak
restore() { cp /tmp/librr_emu.so.bak $ROOT/tests/emu/librr_emu.so; cp /tmp/librr_emu_exact.so.bak $ROOT/tests/emu/librr_emu_exact.so; touch $ROOT/tests/emu/*.so; }
trap restore EXIT
for SAN in "undefined -fno-sanitize-recover=undefined" "address"; do
  g++ -O1 -g -fPIC -ffp-contract=off -std=c++17 -shared -fsanitize=$SAN -o librr_emu.so rr_emu.cpp
  g++ -O1 -g -fPIC -ffp-contract=off -std=c++17 -shared -DRR_EXACT_TRIG=1 -fsanitize=$SAN -o librr_emu_exact.so rr_emu.cpp
  touch librr_emu.so librr_emu_exact.so
  PRE=""; case "$SAN" in address*) PRE=$(gcc -print-file-name=libasan.so);; esac
  echo "== -fsanitize=$SAN"
  (cd $ROOT && ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$PRE python3 -m pytest -q -x -p no:cacheprovider tests/test_parity_build.py tests/test_fixed_point_memo.py \
      tests/test_budgeted_step.py tests/test_emulated_wave.py tests/test_goal_scoring.py -k "not shortcut_equivalence" 2>&1 | tail -2)
done
