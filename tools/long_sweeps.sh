#!/bin/bash
# The long randomised sweeps (tests/stress_*.py, tests/big_shapes.py) on the current build, once with the library's own choice of
# kernel family and once forced onto the throughput kernels. Run on a GPU box:  bash tools/long_sweeps.sh > gpurun_out/sweeps.txt
cd "$(dirname "$0")/.."
for sb in "" 0; do
  export OCTO_TEST_SMALL_BATCH=$sb
  echo "#### OCTO_TEST_SMALL_BATCH='$sb' ($([ -z "$sb" ] && echo "library default: k_small for W*P <= 512" || echo "throughput kernels only"))"
  echo "== stress_parity 250 systems seed 101"; python tests/stress_parity.py 250 101 2>&1 | grep -v "^ *[0-9]* P=" | grep -i "worst\|fail\|error" | tail -3
  echo "== stress_parity 60 systems seed 102, sizes x3"; python tests/stress_parity.py 60 102 3 2>&1 | grep -i "worst\|fail\|error" | tail -3
  echo "== stress_model 100 seed 103"; python tests/stress_model.py 100 103 2>&1 | grep -i "worst\|fail\|error" | tail -3
  echo "== stress_high_e 200 walkers seed 105"; python tests/stress_high_e.py 200 105 2>&1 | grep -i "worst\|fail\|error" | tail -3
  echo "== stress_many_orbits 48 walkers seed 106"; python tests/stress_many_orbits.py 48 106 2>&1 | grep -i "worst\|fail\|error" | tail -3
done
unset OCTO_TEST_SMALL_BATCH
echo "== stress_ofti 80 seed 104"; python tests/stress_ofti.py 80 104 2>&1 | grep -i "worst\|fail\|error" | tail -3
echo "== big_shapes"; python tests/big_shapes.py 2>&1 | grep -i "E=\|ok\|fail\|error" | tail -4
