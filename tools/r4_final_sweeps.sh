#!/bin/bash
# Long randomised sweeps of the final tree of round 4 with new seeds, both kernel families.   bash tools/r4_final_sweeps.sh > gpurun_out/<tag>.txt
cd "$(dirname "$0")/.."
export OCTO_TEST_MAX_P=4
for sb in "" 0; do
  export OCTO_TEST_SMALL_BATCH=$sb
  echo "#### OCTO_TEST_SMALL_BATCH='$sb' max P = 4"
  echo "== stress_parity 1500 systems seed 1041"; python tests/stress_parity.py 1500 1041 2>&1 | grep -i "worst\|fail\|error\|fault" | tail -4
  echo "== stress_model 800 seed 1042"; python tests/stress_model.py 800 1042 2>&1 | grep -i "worst\|fail\|error\|fault" | tail -3
  echo "== stress_high_e 1000 walkers seed 1043"; python tests/stress_high_e.py 1000 1043 2>&1 | grep -i "worst\|fail\|error" | tail -3
done
unset OCTO_TEST_SMALL_BATCH
echo "== stress_ofti 300 seed 1044"; python tests/stress_ofti.py 300 1044 2>&1 | grep -i "worst\|fail\|error" | tail -3
