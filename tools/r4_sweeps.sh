#!/bin/bash
# Round-4 long sweeps of the per-wave orbit constants / per-wave finish in k_small (the shape of the round-3 attempt that faulted on a
# 3-planet sep/PA + relative-RV + HGCA case): random systems of 1-4 planets through both kernel families, every CU's LDS poisoned
# with NaN and with 1e300 ahead of every evaluation, plus random models.   bash tools/r4_sweeps.sh > gpurun_out/<tag>_stress_sweeps.txt
cd "$(dirname "$0")/.."
export OCTO_TEST_MAX_P=4
for poison in nan 1e300 ""; do
  export OCTO_TEST_POISON_LDS=$poison
  for sb in "" 0; do
    export OCTO_TEST_SMALL_BATCH=$sb
    echo "#### poison='$poison' OCTO_TEST_SMALL_BATCH='$sb' max P = 4"
    echo "== stress_parity 500 systems seed 535"; python tests/stress_parity.py 500 535 2>&1 | grep -i "worst\|fail\|error\|fault" | tail -4
    echo "== stress_model 150 seed 536"; python tests/stress_model.py 150 536 2>&1 | grep -i "worst\|fail\|error\|fault" | tail -3
  done
done
