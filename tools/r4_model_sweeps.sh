#!/bin/bash
# Sweeps of the rewritten k_model_fwd (compact Jacobian, closed-form tperi): random standard-parameterisation models on random systems of 1-4
# planets (Campbell and Thiele-Innes bases, nuisance sources, healed priors) forced onto the throughput kernels, and through the library's own
# choice of kernel family, with every CU's LDS poisoned ahead of each evaluation.   bash tools/r4_model_sweeps.sh > gpurun_out/<tag>.txt
cd "$(dirname "$0")/.."
export OCTO_TEST_MAX_P=4
for poison in nan 1e300 ""; do
  export OCTO_TEST_POISON_LDS=$poison
  for sb in 0 ""; do
    export OCTO_TEST_SMALL_BATCH=$sb
    echo "#### poison='$poison' OCTO_TEST_SMALL_BATCH='$sb' max P = 4"
    echo "== stress_model 300 seed 941"; python tests/stress_model.py 300 941 2>&1 | grep -i "worst\|fail\|error\|fault" | tail -4
  done
done
