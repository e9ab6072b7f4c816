#!/bin/bash
# Config 4: the last planet's warm start with a per-row test (this tree) against the static per-wave criterion (lib/variants/…_head.so), same box, two rounds;
# then the parity sweeps that cover the two-planet loops.
V=$PWD/octofitter.jl_amd/lib/variants
{
for r in 1 2; do
  OCTOFITTER_HIP_LIB=$V/liboctofitter_hip_head.so timeout 300 python tools/r6_cfg4_probe.py 2>&1 | grep "us per step"
  timeout 300 python tools/r6_cfg4_probe.py 2>&1 | grep "us per step"
done
} > gpurun_out/r6_cfg4_dyn.txt 2>&1
cat gpurun_out/r6_cfg4_dyn.txt
timeout 900 python tests/stress_round6.py 40 5 2>&1 | tail -3
timeout 600 python -m pytest tests/test_tile_sort.py tests/test_warm_start.py tests/test_configs_gpu.py -m gpu -q -x 2>&1 | tail -4
