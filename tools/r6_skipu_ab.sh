#!/bin/bash
# Same-box A/B: this tree (uniform regions left unstructured: -mllvm -structurizecfg-skip-uniform-regions; the single-planet warm loops as a row-level diamond)
# against the committed library of the step before (lib/variants/…_post.so).
bash tools/r6_ab_variant.sh r6_skipu_ab "post default" "grad fwd nuis two_planet rv_gappy rv_gappy_nuis wide_prior logpost ofti pt"
{
for r in 1 2; do OCTO_PROBE_N=1,4,8 bash tools/ab_strong.sh post default; done
V=$PWD/octofitter.jl_amd/lib/variants
for P in 3 4 5 6 8; do
  OCTOFITTER_HIP_LIB=$V/liboctofitter_hip_post.so python tools/multi_planet_steps.py $P 100 2>&1 | grep "us per step"
  python tools/multi_planet_steps.py $P 100 2>&1 | grep "us per step"
done
OCTOFITTER_HIP_LIB=$V/liboctofitter_hip_post.so python tools/latency_w1.py 2>/dev/null | tail -12
python tools/latency_w1.py 2>/dev/null | tail -12
} > gpurun_out/r6_skipu_more.txt 2>&1
cat gpurun_out/r6_skipu_more.txt
python -m pytest tests/test_warm_start.py tests/test_tile_sort.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3
