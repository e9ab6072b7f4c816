#!/bin/bash
# The multi-planet probes (tools/multi_planet_steps.py: a RA/Dec table of 1 250 rows per planet + 2 500 absolute-RV rows, nuisances, 4 096 walkers, fwd+grad)
# for 2 … 8 planets; with an argument: A/B of four planets against lib/variants/liboctofitter_hip_<arg>.so, two interleaved rounds
tag=${1:-r5_many}
V=$PWD/octofitter.jl_amd/lib/variants/liboctofitter_hip_old.so
{
for r in 1 2; do
  OCTOFITTER_HIP_LIB=$V python tools/multi_planet_steps.py 4 200 2>&1 | grep "us per step"
  python tools/multi_planet_steps.py 4 200 2>&1 | grep "us per step"
done
for P in 2 3 5 6 8; do python tools/multi_planet_steps.py $P 200 2>&1 | grep "us per step"; done
} > gpurun_out/${tag}_steps.txt
cat gpurun_out/${tag}_steps.txt
