#!/bin/bash
# Same-box A/B of the planet-per-wave kernel (octo_mainp.h) against k_main<P> (-DOCTO_MAINP=0) and of its chunk / occupancy settings on the 3- and 4-planet probes
tag=${1:-r5_mainp}; shift
V=$PWD/octofitter.jl_amd/lib/variants
{
for r in 1 2; do
 for lib in "$@"; do
  for P in 3 4; do
   if [ $lib = default ]; then python tools/multi_planet_steps.py $P 200 2>&1 | grep "us per step"; else OCTOFITTER_HIP_LIB=$V/liboctofitter_hip_$lib.so python tools/multi_planet_steps.py $P 200 2>&1 | grep "us per step"; fi
  done
 done
done
} > gpurun_out/${tag}_ab.txt
cat gpurun_out/${tag}_ab.txt
