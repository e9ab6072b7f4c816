#!/bin/bash
# Tiles per block of the planet-per-wave kernel (octo_mainp.h): the multi-planet probe (tools/multi_planet_steps.py) for 3 … 8 planets with one tile per
# block (OCTO_MAINP_TPB=1: the shape of the first round-5 build) against the default table (3 planets x 4 tiles on the mp3 variant, 5-7 planets x 2), and 4 tiles
# where twelve or sixteen waves allow it.   bash tools/r5_tpb_ab.sh <tag>
tag=${1:-r5_tpb}
V3=$PWD/octofitter.jl_amd/lib/variants/liboctofitter_hip_mp3.so
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "planet_per_wave or more_than_four or three_and_four" > gpurun_out/${tag}_tests.txt 2>&1
tail -3 gpurun_out/${tag}_tests.txt
OCTOFITTER_HIP_LIB=$V3 python -m pytest tests/test_gpu_parity.py tests/test_configs_gpu.py -q -x -m gpu -k "three_and_four or three_planet or planets" > gpurun_out/${tag}_tests_mp3.txt 2>&1
tail -3 gpurun_out/${tag}_tests_mp3.txt
{
for r in 1 2; do
  for P in 3 4 5 6 7 8; do
    for tpb in 1 0 4; do
      if [ $P = 3 ]; then export OCTOFITTER_HIP_LIB=$V3; else unset OCTOFITTER_HIP_LIB; fi
      what="default table"; [ $tpb = 1 ] && what="one tile per block"; [ $tpb = 4 ] && what="OCTO_MAINP_TPB=4 (clamped to the block's 4 x waves-per-SIMD)"
      [ $P = 3 ] && [ $tpb = 1 ] && { echo -n "P=3 k_main<3> (default build): "; env -u OCTOFITTER_HIP_LIB python tools/multi_planet_steps.py 3 200 2>&1 | grep "us per step"; }
      echo -n "P=$P $what: "
      OCTO_MAINP_TPB=$tpb python tools/multi_planet_steps.py $P 200 2>&1 | grep "us per step"
    done
  done
done
unset OCTOFITTER_HIP_LIB
} > gpurun_out/${tag}_steps.txt 2>&1
cat gpurun_out/${tag}_steps.txt
