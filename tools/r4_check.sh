#!/bin/bash
# Round-4 GPU check of a build against the round-3 library (octofitter.jl_amd/lib/variants/liboctofitter_hip_r3.so, built from the r3 tree):
# the -m gpu suite, then same-box latencies of the multi-planet / HGCA / model one-θ calls for both builds (and for this build with
# every dataset forced onto the widest kind set, OCTO_KIND_ALL=1: what the narrower sets buy).
#   bash tools/r4_check.sh <tag> [pytest args]     (through gpurun; writes gpurun_out/<tag>_*.txt)
tag=${1:-r4_v0}; shift
mkdir -p gpurun_out
R3=$PWD/octofitter.jl_amd/lib/variants/liboctofitter_hip_r3.so
python -m pytest tests -m gpu -q "$@" > gpurun_out/${tag}_gputests.txt 2>&1; tail -5 gpurun_out/${tag}_gputests.txt
{
for rep in 1 2; do
for v in r3 new kind_all; do
  unset OCTOFITTER_HIP_LIB OCTO_KIND_ALL
  if [ $v = r3 ]; then export OCTOFITTER_HIP_LIB=$R3; fi
  if [ $v = kind_all ]; then export OCTO_KIND_ALL=1; fi
  echo "#### build: $v (round $rep)"
  python tools/latency_multi.py 1 2>&1 | grep "small-batch"
  python tools/latency_hgca_marg.py 2>&1 | grep "W=  1"
  python tools/latency_model_vs_w.py 2>&1 | grep "library default" | head -2
done
done
} > gpurun_out/${tag}_latency_ab.txt 2>&1
cat gpurun_out/${tag}_latency_ab.txt
