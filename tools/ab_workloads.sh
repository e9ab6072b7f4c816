#!/bin/bash
# Same-box A/B of library builds over the bench workloads: bash tools/ab_workloads.sh <variant-tag|default> ...   (run through gpurun)
for v in "$@"; do
  if [ "$v" = default ]; then unset OCTOFITTER_HIP_LIB; else export OCTOFITTER_HIP_LIB=$PWD/octofitter.jl_amd/lib/variants/liboctofitter_hip_$v.so; fi
  for wl in grad fwd two_planet ofti logpost; do
    python bench.py --workload $wl --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-12s %-10s %.4e evals/s  %.4f ms/step' % ('$v', '$wl', d['value'], d['ms_per_step']))"
  done
done
