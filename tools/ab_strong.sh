#!/bin/bash
# Same-box A/B of library builds on the strong-scaling shares (tools/strong_probe.py): bash tools/ab_strong.sh <variant-tag|default> ...
for v in "$@"; do
  if [ "$v" = default ]; then unset OCTOFITTER_HIP_LIB; else export OCTOFITTER_HIP_LIB=$PWD/octofitter.jl_amd/lib/variants/liboctofitter_hip_$v.so; fi
  OCTO_PROBE_N=${OCTO_PROBE_N:-1,4,8} python tools/strong_probe.py 2>/dev/null | grep "grad=1"
done
