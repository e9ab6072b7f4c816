#!/bin/bash
# Same-box A/B: the warm solve as a triangle (+ the odd-row exit hinted unlikely) — this tree — against the round's earlier build (lib/variants/…_head.so: diamond).
[ "$1" = strong ] || bash tools/r6_ab_variant.sh r6_tri2_ab "head default" "grad nuis two_planet rv_gappy wide_prior fwd"
for r in 1 2; do OCTO_PROBE_N=1,4,8 bash tools/ab_strong.sh head default; done > gpurun_out/r6_tri2_strong.txt 2>&1
cat gpurun_out/r6_tri2_strong.txt
