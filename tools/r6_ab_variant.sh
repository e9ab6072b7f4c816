#!/bin/bash
# Same-box A/B of library variants (lib/variants/liboctofitter_hip_<v>.so; `default` = the tree's library) on bench workloads, two interleaved rounds.
#   bash tools/r6_ab_variant.sh <tag> "<variants>" "<workloads>" [extra bench args]
tag=$1; vars=$2; wls=$3; shift 3
out=gpurun_out/${tag}.txt
: > $out
for rep in 1 2; do
for wl in $wls; do
  for v in $vars; do
    if [ $v = default ]; then unset OCTOFITTER_HIP_LIB; else export OCTOFITTER_HIP_LIB=$PWD/octofitter.jl_amd/lib/variants/liboctofitter_hip_$v.so; fi
    python bench.py --workload $wl --steps 100 --warmup 10 --no-extras --no-cpu-baseline "$@" 2>gpurun_out/${tag}_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v'.rjust(10), '$wl'.ljust(14), 'ms/step %.4f  median %.4f  k_main %.4f  evals/s %.3e' % (r['ms_per_step'], r['ms_per_step_median_events'], r['roofline']['kernel_avg_ms'], r['value']), r.get('tile_sort',{}).get('on'))
" >> $out || tail -3 gpurun_out/${tag}_err.txt >> $out
  done
done
done
cat $out
