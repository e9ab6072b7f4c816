#!/bin/bash
# Same-box A/B of the round-6 library against the round-5 one (lib/variants/liboctofitter_hip_r5.so, built from the round-5 tree) on config 3 and
# the non-uniform workloads of VERDICT r5 item 1.   bash tools/r6_ab_workloads.sh <tag> [workloads...]
tag=${1:-r6_ab}; shift
wls=${@:-grad wide_prior rv_gappy rv_gappy_nuis}
out=gpurun_out/${tag}.txt
: > $out
for rep in 1 2; do
for wl in $wls; do
  for lib in r5 default; do
    if [ $lib = r5 ]; then export OCTOFITTER_HIP_LIB=$PWD/octofitter.jl_amd/lib/variants/liboctofitter_hip_r5.so; else unset OCTOFITTER_HIP_LIB; fi
    python bench.py --workload $wl --steps 100 --warmup 10 --no-extras --no-cpu-baseline 2>gpurun_out/${tag}_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib'.rjust(8), '$wl'.ljust(14), 'ms/step %.4f  median %.4f  k_main %.4f  evals/s %.3e' % (r['ms_per_step'], r['ms_per_step_median_events'], r['roofline']['kernel_avg_ms'], r['value']))
" >> $out || tail -3 gpurun_out/${tag}_err.txt >> $out
  done
done
done
cat $out
