#!/bin/bash
# Same-box A/B of the throughput path, round-3 library against the current build: the strong-scaling shares (tools/strong_probe.py),
# the batch sizes of tools/ab.py and the bench workloads.   bash tools/r4_ab_throughput.sh <tag>   (through gpurun)
tag=${1:-r4_ab}
R3=$PWD/octofitter.jl_amd/lib/variants/liboctofitter_hip_r3.so
{
for rep in 1 2; do
for v in r3 new; do
  unset OCTOFITTER_HIP_LIB
  if [ $v = r3 ]; then export OCTOFITTER_HIP_LIB=$R3; fi
  echo "#### build: $v (round $rep)"
  OCTO_PROBE_N=1,2,4,8 python tools/strong_probe.py 2>/dev/null | grep "grad=1"
  python tools/ab.py 2>/dev/null
  for wl in two_planet logpost nuis; do
    python bench.py --workload $wl --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-6s %-10s %.4e evals/s  %.4f ms/step' % ('$v', '$wl', d['value'], d['ms_per_step']))"
  done
done
done
} > gpurun_out/${tag}_throughput_ab.txt 2>&1
cat gpurun_out/${tag}_throughput_ab.txt
