#!/bin/bash
# Same-box A/B: the warm-started row loop also in the two-planet kernels and in the nuisance kernels with sep/PA or RV rows (plain scalar row loads there):
# -DOCTO_WARM_P=2 -DOCTO_WARM_PLAIN=1 (lib/variants/liboctofitter_hip_warm2.so) against the default build.   bash tools/r5_warm2_ab.sh <tag>
tag=${1:-r5_warm2}
V=$PWD/octofitter.jl_amd/lib/variants/liboctofitter_hip_warm2.so
OCTOFITTER_HIP_LIB=$V python -m pytest tests/test_warm_start.py tests/test_gpu_parity.py tests/test_configs_gpu.py -q -x -m gpu -k "not bench" > gpurun_out/${tag}_tests.txt 2>&1
tail -3 gpurun_out/${tag}_tests.txt
{
for r in 1 2; do
  for v in default warm2; do
    if [ "$v" = default ]; then unset OCTOFITTER_HIP_LIB; else export OCTOFITTER_HIP_LIB=$V; fi
    for wl in two_planet nuis fwd; do
      python bench.py --workload $wl --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-12s %-10s %.4e evals/s  %.4f ms/step' % ('$v', '$wl', d['value'], d['ms_per_step']))"
    done
  done
done
} > gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt
