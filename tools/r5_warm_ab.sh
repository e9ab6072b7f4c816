#!/bin/bash
# Same-box A/B of the warm-started row loop (default build) against the cold-only build (-DOCTO_WARM=0: lib/variants/liboctofitter_hip_cold.so),
# two interleaved rounds, then the whole -m gpu suite.   bash tools/r5_warm_ab.sh <tag>
tag=${1:-r5_warm}
V=$PWD/octofitter.jl_amd/lib/variants/liboctofitter_hip_cold.so
python -m pytest tests/test_warm_start.py -q -x -m gpu > gpurun_out/${tag}_warmtests.txt 2>&1
tail -3 gpurun_out/${tag}_warmtests.txt
{
for r in 1 2; do
  OCTOFITTER_HIP_LIB=$V python tools/ab.py 2>&1 | grep "us/step"
  python tools/ab.py 2>&1 | grep "us/step"
done
} > gpurun_out/${tag}_ab.txt
python -m pytest tests -q -m gpu -k "not bench" > gpurun_out/${tag}_gputests.txt 2>&1
tail -3 gpurun_out/${tag}_gputests.txt
cat gpurun_out/${tag}_ab.txt
