#!/bin/bash
# Kernel trace of the one-θ call beyond four planets (tools/latency_multi.py: five RA/Dec tables of 60 rows + 200 RV rows, W = 1): which launches the
# 59 µs are made of.   bash tools/r6_manyp_latency_trace.sh
ROOT=$PWD; out=$ROOT/gpurun_out/r6_manyp_trace; mkdir -p $out
python tools/latency_multi.py 1 5 4,5,8 2>&1 | grep "fwd+grad" > $out/latency.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $out/stats -o k -- python $ROOT/tools/latency_multi.py 1 5 5 > $out/run.log 2>&1
cd $ROOT
python profiles/summarize_rocpd.py gpurun_out/r6_manyp_trace gpurun_out/r6_manyp_trace | head -12 > $out/kernels.txt
cat $out/latency.txt $out/kernels.txt
rm -rf $out/stats
