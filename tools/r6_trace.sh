#!/bin/bash
# rocprofv3 kernel trace (no counters) of one bench workload: per-kernel average durations.   bash tools/r6_trace.sh <tag> <bench args...>
tag=$1; shift
ROOT=$PWD
out=$ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/stats -o k -- python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras "$@" > $out/bench.log 2>&1
cd $ROOT
python profiles/summarize_rocpd.py gpurun_out/$tag gpurun_out/$tag | head -30
tail -1 $out/bench.log | cut -c1-300
rm -rf $out/stats
