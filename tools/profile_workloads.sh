#!/bin/bash
# Kernel-trace stats of the secondary workloads (one rocprofv3 run each) -> gpurun_out/<tag>_workloads.txt
tag=${1:-r1_v6}
ROOT=$PWD; out=$ROOT/gpurun_out/${tag}_wl; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
: > $ROOT/gpurun_out/${tag}_workloads.txt
for wl in fwd two_planet pt ofti logpost; do
  rocprofv3 --kernel-trace --stats -d $out/$wl -o k -- python $ROOT/bench.py --workload $wl --steps 200 --warmup 50 --no-cpu-baseline > $out/$wl.log 2>&1
  echo "=== bench.py --workload $wl" >> $ROOT/gpurun_out/${tag}_workloads.txt
  grep "^{" $out/$wl.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%s\n  value %.4g %s, %.1f us/step' % (d['config']['workload'], d['value'], d['unit'], d['ms_per_step']*1e3))" >> $ROOT/gpurun_out/${tag}_workloads.txt
  (cd $ROOT && python profiles/summarize_rocpd.py gpurun_out/${tag}_wl/$wl /tmp/x_$wl | grep -v "^==" | grep -v copyBuffer >> gpurun_out/${tag}_workloads.txt)
done
[ -n "${OCTO_KEEP_DB:-}" ] || rm -rf $out
cat $ROOT/gpurun_out/${tag}_workloads.txt
