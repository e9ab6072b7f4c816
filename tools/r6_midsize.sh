#!/bin/bash
# Mid-size callbacks: octo_model_logpost at 256-2048 theta_t x 50 / 300 epochs with the one-task launch finishing inside k_main (default)
# and with the k_finish launch (OCTO_FIN_FUSED=0), same box; then a kernel trace of each.   bash tools/r6_midsize.sh <tag>
tag=${1:-r6_midsize}
out=gpurun_out/$tag.txt
: > $out
for ff in 1 0; do
  echo "== OCTO_FIN_FUSED=$ff" >> $out
  OCTO_FIN_FUSED=$ff python tools/r5_midsize_small.py 2>&1 | grep "E=  50" >> $out
done
export TMPDIR=/tmp
for ff in 1 0; do
  (cd /tmp && OCTO_FIN_FUSED=$ff rocprofv3 --kernel-trace --stats -d $PWD/gpurun_out/${tag}_t$ff -o k -- python $OLDPWD/tools/r5_midsize_small.py > /dev/null 2>&1)
  echo "== kernel trace, OCTO_FIN_FUSED=$ff" >> $out
  python profiles/summarize_rocpd.py gpurun_out/${tag}_t$ff gpurun_out/${tag}_t$ff | grep -v "^==" | head -12 >> $out
  rm -rf gpurun_out/${tag}_t$ff gpurun_out/${tag}_t${ff}_rocprof_summary.txt
done
cat $out
