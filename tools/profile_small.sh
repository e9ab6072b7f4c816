#!/bin/bash
# rocprofv3 kernel-trace of the small-batch path, one configuration per run (the kernel name is the same for all of them).
set -u
tag=${1:-r2_small}
ROOT=$PWD; out=$ROOT/gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
cd /tmp
for cfg in "50 1" "10000 1" "50 32" "10000 32"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats -d $out/E$1_W$2 -o k -- python $ROOT/tools/small_batch_one.py $1 $2 > $out/E$1_W$2.log 2>&1
  tail -1 $out/E$1_W$2.log
done
cd $ROOT
python profiles/summarize_rocpd.py gpurun_out/$tag gpurun_out/$tag | grep -E "^==|k_small"
