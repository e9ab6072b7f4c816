#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun): kernel-trace stats of the default bench
# command, then PMC passes (each in its own run, never combined with trace domains other than --kernel-trace).
# Usage: bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>/..., summary in gpurun_out/<tag>_rocprof_summary.txt
set -u
tag=${1:-r2_v0}
ROOT=$PWD
out=$ROOT/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras"
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/stats -o k -- python $ROOT/bench.py > $out/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $out/pmc_fetch -o k -- $B > $out/b_fetch.log 2>&1
rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $out/pmc_write -o k -- $B > $out/b_write.log 2>&1
rocprofv3 --kernel-trace --stats --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 -d $out/pmc_mix -o k -- $B > $out/b_mix.log 2>&1
rocprofv3 --kernel-trace --stats --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $out/pmc_sq -o k -- $B > $out/b_sq.log 2>&1
# the per-GPU launch shapes of a strong-scaled run (1e4 walkers over 2 / 4 / 8 GPUs): counter passes of the same command at --walkers W
MIX="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"
for Wn in 5000 2500 1250; do
  Bn="$B --steps 100 --walkers $Wn"
  rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $out/w$Wn/pmc_fetch -o k -- $Bn > $out/b_w${Wn}_fetch.log 2>&1
  rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $out/w$Wn/pmc_write -o k -- $Bn > $out/b_w${Wn}_write.log 2>&1
  rocprofv3 --kernel-trace --stats --pmc $MIX -d $out/w$Wn/pmc_mix -o k -- $Bn > $out/b_w${Wn}_mix.log 2>&1
done
cd $ROOT
python profiles/summarize_rocpd.py gpurun_out/$tag gpurun_out/$tag > /dev/null
python tools/make_pmc_json.py gpurun_out/$tag gpurun_out/${tag}_pmc_traffic.json > /dev/null
mkdir -p gpurun_out/${tag}_logs && cp $out/*.log gpurun_out/${tag}_logs/
[ -n "${OCTO_KEEP_DB:-}" ] || rm -rf $out      # the rocpd databases (~45 MB): gpurun merges at most 64 MiB back; the summaries above are what is committed
tail -1 gpurun_out/${tag}_logs/bench_under_rocprof.log | cut -c1-600
