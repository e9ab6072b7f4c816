#!/bin/bash
# Counter evidence for every kernel of DESIGN.md §3's table that tools/profile_round.sh does not cover (it profiles config 3's
# k_main<1, true, false, 1>): three rocprofv3 passes per workload — FETCH_SIZE, WRITE_SIZE, instruction mix — each with
# --kernel-trace --stats only (gpurun refuses --pmc combined with other trace domains), then tools/make_kernel_table.py.
# Usage (through gpurun): bash tools/profile_kernels.sh <tag>   -> gpurun_out/<tag>_kernels/<workload>/{fetch,write,mix}/*.db,
#                                                                  gpurun_out/<tag>_kernel_table.md
set -u
tag=${1:-r3_v0}
ROOT=$PWD
out=$ROOT/gpurun_out/${tag}_kernels
mkdir -p "$out"
export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras"
MIX="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"
cd /tmp
run() {   # run <name> <command...>
  local name=$1; shift
  mkdir -p $out/$name
  rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $out/$name/fetch -o k -- "$@" > $out/$name/fetch.log 2>&1
  rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $out/$name/write -o k -- "$@" > $out/$name/write.log 2>&1
  rocprofv3 --kernel-trace --stats --pmc $MIX -d $out/$name/mix -o k -- "$@" > $out/$name/mix.log 2>&1
}
run two_planet $B --workload two_planet
run nuis $B --workload nuis
run fwd $B --workload fwd
run wide_prior $B --workload wide_prior
run rv_gappy $B --workload rv_gappy
run rv_gappy_nuis $B --workload rv_gappy_nuis
run ofti $B --workload ofti
run logpost $B --workload logpost
run three_planet python $ROOT/tools/multi_planet_steps.py 3 60
run four_planet python $ROOT/tools/multi_planet_steps.py 4 60
run five_planet python $ROOT/tools/multi_planet_steps.py 5 40
run eight_planet python $ROOT/tools/multi_planet_steps.py 8 30
run shard_1250 $B --steps 100 --walkers 1250
run shard_2500 $B --steps 100 --walkers 2500
run small_w1 python $ROOT/tools/small_batch_one.py 10000 1 600
run small_w512 python $ROOT/tools/small_batch_one.py 10000 512 300
cd $ROOT
python tools/make_kernel_table.py gpurun_out/${tag}_kernels gpurun_out/${tag}_kernel_table.md
python profiles/summarize_rocpd.py gpurun_out/${tag}_kernels gpurun_out/${tag}_kernels > /dev/null
[ -n "${OCTO_KEEP_DB:-}" ] || rm -rf $out
