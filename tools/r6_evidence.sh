#!/bin/bash
# Round 6's evidence in one gpurun call (after tools/r5_evidence.sh): rocprofv3 passes of the default bench command and of the strong-scaling shard shapes
# (tools/profile_round.sh -> pmc_traffic.json), counter passes of every other kernel incl. the non-uniform workloads (tools/profile_kernels.sh -> kernel
# table), the -m gpu suite, the driver-comparable bench line, the two-rank gloo dry run, every workload's bench line on this build AND on the round-5
# library (same box: before / after), the latency family, the strong-scaling shares, and the randomised sweeps.   bash tools/r6_evidence.sh <tag>
tag=${1:-r6_v7}
export OCTO_KEEP_DB=1
bash tools/profile_round.sh $tag > gpurun_out/${tag}_profile.log 2>&1
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic.json
bash tools/profile_kernels.sh $tag > gpurun_out/${tag}_kernels.log 2>&1
python tools/make_kernel_table.py gpurun_out/${tag}_kernels gpurun_out/${tag}_kernel_table.md gpurun_out/$tag > /dev/null 2>&1
rm -rf gpurun_out/$tag gpurun_out/${tag}_kernels
unset OCTO_KEEP_DB
python -m pytest tests -q -m gpu > gpurun_out/${tag}_gputests.txt 2>&1
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python bench.py --gpus 2 --backend gloo --device 0 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_2rank_gloo_dryrun.json 2> gpurun_out/${tag}_2rank.err
{
  for lib in r5 default; do
    if [ $lib = r5 ]; then export OCTOFITTER_HIP_LIB=$PWD/octofitter.jl_amd/lib/variants/liboctofitter_hip_r5.so; else unset OCTOFITTER_HIP_LIB; fi
    echo "== library: $lib"
    for w in grad fwd nuis wide_prior rv_gappy rv_gappy_nuis two_planet logpost ofti pt; do python bench.py --workload $w --steps 100 --no-extras --no-cpu-baseline 2>/dev/null | cut -c1-520; done
  done
  unset OCTOFITTER_HIP_LIB
} > gpurun_out/${tag}_workloads.txt
{
  echo "# one parameter set per call and mid-size batches: tools/latency_w1.py, tools/r5_midsize_small.py"
  python tools/latency_w1.py 2>/dev/null | tail -12
  python tools/r5_midsize_small.py 2>/dev/null
  echo "# the same mid-size callbacks with the k_finish launch (OCTO_FIN_FUSED=0)"
  OCTO_FIN_FUSED=0 python tools/r5_midsize_small.py 2>/dev/null | grep "E=  50"
  echo "# strong-scaling shares (tools/strong_probe.py)"
  python tools/strong_probe.py 2>/dev/null | grep "grad=1"
  echo "# many planets (tools/multi_planet_steps.py)"
  for P in 2 3 4 5 6 8; do python tools/multi_planet_steps.py $P 100 2>/dev/null | tail -1; done
  echo "# config 4 and the last-planet-always-warm loop (tools/r6_cfg4_probe.py)"
  python tools/r6_cfg4_probe.py 2>/dev/null | grep "us per step"
} > gpurun_out/${tag}_latency.txt 2>&1
[ -n "${OCTO_SKIP_SWEEPS:-}" ] || {
  python tests/stress_round6.py 150 611 2>&1 | tail -4
  OCTO_TEST_POISON_LDS=nan python tests/stress_round6.py 60 612 2>&1 | tail -4
  python tests/stress_round5.py 200 601 2>&1 | tail -4
  export OCTO_TEST_MAX_P=4
  for sb in "" 0; do
    export OCTO_TEST_SMALL_BATCH=$sb
    echo "#### OCTO_TEST_SMALL_BATCH='$sb' max P = 4"
    echo "== stress_parity 400 systems seed 1061"; python tests/stress_parity.py 400 1061 2>&1 | grep -i "worst\|fail\|error\|fault" | tail -4
    echo "== stress_model 200 seed 1062"; python tests/stress_model.py 200 1062 2>&1 | grep -i "worst\|fail\|error\|fault" | tail -3
    echo "== stress_high_e 400 walkers seed 1063"; python tests/stress_high_e.py 400 1063 2>&1 | grep -i "worst\|fail\|error" | tail -3
  done
} > gpurun_out/${tag}_stress_sweeps.txt 2>&1
tail -3 gpurun_out/${tag}_gputests.txt | cut -c1-200
[ -n "${OCTO_SKIP_SWEEPS:-}" ] || cat gpurun_out/${tag}_stress_sweeps.txt | cut -c1-300
