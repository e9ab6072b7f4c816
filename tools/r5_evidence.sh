#!/bin/bash
# Round 5's evidence in one gpurun call: rocprofv3 passes of the default bench command and of the strong-scaling shard shapes (tools/profile_round.sh ->
# pmc_traffic.json), counter passes of every other kernel (tools/profile_kernels.sh -> kernel table incl. config 3's row), the -m gpu suite, the
# two-rank gloo dry run of the N > 1 bench line, the other workloads' bench lines, and the randomised sweeps.   bash tools/r5_evidence.sh <tag>
tag=${1:-r5_v2}
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
for w in fwd nuis two_planet logpost ofti pt; do python bench.py --workload $w --steps 100 --no-extras --no-cpu-baseline 2>/dev/null | cut -c1-420; done > gpurun_out/${tag}_workloads.txt
[ -n "${OCTO_SKIP_SWEEPS:-}" ] || {
  python tests/stress_round5.py 400 601 2>&1 | tail -4
  OCTO_TEST_POISON_LDS=nan python tests/stress_round5.py 200 602 2>&1 | tail -4
  export OCTO_TEST_MAX_P=4
  for sb in "" 0; do
    export OCTO_TEST_SMALL_BATCH=$sb
    echo "#### OCTO_TEST_SMALL_BATCH='$sb' max P = 4"
    echo "== stress_parity 600 systems seed 1051"; python tests/stress_parity.py 600 1051 2>&1 | grep -i "worst\|fail\|error\|fault" | tail -4
    echo "== stress_model 300 seed 1052"; python tests/stress_model.py 300 1052 2>&1 | grep -i "worst\|fail\|error\|fault" | tail -3
    echo "== stress_high_e 600 walkers seed 1053"; python tests/stress_high_e.py 600 1053 2>&1 | grep -i "worst\|fail\|error" | tail -3
  done
} > gpurun_out/${tag}_stress_sweeps.txt 2>&1
tail -3 gpurun_out/${tag}_gputests.txt | cut -c1-200
[ -n "${OCTO_SKIP_SWEEPS:-}" ] || cat gpurun_out/${tag}_stress_sweeps.txt | cut -c1-300
