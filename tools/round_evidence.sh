#!/bin/bash
# Everything the round's record needs for ONE source hash, in one gpurun call:  bash tools/round_evidence.sh <tag>
#   -m gpu suite -> gpurun_out/<tag>_gputests.txt; rocprofv3 passes of the default bench (kernel trace + PMC) -> <tag>_rocprof_summary.txt,
#   <tag>_pmc_traffic.json; counter passes of every other kernel -> <tag>_kernel_table.md; bench lines of the default command (20 and 200 steps)
#   and of the logpost / two_planet / fwd workloads -> <tag>_bench*.json.
tag=${1:-r4_v0}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --deselect tests/test_configs_gpu.py::test_bench_line_contract > gpurun_out/${tag}_gputests_pre.txt 2>&1
bash tools/profile_round.sh $tag > gpurun_out/${tag}_profile_tail.txt 2>&1
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic.json      # (on the box: the contract test below and bench.py read it)
bash tools/profile_kernels.sh $tag > gpurun_out/${tag}_kernels_tail.txt 2>&1
python -m pytest tests -m gpu -q > gpurun_out/${tag}_gputests.txt 2>&1; tail -3 gpurun_out/${tag}_gputests.txt
python bench.py 2>gpurun_out/${tag}_bench20.err | tail -1 > gpurun_out/${tag}_bench20.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench200.json
for wl in logpost two_planet fwd; do
  python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_$wl.json
done
for f in bench20 bench200 bench_logpost bench_two_planet bench_fwd; do
  python -c "
import json; d=json.load(open('gpurun_out/${tag}_$f.json')); print('$f', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('value_pcie_inclusive'), d.get('value_pcie_inclusive_blocking_call'), d.get('value_pcie_inclusive_pipelined'), (d.get('strong_scaling_projection') or {}).get('by_n_gpus',{}).get('8'))"
done
