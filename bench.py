#!/usr/bin/env python
"""
bench.py — throughput of the epoch-loop likelihood hot path on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path (k_setup -> k_main -> k_finish through octo_eval_device) over one
batch of synthetic input that is already resident in HBM: BASELINE.json config 3 — 1 planet, 1e4 RA/Dec
epochs x 1e4 prior-drawn walkers, forward log-likelihood + reverse gradient w.r.t. the orbital elements.
Metric: epoch-likelihood evaluations per second = walkers x rows x steps / wall time, whole job.
Multi-GPU: walkers are independent -> each rank owns its own 1e4 walkers (weak scaling), the dataset is
replicated, and there is NO collective on the data path (the only collective of the path is the
parallel-tempering swap step, exercised with --workload pt).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      HBM bound as contracted by north_star: achieved = algorithmic bytes per launch
                (40 B per RA/Dec row per walker + 136 B per walker) / average duration of the
                dominant kernel k_main measured with HIP events on its launch stream.
  cpu_baseline  the CPU restatement (oracle/, kind "port") timed on this host's cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

TIMED_EVERY = 8   # k_main is bracketed with HIP events on every 8th step of the timed region (an event pair costs ~µs of stream time)

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
BYTES_PER_ROW = 40.0          # epoch, ra, dec, σ_ra, σ_dec  
BYTES_PER_WALKER = 64.0 + 72.0  # read 8 elements, write ll + 8 adjoints


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--epochs", type=int, default=10_000)
    ap.add_argument("--walkers", type=int, default=None, help="walkers per GPU (default 10000; 8192 = 8 temperatures x 1024 for --workload pt)")
    ap.add_argument("--workload", choices=["grad", "fwd", "two_planet", "pt", "ofti", "logpost"], default="grad")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for single-GPU dry runs")
    ap.add_argument("--device", type=int, default=None, help="force the HIP device index (dry runs of the N>1 path on one GPU)")
    args = ap.parse_args()
    if args.walkers is None:      # BASELINE config 5: 64 temperatures x 1024 walkers over 8 GPUs = 8 x 1024 per GPU
        args.walkers = 8192 if args.workload == "pt" else 10_000
    return args


def cpu_baseline(cfg, obs_tables, planets, seconds):
    """Oracle (reference-order restatement, forward-mode duals = ForwardDiff chunk) on all host cores, on a bounded
    sample of the same workload: whole passes over a subset of the walkers until ~`seconds` of CPU work."""
    import oracle_binding as ob
    import synth
    cores = os.cpu_count() or 1
    E = cfg["n_epochs"]
    mask = synth.active_mask(1, 1, mass=False, nuis=False)
    probe = min(cfg["n_walkers"], 4 * cores)
    ob.oracle_eval(obs_tables, planets, cfg["elems"][:, :probe], None, grad=True, active=mask, n_threads=0)   # warm
    t0 = time.perf_counter()
    ob.oracle_eval(obs_tables, planets, cfg["elems"][:, :probe], None, grad=True, active=mask, n_threads=0)
    rate = probe * E / max(time.perf_counter() - t0, 1e-6)
    n = int(min(cfg["n_walkers"], max(cores, rate * seconds / E)))
    n = max(cores, n // cores * cores)
    reps = max(1, int(round(rate * seconds / (n * E))))
    t0 = time.perf_counter()
    for _ in range(reps):
        ob.oracle_eval(obs_tables, planets, cfg["elems"][:, :n], None, grad=True, active=mask, n_threads=0)
    dt = time.perf_counter() - t0
    return {"value": reps * n * E / dt, "unit": "epoch-likelihood evals/s (fwd+grad)", "cores": cores, "kind": "port",
            "sample": f"{reps} pass(es) over {n} walkers x {E} epochs of the same workload, oracle/liboctooracle.so "
                      f"(gcc -O3 -march=native, OpenMP over walkers, 8 forward-mode partials per dual), {dt:.1f} s"}


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes on this driver
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dev_index = local_rank if args.device is None else args.device
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from __graft_entry__ import load_package
    import synth
    pkg = load_package()
    capi = pkg.capi

    grad = args.workload in ("grad", "two_planet")
    if args.workload == "ofti":
        # batched ofti_linear_solve (src/parameterizations.jl:318-405), forward marginal likelihood
        cfg0 = synth.config_astrom(n_epochs=args.epochs, n_walkers=args.walkers, cfg=6)
        t = cfg0["table"]
        solver = pkg.OftiLinearSolver(t["epoch"], t["ra"], t["dec"], t["σ_ra"], t["σ_dec"], None, 1000.0, device=dev_index)
        el = cfg0["elems"]
        nl = torch.tensor(np.stack([el[1], el[0], el[5], el[6], el[7]]), device=dev)
        for _ in range(args.warmup):
            solver.eval_device(nl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            solver.eval_device(nl)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({"metric": "OFTI marginal-likelihood epoch evals/sec (fwd)", "value": args.epochs * args.walkers * args.steps / dt,
                              "unit": "evals/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": f"ofti_linear_solve: {args.epochs} RA/Dec epochs x {args.walkers} walkers, forward"}}))
        solver.close()
        return
    if args.workload == "logpost":
        # the whole ∇ℓπcallback (src/logdensitymodel.jl:169-177) on the device for the D = 11 model of the
        # reference's tests: θ_t -> invlink -> priors -> elements -> likelihood + gradient -> chain rule back to θ_t
        cfg0 = synth.config_astrom(n_epochs=args.epochs, n_walkers=args.walkers, cfg=3)
        astrom = pkg.PlanetRelAstromObs(cfg0["table"], name="astrom")
        b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[astrom],
                       variables=pkg.variables(a=pkg.LogUniform(1, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                               Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
        sysm = pkg.System(name="bench", companions=[b], variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1),
                                                                              plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1)))
        model = pkg.LogDensityModel(sysm, device=dev_index)
        θt = torch.tensor(model.link(model.sample_priors(np.random.default_rng(20260929 + 7), args.walkers)), device=dev)
        for _ in range(args.warmup):
            model.logpost_device(θt, grad=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model.logpost_device(θt, grad=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({"metric": "epoch-likelihood evals/sec, full log-posterior + gradient w.r.t. theta_t (D=11)",
                              "value": args.epochs * args.walkers * args.steps / dt, "unit": "evals/s", "n_gpus": 1, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": f"LogDensityModel D={model.D}: {args.epochs} RA/Dec epochs x {args.walkers} walkers, theta_t resident in HBM"}}))
        model.close()
        return
    if args.workload == "two_planet":
        c4 = synth.config_two_planet()
        astrom = pkg.PlanetRelAstromObs(c4["astrom"], name="astrom")
        rv = pkg.StarAbsoluteRVObs(c4["rv"], name="rv")
        b = pkg.Planet(name="b", observations=[])
        c = pkg.Planet(name="c", observations=[astrom])
        system = pkg.System(name="cfg4", companions=[b, c], observations=[rv])
        θex = dict(M=1.2, plx=50.0, planets=dict(b=dict(a=3, e=0.1, i=1, ω=1, Ω=2, tp=5e4, mass=5), c=dict(a=15, e=0.3, i=1, ω=.5, Ω=2, tp=5e4, mass=10)))
        fn = pkg.make_ln_like(system, θex, device=dev_index)
        # nuisance rows follow the evaluation order: planet observations first, then system observations
        elems_h, nuis_h = c4["elems"], c4["nuis"]
        n_rows, W = c4["n_rows"], c4["n_walkers"]
        workload = "config4: 2 planets, 2500 RA/Dec + 2500 abs-RV epochs x 4096 walkers, fwd+grad"
        bytes_per_launch = W * (2500 * 40.0 + 2500 * 24.0) + W * 8.0 * (18 + 6 + 1 + 18 + 6)
        cfg = None
    else:
        cfg = synth.config_astrom(n_epochs=args.epochs, n_walkers=args.walkers, cfg=3 if grad else 2,
                                  seed=None if world == 1 else 20260929 + 3 + 1000 * rank)
        obs, planet = synth.to_mirror(pkg, cfg)
        system = pkg.System(name="bench", companions=[planet], observations=[])
        fn = pkg.make_ln_like(system, cfg["theta_example"], device=dev_index)
        elems_h, nuis_h = cfg["elems"], None
        n_rows, W = cfg["n_epochs"], cfg["n_walkers"]
        workload = f"config{'3' if grad else '2'}: 1 planet, {n_rows} RA/Dec epochs x {W} walkers/GPU, {'fwd+reverse-grad' if grad else 'fwd only'}"
        bytes_per_launch = W * n_rows * BYTES_PER_ROW + W * (BYTES_PER_WALKER if grad else 72.0)

    elems = torch.tensor(elems_h, device=dev)
    nuis = torch.tensor(nuis_h, device=dev) if nuis_h is not None else None
    out = (torch.empty(W, dtype=torch.float64, device=dev),
           torch.empty_like(elems) if grad else None,
           torch.empty_like(nuis) if (grad and nuis is not None) else None)

    pt = None
    if args.workload == "pt":
        # config 5: temperatures sharded over ranks; every step = fwd+grad-free explorer evaluation, RCCL
        # all_gather of the per-replica log-likelihoods, deterministic neighbour swap of β labels.
        from octofitter_jl_amd.host.tempering import TemperedSwap
        n_temps_total = 8 * world
        chains = W // 8
        pt = TemperedSwap(fn, n_temps_total=n_temps_total, n_chains=chains, rank=rank, world=world, device=dev, seed=20260929)
        grad = False
        out = (out[0], None, None)
        workload = f"config5: {n_temps_total} temperatures x {chains} walkers x {n_rows} epochs, sharded by temperature, RCCL all_gather swap"

    def run_step(i):
        if grad:
            fn.ln_like_device(elems, nuis, grad=True, out=out)
        else:
            fn.ln_like_device(elems, nuis, grad=False, out=(out[0], None, None))
        if pt is not None:
            pt.swap_step(out[0], i)

    for i in range(args.warmup):
        run_step(i)
    torch.cuda.synchronize()
    fn.timing_enable(TIMED_EVERY)      # HIP events around k_main of every TIMED_EVERY-th step of the timed region
    fn.timing_read(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    kern_ms, kern_n = fn.timing_read(reset=True)
    fn.timing_enable(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    evals = float(W) * n_rows * args.steps * world
    value = evals / dt
    res = {
        "metric": "epoch-likelihood evals/sec (fwd+grad), 1e4 epochs x 1e4 walkers" if args.workload == "grad" else f"epoch-likelihood evals/sec ({args.workload})",
        "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "walkers_per_gpu": W, "rows": n_rows, "parallelism": f"walkers sharded x{world}, dataset replicated, no data-path collective"},
    }
    if rank == 0:
        ach = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
        traffic, flops_per_eval, issue = None, None, None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists() and args.workload == "grad" and (n_rows, W) == (10_000, 10_000):
            try:      # PMC figures are per launch of exactly this workload; measured off-line (separate --pmc passes)
                j = json.loads(pmc.read_text())
                traffic, flops_per_eval = j.get("hbm_bytes_per_launch"), j.get("fp64_flops_per_eval")
                issue = j.get("issue_model")
            except Exception:
                traffic = None
        res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": (ach / HBM_PEAK_GBPS) if ach else None, "traffic": traffic,
                           "kernel": "k_main", "kernel_avg_ms": kern_ms, "kernel_launches_timed": kern_n,
                           "algorithmic_bytes_per_launch": bytes_per_launch,
                           "note": "algorithmic bytes (SURVEY 8d) over the live k_main duration, as north_star contracts; rows are "
                                   "reused across 64 lanes from the scalar cache, so real HBM traffic is `traffic` bytes per launch "
                                   "and frac may exceed 1. The kernel's true bound is FP64 VALU issue (see `valu`, DESIGN.md)"}
        if flops_per_eval and kern_ms > 0:
            tf = flops_per_eval * float(W) * n_rows / (kern_ms * 1e-3) / 1e12
            res["roofline"]["valu"] = {"bound": "fp64_vector", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
                                       "fp64_flops_per_eval": flops_per_eval,
                                       "note": "FP64 flops only (FMA = 2); VALU issue slots are ~100 % busy at the sustained ~2.1 GHz clock"}
            if issue:
                # time the chip needs just to ISSUE this kernel's VALU instructions (measured mix x measured cost per class)
                t_issue = issue["ns_per_row_per_wave"] * 1e-6 * n_rows * ((W + 63) // 64) / issue["simds"]
                res["roofline"]["valu"].update({"issue_bound_ms": t_issue, "issue_frac": t_issue / kern_ms})
        if not args.no_cpu_baseline and cfg is not None and world == 1 and args.workload in ("grad",):
            try:
                res["cpu_baseline"] = cpu_baseline(cfg, fn.obs_tables, fn.planet_desc, args.cpu_seconds)
            except Exception as ex:  # the checker is optional for the measurement itself
                res["cpu_baseline"] = {"value": None, "unit": "evals/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {ex}"}
        print(json.dumps(res), flush=True)
    fn.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
