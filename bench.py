#!/usr/bin/env python
"""
bench.py — throughput of the epoch-loop likelihood hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under
                                                          python -m torch.distributed.run --nproc-per-node N)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path (k_main with the orbit constructors in its prologue -> k_finish, through octo_eval_device) over
one batch of synthetic input that is already resident in HBM: BASELINE.json config 3 — 1 planet, 1e4 RA/Dec epochs x 1e4 prior-drawn
walkers, forward log-likelihood + reverse gradient w.r.t. the orbital elements. Metric: epoch-likelihood evaluations per second =
walkers x rows x steps / wall time, whole job. `value` is that HBM-resident rate (the bench contract); SURVEY.md §8(d) defines the
metric with the H2D of the elements and the D2H of ll + gradient inside the timed call, on the DEVICE's clock (hipEvent): that rate travels on
the same line as `value_pcie_inclusive` (octo_eval on host arrays the caller registered once, events around the whole call inside the library),
next to `value_pcie_inclusive_blocking_call` (the same calls by the host's clock) and the breakdown in `pcie_inclusive`.
Multi-GPU: walkers are independent, the dataset is replicated, and there is NO collective on the data path (the only collective of the
path is the parallel-tempering swap step, exercised with --workload pt).
  --scaling strong  (default for the contracted workloads grad / fwd / nuis) SURVEY §8(d) "Scaling runs": the SAME 1e4 walkers split evenly
                    over the ranks (1 250 per GPU at 8), so that the metric's "1e4 epochs x 1e4 walkers" is literally what ran at every N;
  --scaling weak    every rank owns its own --walkers (the default of the per-GPU-shaped workloads pt / two_planet / ofti / logpost).
At N = 1 the line also carries `strong_scaling_projection`: the per-GPU shares of a strong-scaled run (W/2, W/4, W/8 walkers) measured
on this one GPU — shards are independent, so N x rate(W/N) / rate(W) is what N GPUs deliver short of launch jitter. At N > 1 the default
(strong) run also MEASURES the weak-scaling point (`weak_scaling_measured`: every rank its own 1e4 walkers, max over ranks), and a
`--scaling weak` run the strong one (`strong_scaling_measured`). Every N > 1 line carries `cpu_baseline` (rank 0, the full 1e4-walker
workload) and a `roofline` for the per-GPU launch shape (profiles/pmc_traffic.json holds counter passes per shard size).

Timed region: W warm-up steps, then an untimed spin-up until the device has been busy for >= 0.3 s (clocks ramped, so that a
20-step run measures the same thing as a 200-step run), barrier + synchronize, EXACTLY K steps, synchronize + barrier, MAX over
ranks. Beside the wall-clock value the line carries the median per-step time from HIP events on the launch stream (one event
after every 4th step: group means).

Rank 0 prints ONE JSON line (contract in the task statement) with these extra objects:
  roofline      the bound that BINDS the dominant kernel k_main: FP64 vector issue. achieved = FP64 flops per evaluation (from
                the committed PMC instruction counts of exactly this kernel source, profiles/pmc_traffic.json) x evaluations per
                launch / the kernel's average duration measured live with HIP events on its launch stream; peak = 78.6 TFLOP/s
                (MI355X FP64 vector). `hbm`: the real HBM rate (PMC FETCH/WRITE bytes per launch / the same live duration)
                against 8 TB/s — 1-2 % — and, clearly named, the north_star's ALGORITHMIC streaming figure (40 B per row per
                walker), which exceeds the HBM peak because rows are served to 64 lanes from the scalar cache.
  pcie_inclusive the same batch through octo_eval with HOST buffers (H2D of elems, D2H of ll and gradient inside the timed
                region): SURVEY.md §8(d)'s definition of the metric; never `value`.
  parity        max relative error of ll and gradient on 8 walkers of the batch just timed, against the oracle.
  cpu_baseline  the CPU restatement (oracle/, kind "port") timed on this host's cores on a bounded sample.
  config1       BASELINE config 1 (D = 11 model, 50 epochs, one θ per call): µs per call of octo_model_logpost (value +
                gradient) next to the single-thread CPU restatement's µs per call.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

TIMED_EVERY = 4   # k_main is bracketed with HIP events on every 4th step of the timed region (an event pair costs ~µs of stream time)
SPINUP_SECONDS = 0.3
STEP_EVENTS_EVERY = 4   # the timed region carries a HIP event after every 4th step; per-step times are the group means

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VECTOR_PEAK_TFLOPS = 78.6
BYTES_PER_ROW = 40.0          # epoch, ra, dec, σ_ra, σ_dec  
BYTES_PER_WALKER = 64.0 + 72.0  # read 8 elements, write ll + 8 adjoints


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--epochs", type=int, default=10_000)
    ap.add_argument("--walkers", type=int, default=None, help="walkers per GPU (default 10000; 8192 = 8 temperatures x 1024 for --workload pt)")
    ap.add_argument("--workload", choices=["grad", "fwd", "nuis", "two_planet", "pt", "ofti", "logpost", "wide_prior", "rv_gappy", "rv_gappy_nuis"], default="grad",
                    help="grad (default): BASELINE config 3, the metric. Round 6: wide_prior = config 3's table with a ~ LogU(0.3, 100) AU; rv_gappy(_nuis) = one planet, "
                         "1e4 absolute-RV epochs in nightly runs with seasonal gaps (with per-walker offset + jitter): tests/synth.py")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="strong (default for grad / fwd / nuis): --walkers in total, split evenly over the ranks (SURVEY 8d 'Scaling runs'); "
                         "weak (default for the other workloads): --walkers per GPU")
    ap.add_argument("--pt-comm", choices=["c_abi", "torch"], default="c_abi",
                    help="--workload pt: all-gather inside the library (octo_pt_step_device, RCCL bound by the C ABI) or through torch.distributed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive, parity and config-1 legs (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for single-GPU dry runs")
    ap.add_argument("--device", type=int, default=None, help="force the HIP device index (dry runs of the N>1 path on one GPU)")
    args = ap.parse_args()
    if args.walkers is None:      # BASELINE config 5: 64 temperatures x 1024 walkers over 8 GPUs = 8 x 1024 per GPU
        args.walkers = 8192 if args.workload == "pt" else 10_000
    if args.scaling is None:
        args.scaling = "strong" if args.workload in ("grad", "fwd", "nuis") else "weak"
    if args.scaling == "strong" and args.workload not in ("grad", "fwd", "nuis"):
        raise SystemExit(f"--scaling strong is defined for the walker-sharded workloads grad / fwd / nuis, not for {args.workload}")
    return args


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run this same command line under torch.distributed.run, one rank per GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if os.environ.get("OCTO_BENCH_PRINT_LAUNCH"):      # tests: show the command instead of running it
        print(json.dumps(cmd))
        raise SystemExit(0)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def kernel_source_hash():
    from __graft_entry__ import kernel_source_hash as h
    return h()


def cpu_baseline(cfg, obs_tables, planets, seconds):
    """SURVEY §8(d) "CPU baseline beside it": the C restatement of the reference path (oracle/, kind "port"; forward-mode duals = a
    ForwardDiff chunk of 8) on a bounded sample of the same workload — (a) single thread and (b) OpenMP over walkers on all host cores,
    forward-only and fwd+gradient each. `value` is (b) fwd+grad, the metric's definition; the other three travel beside it. The
    reference's own remarks for scale: 32-200 ns per orbit solve (src/likelihoods/system.jl:244-249)."""
    import oracle_binding as ob
    import synth
    cores = os.cpu_count() or 1
    E = cfg["n_epochs"]
    mask = synth.active_mask(1, 1, mass=False, nuis=False)

    def timed(n_threads, grad, budget):
        width = cores if n_threads == 0 else 1
        n_threads = width      # explicit team size: torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks, which "all cores" (0) would obey
        probe = min(cfg["n_walkers"], 4 * width)
        ob.oracle_eval(obs_tables, planets, cfg["elems"][:, :probe], None, grad=grad, active=mask, n_threads=n_threads)   # warm
        t0 = time.perf_counter()
        ob.oracle_eval(obs_tables, planets, cfg["elems"][:, :probe], None, grad=grad, active=mask, n_threads=n_threads)
        rate = probe * E / max(time.perf_counter() - t0, 1e-6)
        n = int(min(cfg["n_walkers"], max(width, rate * budget / E)))
        n = max(width, n // width * width)
        reps = max(1, int(round(rate * budget / (n * E))))
        t0 = time.perf_counter()
        for _ in range(reps):
            ob.oracle_eval(obs_tables, planets, cfg["elems"][:, :n], None, grad=grad, active=mask, n_threads=n_threads)
        dt = time.perf_counter() - t0
        return reps * n * E / dt, f"{reps} pass(es) over {n} walkers x {E} epochs, {dt:.1f} s"

    v_all, s_all = timed(0, True, 0.40 * seconds)
    v_all_f, s_all_f = timed(0, False, 0.15 * seconds)
    v_one, s_one = timed(1, True, 0.30 * seconds)
    v_one_f, s_one_f = timed(1, False, 0.15 * seconds)
    return {"value": v_all, "unit": "epoch-likelihood evals/s (fwd+grad)", "cores": cores, "kind": "port",
            "sample": f"{s_all} of the same workload, oracle/liboctooracle.so (gcc -O3 -march=native, OpenMP over walkers, 8 forward-mode partials per dual)",
            "forward_only": {"value": v_all_f, "unit": "evals/s (fwd)", "cores": cores, "sample": s_all_f},
            "single_thread": {"value": v_one, "unit": "evals/s (fwd+grad)", "cores": 1, "ns_per_eval": 1e9 / v_one, "sample": s_one},
            "single_thread_forward_only": {"value": v_one_f, "unit": "evals/s (fwd)", "cores": 1, "ns_per_eval": 1e9 / v_one_f, "sample": s_one_f},
            "reference_remarks": "the reference's own comment puts one Kepler solve at ~200 ns on a CPU core (src/likelihoods/system.jl:245-246; "
                                 "SURVEY.md section 6 infers 32-200 ns); one evaluation here = one solve + projection + density (+ 8 partials with the gradient)"}


def parity_sample(fn, cfg, ll_dev, g_dev, n=8):
    """ll and gradient of n walkers of the batch just timed against the oracle (the checker)."""
    import oracle_binding as ob
    import synth
    idx = np.random.default_rng(0).choice(cfg["n_walkers"], n, replace=False)
    ll_o, g_o, _ = ob.oracle_eval(fn.obs_tables, fn.planet_desc, cfg["elems"][:, idx], None, grad=True,
                                  active=synth.active_mask(1, 1, mass=False, nuis=False))
    e_ll = float(np.max(np.abs(ll_dev[idx] - ll_o) / np.maximum(1.0, np.abs(ll_o))))
    sc = np.maximum(np.abs(g_o[:8]).max(axis=1, keepdims=True), 1e-300)
    e_g = float(np.max(np.abs(g_dev[:8, idx] - g_o[:8]) / sc))
    return {"max_rel_err_ll": e_ll, "max_rel_err_grad": e_g, "walkers_checked": int(n), "against": "oracle/liboctooracle.so (CPU restatement)",
            "tolerance": 1e-8, "ok": bool(e_ll < 1e-8 and e_g < 1e-8)}


def config1_latency(pkg, dev_index):
    """BASELINE config 1 / SURVEY §8(d): D = 11 model, 50 RA/Dec epochs, ONE θ_t per call — what NUTS pays per gradient."""
    import ctypes as C
    import oracle_binding as ob
    case = json.loads((ROOT / "tests" / "golden" / "config1.json").read_text())["cases"][0]
    o = case["obs"][0]
    table = dict(epoch=o["epoch"], ra=o["y1"], dec=o["y2"], σ_ra=o["s1"], σ_dec=o["s2"])
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[pkg.PlanetRelAstromLikelihood(table, name="astrom")],
                   variables=pkg.variables(a=pkg.Uniform(0, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                           Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
    model = pkg.LogDensityModel(pkg.System(name="cfg1", companions=[b], observations=[],
                                variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1))),
                                device=dev_index)
    capi = pkg.capi
    fn = model.ln_like
    th = np.ascontiguousarray(np.asarray(case["theta_t"])[:, :1]); lp = np.empty(1); g = np.empty_like(th)
    args = (fn._ctx, model._m, capi._dptr(th), 1, 1, capi._dptr(lp), capi._dptr(g))
    for _ in range(300):
        fn.lib.octo_model_logpost(*args)
    n = 1000
    gpu_us = 1e9
    for _ in range(5):      # best of 5 blocks of 1000 calls: a latency, so the quietest block is the measurement
        t0 = time.perf_counter()
        for _ in range(n):
            fn.lib.octo_model_logpost(*args)
        gpu_us = min(gpu_us, (time.perf_counter() - t0) / n * 1e6)
    ok = abs(lp[0] - case["lp"][0]) <= 1e-10 * abs(case["lp"][0])
    obs = [dict(kind=0, planet=0, epoch=np.asarray(o["epoch"]), y1=np.asarray(o["y1"]), y2=np.asarray(o["y2"]), s1=np.asarray(o["s1"]),
                s2=np.asarray(o["s2"]), cor=None)]
    for _ in range(20):
        ob.oracle_model_logpost(obs, case["planets"], model._c_priors, model._c_esrc, None, th)
    m = 300
    t0 = time.perf_counter()
    for _ in range(m):
        ob.oracle_model_logpost(obs, case["planets"], model._c_priors, model._c_esrc, None, th)
    cpu_us = (time.perf_counter() - t0) / m * 1e6
    model.close()
    first = None
    try:      # cold start of the library in a fresh process (module load + first call): tools/first_call.py
        r = subprocess.run([sys.executable, str(ROOT / "tools" / "first_call.py")], capture_output=True, text=True, timeout=300)
        first = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as ex:
        first = {"error": str(ex)}
    return {"first_call": first, "first_call_ms": (first or {}).get("first_call_ms"), "lib_bytes": (first or {}).get("lib_bytes"),
            "workload": "config1: D=11 model (test/integration/sampling.jl:29-64), 50 RA/Dec epochs, one theta_t per call, value + gradient",
            "gpu_us_per_call": gpu_us, "gpu_entry": "octo_model_logpost (host buffers, blocking)", "gpu_matches_fixture": bool(ok),
            "cpu_us_per_call": cpu_us, "cpu_entry": "oracle/ octo_oracle_model_logpost, 1 thread, forward-mode duals (includes ~10 us of ctypes marshalling)"}


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes on this driver
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    # stdout carries exactly ONE line, the JSON: anything a library prints on the way (RCCL's version banner at communicator
    # creation, for one) is sent to stderr by pointing fd 1 there and keeping the real stdout aside for the final line
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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

    def timed_loop(step_fn, on_timed_start=None):
        """warm-up, spin-up, barrier + sync, K steps (per-step events on the launch stream), sync + barrier; max over ranks."""
        for i in range(args.warmup):
            step_fn(i)
        torch.cuda.synchronize()
        t_spin = time.perf_counter()
        n_spin = 0
        while time.perf_counter() - t_spin < SPINUP_SECONDS:
            for _ in range(8):
                step_fn(args.warmup + n_spin); n_spin += 1
            torch.cuda.synchronize()
        if on_timed_start is not None:
            on_timed_start()
        # an event record costs ~1.5 µs of stream time: one after every step is 1 % of a 20-step run, one after every 4th half of that
        ev_every = int(os.environ.get("OCTO_BENCH_EVENTS_EVERY", str(STEP_EVENTS_EVERY)))
        marks = [i for i in range(args.steps + 1) if i % ev_every == 0 or i == args.steps] if ev_every > 0 else []
        evs = {i: torch.cuda.Event(enable_timing=True) for i in marks}
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if 0 in evs:
            evs[0].record()
        for i in range(args.steps):
            step_fn(args.warmup + n_spin + i)
            if i + 1 in evs:
                evs[i + 1].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        per_step = np.array([evs[a].elapsed_time(evs[b]) / (b - a) for a, b in zip(marks[:-1], marks[1:])]) if marks else np.array([dt / args.steps * 1e3])
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, per_step, n_spin

    def base_line(metric, value, dt, per_step, n_spin, workload, extra_cfg=None):
        cfgd = {"workload": workload}
        cfgd.update(extra_cfg or {})
        return {"metric": metric, "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": dt / args.steps * 1e3, "ms_per_step_median_events": float(np.median(per_step)),
                "ms_per_step_min_events": float(per_step.min()), "step_events_every": int(os.environ.get("OCTO_BENCH_EVENTS_EVERY", str(STEP_EVENTS_EVERY))),
                "spinup_steps_untimed": n_spin,
                "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": cfgd}

    grad = args.workload in ("grad", "two_planet", "nuis", "wide_prior", "rv_gappy", "rv_gappy_nuis")
    if args.workload == "ofti":
        # batched ofti_linear_solve (src/parameterizations.jl:318-405), forward marginal likelihood
        cfg0 = synth.config_astrom(n_epochs=args.epochs, n_walkers=args.walkers, cfg=6)
        t = cfg0["table"]
        solver = pkg.OftiLinearSolver(t["epoch"], t["ra"], t["dec"], t["σ_ra"], t["σ_dec"], None, 1000.0, device=dev_index)
        el = cfg0["elems"]
        nl = torch.tensor(np.stack([el[1], el[0], el[5], el[6], el[7]]), device=dev)
        dt, per_step, n_spin = timed_loop(lambda i: solver.eval_device(nl))
        if rank == 0:
            emit(base_line("OFTI marginal-likelihood epoch evals/sec (fwd)", args.epochs * args.walkers * args.steps * world / dt, dt, per_step, n_spin,
                           f"ofti_linear_solve: {args.epochs} RA/Dec epochs x {args.walkers} walkers, forward"))
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
        dt, per_step, n_spin = timed_loop(lambda i: model.logpost_device(θt, grad=True))
        if rank == 0:
            emit(base_line("epoch-likelihood evals/sec, full log-posterior + gradient w.r.t. theta_t (D=11)",
                           args.epochs * args.walkers * args.steps * world / dt, dt, per_step, n_spin,
                           f"LogDensityModel D={model.D}: {args.epochs} RA/Dec epochs x {args.walkers} walkers, theta_t resident in HBM"))
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
        cfg = cfg_full = None
    else:
        # the whole batch of the metric (one seed): what a strong-scaled run splits, what the CPU baseline samples, and (N > 1, strong) what
        # the weak-scaling point gives every rank
        cfg_full = synth.config_astrom(n_epochs=args.epochs, n_walkers=args.walkers, cfg=3 if grad else 2)
        if args.scaling == "strong":
            # SURVEY §8(d) "Scaling runs": the SAME --walkers split evenly over the ranks (contiguous shards, host/sharding.py:shard_range —
            # what octo_eval_multi does inside one process); every rank draws the whole batch with the one seed and keeps its slice
            lo, hi = pkg.shard_range(args.walkers, rank, world)
            cfg = dict(cfg_full, elems=np.ascontiguousarray(cfg_full["elems"][:, lo:hi]), n_walkers=hi - lo)
        else:
            cfg = cfg_full if world == 1 else synth.config_astrom(n_epochs=args.epochs, n_walkers=args.walkers, cfg=3 if grad else 2,
                                                                  seed=20260929 + 3 + 1000 * rank)
        if args.workload in ("wide_prior", "rv_gappy", "rv_gappy_nuis"):      # round 6: non-uniform workloads (weak scaling only: every rank its own draw)
            sd = None if world == 1 else 20260929 + 60 + 1000 * rank
            cfg = cfg_full = (synth.config_wide_prior(n_epochs=args.epochs, n_walkers=args.walkers, seed=sd) if args.workload == "wide_prior"
                              else synth.config_rv_gappy(n_epochs=args.epochs, n_walkers=args.walkers, nuis=args.workload == "rv_gappy_nuis", seed=sd))
        if args.workload.startswith("rv_gappy"):
            planet = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[])
            system = pkg.System(name="bench", companions=[planet], observations=[pkg.StarAbsoluteRVObs(cfg["table"], name="rv")])
        else:
            obs, planet = synth.to_mirror(pkg, cfg)
            system = pkg.System(name="bench", companions=[planet], observations=[])
        fn = pkg.make_ln_like(system, cfg["theta_example"], device=dev_index)
        elems_h, nuis_h = cfg["elems"], cfg.get("nuis")
        n_rows, W = cfg["n_epochs"], cfg["n_walkers"]
        if args.workload == "nuis":      # config 3 with per-walker jitter, platescale and northangle: the raw-σ branch of relative-astrometry.jl:234-252
            rng = np.random.default_rng(1)
            nuis_h = np.stack([rng.uniform(0, 3, W), rng.normal(1, 0.01, W), rng.normal(0, 0.02, W)])
        wl_name = {"wide_prior": "wide_prior (config 3's table, a ~ LogU(0.3, 100) AU)", "rv_gappy": "rv_gappy (absolute RV, nightly runs with seasonal gaps, jitter == 0 path)",
                   "rv_gappy_nuis": "rv_gappy_nuis (absolute RV, nightly runs with seasonal gaps, per-walker offset + jitter)"}.get(args.workload, f"config{'3' if grad else '2'}")
        workload = (f"{wl_name}: 1 planet, {n_rows} {'RV' if args.workload.startswith('rv_gappy') else 'RA/Dec'} epochs x "
                    + (f"{args.walkers} walkers in total ({args.walkers} walkers split over {world} GPUs: {W} walkers/GPU)" if args.scaling == "strong"
                       else f"{W} walkers/GPU")
                    + (", per-walker jitter/platescale/northangle" if args.workload == "nuis" else "")
                    + f", {'fwd+reverse-grad' if grad else 'fwd only'}, inputs and outputs resident in HBM (PCIe-inclusive rate: value_pcie_inclusive)")
        bytes_per_launch = W * n_rows * BYTES_PER_ROW + W * (BYTES_PER_WALKER if grad else 72.0)

    elems = torch.tensor(elems_h, device=dev)
    nuis = torch.tensor(nuis_h, device=dev) if nuis_h is not None else None
    out = (torch.empty(W, dtype=torch.float64, device=dev),
           torch.empty_like(elems) if grad else None,
           torch.empty_like(nuis) if (grad and nuis is not None) else None)

    pt = None
    parallelism = f"walkers sharded x{world}, dataset replicated, no data-path collective"
    if args.workload == "pt":
        # config 5: temperatures sharded over ranks; every step = fwd+grad-free explorer evaluation, all-gather of the
        # per-replica log-likelihoods (RCCL; nothing to gather on one rank), deterministic neighbour swap of β labels.
        from octofitter_jl_amd.host.tempering import TemperedSwap
        n_temps_total = 8 * world
        chains = W // 8
        pt = TemperedSwap(fn, n_temps_total=n_temps_total, n_chains=chains, rank=rank, world=world, device=dev, seed=20260929, comm=args.pt_comm)
        if args.pt_comm == "c_abi":
            pt.create_comm()
        grad = False
        out = (out[0], None, None)
        workload = (f"config5: {n_temps_total} temperatures x {chains} walkers x {n_rows} epochs, sharded by temperature, "
                    + ((f"ncclAllGather of log-likelihoods ({'octo_pt_step_device, C ABI' if args.pt_comm == 'c_abi' else 'torch.distributed'}) + swap kernel")
                       if world > 1 else "swap kernel (single rank: no collective ran)"))
        parallelism = f"temperatures sharded x{world}, dataset replicated, one all_gather per swap step" if world > 1 else "single rank"

    def run_step(i):
        if grad:
            fn.ln_like_device(elems, nuis, grad=True, out=out)
        else:
            fn.ln_like_device(elems, nuis, grad=False, out=(out[0], None, None))
        if pt is not None:
            pt.swap_step(out[0], i)

    def swap_latency(n=200):
        """SURVEY §8(d) config 5 "swap-step latency": the all-gather (N > 1) + swap kernel alone, one step at a time, host-synchronised."""
        for i in range(20):
            pt.swap_step(out[0], 10_000 + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ts = []
        for i in range(n):
            t1 = time.perf_counter(); pt.swap_step(out[0], 20_000 + i); torch.cuda.synchronize(); ts.append(time.perf_counter() - t1)
        t = torch.tensor([float(np.median(ts))], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) * 1e6

    def strong_projection():
        """Per-GPU shares of a strong-scaled run of THIS workload (W/N walkers, N = 2, 4, 8) measured on this one GPU, device-resident,
        back-to-back like the timed region: speed-up N GPUs would deliver = N x rate(W/N) / rate(W) (shards are independent)."""
        outp = {}
        base = None
        for N in (1, 2, 4, 8):
            Wn = W // N
            el_n = elems[:, :Wn].contiguous()
            out_n = (torch.empty(Wn, dtype=torch.float64, device=dev), torch.empty_like(el_n) if grad else None, None)
            for _ in range(30):
                fn.ln_like_device(el_n, None, grad=grad, out=out_n)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(100):
                    fn.ln_like_device(el_n, None, grad=grad, out=out_n)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t1) / 100)
            rate = Wn * n_rows / best
            base = base or rate
            outp[str(N)] = {"walkers_per_gpu": Wn, "us_per_step": best * 1e6, "evals_per_s_per_gpu": rate, "projected_speedup": N * rate / base}
        return {"what": "strong scaling of this workload (SURVEY 8d: walkers split evenly, dataset replicated) projected from the per-GPU share "
                        "measured on one GPU: N x rate(W/N) / rate(W); no collective on this path, so the only thing a real N-GPU run adds is launch jitter",
                "by_n_gpus": outp}

    def strong_measured():
        """N > 1, default (weak) run: the SURVEY §8(d) strong-scaling point of this job measured in the same launch — the W walkers of ONE
        rank's batch split evenly over the N ranks (rank r evaluates columns shard_range(W, r, N) of its own batch: same shapes and launches
        as a true split), barrier + synchronize around 100 back-to-back steps, MAX over ranks; speed-up against this rank-0 GPU's own
        full-W step time from the timed region above. No collective on the data path, so the only thing N ranks add is launch jitter."""
        lo, hi = pkg.shard_range(W, rank, world)
        el_n = elems[:, lo:hi].contiguous()
        out_n = (torch.empty(hi - lo, dtype=torch.float64, device=dev), torch.empty_like(el_n) if grad else None, None)
        for _ in range(30):
            fn.ln_like_device(el_n, None, grad=grad, out=out_n)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); dist.barrier(); t1 = time.perf_counter()
            for _ in range(100):
                fn.ln_like_device(el_n, None, grad=grad, out=out_n)
            torch.cuda.synchronize(); dist.barrier()
            tt = torch.tensor([(time.perf_counter() - t1) / 100], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            best = min(best, float(tt.item()))
        return {"what": "strong scaling measured in this run: one rank's W walkers split evenly over the N ranks (contiguous shards, dataset replicated, "
                        "no collective), 100 back-to-back steps between barriers, max over ranks, best of 3; speedup = full-W step time of the timed "
                        "region / this",
                "n_gpus": world, "walkers_total": W, "walkers_per_gpu": hi - lo, "us_per_step": best * 1e6, "value": W * n_rows / best, "unit": "evals/s"}

    def weak_measured():
        """N > 1, default (strong) run: the weak-scaling point of this job measured in the same launch — EVERY rank evaluates its own batch of
        --walkers walkers (the whole batch of the metric; the ranks' batches differ only by a rotation of the walker order), barrier +
        synchronize around 100 back-to-back steps, MAX over ranks, best of 3."""
        Wf = cfg_full["n_walkers"]
        el_f = torch.tensor(np.roll(cfg_full["elems"], 64 * rank, axis=1), device=dev)
        out_f = (torch.empty(Wf, dtype=torch.float64, device=dev), torch.empty_like(el_f) if grad else None, None)
        for _ in range(10):
            fn.ln_like_device(el_f, None, grad=grad, out=out_f)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); dist.barrier(); t1 = time.perf_counter()
            for _ in range(50):
                fn.ln_like_device(el_f, None, grad=grad, out=out_f)
            torch.cuda.synchronize(); dist.barrier()
            tt = torch.tensor([(time.perf_counter() - t1) / 50], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            best = min(best, float(tt.item()))
        return {"what": "weak scaling measured in this run: every rank its own full batch (walkers per GPU = the metric's walkers in total), dataset "
                        "replicated, no collective, 50 back-to-back steps between barriers, max over ranks, best of 3",
                "n_gpus": world, "walkers_per_gpu": Wf, "us_per_step": best * 1e6, "value": world * Wf * n_rows / best, "unit": "evals/s"}

    fn.timing_enable(TIMED_EVERY)      # HIP events around k_main of every TIMED_EVERY-th evaluation, on its launch stream
    dt, per_step, n_spin = timed_loop(run_step, on_timed_start=lambda: fn.timing_read(reset=True))
    kern_med, kern_min, kern_max, kern_n = fn.timing_stats()
    kern_ms, _ = fn.timing_read(reset=True)
    fn.timing_enable(False)

    if args.scaling == "strong" and cfg is not None:      # ranks may differ by one walker: count what every rank really did
        tw = torch.tensor([float(W)], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tw, op=dist.ReduceOp.SUM)
        evals = float(tw.item()) * n_rows * args.steps
    else:
        evals = float(W) * n_rows * args.steps * world
    value = evals / dt
    metric = "epoch-likelihood evals/sec (fwd+grad), 1e4 epochs x 1e4 walkers" if args.workload == "grad" else f"epoch-likelihood evals/sec ({args.workload})"
    res = base_line(metric, value, dt, per_step, n_spin, workload,
                    {"walkers_per_gpu": W, "walkers_total": (args.walkers if args.scaling == "strong" else W * world), "rows": n_rows, "parallelism": parallelism})
    if world > 1 and args.scaling == "strong" and cfg is not None and args.workload in ("grad", "fwd") and not args.no_extras:
        try:      # every rank takes part (barriers + a MAX all-reduce); rank 0 reports
            res["weak_scaling_measured"] = weak_measured()
        except Exception as ex:
            res["weak_scaling_measured"] = {"error": str(ex)}
    if world > 1 and args.scaling == "weak" and cfg is not None and args.workload in ("grad", "fwd") and not args.no_extras:
        try:      # every rank takes part (barriers + a MAX all-reduce); rank 0 reports
            sm_ = strong_measured()
            sm_["speedup_vs_one_gpu"] = (dt / args.steps) / (sm_["us_per_step"] * 1e-6)
            sm_["one_gpu_us_per_step"] = dt / args.steps * 1e6
            res["strong_scaling_measured"] = sm_
        except Exception as ex:
            res["strong_scaling_measured"] = {"error": str(ex)}
    if pt is not None:
        lat = swap_latency()
        res["swap_step_latency_us"] = lat
        res["swap_step_what"] = ("median over 200 steps, max over ranks: " + ("ncclAllGather of the local replicas' log-likelihoods + " if world > 1 else "")
                                 + "deterministic neighbour-swap kernel, host-synchronised per step")
    if rank == 0:
        is_cfg3 = args.workload == "grad" and n_rows == 10_000 and cfg is not None
        pmc, pmc_ok, pmc_note, pmc_exact = None, False, None, True
        pmc_path = ROOT / "profiles" / "pmc_traffic.json"
        if is_cfg3 and pmc_path.exists():
            try:      # PMC figures are per launch of exactly this kernel and LAUNCH SHAPE (walkers per GPU); measured off-line (separate --pmc passes)
                allp = json.loads(pmc_path.read_text())
                pmc_ok = allp.get("kernel_source_sha256") == kernel_source_hash()
                shapes = {10_000: allp}
                shapes.update({int(k): v for k, v in allp.get("shapes", {}).items()})
                if W in shapes:
                    pmc = shapes[W]
                else:      # a shard size without its own counter pass (an uneven split): the nearest profiled shape's per-evaluation counts, said so
                    near = min(shapes, key=lambda k: abs(np.log(k / W)))
                    pmc, pmc_exact = shapes[near], False
                    pmc_note = f"no counter pass for {W} walkers per GPU: per-evaluation instruction counts of the nearest profiled launch shape ({near} walkers)"
                if not pmc_ok:
                    pmc_note = ("profiles/pmc_traffic.json was collected for a DIFFERENT kernel source (sha256 mismatch): counter-derived "
                                "fields withheld; re-run tools/profile_round.sh + tools/make_pmc_json.py")
                    print("bench.py: " + pmc_note, file=sys.stderr, flush=True)
            except Exception as ex:
                pmc, pmc_note = None, f"profiles/pmc_traffic.json unreadable: {ex}"
        kernel_s = kern_ms * 1e-3 if kern_ms > 0 else None
        alg_gbps = bytes_per_launch / kernel_s / 1e9 if kernel_s else None
        roof = {"bound": "fp64_vector", "achieved": None, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                "kernel": "k_main", "kernel_avg_ms": kern_ms, "kernel_median_ms": kern_med, "kernel_min_ms": kern_min, "kernel_max_ms": kern_max,
                "kernel_launches_timed": kern_n,
                "note": "k_main is bound by FP64 VALU issue, not by HBM: rows reach the 64 lanes of a wave through the scalar cache, so the "
                        "compulsory HBM traffic is ~0.02 B per evaluation. achieved = counter-derived FP64 flops per evaluation (FMA = 2) x "
                        "evaluations per launch / live HIP-event duration of k_main on its launch stream."}
        if pmc is not None and pmc_ok and kernel_s:
            flops_per_eval = pmc["fp64_flops_per_eval"]
            tf = flops_per_eval * float(W) * n_rows / kernel_s / 1e12
            traffic = pmc["hbm_bytes_per_launch"] * (1.0 if pmc_exact else float(W) / pmc["walkers_per_launch"])
            roof.update({"achieved": tf, "frac": tf / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic, "fp64_flops_per_eval": flops_per_eval,
                         "kernel": pmc.get("kernel", "k_main"), "launch_shape": {"walkers": W, "rows": n_rows, "pmc_walkers": pmc.get("walkers_per_launch", 10_000)},
                         "valu_instructions_per_row_per_wave": pmc.get("valu_instructions_per_row_per_wave"),
                         "pmc_source": pmc.get("source"), "pmc_kernel_source_sha256": allp.get("kernel_source_sha256")})
            if pmc_note:
                roof["pmc_note"] = pmc_note
            issue = pmc.get("issue_model")
            if issue:
                # time the chip needs just to ISSUE this kernel's VALU instructions (measured mix x measured cost per class)
                t_issue = issue["ns_per_row_per_wave"] * 1e-6 * n_rows * ((W + 63) // 64) / issue["simds"]
                roof.update({"issue_bound_ms": t_issue, "issue_model_over_measured": t_issue / kern_ms,
                             "issue_note": "per-class issue costs from tools/ubench.hip x the counted instruction mix; within ~6 % of the measured "
                                           "duration either way = the kernel runs at its VALU issue limit"})
            roof["hbm"] = {"bound": "hbm", "achieved": traffic / kernel_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": traffic / kernel_s / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                           "what": "real HBM bytes per launch (PMC FETCH_SIZE x2 on gfx950 + WRITE_SIZE) / live kernel duration"}
        elif pmc_note:
            roof["note"] = pmc_note + " | " + roof["note"]
        if world > 1:
            roof["per_gpu"] = ("rank 0's GPU: its shard's k_main launches timed live on its launch stream; every rank runs the same launch shape "
                               "(walkers split evenly, dataset replicated)")
        roof["north_star_algorithmic_hbm"] = {
            "algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_stream_GBps": alg_gbps,
            "frac_of_hbm_peak": (alg_gbps / HBM_PEAK_GBPS) if alg_gbps else None,
            "what": "SURVEY.md 8(d) streaming definition (every walker re-reads every 40-byte row) over the live kernel duration, as "
                    "north_star words the claim; NOT a bandwidth measurement — it exceeds the 8 TB/s peak because the rows are reused "
                    "by 64 lanes from the scalar cache"}
        res["roofline"] = roof
        ts = fn.tile_state() if hasattr(fn, "tile_state") else None
        if ts is not None:      # round 6: the walker-tile sort (octo_tile.h) — on only where its probe says it pays
            res["tile_sort"] = dict(ts, mode={0: "off", 1: "always", 2: "auto"}.get(fn.get_option(capi.OPT_TILE_SORT), "?"),
                                    what="walkers grouped into tiles of 64 by how often their rows fail the warm start's a-priori test; a probe every 64th "
                                         "evaluation prices the sort against its launch and decides")
        if not args.no_extras and cfg is not None and args.workload == "grad":
            # ---- parity of the batch just timed (N > 1: of rank 0's shard)
            try:
                res["parity"] = parity_sample(fn, cfg, out[0].cpu().numpy(), out[1].cpu().numpy())
                res["max_rel_err"] = max(res["parity"]["max_rel_err_ll"], res["parity"]["max_rel_err_grad"])
            except Exception as ex:
                res["parity"] = {"ok": None, "error": str(ex)}
        if not args.no_extras and cfg is not None and world == 1 and args.workload == "grad":
            # ---- the same batch through octo_eval, host buffers, H2D + D2H inside the timed call
            el_h = np.ascontiguousarray(elems_h); ll_h = np.empty(W); g_h = np.empty_like(el_h)
            a_ = (fn._ctx, fn._ds, capi._dptr(el_h), None, W, W, capi._dptr(ll_h), capi._dptr(g_h), None)
            for _ in range(5):
                fn.lib.octo_eval(*a_)
            ts = []
            for _ in range(25):
                t1 = time.perf_counter(); fn.lib.octo_eval(*a_); ts.append(time.perf_counter() - t1)
            med = float(np.median(ts))
            res["value_pcie_inclusive"] = W * n_rows / med      # replaced below by the registered-arrays rate when that leg succeeds
            res["pcie_inclusive"] = {"value": W * n_rows / med, "unit": "evals/s", "ms_per_call_median": med * 1e3, "calls": len(ts),
                                     "what": "octo_eval with PAGEABLE host buffers: H2D of elems and D2H of ll + gradient inside the call (SURVEY 8d definition); median of 25"}
            # the same call with the caller's arrays registered once (octo_host_register): copy kernel in, results written in place
            try:
                ll_ref = ll_h.copy()
                fn.host_register(el_h, ll_h, g_h)
                ll_h[:] = np.nan
                for _ in range(5):
                    fn.lib.octo_eval(*a_)
                ts = []
                for _ in range(25):
                    t1 = time.perf_counter(); fn.lib.octo_eval(*a_); ts.append(time.perf_counter() - t1)
                medr = float(np.median(ts))
                # SURVEY 8(d)'s clock: DEVICE time of the call by HIP events, from ahead of the copy-in to behind k_finish's stores into the
                # caller's arrays (octo_timing_enable(ctx, -1)); the blocking call's wall time adds the launch + synchronisation of one call
                fn.timing_read(reset=True)
                fn.timing_enable(-1)
                for _ in range(25):
                    fn.lib.octo_eval(*a_)
                dev_med, dev_min, dev_max, dev_n = fn.timing_stats()
                fn.timing_read(reset=True)
                fn.timing_enable(0)
                fn.host_unregister(el_h, ll_h, g_h)
                # key meanings: `value_pcie_inclusive` = the BLOCKING call by the host's clock, as in BENCH_r01..r03 (r04 carried the
                # device-clock figure under this key); the device-clock figure of SURVEY 8(d)'s wording travels as `value_pcie_inclusive_device_clock`
                res["value_pcie_inclusive"] = W * n_rows / medr
                res["value_pcie_inclusive_device_clock"] = W * n_rows / (dev_med * 1e-3)
                res["value_pcie_inclusive_what"] = ("octo_eval on host arrays registered once by the caller, H2D of the elements and D2H of ll + gradient inside the call; "
                                                    "median of 25 calls. value_pcie_inclusive: wall time of the blocking call by the host's clock (launch + "
                                                    "synchronisation included; the meaning of this key in BENCH_r01..r03). value_pcie_inclusive_device_clock: SURVEY.md "
                                                    "8(d)'s clock, device time by HIP events on the call's stream (BENCH_r04 carried this one as value_pcie_inclusive)")
                res["value_pcie_inclusive_blocking_call"] = W * n_rows / medr
                res["pcie_inclusive"]["registered"] = {
                    "value": W * n_rows / (dev_med * 1e-3), "unit": "evals/s", "device_ms_per_call_median": dev_med, "device_ms_min": dev_min, "device_ms_max": dev_max,
                    "calls_timed": int(dev_n), "blocking_call_value": W * n_rows / medr, "ms_per_call_median": medr * 1e3, "calls": len(ts),
                    "bit_identical_to_pageable": bool(np.array_equal(ll_h, ll_ref, equal_nan=True)),
                    "what": "the same octo_eval call with elems, ll and gradient arrays registered once by the caller (octo_host_register: "
                            "page-locked + mapped): no copy engine, inputs by a copy kernel over PCIe, outputs written in place"}
            except Exception as ex:
                res["pcie_inclusive"]["registered"] = {"error": str(ex)}
            # ---- the same host-buffer evaluations PIPELINED: two contexts (two streams), octo_eval_begin on one while the other's batch is in
            # flight — how a driver that holds two walker batches (or two Julia threads, src/initialization.jl:33-48) hides the PCIe legs and
            # the launch + synchronisation of a blocking call: copy-in and write-back of one batch run under the epoch loop of the other
            try:
                fn2 = pkg.make_ln_like(system, cfg["theta_example"], device=dev_index)
                bufs = []
                for f_ in (fn, fn2):
                    e_, l_, g_ = np.array(elems_h, order="C", copy=True), np.empty(W), np.empty((elems_h.shape[0], W))
                    f_.host_register(e_, l_, g_)
                    bufs.append((f_, e_, l_, g_, (f_._ctx, f_._ds, capi._dptr(e_), None, W, W, capi._dptr(l_), capi._dptr(g_), None)))
                lib = fn.lib
                def pipelined(n_calls):
                    assert lib.octo_eval_begin(*bufs[0][4]) == 0
                    for k in range(1, n_calls):
                        assert lib.octo_eval_begin(*bufs[k & 1][4]) == 0
                        assert lib.octo_eval_end(bufs[(k - 1) & 1][0]._ctx) == 0
                    assert lib.octo_eval_end(bufs[(n_calls - 1) & 1][0]._ctx) == 0
                pipelined(6)
                n_calls = 40
                t1 = time.perf_counter(); pipelined(n_calls); dt2 = time.perf_counter() - t1
                same2 = all(bool(np.array_equal(b[2], ll_ref, equal_nan=True)) for b in bufs)
                for f_, e_, l_, g_, _ in bufs:
                    f_.host_unregister(e_, l_, g_)
                fn2.close()
                res["value_pcie_inclusive_pipelined"] = W * n_rows * n_calls / dt2
                res["pcie_inclusive"]["two_contexts_pipelined"] = {
                    "value": W * n_rows * n_calls / dt2, "unit": "evals/s", "ms_per_call": dt2 / n_calls * 1e3, "calls": n_calls,
                    "bit_identical_to_pageable": same2,
                    "what": "octo_eval_begin / octo_eval_end on two contexts alternately, registered host arrays, every call a full 1e4-walker "
                            "batch with its H2D and D2H inside: wall time of 40 calls / 40"}
            except Exception as ex:
                res["pcie_inclusive"]["two_contexts_pipelined"] = {"error": str(ex)}
        if not args.no_extras and world == 1 and args.workload == "grad":      # before the CPU baseline: its OpenMP team keeps spinning for a while and would disturb a latency measurement
            try:
                res["strong_scaling_projection"] = strong_projection()
            except Exception as ex:
                res["strong_scaling_projection"] = {"error": str(ex)}
            try:
                res["config1"] = config1_latency(pkg, dev_index)
            except Exception as ex:
                res["config1"] = {"error": str(ex)}
        if not args.no_cpu_baseline and cfg_full is not None and args.workload == "grad":      # rank 0, at every N: the whole workload of the metric
            try:
                res["cpu_baseline"] = cpu_baseline(cfg_full, fn.obs_tables, fn.planet_desc, args.cpu_seconds)
            except Exception as ex:  # the checker is optional for the measurement itself
                res["cpu_baseline"] = {"value": None, "unit": "evals/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {ex}"}
        emit(res)
    fn.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
