"""Latency of small batches through the host-buffer entry points (what a single-chain sampler pays per gradient, and Pigeons'
32 replicas per sweep): octo_eval at W ∈ {1, 32} for a 50-epoch and a 1e4-epoch table, and octo_model_logpost for one θ_t
(the whole ∇ℓπcallback, src/logdensitymodel.jl:169-177). The C calls are timed with prebuilt ctypes arguments, so the number is the
call itself (+ ~1 µs of ctypes dispatch), not NumPy wrapper overhead. `small_batch=0` rows: the same call forced onto the
throughput kernels (3 launches + staging copies), i.e. the round-1 path.   python tools/latency_w1.py"""
import ctypes as C, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from __graft_entry__ import load_package
import synth
pkg = load_package()
capi = pkg.capi
out = {}


def time_call(f, n=3000, warm=300):
    for _ in range(warm): f()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n // 5): f()
        best = min(best, (time.perf_counter() - t0) / (n // 5))
    return best * 1e6


for E in (50, 10000):
    for W in (1, 32):
        cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3)
        obs, planet = synth.to_mirror(pkg, cfg)
        fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
        el = np.ascontiguousarray(cfg["elems"]); ll = np.empty(W); g = np.empty_like(el)
        for sb in (None, 0):
            fn._check(fn.lib.octo_ctx_set_small_batch(fn._ctx, 32 if sb is None else 0), "set")
            for grad in (False, True):
                args = (fn._ctx, fn._ds, capi._dptr(el), None, W, W, capi._dptr(ll), capi._dptr(g) if grad else None, None)
                us = time_call(lambda: fn.lib.octo_eval(*args))
                key = f"octo_eval E={E} W={W} grad={int(grad)} {'small-batch kernel' if sb is None else 'throughput kernels'}"
                out[key] = us
                print(f"{key:78s} {us:7.1f} us", flush=True)
        fn.close()

case = json.loads((ROOT / "tests" / "golden" / "model.json").read_text())["cases"][0]
o = case["obs"][0]
for E in (8, 50, 10000):
    if E == 8:
        table = dict(epoch=o["epoch"], ra=o["y1"], dec=o["y2"], σ_ra=o["s1"], σ_dec=o["s2"], cor=o["cor"])
    else:
        c0 = synth.config_astrom(n_epochs=E, n_walkers=1, cfg=3)["table"]
        table = dict(epoch=c0["epoch"], ra=c0["ra"], dec=c0["dec"], σ_ra=c0["σ_ra"], σ_dec=c0["σ_dec"])
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[pkg.PlanetRelAstromObs(table, name="astrom")],
                   variables=pkg.variables(a=pkg.Uniform(0, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                           Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
    model = pkg.LogDensityModel(pkg.System(name="T", companions=[b], observations=[],
                                variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1))))
    fn = model.ln_like
    th = np.ascontiguousarray(np.asarray(case["theta_t"])[:, :1]); lp = np.empty(1); g = np.empty_like(th)
    for sb in (None, 0):
        fn._check(fn.lib.octo_ctx_set_small_batch(fn._ctx, 32 if sb is None else 0), "set")
        args = (fn._ctx, model._m, capi._dptr(th), 1, 1, capi._dptr(lp), capi._dptr(g))
        us = time_call(lambda: fn.lib.octo_model_logpost(*args))
        key = f"octo_model_logpost D=11 E={E} one theta_t, value+gradient {'small-batch kernel' if sb is None else 'throughput kernels'}"
        out[key] = us
        print(f"{key:78s} {us:7.1f} us", flush=True)
    model.close()
Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "latency_w1.json").write_text(json.dumps(out, indent=1))
