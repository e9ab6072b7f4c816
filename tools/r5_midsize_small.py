"""Where does the fused small-batch launch (k_small<MODEL>) stop paying for the whole callback? octo_model_logpost (D = 11 model, host buffers,
blocking) at W = 256 … 2048 x 50 and 300 epochs, with the small-batch limit at 512 (round 4's default), at its maximum, and 0 (throughput
kernels).   python tools/r5_midsize_small.py"""
import sys, time, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from __graft_entry__ import load_package
import synth
pkg = load_package(); capi = pkg.capi


def time_call(f, n=600, warm=60):
    for _ in range(warm): f()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n // 5): f()
        best = min(best, (time.perf_counter() - t0) / (n // 5))
    return best * 1e6


case = json.loads((ROOT / "tests" / "golden" / "model.json").read_text())["cases"][0]
rng = np.random.default_rng(3)
for E in (50, 300):
    c0 = synth.config_astrom(n_epochs=E, n_walkers=1, cfg=3)["table"]
    table = dict(epoch=c0["epoch"], ra=c0["ra"], dec=c0["dec"], σ_ra=c0["σ_ra"], σ_dec=c0["σ_dec"])
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[pkg.PlanetRelAstromObs(table, name="astrom")],
                   variables=pkg.variables(a=pkg.Uniform(0, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                           Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
    model = pkg.LogDensityModel(pkg.System(name="T", companions=[b], observations=[],
                                variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1))))
    fn = model.ln_like
    for W in (256, 512, 768, 1024, 1536, 2048):
        th = np.ascontiguousarray(np.asarray(case["theta_t"])[:, :1] + 0.05 * rng.normal(size=(model.D, W))); lp = np.empty(W); g = np.empty_like(th)
        row = []
        for sb in (512, 4096, 0):
            st = fn.lib.octo_ctx_set_small_batch(fn._ctx, sb)
            args = (fn._ctx, model._m, capi._dptr(th), W, W, capi._dptr(lp), capi._dptr(g))
            row.append(time_call(lambda: fn.lib.octo_model_logpost(*args)))
        print(f"octo_model_logpost D=11 E={E:4d} W={W:5d}: small-batch limit 512 {row[0]:7.1f} us | at its maximum {row[1]:7.1f} us | throughput kernels {row[2]:7.1f} us", flush=True)
    model.close()
