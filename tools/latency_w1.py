"""Latency of ONE parameter set through the host-buffer entry point (what a single-chain sampler pays per gradient):
octo_eval with W = 1, for a 50-epoch and a 1e4-epoch table. Development aid."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from __graft_entry__ import load_package
import synth
pkg = load_package()
for E in (50, 10000):
    cfg = synth.config_astrom(n_epochs=E, n_walkers=1, cfg=3)
    obs, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
    el = np.ascontiguousarray(cfg["elems"])
    for grad in (False, True):
        for _ in range(200): fn.ln_like_arrays(el, None, grad=grad)
        t0 = time.perf_counter(); n = 2000
        for _ in range(n): fn.ln_like_arrays(el, None, grad=grad)
        dt = (time.perf_counter() - t0) / n
        print(f"E={E:6d} W=1 grad={grad}: {dt*1e6:7.1f} us per call (Python + ctypes + H2D + 3 kernels + D2H + sync)", flush=True)
    fn.close()

# the whole callback for one θ_t: what AdvancedHMC's leapfrog calls (model.∇ℓπcallback, src/logdensitymodel.jl:169-177)
import json
case = json.loads((ROOT / "tests" / "golden" / "model.json").read_text())["cases"][0]
o = case["obs"][0]
for E in (8, 10000):
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
    th = np.asarray(case["theta_t"])[:, 0].copy()
    for _ in range(200): model.logdensity_and_gradient(th)
    t0 = time.perf_counter(); n = 2000
    for _ in range(n): model.logdensity_and_gradient(th)
    print(f"E={E:6d} D=11 one theta_t, log-posterior + gradient: {(time.perf_counter() - t0) / n * 1e6:7.1f} us per call", flush=True)
    model.close()
