"""Phase timing inside k_small (development build with -DOCTO_SMALL_TRACE: tools/liboctofitter_trace.bin): core-clock stamps of the
finishing block of walker 0 — where one small call's microseconds go."""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("OCTOFITTER_HIP_LIB", str(ROOT / "octofitter.jl_amd" / "lib" / "variants" / "liboctofitter_hip_trace.so"))      # python tools/build_variant.py trace -DOCTO_SMALL_TRACE
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from __graft_entry__ import load_package
import synth
pkg = load_package(); capi = pkg.capi
names = ["start", "setup", "rows+block-reduce", "last-known", "obs-finish", "outputs", "flag"]
for E, W in ((50, 1), (10000, 1), (10000, 32)):
    cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3)
    obs, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
    fn.lib.octo_debug_small_trace.restype = C.POINTER(C.c_uint64); fn.lib.octo_debug_small_trace.argtypes = [C.c_void_p]
    el = np.ascontiguousarray(cfg["elems"]); ll = np.empty(W); g = np.empty_like(el)
    args = (fn._ctx, fn._ds, capi._dptr(el), None, W, W, capi._dptr(ll), capi._dptr(g), None)
    acc = np.zeros(7); n = 0
    for it in range(300):
        fn.lib.octo_eval(*args)
        if it >= 100:
            fn.sync()
            t = np.array([fn.lib.octo_debug_small_trace(fn._ctx)[k] for k in range(7)], dtype=np.float64)
            acc += t - t[0]; n += 1
    acc /= n
    print(f"E={E} W={W}: " + "  ".join(f"{nm} {acc[k] / 2400:.2f}us" for k, nm in enumerate(names)), flush=True)
    fn.close()

# the fused model launch (θ_t in, log-posterior + ∇θ_t out): config 1 and a 1e4-epoch table
import json
case = json.loads((ROOT / "tests" / "golden" / "config1.json").read_text())["cases"][0]
for E in (50, 10000):
    o = case["obs"][0]
    if E == 50:
        table = dict(epoch=o["epoch"], ra=o["y1"], dec=o["y2"], σ_ra=o["s1"], σ_dec=o["s2"])
    else:
        c0 = synth.config_astrom(n_epochs=E, n_walkers=1, cfg=3)["table"]
        table = dict(epoch=c0["epoch"], ra=c0["ra"], dec=c0["dec"], σ_ra=c0["σ_ra"], σ_dec=c0["σ_dec"])
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[pkg.PlanetRelAstromObs(table, name="astrom")],
                   variables=pkg.variables(a=pkg.Uniform(0, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                           Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
    model = pkg.LogDensityModel(pkg.System(name="T", companions=[b], observations=[],
                                variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1))))
    fn = model.ln_like
    fn.lib.octo_debug_small_trace.restype = C.POINTER(C.c_uint64); fn.lib.octo_debug_small_trace.argtypes = [C.c_void_p]
    th = np.ascontiguousarray(np.asarray(case["theta_t"])[:, :1]); lp = np.empty(1); g = np.empty_like(th)
    args = (fn._ctx, model._m, capi._dptr(th), 1, 1, capi._dptr(lp), capi._dptr(g))
    mnames = ["start", "priors", "circ-table", "elements(tperi)", "setup", "rows+block-reduce", "last-known", "obs-finish", "outputs", "flag"]
    acc = np.zeros(len(mnames)); n = 0
    for it in range(300):
        fn.lib.octo_model_logpost(*args)
        if it >= 100:
            fn.sync()
            t = np.array([fn.lib.octo_debug_small_trace(fn._ctx)[k] for k in range(len(mnames))], dtype=np.float64)
            acc += t - t[0]; n += 1
    acc /= n
    print(f"model D=11 E={E} W=1: " + "  ".join(f"{nm} {acc[k] / 2400:.2f}us" for k, nm in enumerate(mnames)), flush=True)
    model.close()
