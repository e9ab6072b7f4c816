"""Config 4 (2 planets, RA/Dec on the outer + absolute RV, nuisances, 4 096 walkers) step time as drawn and with the OUTER planet's eccentricities scaled into
[0, e_max): with e_max = 0.8 every wave is provably safe for the last-planet-always-warm loop (octo_kernels.h: main_warm_last) — the upper bound of what tiling the
walkers by that planet's severity can give. Run under OCTOFITTER_HIP_LIB=<a build> for a same-box A/B.   python tools/r6_cfg4_probe.py"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from __graft_entry__ import load_package
import synth
pkg = load_package()
tag = os.environ.get("OCTOFITTER_HIP_LIB", "default")[-28:]
for e_max in (None, 0.8):
    c4 = synth.config_two_planet()
    if e_max is not None:
        c4["elems"][9 + 1] *= e_max / 0.95
    astrom = pkg.PlanetRelAstromObs(c4["astrom"], name="astrom"); rv = pkg.StarAbsoluteRVObs(c4["rv"], name="rv")
    b = pkg.Planet(name="b", observations=[]); c = pkg.Planet(name="c", observations=[astrom])
    system = pkg.System(name="cfg4", companions=[b, c], observations=[rv])
    θex = dict(M=1.2, plx=50.0, planets=dict(b=dict(a=3, e=0.1, i=1, ω=1, Ω=2, tp=5e4, mass=5), c=dict(a=15, e=0.3, i=1, ω=.5, Ω=2, tp=5e4, mass=10)))
    fn = pkg.make_ln_like(system, θex)
    el = torch.tensor(c4["elems"], device="cuda"); nu = torch.tensor(c4["nuis"], device="cuda")
    out = (torch.empty(el.shape[1], dtype=torch.float64, device="cuda"), torch.empty_like(el), torch.empty_like(nu))
    for _ in range(50): fn.ln_like_device(el, nu, grad=True, out=out)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn.ln_like_device(el, nu, grad=True, out=out)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
    W = el.shape[1]
    print(f"{tag:>28} outer e < {e_max or 0.95}: {best * 1e6:7.1f} us per step  {W * c4['n_rows'] / best:.3e} evals/s  tile sort {fn.tile_state() if hasattr(fn, 'tile_state') else None}", flush=True)
    fn.close()
