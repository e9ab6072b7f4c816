"""Calibration of the planner's row weights (octo_api.hip: row_cost): step time against OCTO_RV_COST (cost of an RV row in percent of an
RA/Dec row) for datasets that mix an RA/Dec table with an absolute-RV table — one and two planets, with and without per-walker
nuisances, fwd+grad. The optimum is where the astrometry and the RV blocks of a one-round launch finish together. Development aid.
   python tools/sweep_rv_cost.py"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from __graft_entry__ import load_package
import synth
pkg = load_package()


def build(n_planets):
    c4 = synth.config_two_planet()
    astrom = pkg.PlanetRelAstromObs(c4["astrom"], name="astrom")
    rv = pkg.StarAbsoluteRVObs(c4["rv"], name="rv")
    if n_planets == 2:
        b = pkg.Planet(name="b", observations=[]); c = pkg.Planet(name="c", observations=[astrom])
        system = pkg.System(name="cfg4", companions=[b, c], observations=[rv])
        θex = dict(M=1.2, plx=50.0, planets=dict(b=dict(a=3, e=0.1, i=1, ω=1, Ω=2, tp=5e4, mass=5), c=dict(a=15, e=0.3, i=1, ω=.5, Ω=2, tp=5e4, mass=10)))
        elems = c4["elems"]
    else:
        c = pkg.Planet(name="c", observations=[astrom])
        system = pkg.System(name="one", companions=[c], observations=[rv])
        θex = dict(M=1.2, plx=50.0, planets=dict(c=dict(a=15, e=0.3, i=1, ω=.5, Ω=2, tp=5e4, mass=10)))
        elems = c4["elems"][9:]
    return system, θex, elems, c4["nuis"], c4["n_rows"]


for n_planets in (2, 1):
    for with_nuis in (True, False):
        for cost in (0, 60, 70, 80, 90, 100, 115, 135, 160):
            if cost: os.environ["OCTO_RV_COST"] = str(cost)
            else: os.environ.pop("OCTO_RV_COST", None)
            system, θex, elems_h, nuis_h, n_rows = build(n_planets)
            fn = pkg.make_ln_like(system, θex)
            W = elems_h.shape[1]
            el = torch.tensor(np.ascontiguousarray(elems_h), device="cuda")
            nu = torch.tensor(nuis_h, device="cuda") if with_nuis else None
            out = (torch.empty(W, dtype=torch.float64, device="cuda"), torch.empty_like(el), torch.empty_like(nu) if nu is not None else None)
            for _ in range(60): fn.ln_like_device(el, nu, grad=True, out=out)
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(150): fn.ln_like_device(el, nu, grad=True, out=out)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 150)
            print(f"P={n_planets} nuis={int(with_nuis)} OCTO_RV_COST={cost or 'model':>5}: {best*1e6:8.1f} us/step  {W*n_rows/best:.3e} evals/s", flush=True)
            fn.close()
