"""Latency of one walker (and of 64) through octo_eval for the two dataset families with their own pre-pass: an RA/Dec table
+ the HGCA proper-motion anomaly (the common joint fit, src/likelihoods/hgca.jl) and an RA/Dec table + a marginalised
absolute-RV table (rv-absolute-margin.jl).   python tools/latency_hgca_marg.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import gpu_binding as gb
import stress_parity as sp
capi = gb.capi


def time_call(f, n=2000, warm=200):
    for _ in range(warm): f()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n // 5): f()
        best = min(best, (time.perf_counter() - t0) / (n // 5))
    return best * 1e6


rng = np.random.default_rng(5)
for what in ("astrometry only", "astrometry + HGCA", "astrometry + marginalised RV"):
    for W in (1, 64):
        planets = [dict(orbit_kind=0, has_mass=True)]
        elems = sp.planet_elems(rng, W, 0, 5, 12)
        n = 60; ep = np.sort(50000 + rng.uniform(0, 4000, n))
        obs = [dict(kind=0, planet=0, epoch=ep, y1=rng.normal(0, 300, n), y2=rng.normal(0, 300, n), s1=rng.uniform(3, 12, n), s2=rng.uniform(3, 12, n), cor=None)]
        if "HGCA" in what:
            rows = []
            for d in np.linspace(-700, 700, 3): rows += [(48348.0 + d, 0, 0), (48414.0 + d, 1, 0)]
            for d in np.linspace(-500, 500, 3): rows += [(57408.0 + d, 0, 1), (57470.0 + d, 1, 1)]
            rows = np.array(rows)
            obs.append(dict(kind=7, planet=-1, epoch=rows[:, 0], y1=rows[:, 1], y2=rows[:, 2], s1=None, s2=None, cor=None, extra=sp.HG))
        if "RV" in what:
            m = 200; ep2 = np.sort(50000 + rng.uniform(0, 4000, m))
            obs.append(dict(kind=3, planet=-1, epoch=ep2, y1=rng.normal(0, 30, m), y2=None, s1=rng.uniform(1, 8, m), s2=None, cor=None))
        nuis = np.zeros((len(obs) * 3, W)); nuis[1] = 1.0
        if len(obs) > 1:
            nuis[3] = 4.3 if "HGCA" in what else 0.0; nuis[4] = -2.0 if "HGCA" in what else 2.0
        with gb.GpuPath(obs, planets) as g:
            el = np.ascontiguousarray(elems); nu = np.ascontiguousarray(nuis); ll = np.empty(W); ge = np.empty_like(el); gn = np.empty_like(nu)
            args = (g.ctx, g.ds, capi._dptr(el), capi._dptr(nu), W, W, capi._dptr(ll), capi._dptr(ge), capi._dptr(gn))
            us = time_call(lambda: g.lib.octo_eval(*args))
            print(f"{what:30s} W={W:3d} fwd+grad {us:7.1f} us  ll[0]={ll[0]:.6f}", flush=True)
