"""Latency of ONE walker through octo_eval for multi-planet systems (P = 1..4 companions, one RA/Dec table of 60 rows per planet
+ one absolute-RV table of 200 rows, per-observation nuisances): the small-batch kernel against the same call forced onto the
throughput kernels.   python tools/latency_multi.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import gpu_binding as gb
import stress_parity as sp
capi = gb.capi


def time_call(f, n=3000, warm=300):
    for _ in range(warm): f()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n // 5): f()
        best = min(best, (time.perf_counter() - t0) / (n // 5))
    return best * 1e6


rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
Ws = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1]
PS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3, 4]
for P, W in [(P, W) for P in PS for W in Ws]:
    planets = [dict(orbit_kind=0, has_mass=True) for _ in range(P)]
    elems = np.concatenate([sp.planet_elems(rng, W, 0, 2 + 6 * i, 6 + 6 * i) for i in range(P)])
    obs = []
    for ip in range(P):
        n = 60; ep = np.sort(50000 + rng.uniform(0, 4000, n))
        obs.append(dict(kind=0, planet=ip, epoch=ep, y1=rng.normal(0, 300, n), y2=rng.normal(0, 300, n), s1=rng.uniform(3, 12, n), s2=rng.uniform(3, 12, n), cor=None))
    n = 200; ep = np.sort(50000 + rng.uniform(0, 4000, n))
    obs.append(dict(kind=2, planet=-1, epoch=ep, y1=rng.normal(0, 30, n), y2=None, s1=rng.uniform(1, 8, n), s2=None, cor=None))
    nuis = np.zeros((len(obs) * 3, W))
    for io in range(P):
        nuis[io * 3] = 1.0; nuis[io * 3 + 1] = 1.0
    nuis[P * 3] = 3.0; nuis[P * 3 + 1] = 2.0
    for sb in (1024, 0):
        with gb.GpuPath(obs, planets, small_batch=sb) as g:
            el = np.ascontiguousarray(elems); nu = np.ascontiguousarray(nuis); ll = np.empty(W); ge = np.empty_like(el); gn = np.empty_like(nu)
            args = (g.ctx, g.ds, capi._dptr(el), capi._dptr(nu), W, W, capi._dptr(ll), capi._dptr(ge), capi._dptr(gn))
            us = time_call(lambda: g.lib.octo_eval(*args), n=1000, warm=100)
            print(f"P={P} W={W} rows={60 * P + 200} fwd+grad {'small-batch kernel' if sb else 'throughput kernels '} {us:7.1f} us  ll={ll[0]:.6f}", flush=True)
