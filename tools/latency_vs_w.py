"""Latency of octo_eval (host buffers, blocking; fwd+grad) as a function of the batch size W, for a 50-, a 300- and a 1e4-epoch
RA/Dec table: where the small-batch kernel hands over to the throughput kernels and how the latter behave for mid-size batches
(an ensemble sampler's 100-1000 walkers).   python tools/latency_vs_w.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import gpu_binding as gb
import synth
capi = gb.capi


def time_call(f, n=1000, warm=100):
    for _ in range(warm): f()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n // 5): f()
        best = min(best, (time.perf_counter() - t0) / (n // 5))
    return best * 1e6


Ws = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 32, 33, 64, 128, 256, 512, 1024, 4096]
SB = int(sys.argv[2]) if len(sys.argv) > 2 else None      # octo_ctx_set_small_batch: largest batch that takes k_small
for E in (50, 300, 10000):
    for W in Ws:
        cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3)
        t = cfg["table"]
        obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
        with gb.GpuPath(obs, [dict(orbit_kind=0, has_mass=False)], small_batch=SB) as g:
            el = np.ascontiguousarray(cfg["elems"]); ll = np.empty(W); ge = np.empty_like(el)
            args = (g.ctx, g.ds, capi._dptr(el), None, W, W, capi._dptr(ll), capi._dptr(ge), None)
            us = time_call(lambda: g.lib.octo_eval(*args), n=1000 if W * E < 4e6 else 200)
            print(f"E={E:6d} W={W:5d} small_batch={SB} fwd+grad {us:8.1f} us   {W * E / us * 1e6:.3e} evals/s", flush=True)
