"""As tools/latency_vs_w.py but through octo_eval_device + octo_sync (inputs and outputs resident in HBM): the kernels' share of
the latency, without staging and PCIe.   python tools/latency_vs_w_device.py [W,W,...] [small_batch]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import gpu_binding as gb
import synth
capi = gb.capi
import ctypes as C


def time_call(f, n=1000, warm=100):
    for _ in range(warm): f()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n // 5): f()
        best = min(best, (time.perf_counter() - t0) / (n // 5))
    return best * 1e6


Ws = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 32, 33, 64, 128, 256, 512, 1024, 4096]
SB = int(sys.argv[2]) if len(sys.argv) > 2 else None
for E in (50, 300, 10000):
    for W in Ws:
        cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3)
        t = cfg["table"]
        obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
        with gb.GpuPath(obs, [dict(orbit_kind=0, has_mass=False)], small_batch=SB) as g:
            el = torch.tensor(cfg["elems"], device="cuda"); ll = torch.empty(W, device="cuda", dtype=torch.float64); ge = torch.empty_like(el)
            p = lambda x: C.c_void_p(x.data_ptr())
            args = (g.ctx, g.ds, p(el), None, W, W, p(ll), p(ge), None, C.c_void_p(-1))

            def f():
                g.lib.octo_eval_device(*args); g.lib.octo_sync(g.ctx)
            us = time_call(f, n=1000 if W * E < 4e6 else 200)
            print(f"E={E:6d} W={W:5d} small_batch={SB} device-resident fwd+grad {us:8.1f} us", flush=True)
