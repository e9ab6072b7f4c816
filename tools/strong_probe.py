"""Per-GPU share of a strong-scaled config 3 (SURVEY §8d "Scaling runs": 1e4 walkers split evenly over N GPUs): step time of the
device-resident call and of octo_eval with registered host buffers at W = 1e4 / N for N = 1, 2, 4, 8, E = 1e4, fwd+grad, and the
projected speed-up N x rate(W/N) / rate(W). One GPU is enough to measure it: the shards are independent (system.jl:206-241).
   python tools/strong_probe.py [--kernel-times]"""
import ctypes as C, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from __graft_entry__ import load_package
import synth
pkg = load_package(); capi = pkg.capi
tag = os.environ.get("OCTOFITTER_HIP_LIB", "default")[-30:]
E = 10000
cfg = synth.config_astrom(n_epochs=E, n_walkers=10000, cfg=3)
obs, planet = synth.to_mirror(pkg, cfg)
fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
base = {}
for grad in (True, False):
    for N in ((1, 2, 4, 8) if not os.environ.get("OCTO_PROBE_N") else tuple(int(x) for x in os.environ["OCTO_PROBE_N"].split(","))):
        W = 10000 // N
        el_h = np.ascontiguousarray(cfg["elems"][:, :W]); el = torch.tensor(el_h, device="cuda")
        out = (torch.empty(W, dtype=torch.float64, device="cuda"), torch.empty_like(el) if grad else None, None)
        for _ in range(100): fn.ln_like_device(el, None, grad=grad, out=out)
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): fn.ln_like_device(el, None, grad=grad, out=out)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
        fn.timing_enable(1)
        for _ in range(50): fn.ln_like_device(el, None, grad=grad, out=out)
        torch.cuda.synchronize()
        kmed = fn.timing_stats()[0]; fn.timing_read(reset=True); fn.timing_enable(0)
        # one call at a time (what a sampler that waits for its gradient sees): launch + sync per call
        lat = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(100):
                fn.ln_like_device(el, None, grad=grad, out=out); torch.cuda.synchronize()
            lat = min(lat, (time.perf_counter() - t0) / 100)
        ll_h = np.empty(W); g_h = np.empty_like(el_h)
        a_ = (fn._ctx, fn._ds, capi._dptr(el_h), None, W, W, capi._dptr(ll_h), capi._dptr(g_h) if grad else None, None)
        fn.host_register(el_h, ll_h, g_h)
        for _ in range(20): fn.lib.octo_eval(*a_)
        ts = []
        for _ in range(60):
            t1 = time.perf_counter(); fn.lib.octo_eval(*a_); ts.append(time.perf_counter() - t1)
        reg = float(np.median(ts))
        fn.host_unregister(el_h, ll_h, g_h)
        rate = W * E / best
        if N == 1: base[grad] = (rate, W * E / reg, W * E / lat)
        print(f"{tag:>30} grad={int(grad)} N={N} W={W:5d}: back-to-back {best*1e6:7.1f} us ({rate:.3e}/s, x{N*rate/base[grad][0]:.2f})  "
              f"k_main {kmed*1e3:6.1f} us  one-at-a-time {lat*1e6:7.1f} us (x{N*W*E/lat/base[grad][2]:.2f})  "
              f"octo_eval registered {reg*1e6:7.1f} us (x{N*W*E/reg/base[grad][1]:.2f})", flush=True)
fn.close()
