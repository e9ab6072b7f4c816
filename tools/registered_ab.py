"""octo_eval on registered host arrays (SURVEY §8d's PCIe-inclusive call) against the device-resident step, config 3 at 1e4 x 1e4:
median of blocking calls; run under OCTOFITTER_HIP_LIB=<build> for a same-box A/B.   python tools/registered_ab.py"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from __graft_entry__ import load_package
import synth
pkg = load_package(); capi = pkg.capi
tag = os.environ.get("OCTOFITTER_HIP_LIB", "default")[-28:]
W, E = 10000, 10000
cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3)
obs, planet = synth.to_mirror(pkg, cfg)
fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
el_h = np.array(cfg["elems"], order="C", copy=True); el = torch.tensor(el_h, device="cuda")      # a COPY: the timed calls perturb el_h[0, 0]
out = (torch.empty(W, dtype=torch.float64, device="cuda"), torch.empty_like(el), None)
for _ in range(100): fn.ln_like_device(el, None, grad=True, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): fn.ln_like_device(el, None, grad=True, out=out)
torch.cuda.synchronize(); dev = (time.perf_counter() - t0) / 200
lat = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(50):
        fn.ln_like_device(el, None, grad=True, out=out); torch.cuda.synchronize()
    lat = min(lat, (time.perf_counter() - t0) / 50)
ll_ref = out[0].cpu().numpy().copy(); g_ref = out[1].cpu().numpy().copy()
ll_h = np.empty(W); g_h = np.empty_like(el_h)
a_ = (fn._ctx, fn._ds, capi._dptr(el_h), None, W, W, capi._dptr(ll_h), capi._dptr(g_h), None)
fn.host_register(el_h, ll_h, g_h)
for _ in range(20): fn.lib.octo_eval(*a_)
ts = []
for _ in range(100):
    el_h[0, 0] += 1e-9      # the inputs change from call to call (a stale cached copy would show)
    t1 = time.perf_counter(); fn.lib.octo_eval(*a_); ts.append(time.perf_counter() - t1)
reg = float(np.median(ts))
fn.timing_read(reset=True); fn.timing_enable(-1)      # SURVEY 8(d)'s clock: device time of the whole call (HIP events inside the library)
for _ in range(100):
    el_h[0, 0] += 1e-9
    fn.lib.octo_eval(*a_)
dev_med, dev_min, dev_max, dev_n = fn.timing_stats()
fn.timing_read(reset=True); fn.timing_enable(0)
el_h[:] = cfg["elems"]; ll_h[:] = np.nan
fn.lib.octo_eval(*a_)
same = bool(np.array_equal(ll_h, ll_ref, equal_nan=True) and np.array_equal(g_h, g_ref, equal_nan=True))
if not same:
    bad = ~((ll_h == ll_ref) | (np.isnan(ll_h) & np.isnan(ll_ref)))
    print("ll differs at", int(bad.sum()), "walkers; first:", np.flatnonzero(bad)[:5], ll_h[bad][:3], ll_ref[bad][:3],
          "| gradient entries that differ:", int((~((g_h == g_ref) | (np.isnan(g_h) & np.isnan(g_ref)))).sum()), flush=True)
fn.host_unregister(el_h, ll_h, g_h)
tag += " ahead=" + os.environ.get("OCTO_STAGE_AHEAD", "default")
print(f"{tag:>36}: device-resident back-to-back {dev*1e6:7.1f} us, one at a time {lat*1e6:7.1f} us | octo_eval registered: device clock {dev_med*1e3:7.1f} us "
      f"(ratio {dev/(dev_med*1e-3):.3f}), blocking call {reg*1e6:7.1f} us (ratio {dev/reg:.3f}) bit-identical to the device-resident result: {same}", flush=True)
fn.close()
