"""octo_eval with registered host arrays at 1e4 x 1e4 (fwd+grad): per-call time; run under rocprofv3 --kernel-trace --stats to see what the
PCIe legs cost inside k_copy_in and k_finish.   python tools/registered_probe.py [n_calls]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from __graft_entry__ import load_package
import synth
pkg = load_package(); capi = pkg.capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = synth.config_astrom(n_epochs=10000, n_walkers=10000, cfg=3)
obs, planet = synth.to_mirror(pkg, cfg)
fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
W = 10000
el_h = np.ascontiguousarray(cfg["elems"]); ll_h = np.empty(W); g_h = np.empty_like(el_h)
a_ = (fn._ctx, fn._ds, capi._dptr(el_h), None, W, W, capi._dptr(ll_h), capi._dptr(g_h), None)
fn.host_register(el_h, ll_h, g_h)
for _ in range(20): fn.lib.octo_eval(*a_)
ts = []
for _ in range(n):
    t1 = time.perf_counter(); fn.lib.octo_eval(*a_); ts.append(time.perf_counter() - t1)
print(f"registered octo_eval 1e4 x 1e4 fwd+grad: median {np.median(ts)*1e6:.1f} us, min {np.min(ts)*1e6:.1f} us over {n} calls")
fn.host_unregister(el_h, ll_h, g_h)
fn.close()
