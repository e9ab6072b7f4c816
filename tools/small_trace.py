"""Phase timing inside k_small (development build with -DOCTO_SMALL_TRACE: tools/liboctofitter_trace.bin): core-clock stamps of the
finishing block of walker 0 — where one small call's microseconds go."""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ["OCTOFITTER_HIP_LIB"] = str(ROOT / "tools" / "liboctofitter_trace.bin")
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from __graft_entry__ import load_package
import synth
pkg = load_package(); capi = pkg.capi
names = ["start", "setup", "rows", "block-reduce", "last-known", "obs-finish", "outputs", "flag"]
for E, W in ((50, 1), (10000, 1), (10000, 32)):
    cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3)
    obs, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
    fn.lib.octo_debug_small_trace.restype = C.POINTER(C.c_uint64); fn.lib.octo_debug_small_trace.argtypes = [C.c_void_p]
    el = np.ascontiguousarray(cfg["elems"]); ll = np.empty(W); g = np.empty_like(el)
    args = (fn._ctx, fn._ds, capi._dptr(el), None, W, W, capi._dptr(ll), capi._dptr(g), None)
    acc = np.zeros(8); n = 0
    for it in range(300):
        fn.lib.octo_eval(*args)
        if it >= 100:
            fn.sync()
            t = np.array([fn.lib.octo_debug_small_trace(fn._ctx)[k] for k in range(8)], dtype=np.float64)
            acc += t - t[0]; n += 1
    acc /= n
    print(f"E={E} W={W}: " + "  ".join(f"{nm} {acc[k] / 2400:.2f}us" for k, nm in enumerate(names)), flush=True)
    fn.close()
