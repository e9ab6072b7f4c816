"""Cold-start cost of the library in a FRESH process — what a Julia session pays at `accelerate(system)` and at its first callback
(VERDICT r3 item 7): dlopen of liboctofitter_hip.so, octo_ctx_create (HIP runtime + device initialisation), octo_dataset_create, the
FIRST octo_eval of a one-planet RA/Dec dataset (the runtime loads that translation unit's code object and the kernel's code is fetched
cold), and the second call. Prints one JSON line; bench.py runs it as a subprocess for its `config1.first_call` block.
    python tools/first_call.py"""
import ctypes as C, json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
t0 = time.perf_counter()
from __graft_entry__ import load_package
pkg = load_package(); capi = pkg.capi
lib = capi.load_library()
t_load = time.perf_counter()
ctx = C.c_void_p()
assert lib.octo_ctx_create(C.byref(ctx), 0) == 0
t_ctx = time.perf_counter()
rng = np.random.default_rng(1)
n = 50
ep = 50000.0 + 17.0 * np.arange(n)
obs = [dict(kind=0, planet=0, epoch=ep, y1=rng.normal(0, 300, n), y2=rng.normal(0, 300, n), s1=np.full(n, 10.0), s2=np.full(n, 10.0), cor=None)]
obs_arr, keep = capi.pack_obs(obs)
pl_arr = capi.pack_planets([dict(orbit_kind=0, has_mass=False)])
ds = C.c_void_p()
assert lib.octo_dataset_create(ctx, obs_arr, 1, pl_arr, 1, C.byref(ds)) == 0
t_ds = time.perf_counter()
el = np.array([[10.0], [0.3], [1.0], [0.5], [2.0], [50000.0], [1.2], [50.0], [0.0]]); ll = np.empty(1); g = np.empty_like(el)
args = (ctx, ds, capi._dptr(el), None, 1, 1, capi._dptr(ll), capi._dptr(g), None)
assert lib.octo_eval(*args) == 0
t_first = time.perf_counter()
assert lib.octo_eval(*args) == 0
t_second = time.perf_counter()
for _ in range(200): lib.octo_eval(*args)
t1 = time.perf_counter()
for _ in range(1000): lib.octo_eval(*args)
warm_us = (time.perf_counter() - t1) / 1000 * 1e6
print(json.dumps({"lib_bytes": capi.LIB_PATH.stat().st_size if hasattr(capi.LIB_PATH, "stat") else os.path.getsize(str(capi.LIB_PATH)),
                  "dlopen_ms": (t_load - t0) * 1e3, "ctx_create_ms": (t_ctx - t_load) * 1e3, "dataset_create_ms": (t_ds - t_ctx) * 1e3,
                  "first_call_ms": (t_first - t_ds) * 1e3, "second_call_us": (t_second - t_first) * 1e6, "warm_call_us": warm_us,
                  "finite": bool(np.isfinite(ll[0])),
                  "what": "fresh process: import + dlopen, octo_ctx_create, octo_dataset_create (50 RA/Dec rows), first / second / warm octo_eval (W = 1, fwd+grad)"}))
