"""Thin test helper over the C ABI (ctypes): one context + dataset per call set. Used by -m gpu tests."""
from __future__ import annotations

import ctypes as C

import numpy as np

from __graft_entry__ import load_package

pkg = load_package()
capi = pkg.capi


import os
# tests that want every case through BOTH kernel families set this (None: the library's own choice; the stand-alone sweeps take it
# from OCTO_TEST_SMALL_BATCH, e.g. 0 = everything on the throughput kernels)
DEFAULT_SMALL_BATCH = int(os.environ["OCTO_TEST_SMALL_BATCH"]) if os.environ.get("OCTO_TEST_SMALL_BATCH") else None


# Set to a number (NaN, 1e300) to fill every CU's LDS with it ahead of each evaluation (octo_debug_poison_lds, a test hook of the
# library): a kernel that reads an LDS word it has not written then returns something else than with the usual stale zeros.
POISON_LDS = float(os.environ["OCTO_TEST_POISON_LDS"]) if os.environ.get("OCTO_TEST_POISON_LDS") else None      # the stand-alone sweeps: nan / 1e300


def poison(ctx):
    if POISON_LDS is None:
        return
    lib = capi.load_library()
    lib.octo_debug_poison_lds.restype = C.c_int32
    lib.octo_debug_poison_lds.argtypes = [C.c_void_p, C.c_double]
    st = lib.octo_debug_poison_lds(ctx, float(POISON_LDS))
    if st != 0:
        raise capi.OctoError(st, "octo_debug_poison_lds")


class GpuPath:
    def __init__(self, obs_tables, planets, device=0, consts=None, small_batch=None, options=None):
        if small_batch is None:
            small_batch = DEFAULT_SMALL_BATCH
        self.lib = capi.load_library()
        self.ctx = C.c_void_p()
        st = self.lib.octo_ctx_create(C.byref(self.ctx), device)
        if st != 0:
            raise capi.OctoError(st, "octo_ctx_create")
        if consts is not None:
            self._chk(self.lib.octo_consts_set(self.ctx, C.byref(consts)))
        if small_batch is not None:      # 0: force the throughput kernels (lane = walker) also for tiny batches
            self._chk(self.lib.octo_ctx_set_small_batch(self.ctx, int(small_batch)))
        for opt, val in (options or {}).items():      # octo_ctx_set_option (capi.OPT_*)
            self._chk(self.lib.octo_ctx_set_option(self.ctx, int(opt), int(val)))
        obs_arr, keep = capi.pack_obs(obs_tables)
        pl_arr = capi.pack_planets(planets)
        self.ds = C.c_void_p()
        self._chk(self.lib.octo_dataset_create(self.ctx, obs_arr, len(obs_tables), pl_arr, len(planets), C.byref(self.ds)))
        self.n_obs, self.n_planets = len(obs_tables), len(planets)

    def _chk(self, st):
        if st != 0:
            raise capi.OctoError(st, (self.lib.octo_last_error(self.ctx) or b"").decode())

    def eval(self, elems, nuis=None, grad=True):
        elems = np.ascontiguousarray(elems, dtype=np.float64)
        W = elems.shape[1]
        nu = None if nuis is None else np.ascontiguousarray(nuis, dtype=np.float64)
        ll = np.full(W, np.nan)
        g_el = np.full_like(elems, np.nan) if grad else None
        g_nu = np.full_like(nu, np.nan) if (grad and nu is not None) else None
        poison(self.ctx)
        self._chk(self.lib.octo_eval(self.ctx, self.ds, capi._dptr(elems), capi._dptr(nu), W, W,
                                     capi._dptr(ll), capi._dptr(g_el), capi._dptr(g_nu)))
        return ll, g_el, g_nu

    def tile_state(self):
        """(sorted launches, probes, sort on?, the last probe's estimated saving in µs) — octo_debug_tile_state, a test hook."""
        n, pr, on, sv = C.c_int64(), C.c_int64(), C.c_int32(), C.c_double()
        self.lib.octo_debug_tile_state.restype = C.c_int32
        self.lib.octo_debug_tile_state.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        self._chk(self.lib.octo_debug_tile_state(self.ctx, C.byref(n), C.byref(pr), C.byref(on), C.byref(sv)))
        return n.value, pr.value, bool(on.value), sv.value

    def close(self):
        if self.ds:
            self.lib.octo_dataset_destroy(self.ds); self.ds = None
        if self.ctx:
            self.lib.octo_ctx_destroy(self.ctx); self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def gpu_eval(obs_tables, planets, elems, nuis=None, grad=True, consts=None, small_batch=None, options=None):
    with GpuPath(obs_tables, planets, consts=consts, small_batch=small_batch, options=options) as g:
        return g.eval(elems, nuis, grad)
