"""ctypes binding of oracle/liboctooracle.so — the CHECKER. Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
capi = pkg.capi

ORACLE_DIR = ROOT / "oracle"
ORACLE_LIB = ORACLE_DIR / "liboctooracle.so"
_lib = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR)], check=True)


def load_oracle():
    global _lib
    if _lib is not None:
        return _lib
    if not ORACLE_LIB.exists():
        build_oracle()
    lib = C.CDLL(str(ORACLE_LIB))
    dp = capi.c_double_p
    lib.octo_oracle_eval.restype = C.c_int32
    lib.octo_oracle_eval.argtypes = [C.POINTER(capi.OctoConsts), C.POINTER(capi.OctoObsDesc), C.c_int32,
                                     C.POINTER(capi.OctoPlanetDesc), C.c_int32, dp, dp, C.c_int64, C.c_int64,
                                     dp, dp, dp, C.POINTER(C.c_uint8), C.c_int32]
    lib.octo_oracle_kepler_markley.restype = C.c_double
    lib.octo_oracle_kepler_markley.argtypes = [C.c_double, C.c_double]
    lib.octo_oracle_orbitsolve.restype = C.c_int32
    lib.octo_oracle_orbitsolve.argtypes = [C.POINTER(capi.OctoConsts), C.c_int32, dp, C.c_double, dp]
    lib.octo_oracle_consts_default.restype = C.c_int32
    lib.octo_oracle_consts_default.argtypes = [C.POINTER(capi.OctoConsts)]
    lib.octo_oracle_model_logpost.restype = C.c_int32
    lib.octo_oracle_model_logpost.argtypes = [C.POINTER(capi.OctoConsts), C.POINTER(capi.OctoObsDesc), C.c_int32,
                                              C.POINTER(capi.OctoPlanetDesc), C.c_int32, C.POINTER(capi.OctoPrior), C.c_int32,
                                              C.POINTER(capi.OctoSource), C.POINTER(capi.OctoSource), dp, C.c_int64, C.c_int64, dp, dp, C.c_int32]
    lib.octo_oracle_ofti.restype = C.c_int32
    lib.octo_oracle_ofti.argtypes = [C.POINTER(capi.OctoConsts), dp, dp, dp, dp, dp, dp, C.c_int64, C.c_double, dp, dp]
    _lib = lib
    return lib


def oracle_consts():
    c = capi.OctoConsts()
    assert load_oracle().octo_oracle_consts_default(C.byref(c)) == 0
    return c


def oracle_eval(obs_tables, planets, elems, nuis=None, grad=True, active=None, consts=None, n_threads=1):
    """elems: [n_planets*9, W] float64; nuis: [n_obs*3, W] or None.
    Returns ll[W], g_elems (or None), g_nuis (or None)."""
    lib = load_oracle()
    consts = consts or oracle_consts()
    elems = np.ascontiguousarray(elems, dtype=np.float64)
    W = elems.shape[1]
    obs_arr, keep = capi.pack_obs(obs_tables)
    pl_arr = capi.pack_planets(planets)
    ll = np.empty(W)
    g_el = np.zeros_like(elems) if grad else None
    nu = None if nuis is None else np.ascontiguousarray(nuis, dtype=np.float64)
    g_nu = np.zeros_like(nu) if (grad and nu is not None) else None
    mask = None
    if active is not None:
        mask = np.ascontiguousarray(active, dtype=np.uint8)
    # The restatement carries at most 64 forward-mode partials per pass (a ForwardDiff chunk): systems of more than ~5 planets have more
    # inputs than that, so their gradient is gathered in several passes over disjoint sets of active inputs.
    n_in = elems.shape[0] + (0 if nu is None else nu.shape[0])
    act = np.ones(n_in, dtype=np.uint8) if mask is None else mask.copy()
    if nu is None:
        act[elems.shape[0]:] = 0
    if grad and int(act.sum()) > 64:
        idx = np.nonzero(act)[0]
        for c0 in range(0, idx.size, 64):
            m = np.zeros(n_in, dtype=np.uint8); m[idx[c0:c0 + 64]] = 1
            ll_c, ge_c, gn_c = oracle_eval(obs_tables, planets, elems, nuis, grad=True, active=m, consts=consts, n_threads=n_threads)
            ll = ll_c
            sel = m[:elems.shape[0]].astype(bool)
            g_el[sel] = ge_c[sel]
            if g_nu is not None:
                seln = m[elems.shape[0]:].astype(bool)
                g_nu[seln] = gn_c[seln]
        del keep
        return ll, g_el, g_nu
    st = lib.octo_oracle_eval(C.byref(consts), obs_arr, len(obs_tables), pl_arr, len(planets),
                              capi._dptr(elems), capi._dptr(nu), W, W, capi._dptr(ll), capi._dptr(g_el), capi._dptr(g_nu),
                              mask.ctypes.data_as(C.POINTER(C.c_uint8)) if mask is not None else None, n_threads)
    assert st == 0, st
    del keep
    return ll, g_el, g_nu


def oracle_orbitsolve(el9, t, orbit_kind=0, consts=None):
    lib = load_oracle()
    consts = consts or oracle_consts()
    el9 = np.ascontiguousarray(el9, dtype=np.float64)
    out = np.empty(10)
    assert lib.octo_oracle_orbitsolve(C.byref(consts), orbit_kind, capi._dptr(el9), float(t), capi._dptr(out)) == 0
    return dict(zip(["MA", "EA", "nu", "r", "raoff", "decoff", "radvel", "n", "K", "cart2angle"], out))


def oracle_ofti(epochs, ra, dec, s_ra, s_dec, cor, sigma_abfg, nl, consts=None):
    """nl: [5, W] = e, a, tp, M, plx. Returns abfg [4, W], logml [W] (reference-order restatement, one walker at a time)."""
    lib = load_oracle()
    consts = consts or oracle_consts()
    cols = [np.ascontiguousarray(v, dtype=np.float64) for v in (epochs, ra, dec, s_ra, s_dec)]
    cc = None if cor is None else np.ascontiguousarray(cor, dtype=np.float64)
    nl = np.ascontiguousarray(nl, dtype=np.float64)
    W = nl.shape[1]
    abfg = np.empty((4, W)); lm = np.empty(W)
    out = np.empty(5)
    for w in range(W):
        x = np.ascontiguousarray(nl[:, w])
        assert lib.octo_oracle_ofti(C.byref(consts), *[capi._dptr(v) for v in cols], capi._dptr(cc), len(cols[0]), float(sigma_abfg),
                                    capi._dptr(x), capi._dptr(out)) == 0
        abfg[:, w] = out[:4]; lm[w] = out[4]
    return abfg, lm


def make_priors(priors):
    """list of dict(kind, p0, p1, lo, hi) -> ctypes array (None bounds = open)."""
    arr = (capi.OctoPrior * len(priors))()
    for k, p in enumerate(priors):
        arr[k].kind = int(p["kind"]); arr[k].p0 = float(p["p0"]); arr[k].p1 = float(p["p1"])
        arr[k].lo = -np.inf if p["lo"] is None else float(p["lo"]); arr[k].hi = np.inf if p["hi"] is None else float(p["hi"])
    return arr


def make_sources(srcs):
    arr = (capi.OctoSource * max(len(srcs), 1))()
    for k, s in enumerate(srcs):
        arr[k].kind, arr[k].i0, arr[k].i1, arr[k].flags, arr[k].value = int(s["kind"]), int(s["i0"]), int(s["i1"]), int(s["flags"]), float(s["value"])
    return arr


def oracle_model_logpost(obs_tables, planets, priors, esrc, nsrc, theta_t, grad=True, consts=None, n_threads=1):
    """priors / esrc / nsrc: ctypes arrays (make_priors / make_sources, or a LogDensityModel mirror's _c_* arrays)."""
    lib = load_oracle()
    consts = consts or oracle_consts()
    th = np.ascontiguousarray(theta_t, dtype=np.float64)
    D, W = th.shape
    obs_arr, keep = capi.pack_obs(obs_tables)
    pl_arr = capi.pack_planets(planets)
    lp = np.empty(W)
    g = np.zeros_like(th) if grad else None
    st = lib.octo_oracle_model_logpost(C.byref(consts), obs_arr, len(obs_tables), pl_arr, len(planets), priors, D, esrc, nsrc,
                                       capi._dptr(th), W, W, capi._dptr(lp), capi._dptr(g), n_threads)
    assert st == 0, st
    del keep
    return lp, g
