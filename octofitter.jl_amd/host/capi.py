"""
ctypes binding of the C ABI in include/octofitter_hip.h.

This is plumbing only: structs, symbol signatures and a loader that FAILS LOUDLY when the
HIP shared library is missing — there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent.parent
LIB_PATH = PKG_DIR / "lib" / "liboctofitter_hip.so"

OCTO_OK, OCTO_EINVAL, OCTO_EHIP, OCTO_ENOMEM, OCTO_ENODEV, OCTO_ENOTSUP = 0, 1, 2, 3, 4, 5
STATUS_NAMES = {0: "OCTO_OK", 1: "OCTO_EINVAL", 2: "OCTO_EHIP", 3: "OCTO_ENOMEM", 4: "OCTO_ENODEV", 5: "OCTO_ENOTSUP"}
FALLBACK_STATUSES = (OCTO_ENODEV, OCTO_ENOTSUP)      # what a host-side binding answers by staying on the reference's path; every other status is the caller's to see

ASTROM_RADEC, ASTROM_SEPPA, RV_ABS, RV_ABS_MARG, RV_REL, ONEIL_RADEC, ONEIL_SEPPA, HGCA = 0, 1, 2, 3, 4, 5, 6, 7
HGCA_RA, HGCA_DEC, HGCA_HIP, HGCA_GAIA, HGCA_N_EXTRA = 0, 1, 0, 1, 15
ASTROM_KINDS = (ASTROM_RADEC, ASTROM_SEPPA, ONEIL_RADEC, ONEIL_SEPPA)
ORBIT_VISUAL_KEP, ORBIT_RADVEL, ORBIT_THIELE_INNES, ORBIT_KEP = 0, 1, 2, 3
N_EL, N_NUIS = 9, 3
MAX_PLANETS = 8      # OCTO_MAX_PLANETS
MAX_PLANETS_ALL_KINDS = 4      # OCTO_MAX_PLANETS_ALL_KINDS: beyond it the planet-per-wave throughput kernels only (every observation kind since round 6; no small-batch family)
EL_A, EL_E, EL_I, EL_W, EL_O, EL_TP, EL_M, EL_PLX, EL_MASS = range(9)
NU_JITTER, NU_PLATESCALE, NU_NORTHANGLE = 0, 1, 2
NU_RV_OFFSET, NU_RV_JITTER, NU_RV_TREND = 0, 1, 2
OPT_BATCH_INVARIANT, OPT_WARM_START, OPT_TILE_SORT, OPT_TILE_MIN_WALKERS = 1, 2, 3, 4      # octo_ctx_set_option

c_double_p = C.POINTER(C.c_double)
STREAM_CTX = C.c_void_p(-1)      # OCTO_STREAM_CTX: the context's own stream (NULL = HIP's NULL stream, e.g. torch's default stream)


class OctoConsts(C.Structure):
    _fields_ = [
        ("kepler_year_to_julian_day", C.c_double),
        ("year2day_julian", C.c_double),
        ("au2m", C.c_double),
        ("sec2year_julian", C.c_double),
        ("pc2au", C.c_double),
        ("rad2as", C.c_double),
        ("mjup2msol", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class OctoObsDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("planet", C.c_int32),
        ("n_epochs", C.c_int64),
        ("epoch", c_double_p),
        ("y1", c_double_p),
        ("y2", c_double_p),
        ("s1", c_double_p),
        ("s2", c_double_p),
        ("cor", c_double_p),
        ("extra", c_double_p),
        ("n_extra", C.c_int64),
    ]


class OctoPlanetDesc(C.Structure):
    _fields_ = [("orbit_kind", C.c_int32), ("has_mass", C.c_int32)]


PRIOR_UNIFORM, PRIOR_LOGUNIFORM, PRIOR_NORMAL, PRIOR_TRUNCNORMAL, PRIOR_SINE = 0, 1, 2, 3, 4
SRC_CONST, SRC_THETA, SRC_CIRCULAR, SRC_TPERI = 0, 1, 2, 3
SRC_FLAG_TI = 2
SRC_FLAG_UNITLEN = 1


class OctoPrior(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pad", C.c_int32), ("p0", C.c_double), ("p1", C.c_double), ("lo", C.c_double), ("hi", C.c_double)]


class OctoSource(C.Structure):
    _fields_ = [("kind", C.c_int32), ("i0", C.c_int32), ("i1", C.c_int32), ("flags", C.c_int32), ("value", C.c_double)]


class OctoError(RuntimeError):
    def __init__(self, status, msg=""):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {msg}")
        self.status = status


def _dptr(a):
    return a.ctypes.data_as(c_double_p) if a is not None else c_double_p()


def pack_obs(obs_tables):
    """obs_tables: list of dicts(kind, planet, epoch, y1, y2, s1, s2, cor) with numpy columns.
    Returns (ctypes array, keepalive list)."""
    n = len(obs_tables)
    arr = (OctoObsDesc * max(n, 1))()
    keep = []
    for k, t in enumerate(obs_tables):
        cols = {}
        for name in ("epoch", "y1", "y2", "s1", "s2", "cor", "extra"):
            v = t.get(name)
            cols[name] = None if v is None else np.ascontiguousarray(v, dtype=np.float64)
        keep.append(cols)
        arr[k].kind = int(t["kind"])
        arr[k].planet = int(t["planet"])
        arr[k].n_epochs = int(cols["epoch"].shape[0])
        for name in ("epoch", "y1", "y2", "s1", "s2", "cor", "extra"):
            setattr(arr[k], name, _dptr(cols[name]))
        arr[k].n_extra = 0 if cols["extra"] is None else int(cols["extra"].shape[0])
    return arr, keep


def pack_planets(planets):
    n = len(planets)
    arr = (OctoPlanetDesc * max(n, 1))()
    for k, p in enumerate(planets):
        arr[k].orbit_kind = int(p["orbit_kind"])
        arr[k].has_mass = int(bool(p["has_mass"]))
    return arr


_SIGS = {
    "octo_consts_default": (C.c_int32, [C.POINTER(OctoConsts)]),
    "octo_version": (C.c_int32, [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "octo_ctx_create": (C.c_int32, [C.POINTER(C.c_void_p), C.c_int32]),
    "octo_ctx_destroy": (C.c_int32, [C.c_void_p]),
    "octo_consts_set": (C.c_int32, [C.c_void_p, C.POINTER(OctoConsts)]),
    "octo_ctx_set_small_batch": (C.c_int32, [C.c_void_p, C.c_int32]),
    "octo_ctx_set_option": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int64]),
    "octo_ctx_get_option": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "octo_host_register": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    "octo_host_unregister": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "octo_last_error": (C.c_char_p, [C.c_void_p]),
    "octo_dataset_create": (C.c_int32, [C.c_void_p, C.POINTER(OctoObsDesc), C.c_int32,
                                        C.POINTER(OctoPlanetDesc), C.c_int32, C.POINTER(C.c_void_p)]),
    "octo_dataset_destroy": (C.c_int32, [C.c_void_p]),
    "octo_dataset_n_rows": (C.c_int64, [C.c_void_p]),
    "octo_eval": (C.c_int32, [C.c_void_p, C.c_void_p, c_double_p, c_double_p, C.c_int64, C.c_int64,
                              c_double_p, c_double_p, c_double_p]),
    "octo_eval_begin": (C.c_int32, [C.c_void_p, C.c_void_p, c_double_p, c_double_p, C.c_int64, C.c_int64,
                                    c_double_p, c_double_p, c_double_p]),
    "octo_eval_end": (C.c_int32, [C.c_void_p]),
    "octo_eval_multi": (C.c_int32, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, c_double_p, c_double_p, C.c_int64, C.c_int64,
                                    c_double_p, c_double_p, c_double_p]),
    "octo_comm_unique_id": (C.c_int32, [C.POINTER(C.c_uint8)]),
    "octo_comm_create": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]),
    "octo_comm_destroy": (C.c_int32, [C.c_void_p]),
    "octo_pt_step_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                        C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "octo_pt_step": (C.c_int32, [C.c_void_p, c_double_p, c_double_p, C.POINTER(C.c_int32), C.c_int32, C.c_int64, C.c_int32, C.c_uint64, C.c_uint64,
                                 C.POINTER(C.c_int32)]),
    "octo_eval_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "octo_sync": (C.c_int32, [C.c_void_p]),
    "octo_kepler_solve": (C.c_int32, [C.c_void_p, c_double_p, c_double_p, C.c_int64, c_double_p, c_double_p, c_double_p]),
    "octo_kepler_solve_table": (C.c_int32, [C.c_void_p, c_double_p, c_double_p, C.c_int64, c_double_p, c_double_p, c_double_p]),
    "octo_ofti_create": (C.c_int32, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int64,
                                     C.c_double, C.POINTER(C.c_void_p)]),
    "octo_ofti_destroy": (C.c_int32, [C.c_void_p]),
    "octo_ofti_eval": (C.c_int32, [C.c_void_p, C.c_void_p, c_double_p, C.c_int64, C.c_int64, c_double_p, c_double_p]),
    "octo_ofti_eval_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "octo_model_create": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(OctoPrior), C.c_int32, C.POINTER(OctoSource), C.POINTER(OctoSource),
                                      C.POINTER(C.c_void_p)]),
    "octo_model_destroy": (C.c_int32, [C.c_void_p]),
    "octo_model_logpost": (C.c_int32, [C.c_void_p, C.c_void_p, c_double_p, C.c_int64, C.c_int64, c_double_p, c_double_p]),
    "octo_model_logpost_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "octo_timing_enable": (C.c_int32, [C.c_void_p, C.c_int32]),
    "octo_timing_read": (C.c_int32, [C.c_void_p, c_double_p, C.POINTER(C.c_int64), C.c_int32]),
    "octo_timing_stats": (C.c_int32, [C.c_void_p, c_double_p, c_double_p, c_double_p, C.POINTER(C.c_int64)]),
    "octo_pt_swap_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                        C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None


def load_library(path=None):
    """Load liboctofitter_hip.so. Raises if it has not been built — never falls back."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path or os.environ.get("OCTOFITTER_HIP_LIB", LIB_PATH))
    if not p.exists():
        raise FileNotFoundError(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "The product path has no CPU fallback.")
    lib = C.CDLL(str(p), mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if path is None and os.environ.get("OCTOFITTER_HIP_LIB"):      # an older build selected for a same-box A/B (tools/): its symbols only
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def default_consts(lib=None):
    lib = lib or load_library()
    c = OctoConsts()
    st = lib.octo_consts_default(C.byref(c))
    if st != OCTO_OK:
        raise OctoError(st, "octo_consts_default")
    return c
