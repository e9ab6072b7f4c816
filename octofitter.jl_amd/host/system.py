"""
Host-side mirror of the reference's model containers and of `make_ln_like`, batched.

  Planet, System           src/variables.jl:461-508, 536-594 (containers only; priors/derived stay in Julia)
  make_ln_like             src/likelihoods/system.jl:21-242
  BatchedLnLike.__call__   the generated closure of system.jl:206-241 applied to W parameter sets,
                           i.e. what ℓπcallback adds at src/logdensitymodel.jl:134
  BatchedLnLike.ln_like_and_grad
                           the likelihood part of ∇ℓπcallback (src/logdensitymodel.jl:169-177):
                           value + gradient w.r.t. the resolved orbital elements and nuisances.

θ for a batch of W walkers is the reference's nested NamedTuple with every leaf a length-W
array (or a scalar, broadcast):

    θ = dict(M=..., plx=..., observations={obsname: dict(offset=..., jitter=...)},
             planets={"b": dict(a=..., e=..., i=..., ω=..., Ω=..., tp=..., mass=...,
                                observations={obsname: dict(jitter=..., platescale=..., northangle=...)})})

All compute goes through the C ABI (include/octofitter_hip.h). There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .observations import AbstractObs, normalizename

_EL_KEYS = (("a",), ("e",), ("i",), ("ω", "w", "omega"), ("Ω", "O", "Omega"), ("tp",), ("M",), ("plx",), ("mass",))
# ThieleInnesOrbit(; e, tp, M, plx, A, B, F, G): the constants [mas] travel in the rows of a, i, ω, Ω (include/octofitter_hip.h)
_EL_KEYS_TI = (("A",), ("e",), ("B",), ("F",), ("G",), ("tp",), ("M",), ("plx",), ("mass",))
_BASIS = {"Visual{KepOrbit}": capi.ORBIT_VISUAL_KEP, "RadialVelocityOrbit": capi.ORBIT_RADVEL, "ThieleInnesOrbit": capi.ORBIT_THIELE_INNES,
          "KepOrbit": capi.ORBIT_KEP}      # plain KepOrbit (a, e, i, ω, Ω, tp, M): no parallax, so RV tables only


def el_keys(basis):
    return _EL_KEYS_TI if _BASIS[basis] == capi.ORBIT_THIELE_INNES else _EL_KEYS


class Planet:
    def __init__(self, *, name, basis="Visual{KepOrbit}", observations=(), variables=None):
        if basis not in _BASIS:
            raise NotImplementedError(f"basis {basis!r} is not on the HIP path (supported: {sorted(_BASIS)})")
        self.name = str(name)
        self.basis = basis
        self.observations = tuple(observations)
        self.variables = variables
        for o in self.observations:
            if not isinstance(o, AbstractObs):
                raise TypeError(f"planet observation {o!r} is not an AbstractObs")
            if o.kind in (capi.RV_ABS, capi.RV_ABS_MARG, capi.HGCA):
                raise ValueError(f"{type(o).__name__} is a system-level observation")


class System:
    def __init__(self, *, name, companions=(), observations=(), variables=None):
        self.name = str(name)
        self.planets = tuple(companions)
        self.observations = tuple(observations)
        self.variables = variables
        names = [p.name for p in self.planets]
        if len(set(names)) != len(names):
            raise ValueError("planet names must be unique")


def _lookup(d, keys):
    for k in keys:
        if k in d:
            return d[k]
    return None


class BatchedLnLike:
    """Callable returned by make_ln_like. Holds one HIP context and the uploaded dataset."""

    def __init__(self, system: System, θ_example: dict, device: int = 0, consts: capi.OctoConsts | None = None):
        self.system = system
        self.device_index = int(device)
        self.lib = capi.load_library()
        # ---- epoch gather in the reference's standardised order (system.jl:35-54) ----------------
        self.all_epochs = []
        self.epoch_start_index_mapping = {}
        j = 1
        for obs in system.observations:
            self.epoch_start_index_mapping[id(obs)] = j
            j += len(obs)
            self.all_epochs.extend(obs.table["epoch"].tolist())
        for pl in system.planets:
            for obs in pl.observations:
                self.epoch_start_index_mapping[id(obs)] = j
                j += len(obs)
                self.all_epochs.extend(obs.table["epoch"].tolist())
        # ---- evaluation order: planet observations planet by planet, then system (system.jl:229-235)
        self.obs_entries = []   # (obs, planet_index or -1, planet_name or None, θ_obs key)
        for ip, pl in enumerate(system.planets):
            for obs in pl.observations:
                if obs.kind in (capi.RV_ABS, capi.RV_ABS_MARG, capi.HGCA):
                    raise ValueError(f"{type(obs).__name__} is a system-level observation")
                self.obs_entries.append((obs, ip, pl.name, normalizename(obs.likelihoodname())))
        for obs in system.observations:
            if obs.kind not in (capi.RV_ABS, capi.RV_ABS_MARG, capi.HGCA):
                raise ValueError(f"{type(obs).__name__} must be attached to a planet")
            self.obs_entries.append((obs, -1, None, normalizename(obs.likelihoodname())))
        self.n_planets = len(system.planets)
        if self.n_planets < 1:
            raise ValueError("the HIP path needs at least one planet")
        planets_ex = θ_example.get("planets", {})
        self.planet_desc = []
        for pl in system.planets:
            θp = planets_ex.get(pl.name, {})
            self.planet_desc.append(dict(orbit_kind=_BASIS[pl.basis], has_mass="mass" in θp))
        # every planet contributes to absolute RV and requires a mass (rv-absolute.jl:146-155)
        if any(e[0].kind in (capi.RV_ABS, capi.RV_ABS_MARG) for e in self.obs_entries):
            for pl, d in zip(system.planets, self.planet_desc):
                if not d["has_mass"]:
                    raise KeyError(f"planet {pl.name} has no `mass` variable but the system has absolute RV data")
        # HGCA reads mass * mjup2msol of every Visual{KepOrbit} planet (hgca.jl:279-290)
        if any(e[0].kind == capi.HGCA for e in self.obs_entries):
            for pl, d in zip(system.planets, self.planet_desc):
                if d["orbit_kind"] == capi.ORBIT_VISUAL_KEP and not d["has_mass"]:
                    raise KeyError(f"planet {pl.name} has no `mass` variable but the system has HGCA data")
        # RV trend closures: classified against the θ_obs variables the model declares (or the example θ carries)
        for obs, ip, plname, key in self.obs_entries:
            if getattr(obs, "trend_function", None) is not None:
                if ip >= 0:
                    θobs_ex = planets_ex.get(plname, {}).get("observations", {}).get(key, {})
                else:
                    θobs_ex = θ_example.get("observations", {}).get(key, {})
                names = list(getattr(obs, "variables", None) or {}) or list(θobs_ex)
                obs.classify_trend(names)
        self.obs_tables = [e[0]._c_table(e[1]) for e in self.obs_entries]
        # ---- C side ------------------------------------------------------------------------------
        self._ctx = C.c_void_p()
        self._ds = C.c_void_p()
        try:
            self._check(self.lib.octo_ctx_create(C.byref(self._ctx), int(device)), "octo_ctx_create")
            if consts is not None:
                self._check(self.lib.octo_consts_set(self._ctx, C.byref(consts)), "octo_consts_set")
            obs_arr, keep = capi.pack_obs(self.obs_tables)
            pl_arr = capi.pack_planets(self.planet_desc)
            self._check(self.lib.octo_dataset_create(self._ctx, obs_arr, len(self.obs_tables), pl_arr, self.n_planets,
                                                     C.byref(self._ds)), "octo_dataset_create")
            del keep
        except Exception:
            self.close()      # a refused dataset must not leak the context it was offered to
            raise
        self.n_obs = len(self.obs_tables)
        self.n_rows = int(self.lib.octo_dataset_n_rows(self._ds))

    # -- lifecycle ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ds", None):
            self.lib.octo_dataset_destroy(self._ds)
            self._ds = None
        if getattr(self, "_ctx", None):
            self.lib.octo_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status, what):
        if status != capi.OCTO_OK:
            msg = self.lib.octo_last_error(self._ctx) if self._ctx else b""
            raise capi.OctoError(status, f"{what}: {(msg or b'').decode()}")

    # -- θ <-> SoA ---------------------------------------------------------------------------------
    def _batch_size(self, θ):
        W = 1
        def visit(d):
            nonlocal W
            for v in d.values():
                if isinstance(v, dict):
                    visit(v)
                else:
                    n = np.size(v)
                    if n != 1:
                        if W != 1 and n != W:
                            raise ValueError("inconsistent batch sizes in θ")
                        W = n
        visit(θ)
        return W

    def pack(self, θ):
        """merge(θ_system, θ_planet) per planet (system.jl:117) -> elems [P*9, W]; θ_obs -> nuis [n_obs*3, W]."""
        W = self._batch_size(θ)
        elems = np.zeros((self.n_planets * capi.N_EL, W))
        for ip, pl in enumerate(self.system.planets):
            θp = θ.get("planets", {}).get(pl.name, {})
            for k, keys in enumerate(el_keys(pl.basis)):
                v = _lookup(θp, keys)
                if v is None:
                    v = _lookup(θ, keys)      # planet-level wins, as in merge(θ_system, θ_planet)
                if v is None:
                    radvel_unused = self.planet_desc[ip]["orbit_kind"] == capi.ORBIT_RADVEL and k in (capi.EL_I, capi.EL_O, capi.EL_PLX)
                    kep_unused = self.planet_desc[ip]["orbit_kind"] == capi.ORBIT_KEP and k == capi.EL_PLX
                    if k == capi.EL_MASS or radvel_unused or kep_unused:
                        v = 0.0
                    else:
                        raise KeyError(f"planet {pl.name}: missing orbital element {keys[0]}")
                elems[ip * capi.N_EL + k, :] = v
        nuis = None
        any_nuis = False
        buf = np.zeros((self.n_obs * capi.N_NUIS, W))
        for io, (obs, ip, plname, key) in enumerate(self.obs_entries):
            if ip >= 0:
                θobs = θ.get("planets", {}).get(plname, {}).get("observations", {}).get(key, {})
            else:
                θobs = θ.get("observations", {}).get(key, {})
            if obs.kind == capi.HGCA:
                # its two per-walker inputs are SYSTEM variables, θ_system.pmra / .pmdec (hgca.jl:266-267)
                for k, nm in enumerate(("pmra", "pmdec")):
                    if nm not in θ:
                        raise KeyError(f"HGCAInstantaneousObs requires the system variable `{nm}`")
                    buf[io * capi.N_NUIS + k, :] = θ[nm]
                any_nuis = True
                continue
            if obs.kind in capi.ASTROM_KINDS:
                defaults = (("jitter", 0.0), ("platescale", 1.0), ("northangle", 0.0))   # relative-astrometry.jl:170-172
            else:
                defaults = (("offset", 0.0), ("jitter", 0.0), (getattr(obs, "trend_coef", None), 0.0))      # row 2: OCTO_NU_RV_TREND
                if obs.kind == capi.RV_ABS_MARG and "jitter" not in θobs:
                    raise KeyError("MarginalizedStarAbsoluteRVObs requires θ_obs.jitter (rv-absolute-margin.jl:149)")
            for k, (nm, dv) in enumerate(defaults):
                if nm is not None and nm in θobs:
                    any_nuis = True
                    buf[io * capi.N_NUIS + k, :] = θobs[nm]
                else:
                    buf[io * capi.N_NUIS + k, :] = dv
        if any_nuis:
            nuis = buf
        return elems, nuis

    def unpack_grad(self, g_elems, g_nuis):
        out = dict(planets={}, observations={})
        for ip, pl in enumerate(self.system.planets):
            out["planets"][pl.name] = {keys[0]: g_elems[ip * capi.N_EL + k] for k, keys in enumerate(el_keys(pl.basis))}
            out["planets"][pl.name]["observations"] = {}
        if g_nuis is not None:
            for io, (obs, ip, plname, key) in enumerate(self.obs_entries):
                if obs.kind == capi.HGCA:
                    out["pmra"] = out.get("pmra", 0.0) + g_nuis[io * capi.N_NUIS]
                    out["pmdec"] = out.get("pmdec", 0.0) + g_nuis[io * capi.N_NUIS + 1]
                    continue
                names = ("jitter", "platescale", "northangle") if obs.kind in capi.ASTROM_KINDS else ("offset", "jitter", getattr(obs, "trend_coef", None))
                d = {}
                for k, nm in enumerate(names):      # summed per NAME: a trend coefficient may be the variable that is also `offset` or `jitter`
                    if nm is not None:
                        d[nm] = d.get(nm, 0.0) + g_nuis[io * capi.N_NUIS + k]
                if ip >= 0:
                    out["planets"][plname]["observations"][key] = d
                else:
                    out["observations"][key] = d
        return out

    # -- evaluation: host arrays ---------------------------------------------------------------------
    def ln_like_arrays(self, elems, nuis=None, grad=False):
        elems = np.ascontiguousarray(elems, dtype=np.float64)
        if elems.ndim != 2 or elems.shape[0] != self.n_planets * capi.N_EL:
            raise ValueError(f"elems must be [{self.n_planets * capi.N_EL}, W]")
        W = elems.shape[1]
        nu = None
        if nuis is not None:
            nu = np.ascontiguousarray(nuis, dtype=np.float64)
            if nu.shape != (self.n_obs * capi.N_NUIS, W):
                raise ValueError(f"nuis must be [{self.n_obs * capi.N_NUIS}, {W}]")
        ll = np.empty(W)
        g_el = np.empty_like(elems) if grad else None
        g_nu = np.empty_like(nu) if (grad and nu is not None) else None
        self._check(self.lib.octo_eval(self._ctx, self._ds, capi._dptr(elems), capi._dptr(nu), W, W,
                                       capi._dptr(ll), capi._dptr(g_el), capi._dptr(g_nu)), "octo_eval")
        return (ll, g_el, g_nu) if grad else ll

    def ln_like_into(self, elems, nuis, ll, g_elems=None, g_nuis=None):
        """octo_eval into arrays the caller owns (C-contiguous float64, walker index fastest): with all of them registered
        (`host_register`) a big batch crosses PCIe without the copy engine and lands in place."""
        W = elems.shape[1]
        self._check(self.lib.octo_eval(self._ctx, self._ds, capi._dptr(elems), capi._dptr(nuis), W, W,
                                       capi._dptr(ll), capi._dptr(g_elems), capi._dptr(g_nuis)), "octo_eval")
        return ll

    def host_register(self, *arrays):
        """Page-lock and map arrays the caller keeps alive across calls (octo_host_register)."""
        for a in arrays:
            if a is None:
                continue
            if not (a.flags["C_CONTIGUOUS"] and a.dtype == np.float64):
                raise ValueError("host_register: C-contiguous float64 arrays only")
            self._check(self.lib.octo_host_register(self._ctx, a.ctypes.data, a.nbytes), "octo_host_register")

    def host_unregister(self, *arrays):
        for a in arrays:
            if a is not None:
                self._check(self.lib.octo_host_unregister(self._ctx, a.ctypes.data), "octo_host_unregister")

    def __call__(self, θ):
        elems, nuis = self.pack(θ)
        return self.ln_like_arrays(elems, nuis)

    def ln_like_and_grad(self, θ):
        elems, nuis = self.pack(θ)
        ll, g_el, g_nu = self.ln_like_arrays(elems, nuis, grad=True)
        return ll, self.unpack_grad(g_el, g_nu)

    # -- evaluation: device-resident torch tensors (plumbing for bench / sharded drivers) -------------
    def ln_like_device(self, elems_t, nuis_t=None, grad=False, out=None, stream=None):
        """elems_t: torch float64 CUDA tensor [P*9, W] (contiguous). Enqueues on torch's CURRENT stream — its raw handle is
        handed to the C ABI as it is (0 = HIP's NULL stream when no torch stream is active), so the kernels are ordered
        after the torch ops that produced the inputs and before the ones that consume the outputs — unless `stream`
        (a raw hipStream_t integer, or capi.STREAM_CTX for the context's own stream) is given. Asynchronous."""
        import torch
        assert elems_t.is_cuda and elems_t.dtype == torch.float64 and elems_t.is_contiguous()
        W = elems_t.shape[1]
        if out is None:
            ll = torch.empty(W, dtype=torch.float64, device=elems_t.device)
            g_el = torch.empty_like(elems_t) if grad else None
            g_nu = torch.empty_like(nuis_t) if (grad and nuis_t is not None) else None
        else:
            ll, g_el, g_nu = out
        if stream is None:
            stream = torch.cuda.current_stream(elems_t.device).cuda_stream
        if not isinstance(stream, C.c_void_p):
            stream = C.c_void_p(stream)
        self._check(self.lib.octo_eval_device(
            self._ctx, self._ds, elems_t.data_ptr(), nuis_t.data_ptr() if nuis_t is not None else None, W, W,
            ll.data_ptr(), g_el.data_ptr() if g_el is not None else None, g_nu.data_ptr() if g_nu is not None else None,
            stream), "octo_eval_device")
        return (ll, g_el, g_nu) if grad else ll

    def sync(self):
        self._check(self.lib.octo_sync(self._ctx), "octo_sync")

    def timing_enable(self, on=True):
        self._check(self.lib.octo_timing_enable(self._ctx, int(on)), "octo_timing_enable")

    def set_option(self, option, value):
        """octo_ctx_set_option (capi.OPT_*): batch-invariant results, the warm start, the walker-tile sort."""
        self._check(self.lib.octo_ctx_set_option(self._ctx, int(option), int(value)), "octo_ctx_set_option")

    def get_option(self, option):
        v = C.c_int64()
        self._check(self.lib.octo_ctx_get_option(self._ctx, int(option), C.byref(v)), "octo_ctx_get_option")
        return v.value

    def tile_state(self):
        """The walker-tile sort of this context (octo_debug_tile_state, a measurement hook outside the C ABI): evaluations that ran sorted, probes
        taken, whether the sort is on for the current (dataset, batch size), the last probe's estimated saving [µs per evaluation]. None for a
        library without the hook (an older build selected with OCTOFITTER_HIP_LIB)."""
        try:
            f = self.lib.octo_debug_tile_state
        except AttributeError:
            return None
        f.restype = C.c_int32
        f.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        n, pr, on, sv = C.c_int64(), C.c_int64(), C.c_int32(), C.c_double()
        self._check(f(self._ctx, C.byref(n), C.byref(pr), C.byref(on), C.byref(sv)), "octo_debug_tile_state")
        return {"evaluations_sorted": n.value, "probes": pr.value, "on": bool(on.value), "last_probe_saving_us": sv.value}

    def timing_stats(self):
        """(median_ms, min_ms, max_ms, n) over the individual timed launches since the last reset."""
        med, lo, hi, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        self._check(self.lib.octo_timing_stats(self._ctx, C.byref(med), C.byref(lo), C.byref(hi), C.byref(n)), "octo_timing_stats")
        return med.value, lo.value, hi.value, n.value

    def timing_read(self, reset=True):
        ms = C.c_double()
        n = C.c_int64()
        self._check(self.lib.octo_timing_read(self._ctx, C.byref(ms), C.byref(n), int(reset)), "octo_timing_read")
        return ms.value, n.value


def not_on_device(system: System):
    """Why a system cannot go to the device at all, or None (julia/OctofitterHIP.jl: _not_on_device): the library compiles its epoch-loop
    kernels for 1 … OCTO_MAX_PLANETS planets; the reference unrolls over any number (src/likelihoods/system.jl:116-118,156-170)."""
    n = len(system.planets)
    if n < 1:
        return "the system has no planet"
    if n > capi.MAX_PLANETS:
        return f"{n} planets: the device kernels are compiled for at most {capi.MAX_PLANETS}"
    return None


def accelerate(system: System, θ_example: dict, device: int = 0, consts=None, verbosity: int = 1):
    """Mirror of `OctofitterHIP.accelerate(system)` ("falls back to the reference closure" instead of throwing): the batched
    device evaluator of `system` — or, when the system cannot go to the device (more planets than the kernels are compiled for, no usable
    HIP device, a dataset the library refuses), the SAME `system` object back, untouched, with the reason in `system.hip_fallback_reason`
    and one log line. What evaluates an un-accelerated system is the caller's business — in Julia, the reference itself; this package has no
    CPU evaluator and never pretends to (a missing library is still an error: load_library raises). Nothing is leaked on the fallback paths."""
    import logging
    why = not_on_device(system)
    if why is None:
        try:
            fn = BatchedLnLike(system, θ_example, device=device, consts=consts)
            system.hip_fallback_reason = None
            return fn
        except capi.OctoError as ex:
            # OCTO_ENODEV (no usable device) and OCTO_ENOTSUP (a valid system the device path does not take) are the fallback's reasons; bad input
            # (OCTO_EINVAL: σ <= 0, non-finite epochs, |cor| >= 1 …), OCTO_EHIP and OCTO_ENOMEM are the caller's to see
            if ex.status not in capi.FALLBACK_STATUSES:
                raise
            why = str(ex)
    system.hip_fallback_reason = why
    if verbosity >= 1:
        logging.getLogger("octofitter_hip").info("OctofitterHIP: %s — the system stays on the host path", why)
    return system


def make_ln_like(system: System, θ_system: dict, device: int = 0, consts=None) -> BatchedLnLike:
    """Mirror of `make_ln_like(system::System, θ_system)` (src/likelihoods/system.jl:21): θ_system is an
    example parameter set used, as in the reference, only to discover which variables exist."""
    return BatchedLnLike(system, θ_system, device=device, consts=consts)
