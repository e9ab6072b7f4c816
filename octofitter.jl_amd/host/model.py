"""
Host-side mirror of `Octofitter.LogDensityModel` (src/logdensitymodel.jl:5-24) for the standard parameterisation
(SURVEY.md §8 f1), batched: every callback takes θ_t as a [D, W] array (or [D]) and runs on the device.

    model = LogDensityModel(system)
    model.D                                 # 11 for the model of test/integration/sampling.jl:29-64
    lp = model.ℓπcallback(θ_t)              # log-posterior in the unconstrained space, src/logdensitymodel.jl:110-146
    lp, ∇ = model.∇ℓπcallback(θ_t)          # value and gradient, :169-177
    θ = model.sample_priors(rng, n); θ_t = model.link(θ); θ = model.invlink(θ_t); nt = model.arr2nt(θ)

Flattening order of θ (src/variables.jl:1205-1347, :1372-1431): system priors, system-observation priors, then per
planet its priors and its observations' priors; `x ~ UniformCircular()` contributes (xx, xy) in place.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import capi
from .observations import normalizename
from .priors import Prior, UniformCircular, θ_at_epoch_to_tperi
from .system import BatchedLnLike, System, _BASIS, el_keys

_K_YR, _YD = 365.2568983840419, 365.25


class LogDensityModel:
    def __init__(self, system: System, device: int = 0, consts: capi.OctoConsts | None = None, verbosity: int = 0):
        self.system = system
        self.names: list[str] = []
        self.priors: list[Prior] = []
        self._circ: dict[tuple, tuple[int, int, float]] = {}    # (scope, varname) -> (ix, iy, domain)
        self._index: dict[tuple, int] = {}                      # (scope, varname) -> θ index
        sysvars = system.variables or {}
        self._add_block(("sys",), "", sysvars)
        for obs in system.observations:
            self._add_block(("sysobs", id(obs)), normalizename(obs.likelihoodname()) + "_", getattr(obs, "variables", None) or {})
        for pl in system.planets:
            self._add_block(("pl", pl.name), pl.name + "_", pl.variables or {})
            for obs in pl.observations:
                self._add_block(("plobs", pl.name, id(obs)), f"{pl.name}_{normalizename(obs.likelihoodname())}_", getattr(obs, "variables", None) or {})
        self.D = len(self.priors)
        if self.D == 0:
            raise ValueError("Model includes no free variables")          # variables.jl:1349-1351
        # ---- example θ for make_ln_like (only the set of variable names matters, system.jl:21)
        θex = dict(planets={pl.name: {k: 0.0 for k in (pl.variables or {})} for pl in system.planets})
        self.ln_like = BatchedLnLike(system, θex, device=device, consts=consts)
        fn = self.ln_like
        try:
            self._build(system, fn, sysvars)
        except Exception:
            fn.close()      # a model the library (or this classifier) refuses must not leak the context and dataset created for it
            raise

    def _build(self, system, fn, sysvars):
        # ---- kernel-input sources
        used_circ = set()
        esrc = []
        for ip, pl in enumerate(system.planets):
            pv = pl.variables or {}
            radvel = _BASIS[pl.basis] == capi.ORBIT_RADVEL
            kep = _BASIS[pl.basis] == capi.ORBIT_KEP
            for k, keys in enumerate(el_keys(pl.basis)):
                spec, scope, name = None, None, None
                for key in keys:
                    if key in pv:
                        spec, scope, name = pv[key], ("pl", pl.name), key
                        break
                if spec is None:
                    for key in keys:
                        if key in sysvars:
                            spec, scope, name = sysvars[key], ("sys",), key
                            break
                if spec is None:
                    if k == capi.EL_MASS or (radvel and k in (capi.EL_I, capi.EL_O, capi.EL_PLX)) or (kep and k == capi.EL_PLX):
                        esrc.append((capi.SRC_CONST, 0, 0, 0, 0.0))
                        continue
                    raise KeyError(f"planet {pl.name}: missing orbital element {keys[0]}")
                esrc.append(self._source(spec, scope, name, pv, ("pl", pl.name), used_circ,
                                         ti=_BASIS[pl.basis] == capi.ORBIT_THIELE_INNES))
        nsrc = []
        for obs, ip, plname, key in fn.obs_entries:
            ov = getattr(obs, "variables", None) or {}
            scope = ("plobs", plname, id(obs)) if ip >= 0 else ("sysobs", id(obs))
            if obs.kind == capi.HGCA:
                # θ_system.pmra / .pmdec (hgca.jl:266-267): system variables, not observation variables
                for nm in ("pmra", "pmdec"):
                    if nm not in sysvars:
                        raise KeyError(f"HGCAInstantaneousObs requires the system variable `{nm}`")
                    spec = sysvars[nm]
                    nsrc.append((capi.SRC_CONST, 0, 0, 0, float(spec)) if isinstance(spec, (int, float))
                                else self._source(spec, ("sys",), nm, sysvars, ("sys",), used_circ))
                nsrc.append((capi.SRC_CONST, 0, 0, 0, 0.0))
                continue
            if obs.kind in capi.ASTROM_KINDS:
                rows = (("jitter", 0.0), ("platescale", 1.0), ("northangle", 0.0))
            else:
                rows = (("offset", 0.0), ("jitter", 0.0), (getattr(obs, "trend_coef", None), 0.0))      # row 2: OCTO_NU_RV_TREND
            for nm, dv in rows:
                if nm is not None and nm in ov:
                    nsrc.append(self._source(ov[nm], scope, nm, ov, scope, used_circ))
                else:
                    nsrc.append((capi.SRC_CONST, 0, 0, 0, dv))
        unused = set(self._circ) - used_circ
        if unused:
            raise ValueError(f"UniformCircular variables {sorted(v[-1] for v in unused)} are not used by any element or derived variable")
        self._esrc, self._nsrc = esrc, nsrc
        # ---- C side
        lib = fn.lib
        pr = (capi.OctoPrior * self.D)()
        for k, p in enumerate(self.priors):
            pr[k].kind = p.kind
            pr[k].p0, pr[k].p1, pr[k].lo, pr[k].hi = p.c_params()
        es = (capi.OctoSource * len(esrc))(*[capi.OctoSource(*t) for t in esrc])
        ns = (capi.OctoSource * max(len(nsrc), 1))(*[capi.OctoSource(*t) for t in nsrc])
        self._c_priors, self._c_esrc, self._c_nsrc = pr, es, ns      # also used by the oracle binding in the tests
        self._m = C.c_void_p()
        fn._check(lib.octo_model_create(fn._ctx, fn._ds, pr, self.D, es, ns if nsrc else None, C.byref(self._m)), "octo_model_create")

    # ------------------------------------------------------------------------------------------------ construction
    def _add_block(self, scope, prefix, block):
        for name, spec in block.items():
            if isinstance(spec, Prior):
                self._index[scope + (name,)] = len(self.priors)
                self.priors.append(spec)
                self.names.append(prefix + name)
            elif isinstance(spec, UniformCircular):
                from .priors import Normal
                ix = len(self.priors)
                self.priors += [Normal(0, 1), Normal(0, 1)]                 # variables.jl:290-293
                self.names += [prefix + name + "x", prefix + name + "y"]
                self._circ[scope + (name,)] = (ix, ix + 1, spec.domain)

    def _source(self, spec, scope, name, block, block_scope, used_circ, ti=False):
        if isinstance(spec, Prior):
            return (capi.SRC_THETA, self._index[scope + (name,)], 0, 0, 0.0)
        if isinstance(spec, UniformCircular):
            key = scope + (name,)
            ix, iy, dom = self._circ[key]
            flag = 0 if key in used_circ else capi.SRC_FLAG_UNITLEN
            used_circ.add(key)
            return (capi.SRC_CIRCULAR, ix, iy, flag, dom)
        if isinstance(spec, θ_at_epoch_to_tperi):
            key = block_scope + (spec.θ,)
            if key not in self._circ:
                raise KeyError(f"θ_at_epoch_to_tperi: `{spec.θ}` must be a UniformCircular variable of the same block")
            ix, iy, dom = self._circ[key]
            if dom != 2 * math.pi:
                raise ValueError("θ_at_epoch_to_tperi expects θ ~ UniformCircular() (domain 2π)")
            flag = 0 if key in used_circ else capi.SRC_FLAG_UNITLEN
            used_circ.add(key)
            # a ThieleInnesOrbit planet uses θ_at_epoch_to_tperi(θ, epoch; plx, M, e, A, B, F, G)   (parameterizations.jl:8-19)
            return (capi.SRC_TPERI, ix, iy, flag | (capi.SRC_FLAG_TI if ti else 0), spec.theta_epoch)
        return (capi.SRC_CONST, 0, 0, 0, float(spec))

    # ------------------------------------------------------------------------------------------------ callbacks
    def _call(self, θ_t, grad):
        fn = self.ln_like
        θ_t = np.asarray(θ_t, dtype=np.float64)
        single = θ_t.ndim == 1
        th = np.ascontiguousarray(θ_t.reshape(self.D, -1))
        W = th.shape[1]
        lp = np.empty(W)
        g = np.empty_like(th) if grad else None
        fn._check(fn.lib.octo_model_logpost(fn._ctx, self._m, capi._dptr(th), W, W, capi._dptr(lp), capi._dptr(g)), "octo_model_logpost")
        if single:
            return (lp[0], g[:, 0]) if grad else lp[0]
        return (lp, g) if grad else lp

    def ℓπcallback(self, θ_t):
        return self._call(θ_t, False)

    def ᐁℓπcallback(self, θ_t):
        return self._call(θ_t, True)

    # `∇` is not a Python identifier character; keep the reference's spelling reachable through getattr
    def __getattr__(self, name):
        if name == "∇ℓπcallback":
            return self.ᐁℓπcallback
        raise AttributeError(name)

    logdensity = ℓπcallback                     # LogDensityProblems.logdensity, src/logdensitymodel.jl:252

    def __call__(self, θ_t):
        """`(model::LogDensityModel)(θ)` — what Pigeons' explorers call (ext/OctofitterPigeonsExt/OctofitterPigeonsExt.jl:10-12); a [D, W]
        array evaluates all W replicas in one device call."""
        return self._call(θ_t, False)

    def logdensity_and_gradient(self, θ_t):     # :253
        return self._call(θ_t, True)

    def logpost_device(self, θ_t_tensor, grad=True, stream=None):
        """θ_t_tensor: torch float64 CUDA [D, W]. Returns (lp, grad) tensors; asynchronous."""
        import torch
        fn = self.ln_like
        W = θ_t_tensor.shape[1]
        lp = torch.empty(W, dtype=torch.float64, device=θ_t_tensor.device)
        g = torch.empty_like(θ_t_tensor) if grad else None
        if stream is None:
            stream = torch.cuda.current_stream(θ_t_tensor.device).cuda_stream
        fn._check(fn.lib.octo_model_logpost_device(fn._ctx, self._m, θ_t_tensor.data_ptr(), W, W, lp.data_ptr(),
                                                   g.data_ptr() if grad else None, C.c_void_p(stream)), "octo_model_logpost_device")
        return lp, g

    # ------------------------------------------------------------------------------------------------ host utilities
    def sample_priors(self, rng, n=None):
        m = 1 if n is None else int(n)
        θ = np.stack([p.sample(rng, m) for p in self.priors])
        return θ[:, 0] if n is None else θ

    def link(self, θ):
        θ = np.asarray(θ, dtype=np.float64)
        return np.stack([p.link(θ[k]) for k, p in enumerate(self.priors)])

    def invlink(self, θ_t):
        θ_t = np.asarray(θ_t, dtype=np.float64)
        return np.stack([p.invlink(θ_t[k]) for k, p in enumerate(self.priors)])

    def kernel_inputs(self, θ):
        """Natural θ [D, W] -> (elems [P*9, W], nuis [n_obs*3, W]): the Derived variables, on the host (NumPy)."""
        θ = np.asarray(θ, dtype=np.float64).reshape(self.D, -1)
        W = θ.shape[1]
        fn = self.ln_like
        n_el = fn.n_planets * capi.N_EL

        def resolve(src, elems, p):
            kind, i0, i1, _flag, val = src
            if kind == capi.SRC_CONST:
                return np.full(W, val)
            if kind == capi.SRC_THETA:
                return θ[i0]
            ang = np.arctan2(θ[i1], θ[i0])
            if kind == capi.SRC_CIRCULAR:
                return ang / (2 * np.pi) * val
            e_ = elems[p * capi.N_EL:(p + 1) * capi.N_EL]
            if src[3] & capi.SRC_FLAG_TI:
                return _tperi(ang, val, e_[capi.EL_M], e_[capi.EL_E], None, None, None, None,
                              abfg=(e_[capi.EL_A], e_[capi.EL_I], e_[capi.EL_W], e_[capi.EL_O]), plx=e_[capi.EL_PLX])
            return _tperi(ang, val, e_[capi.EL_M], e_[capi.EL_E], e_[capi.EL_A], e_[capi.EL_I], e_[capi.EL_W], e_[capi.EL_O])
        elems = np.zeros((n_el, W))
        for want_tperi in (False, True):
            for k, src in enumerate(self._esrc):
                if (src[0] == capi.SRC_TPERI) == want_tperi:
                    elems[k] = resolve(src, elems, k // capi.N_EL)
        nuis = np.stack([resolve(src, elems, 0) for src in self._nsrc]) if self._nsrc else None
        return elems, nuis

    def close(self):
        if getattr(self, "_m", None):
            self.ln_like.lib.octo_model_destroy(self._m)
            self._m = None
        self.ln_like.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _tperi(θ, theta_epoch, M, e, a, i, ω, Ω, abfg=None, plx=None):
    """NumPy θ_at_epoch_to_tperi (src/parameterizations.jl:6-69) — host convenience for arr2nt-style inspection."""
    if abfg is not None:
        A, B, F, G = abfg
        pp = ((A + G) ** 2 + (B - F) ** 2) / 2          # u + v and u − v of src/parameterizations.jl:15-18 as sums of squares:
        mm = ((A - G) ** 2 + (B + F) ** 2) / 2          # α = √(u + √(u² − v²)) = (√(u+v) + √(u−v))/√2, no cancellation near face-on
        a = (np.sqrt(pp) + np.sqrt(mm)) / np.sqrt(2.0) / plx
    else:
        A = np.cos(Ω) * np.cos(ω) - np.sin(Ω) * np.sin(ω) * np.cos(i)
        B = np.sin(Ω) * np.cos(ω) + np.cos(Ω) * np.sin(ω) * np.cos(i)
        F = -np.cos(Ω) * np.sin(ω) - np.sin(Ω) * np.cos(ω) * np.cos(i)
        G = -np.sin(Ω) * np.sin(ω) + np.cos(Ω) * np.cos(ω) * np.cos(i)
    det = A * G - F * B
    xr = (G * np.cos(θ) - F * np.sin(θ)) / det
    yr = (A * np.sin(θ) - B * np.cos(θ)) / det
    ν = np.arctan2(yr, xr)
    s1 = np.sqrt(1 - e ** 2)
    MA = np.arctan2(-s1 * np.sin(ν), -e - np.cos(ν)) + np.pi - e * s1 * np.sin(ν) / (1 + e * np.cos(ν))
    n = 2 * np.pi / (np.sqrt(a ** 3 / M) * _K_YR / _YD)
    return theta_epoch - MA / n * _YD
