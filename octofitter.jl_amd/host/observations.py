"""
Host-side mirror of the reference's observation types on the hot path.

Same names, column conventions and constructor error behaviour as the reference, so that the
parity tests read like the reference's own tests:

  PlanetRelAstromObs              src/likelihoods/relative-astrometry.jl:20-95
  StarAbsoluteRVObs               OctofitterRadialVelocity/src/rv-absolute.jl:56-113
  MarginalizedStarAbsoluteRVObs   OctofitterRadialVelocity/src/rv-absolute-margin.jl:60-84
  PlanetRelativeRVObs             OctofitterRadialVelocity/src/rv-relative.jl:60-101
  HGCAInstantaneousObs            src/likelihoods/hgca.jl:28-152

A "table" is anything column-like: a dict of equal-length sequences, or a list of row dicts.
Priors / Derived variable blocks are host-side model specification and stay in the reference
(SURVEY.md §2 rows 4, 9): here an observation only carries the NAMES of its nuisance
variables so that `make_ln_like` can look them up in θ_obs.
"""
from __future__ import annotations

import re
import unicodedata
import warnings

import numpy as np

from . import capi

astrom_cols1 = ("epoch", "ra", "dec", "σ_ra", "σ_dec")      # relative-astrometry.jl:3
astrom_cols3 = ("epoch", "pa", "sep", "σ_pa", "σ_sep")      # relative-astrometry.jl:4
rv_cols = ("epoch", "rv", "σ_rv")

_ASCII = {"σ_ra": "sigma_ra", "σ_dec": "sigma_dec", "σ_pa": "sigma_pa", "σ_sep": "sigma_sep", "σ_rv": "sigma_rv"}
_MJD_1950, _MJD_2050 = 33282.0, 69807.0   # mjd("1950"), mjd("2050")


def normalizename(name: str) -> str:
    """src/variables.jl:1068-1073."""
    uname = unicodedata.normalize("NFC", name).strip()
    ident = uname if uname.isidentifier() else "".join(ch if (ch.isalnum() or ch == "_") else "_" for ch in uname)
    if not ident or not (ident[0].isalpha() or ident[0] == "_"):
        ident = "_" + ident
    return re.sub(r"(_)\1+", "_", ident)


def _as_table(observations) -> dict:
    if isinstance(observations, dict):
        tab = {k: np.atleast_1d(np.asarray(v)) for k, v in observations.items()}
    else:
        rows = list(observations)
        if not rows:
            raise ValueError("empty observation table")
        keys = list(rows[0].keys())
        tab = {k: np.asarray([r[k] for r in rows]) for k in keys}
    for ascii_name_of, uni in ((v, k) for k, v in _ASCII.items()):
        if ascii_name_of in tab and uni not in tab:
            tab[uni] = tab.pop(ascii_name_of)
    return tab


def _equal_length_cols(tab) -> bool:
    return len({len(v) for v in tab.values()}) <= 1


def _warn_epoch_range(epoch):
    if np.any(epoch >= _MJD_2050) or np.any(epoch <= _MJD_1950):
        warnings.warn("The data you entered fell outside the range year 1950 to year 2050. The expected input "
                      "format is MJD (modified julian date). We suggest you double check your input data!")


class AbstractObs:
    kind = None
    nuisance_names: tuple = ()

    def likelihoodname(self):
        return self.name

    def _c_table(self, planet_index):
        raise NotImplementedError


class PlanetRelAstromObs(AbstractObs):
    """Relative astrometry between a host star and a secondary body (mas / radians)."""
    nuisance_names = ("jitter", "platescale", "northangle")

    def __init__(self, observations, *, name, variables=None):
        table = _as_table(observations)
        if not _equal_length_cols(table):
            raise ValueError("The columns in the input data do not all have the same length")
        has1 = set(astrom_cols1) <= set(table)
        has3 = set(astrom_cols3) <= set(table)
        if not has1 and not has3:
            raise ValueError(f"Expected columns {astrom_cols1} or {astrom_cols3}")
        table = {k: np.asarray(v, dtype=np.float64) for k, v in table.items()}
        _warn_epoch_range(table["epoch"])
        ii = np.argsort(table["epoch"], kind="stable")          # relative-astrometry.jl:46-47
        table = {k: v[ii] for k, v in table.items()}
        self.is_seppa = "pa" in table and "sep" in table         # :53 — pa/sep takes precedence
        if self.is_seppa:
            if np.any(table["pa"] >= 2 * np.pi) or np.any(table["pa"] <= -2 * np.pi):
                warnings.warn("The data you entered fell outside the range [-2pi, +2pi]. The expected input format "
                              "is radians (you can use `deg2rad` to convert). We suggest you double check your input data!")
        if "cor" in table and np.any(np.abs(table["cor"]) > 1 - 1e-5):   # :70-72
            raise ValueError(f"Correlation values may not be well-specified: {table['cor']}")
        self.table = table
        self.name = name
        self.variables = variables
        self.kind = capi.ASTROM_SEPPA if self.is_seppa else capi.ASTROM_RADEC

    def __len__(self):
        return len(self.table["epoch"])

    def _c_table(self, planet_index):
        t = self.table
        if self.is_seppa:
            y1, y2, s1, s2 = t["pa"], t["sep"], t["σ_pa"], t["σ_sep"]
        else:
            y1, y2, s1, s2 = t["ra"], t["dec"], t["σ_ra"], t["σ_dec"]
        return dict(kind=self.kind, planet=planet_index, epoch=t["epoch"], y1=y1, y2=y2, s1=s1, s2=s2, cor=t.get("cor"))


PlanetRelAstromLikelihood = PlanetRelAstromObs   # backwards-compat alias, relative-astrometry.jl:98


class ObsPriorAstromONeil2019(AbstractObs):
    """Observable-based prior of O'Neil et al. 2019 wrapped around a relative-astrometry table
    (src/likelihoods/prior-observable.jl:56-76): ln_like = ln_like(wrapped) + 2 log(Σ_j |…| ∛P / √(1−e²)).
    As in the reference it exposes the wrapped table (so its epochs are gathered a second time when both objects are
    attached) and the wrapped observation's variables, under the name "obspri_<name>"."""
    nuisance_names = ("jitter", "platescale", "northangle")

    def __init__(self, obs):
        if not isinstance(obs, PlanetRelAstromObs):
            raise NotImplementedError("ObsPriorAstromONeil2019 is on the HIP path for PlanetRelAstromObs only")
        self.wrapped_like = obs
        self.table = obs.table
        self.variables = obs.variables
        self.is_seppa = obs.is_seppa
        self.kind = capi.ONEIL_SEPPA if obs.is_seppa else capi.ONEIL_RADEC

    @property
    def name(self):
        return self.likelihoodname()

    def likelihoodname(self):
        return "obspri_" + self.wrapped_like.likelihoodname()      # prior-observable.jl:68

    def __len__(self):
        return len(self.wrapped_like)

    def _c_table(self, planet_index):
        t = self.wrapped_like._c_table(planet_index)
        t["kind"] = self.kind
        return t


class θObs(dict):
    """θ_obs as a `trend_function(θ_obs, epoch)` receives it: the reference passes a NamedTuple, so the closure reads its
    variables as fields (`θ_obs.trend_slope`, rv-absolute.jl:26). Item access works too."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


class _RVBase(AbstractObs):
    nuisance_names = ("offset", "jitter")

    def __init__(self, observations, *, name, variables=None, trend_function=None, gaussian_process=None):
        if gaussian_process is not None:
            # The GP branch (rv-absolute.jl:205-315) is host-side Julia code and out of scope for the kernel; the Julia
            # shim leaves such an observation on the reference's CPU path.
            raise NotImplementedError("gaussian_process observations are not on the HIP path")
        if trend_function is not None and not callable(trend_function):
            raise TypeError("trend_function must be callable: (θ_obs, epoch) -> m/s")
        # rv-absolute.jl:69, rv-relative.jl:64, rv-absolute-margin.jl:52: default (θ_obs, epoch) -> 0
        self.trend_function = trend_function
        self.trend_coef = None        # the θ_obs variable the trend is linear in      } set by classify_trend, as the Julia shim's
        self.trend_basis = None       # trend_function(θ_obs with that variable = 1, epoch_j) } `_trend_basis` does (OctofitterHIP.jl)
        self._trend_checked = trend_function is None
        table = _as_table(observations)
        if not _equal_length_cols(table):
            raise ValueError("The columns in the input data do not all have the same length")
        if not set(rv_cols) <= set(table):
            raise ValueError(f"Expected columns {rv_cols}")
        if "inst_idx" in table and len(np.unique(table["inst_idx"])) > 1:
            raise ValueError("Deprecated: data from separate RV instruments should now be placed into different "
                             "StarAbsoluteRVLikelihood likelihood objects, rather than specified by an inst_idx parameter.")
        table = {k: np.asarray(v, dtype=np.float64) for k, v in table.items() if k in rv_cols}
        ii = np.argsort(table["epoch"], kind="stable")
        table = {k: v[ii] for k, v in table.items()}
        _warn_epoch_range(table["epoch"])
        self.table = table
        self.name = name
        self.variables = variables

    def __len__(self):
        return len(self.table["epoch"])

    def classify_trend(self, names, n_probe=3, seed=20260929):
        """Which device form does `trend_function` have? The closure is arbitrary host code; the kernels carry a trend as ONE θ_obs
        variable times a per-row basis column (include/octofitter_hip.h: OCTO_NU_RV_TREND). Probed numerically, exactly as the Julia
        shim does: evaluate the closure at every table epoch for a few draws of θ_obs;
          * identically zero (the default closure)          -> no trend on the device;
          * θ_obs[c] · b(epoch) for one variable c           -> coefficient c, basis b = closure at c = 1;
          * anything else                                    -> NotImplementedError (the Julia shim leaves the observation on the CPU).
        `names`: the observation's θ_obs variables."""
        self.trend_coef, self.trend_basis = None, None
        self._trend_checked = True
        if self.trend_function is None:
            return
        epochs = self.table["epoch"]
        names = list(names)
        rng = np.random.default_rng(seed)
        f = lambda θ: np.array([float(self.trend_function(θObs(θ), float(t))) for t in epochs], dtype=np.float64)
        draws = [{n: float(rng.uniform(0.5, 2.0)) for n in names} for _ in range(n_probe)]
        vals = [f(d) for d in draws]
        if all(not np.any(v) for v in vals):
            return                                             # the zero trend
        for c in names:
            b = f({**draws[0], c: 1.0})
            z = f({**draws[0], c: 0.0})
            scale = max(float(np.max(np.abs(b))), 1e-300)
            if np.any(z) or not np.all(np.isfinite(b)):
                continue
            if all(np.all(np.abs(v - d[c] * b) <= 1e-12 * scale * max(1.0, abs(d[c]))) for v, d in zip(vals, draws)):
                self.trend_coef, self.trend_basis = c, b
                return
        raise NotImplementedError(f"{type(self).__name__} {self.name!r}: trend_function is not (one θ_obs variable) × (a function of the "
                                  "epoch): not on the HIP path — the Julia shim keeps such an observation on the reference's CPU path")

    def _c_table(self, planet_index):
        t = self.table
        if not self._trend_checked:
            raise RuntimeError("classify_trend must run before the table is uploaded (make_ln_like does it)")
        return dict(kind=self.kind, planet=planet_index, epoch=t["epoch"], y1=t["rv"], y2=None, s1=t["σ_rv"], s2=None, cor=None,
                    extra=self.trend_basis)


class StarAbsoluteRVObs(_RVBase):
    kind = capi.RV_ABS


class MarginalizedStarAbsoluteRVObs(_RVBase):
    kind = capi.RV_ABS_MARG
    nuisance_names = ("jitter",)


class PlanetRelativeRVObs(_RVBase):
    kind = capi.RV_REL


StarAbsoluteRVLikelihood = StarAbsoluteRVObs
PlanetRelativeRVLikelihood = PlanetRelativeRVObs


class HGCAInstantaneousObs(AbstractObs):
    """Hipparcos-Gaia Catalog of Accelerations, instantaneous model (src/likelihoods/hgca.jl:28-152): the proper
    motion and position of the primary at 1..N_ave epochs around the Hipparcos and Gaia epochs, averaged.

    The reference looks the star up by `gaia_id` in the HGCA FITS catalogue (a DataDeps download). There is no network
    here, so the catalogue ROW is passed in: `hgca` is a mapping with the catalogue's own column names
    (epoch_ra_hip, epoch_dec_hip, epoch_ra_gaia, epoch_dec_gaia [Julian years]; pmra_hip, pmdec_hip, pmra_hip_error,
    pmdec_hip_error, pmra_pmdec_hip, and the same for _hg and _gaia). A system-level observation; it reads
    θ_system.pmra / .pmdec (hgca.jl:266-267) and every Visual{KepOrbit} planet's `mass`."""
    kind = capi.HGCA
    nuisance_names = ()
    name = "HGCA"                                           # likelihoodname(::HGCAInstantaneousObs), hgca.jl:41
    _J2000_MJD = 51544.5
    _JULIAN_YEAR = 365.25

    def __init__(self, *, hgca=None, gaia_id=None, N_ave=1, factor=1, variables=None):
        if hgca is None:
            raise NotImplementedError("the HGCA catalogue (DataDeps download) is not available here: pass the "
                                      f"catalogue row as hgca=dict(...) (gaia_id={gaia_id!r})")
        h = {k: float(np.asarray(v).reshape(-1)[0]) for k, v in dict(hgca).items()}
        self.hgca = h
        self.variables = variables
        self.N_ave, self.factor = int(N_ave), float(factor)
        to_mjd = lambda yr: (yr - 2000.0) * self._JULIAN_YEAR + self._J2000_MJD          # hgca.jl:80-84
        e_ra_hip, e_dec_hip = to_mjd(h["epoch_ra_hip"]), to_mjd(h["epoch_dec_hip"])
        e_ra_gaia, e_dec_gaia = to_mjd(h["epoch_ra_gaia"]), to_mjd(h["epoch_dec_gaia"])
        dt_gaia, dt_hip = 1038.0, 4 * 365.25                                              # :87-88
        if self.N_ave == 1:
            δ_hip = δ_gaia = [0.0]
        else:
            δ_hip = np.linspace(-dt_hip / 2, dt_hip / 2, self.N_ave)
            δ_gaia = np.linspace(-dt_gaia / 2, dt_gaia / 2, self.N_ave)
        rows = []
        for δ in δ_hip:                                                                   # :101-110, row order kept
            rows.append((e_ra_hip + δ, capi.HGCA_RA, capi.HGCA_HIP))
            rows.append((e_dec_hip + δ, capi.HGCA_DEC, capi.HGCA_HIP))
        for δ in δ_gaia:
            rows.append((e_ra_gaia + δ, capi.HGCA_RA, capi.HGCA_GAIA))
            rows.append((e_dec_gaia + δ, capi.HGCA_DEC, capi.HGCA_GAIA))
        rows = np.asarray(rows, dtype=np.float64)
        self.table = {"epoch": rows[:, 0], "meas": rows[:, 1], "inst": rows[:, 2]}
        ex = []
        for tag in ("hip", "hg", "gaia"):                                                 # dist_hip, dist_hg, dist_gaia :127-145
            ex += [h[f"pmra_{tag}"], h[f"pmdec_{tag}"], h[f"pmra_{tag}_error"] * self.factor,
                   h[f"pmdec_{tag}_error"] * self.factor, h[f"pmra_pmdec_{tag}"]]
        self.extra = np.asarray(ex, dtype=np.float64)

    def __len__(self):
        return len(self.table["epoch"])

    def _c_table(self, planet_index):
        t = self.table
        return dict(kind=self.kind, planet=-1, epoch=t["epoch"], y1=t["meas"], y2=t["inst"], s1=None, s2=None, cor=None,
                    extra=self.extra)


HGCAInstantaneousLikelihood = HGCAInstantaneousObs
