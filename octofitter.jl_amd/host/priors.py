"""
Host-side mirror of the reference's variable blocks for the STANDARD parameterisation.

    variables(M=truncated(Normal(1.2, 0.1), lower=0.1), plx=truncated(Normal(50.0, 0.02), lower=0.1))
    variables(a=Uniform(0, 100), e=Uniform(0.0, 0.99), i=Sine(), ω=UniformCircular(), Ω=UniformCircular(),
              θ=UniformCircular(), tp=θ_at_epoch_to_tperi("θ", 50000))

mirrors `@variables begin … end` (src/macros.jl) for the building blocks every reference test model uses:
priors `~` (Uniform, LogUniform, Normal, truncated Normal, Sine, UniformCircular — src/variables.jl:279-299) and the
derived `tp = θ_at_epoch_to_tperi(θ, epoch; M, e, a, i, ω, Ω)` (src/parameterizations.jl:6-69), plus constants.
Arbitrary `Derived` Julia expressions are out of scope (they stay on the host in the reference).

The NumPy methods here (sample / link / invlink) are host conveniences of the mirror — `model.sample_priors`,
`model.link` in the reference (src/logdensitymodel.jl:18-24); the hot path (invlink + log-prior + likelihood +
gradient for a batch) runs on the device through `octo_model_logpost`.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

from . import capi

_EPS = 2.220446049250313e-16


class Prior:
    kind = None
    lo, hi = -math.inf, math.inf

    def bounds(self):
        return self.lo, self.hi

    def c_params(self):
        raise NotImplementedError

    def sample(self, rng, n):
        raise NotImplementedError

    # Bijectors.link / invlink (TruncatedBijector) — host convenience
    def link(self, x):
        a, b = self.bounds()
        x = np.asarray(x, dtype=np.float64)
        if math.isfinite(a) and math.isfinite(b):
            u = (x - a) / (b - a)
            return np.log(u) - np.log1p(-u)
        if math.isfinite(a):
            return np.log(x - a)
        if math.isfinite(b):
            return np.log(b - x)
        return x

    def invlink(self, y):
        a, b = self.bounds()
        y = np.asarray(y, dtype=np.float64)
        if math.isfinite(a) and math.isfinite(b):
            return (b - a) / (1.0 + np.exp(-y)) + a
        if math.isfinite(a):
            return np.exp(y) + a
        if math.isfinite(b):
            return b - np.exp(y)
        return y


class Uniform(Prior):
    kind = capi.PRIOR_UNIFORM

    def __init__(self, a, b):
        self.lo, self.hi = float(a), float(b)

    def c_params(self):
        return self.lo, self.hi, self.lo, self.hi

    def sample(self, rng, n):
        return rng.uniform(self.lo, self.hi, n)


class LogUniform(Prior):
    kind = capi.PRIOR_LOGUNIFORM

    def __init__(self, a, b):
        self.lo, self.hi = float(a), float(b)

    def c_params(self):
        return self.lo, self.hi, self.lo, self.hi

    def sample(self, rng, n):
        return np.exp(rng.uniform(math.log(self.lo), math.log(self.hi), n))


class Normal(Prior):
    kind = capi.PRIOR_NORMAL

    def __init__(self, μ, σ):
        self.μ, self.σ = float(μ), float(σ)

    def c_params(self):
        return self.μ, self.σ, -math.inf, math.inf

    def sample(self, rng, n):
        return rng.normal(self.μ, self.σ, n)


class TruncatedNormal(Prior):
    kind = capi.PRIOR_TRUNCNORMAL

    def __init__(self, μ, σ, lower=None, upper=None):
        self.μ, self.σ = float(μ), float(σ)
        self.lo = -math.inf if lower is None else float(lower)
        self.hi = math.inf if upper is None else float(upper)

    def c_params(self):
        return self.μ, self.σ, self.lo, self.hi

    def sample(self, rng, n):
        out = np.empty(n)
        k = 0
        while k < n:
            x = rng.normal(self.μ, self.σ, max(n - k, 16))
            x = x[(x >= self.lo) & (x <= self.hi)][: n - k]
            out[k:k + len(x)] = x
            k += len(x)
        return out


def truncated(d, lower=None, upper=None):
    """Distributions.truncated(Normal(μ, σ); lower, upper)."""
    if not isinstance(d, Normal):
        raise NotImplementedError("only truncated(Normal(...)) is on the HIP path")
    return TruncatedNormal(d.μ, d.σ, lower, upper)


class Sine(Prior):
    """Octofitter.Sine(): pdf sin(x)/2 on (0, π)   (src/distributions.jl:14-39)."""
    kind = capi.PRIOR_SINE
    lo, hi = 0.0 + _EPS, math.pi - _EPS       # minimum / maximum, distributions.jl:31-32

    def c_params(self):
        return 0.0, 0.0, self.lo, self.hi

    def sample(self, rng, n):
        return np.arccos(1.0 - 2.0 * rng.uniform(0.0, 1.0, n))    # quantile, distributions.jl:39


class UniformCircular:
    """`ω ~ UniformCircular()` expands to ωx, ωy ~ Normal(0, 1), ω = atan(ωy, ωx)/2π·domain and a UnitLengthPrior
    likelihood term (src/variables.jl:260-299)."""

    def __init__(self, domain=2 * math.pi):
        self.domain = float(domain)


class θ_at_epoch_to_tperi:
    """`tp = θ_at_epoch_to_tperi(θ, epoch; M=system.M, e, a, i, ω, Ω)` (src/parameterizations.jl:6-69). `θ` names a
    UniformCircular variable of the same planet."""

    def __init__(self, θ: str, theta_epoch: float):
        self.θ, self.theta_epoch = str(θ), float(theta_epoch)


def variables(**kw) -> "OrderedDict":
    """An ordered variable block: name -> Prior | UniformCircular | θ_at_epoch_to_tperi | number (constant)."""
    out = OrderedDict()
    for k, v in kw.items():
        if not isinstance(v, (Prior, UniformCircular, θ_at_epoch_to_tperi, int, float)):
            raise TypeError(f"variable {k}: {type(v).__name__} is not a standard-parameterisation building block")
        out[k] = v
    return out
