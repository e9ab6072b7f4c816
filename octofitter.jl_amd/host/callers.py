"""
Batched callers of the hot path: the reference's drivers that already hold a walker batch on the
host, re-expressed on the LogDensityModel mirror so that every likelihood evaluation goes through ONE device call.

  guess_starting_position(rng, model, N)     src/initialization.jl:14-66    N prior draws -> link -> ℓπcallback -> argmax
  octofit_rejection(rng, model, draws)       src/sampling.jl:168-256        prior draws, accept with prob exp(ll − max ll)
  rejection_evaluate_likelihoods(model, θ)   src/sampling.jl:260-268        the inner batch: non-finite -> -Inf

The accept/reject and argmax logic is the reference's, line for line; random numbers come from NumPy's Generator
(the reference uses Julia's Xoshiro), so individual draws differ while the sampled distribution is the same.
"""
from __future__ import annotations

import numpy as np


def guess_starting_position(rng, model, N=500_000, batch=250_000, prior_samples=None):
    """Sample IID from the prior N times and return the highest-posterior sample: (bestparams, bestlogpost).
    prior_samples ([D, N], natural domain): use these draws instead of drawing (parity tests feed the same draws to the oracle)."""
    if prior_samples is not None:
        prior_samples = np.asarray(prior_samples, dtype=np.float64)
        N = prior_samples.shape[1]
    bestparams = model.sample_priors(rng) if prior_samples is None else prior_samples[:, 0].copy()
    bestlogpost = -np.inf
    done = 0
    while done < N:
        n = min(batch, N - done)
        params = model.sample_priors(rng, n) if prior_samples is None else prior_samples[:, done:done + n]
        logpost = model.ℓπcallback(model.link(params))
        k = int(np.argmax(logpost))
        if logpost[k] > bestlogpost:                      # initialization.jl:41-44
            bestlogpost, bestparams = float(logpost[k]), params[:, k].copy()
        done += n
    return bestparams, bestlogpost


def rejection_evaluate_likelihoods(model, prior_samples):
    """src/sampling.jl:260-268 for a [D, n] batch of natural-domain prior draws: ll per draw, non-finite -> -Inf."""
    elems, nuis = model.kernel_inputs(prior_samples)
    ll = model.ln_like.ln_like_arrays(elems, nuis)
    # epoch-free likelihood terms of the standard parameterisation (UnitLengthPrior, variables.jl:309-323)
    ll = ll + _unit_length_terms(model, prior_samples)
    return np.where(np.isfinite(ll), ll, -np.inf)


def _unit_length_terms(model, θ):
    tot = np.zeros(θ.shape[1])
    for (kind, i0, i1, flag, _v) in list(model._esrc) + list(model._nsrc):
        if kind in (2, 3) and (flag & 1):
            r = np.sqrt(θ[i0] ** 2 + θ[i1] ** 2)
            tot += -np.log(r) - np.log(0.1 * np.sqrt(2 * np.pi)) - np.log(r) ** 2 / (2 * 0.01)
    return tot


def octofit_rejection(rng, model, draws=100_000, verbosity=0, prior_samples=None, uniforms=None):
    """Rejection sampling with the prior as proposal. Returns dict(samples [D, n_accepted] (natural domain), loglike,
    logpost, draws, n_accepted, acceptance_rate, accept) — the chain the reference packs into MCMCChains.
    prior_samples / uniforms: use these draws (parity tests feed the same ones to the oracle)."""
    if prior_samples is None:
        prior_samples = model.sample_priors(rng, draws)                       # sampling.jl:178
    else:
        prior_samples = np.asarray(prior_samples, dtype=np.float64)
        draws = prior_samples.shape[1]
    log_likes = rejection_evaluate_likelihoods(model, prior_samples)          # :189-191
    max_ll = np.max(log_likes)                                                # :194
    if not np.isfinite(max_ll):
        raise RuntimeError(f"All {draws} prior samples produced non-finite log-likelihoods. Check your model and priors.")
    u = rng.uniform(0.0, 1.0, draws) if uniforms is None else np.asarray(uniforms, dtype=np.float64)
    with np.errstate(over="ignore"):
        accept = (log_likes != -np.inf) & (u < np.exp(log_likes - max_ll))    # :202-210
    idx = np.nonzero(accept)[0]
    if idx.size == 0:
        raise RuntimeError(f"No samples were accepted out of {draws} draws. The posterior may be extremely concentrated relative "
                           "to the prior. Consider increasing `draws` or using a different sampler.")
    samples = prior_samples[:, idx]
    logpost = model.ℓπcallback(model.link(samples))                           # _rejection_build_chain, :270-
    return dict(samples=samples, loglike=log_likes[idx], logpost=logpost, draws=draws, n_accepted=int(idx.size),
                acceptance_rate=idx.size / draws, names=list(model.names), accept=accept, all_loglike=log_likes)


def pointwise_like(model, θ_samples):
    """`Octofitter.pointwise_like` (src/cross-validation.jl:17-46) on the batch path: the log-likelihood of every posterior sample under
    EACH observation table separately — LL_out[n_samples, n_observations], one device call per observation over all samples (the
    reference builds one single-observation system per table and loops over samples on the CPU). θ_samples: [D, n] natural domain.
    Returns (LL_out, names)."""
    from .system import BatchedLnLike, Planet, System
    θ_samples = np.asarray(θ_samples, dtype=np.float64).reshape(model.D, -1)
    elems, nuis = model.kernel_inputs(θ_samples)
    fn = model.ln_like
    n = θ_samples.shape[1]
    out = np.zeros((n, len(fn.obs_entries)))
    names = []
    θex = dict(planets={pl.name: {k: 0.0 for k in (pl.variables or {})} for pl in model.system.planets})
    for io, (obs, ip, plname, key) in enumerate(fn.obs_entries):
        planets = [Planet(name=pl.name, basis=pl.basis, observations=[obs] if (ip >= 0 and pl.name == plname) else [], variables=pl.variables)
                   for pl in model.system.planets]
        sub = System(name=f"{model.system.name}_{key}", companions=planets, observations=[obs] if ip < 0 else [], variables=model.system.variables)
        one = BatchedLnLike(sub, θex, device=fn.device_index, consts=None)
        try:
            nu = None if nuis is None else np.ascontiguousarray(nuis[io * 3:(io + 1) * 3])
            out[:, io] = one.ln_like_arrays(elems, nu)
        finally:
            one.close()
        names.append(key)
    return out, names
