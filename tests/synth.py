"""
Synthetic inputs for tests and bench.py — the workloads BASELINE.json / SURVEY.md §8(d) name.

Input generation only (NumPy): a Newton Kepler solve to place the "truth" companion, Gaussian noise,
prior-drawn walkers. Nothing here is a parity claim; results are always compared through
oracle/ (tests) or not compared at all (bench timing).
"""
from __future__ import annotations

import numpy as np

K_YR = 365.2568983840419
N_EL, N_NUIS = 9, 3
TRUTH = dict(a=10.0, e=0.3, i=1.0, w=0.5, O=2.0, tp=50000.0, M=1.2, plx=50.0)   # examples/ofti_rejection_sampling.jl:26-35


def _kepler(MA, e):
    M = MA - 2 * np.pi * np.round(MA / (2 * np.pi))
    E = M + e * np.sin(M)
    for _ in range(60):
        E = E - (E - e * np.sin(E) - M) / (1 - e * np.cos(E))
    return E


def truth_radec(t, el=TRUTH):
    P_d = K_YR * np.sqrt(el["a"] ** 3 / el["M"])
    E = _kepler(2 * np.pi * (t - el["tp"]) / P_d, el["e"])
    X = el["a"] * (np.cos(E) - el["e"])
    Y = el["a"] * np.sqrt(1 - el["e"] ** 2) * np.sin(E)
    cw, sw, ci = np.cos(el["w"]), np.sin(el["w"]), np.cos(el["i"])
    cO, sO = np.cos(el["O"]), np.sin(el["O"])
    east = X * (cw * sO + sw * ci * cO) + Y * (-sw * sO + cw * ci * cO)
    north = X * (cw * cO - sw * ci * sO) + Y * (-sw * cO - cw * ci * sO)
    return east * el["plx"], north * el["plx"]


def truth_rv_star(t, el, mass_mjup, mjup2msol=0.0009545942339693249):
    """Stellar reflex RV [m/s] induced by a companion of mass_mjup (negative of the companion's scaled RV)."""
    P_d = K_YR * np.sqrt(el["a"] ** 3 / el["M"])
    E = _kepler(2 * np.pi * (t - el["tp"]) / P_d, el["e"])
    nu = 2 * np.arctan2(np.sqrt(1 + el["e"]) * np.sin(E / 2), np.sqrt(1 - el["e"]) * np.cos(E / 2))
    K = 2 * np.pi * el["a"] * np.sin(el["i"]) / ((P_d / 365.25) * np.sqrt(1 - el["e"] ** 2)) * 1.495978707e11 * 3.168808781402895e-8
    return -(mass_mjup * mjup2msol / el["M"]) * K * (np.cos(nu + el["w"]) + el["e"] * np.cos(el["w"]))


def draw_walkers(rng, W, a_lo=1.0, a_hi=100.0, with_mass=False):
    """SURVEY §8(d) config-2 prior draws -> elems [9, W]."""
    a = np.exp(rng.uniform(np.log(a_lo), np.log(a_hi), W))
    e = rng.uniform(0.0, 0.95, W)
    inc = np.arccos(rng.uniform(-1.0, 1.0, W))                 # Sine() prior on [0, π]
    w = rng.uniform(0.0, 2 * np.pi, W)
    O = rng.uniform(0.0, 2 * np.pi, W)
    M = np.abs(rng.normal(1.2, 0.1, W)) + 1e-3
    plx = rng.normal(50.0, 0.02, W)
    P_d = K_YR * np.sqrt(a ** 3 / M)
    tp = 50000.0 + rng.uniform(0.0, 1.0, W) * P_d
    mass = rng.uniform(0.0, 20.0, W) if with_mass else np.zeros(W)
    return np.stack([a, e, inc, w, O, tp, M, plx, mass])


def config_astrom(n_epochs=10_000, n_walkers=10_000, cfg=2, sigma=10.0, seed=None):
    """BASELINE configs 2/3: 1 planet, RA/Dec epochs t_j = 50000 + j, σ = 10 mas, prior-drawn walkers."""
    rng = np.random.default_rng(20260929 + cfg if seed is None else seed)
    t = 50000.0 + np.arange(n_epochs, dtype=np.float64)
    ra, dec = truth_radec(t)
    ra = ra + rng.normal(0.0, sigma, n_epochs)
    dec = dec + rng.normal(0.0, sigma, n_epochs)
    table = dict(epoch=t, ra=ra, dec=dec, σ_ra=np.full(n_epochs, sigma), σ_dec=np.full(n_epochs, sigma))
    elems = draw_walkers(rng, n_walkers)
    return dict(n_epochs=n_epochs, n_walkers=n_walkers, table=table, elems=elems,
                theta_example=dict(M=1.2, plx=50.0, planets=dict(b=dict(a=10.0, e=0.3, i=1.0, ω=0.5, Ω=2.0, tp=50000.0))))


def config_wide_prior(n_epochs=10_000, n_walkers=10_000, a_lo=0.3, a_hi=100.0, seed=None):
    """Round 6 workload `wide_prior`: config 3's daily table with a ~ LogUniform(0.3, 100) AU — 14 % of the walkers have periods below ~200 days,
    too short for a warm start at a one-day cadence (ΔM > 0.0315 rad); drawn at random every tile of 64 holds some of them."""
    cfg = config_astrom(n_epochs=n_epochs, n_walkers=n_walkers, cfg=3, seed=seed)
    rng = np.random.default_rng(20260929 + 61 if seed is None else seed + 1)
    cfg["elems"] = draw_walkers(rng, n_walkers, a_lo, a_hi)
    return cfg


def gappy_epochs(n_epochs, per_night=4, nights_per_season=125, intra_night=0.02, season=365.25, t0=50000.0, rng=None):
    """Epochs of a ground-based RV campaign: `per_night` exposures `intra_night` days apart, one night after the other (with a little
    scatter in the start of a night) for `nights_per_season` nights, then nothing until the next season. Sorted."""
    rng = rng or np.random.default_rng(0)
    t = []
    s = 0
    while len(t) < n_epochs:
        for night in range(nights_per_season):
            start = t0 + s * season + night + rng.uniform(-0.05, 0.05)
            t.extend(start + intra_night * np.arange(per_night))
        s += 1
    return np.sort(np.asarray(t[:n_epochs], dtype=np.float64))


def config_rv_gappy(n_epochs=10_000, n_walkers=10_000, nuis=False, seed=None):
    """Round 6 workload `rv_gappy`: ONE planet, an absolute-RV table of nightly runs with seasonal gaps (gappy_epochs: 1e4 epochs = 20 seasons of
    125 nights x 4 exposures; steps of 0.02 d, ~0.94 d and ~241 d), walkers from config 3's prior with a mass. nuis: per-walker offset and jitter
    (rv-absolute.jl:172-204's θ_obs) — otherwise the jitter == 0 path with the offset folded into the data."""
    rng = np.random.default_rng(20260929 + 62 if seed is None else seed)
    t = gappy_epochs(n_epochs, rng=rng)
    rv = truth_rv_star(t, TRUTH, 8.0) + rng.normal(0.0, 3.0, n_epochs)
    table = dict(epoch=t, rv=rv, σ_rv=np.full(n_epochs, 3.0))
    elems = draw_walkers(rng, n_walkers, with_mass=True)
    nz = None
    if nuis:
        nz = np.zeros((N_NUIS, n_walkers))
        nz[0] = rng.normal(0.0, 3.0, n_walkers)
        nz[1] = np.exp(rng.uniform(np.log(0.1), np.log(10.0), n_walkers))
    return dict(n_epochs=n_epochs, n_walkers=n_walkers, table=table, elems=elems, nuis=nz,
                theta_example=dict(M=1.2, plx=50.0, planets=dict(b=dict(a=10.0, e=0.3, i=1.0, ω=0.5, Ω=2.0, tp=50000.0, mass=8.0))))


def config_small(n_epochs=96, n_walkers=257, seed=7):
    return config_astrom(n_epochs=n_epochs, n_walkers=n_walkers, seed=seed)


def to_mirror(pkg, cfg, name="astrom"):
    obs = pkg.PlanetRelAstromObs(cfg["table"], name=name)
    planet = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[obs])
    return obs, planet


def config_two_planet(n_astrom=2500, n_rv=2500, n_walkers=4096, seed=None):
    """BASELINE config 4: 2 planets, RA/Dec on the outer one + absolute RV, inner-barycentre term active."""
    rng = np.random.default_rng(20260929 + 4 if seed is None else seed)
    inner = dict(a=3.0, e=0.1, i=1.0, w=1.0, O=2.0, tp=50100.0, M=1.2, plx=50.0)
    outer = dict(TRUTH, a=15.0)
    m_in, m_out = 5.0, 10.0
    t_a = 50000.0 + 4.0 * np.arange(n_astrom, dtype=np.float64)
    ra, dec = truth_radec(t_a, outer)
    ra_i, dec_i = truth_radec(t_a, inner)
    f = m_in * 0.0009545942339693249 / inner["M"]
    ra = ra + f * ra_i + rng.normal(0, 10.0, n_astrom)
    dec = dec + f * dec_i + rng.normal(0, 10.0, n_astrom)
    t_r = 50001.0 + 4.0 * np.arange(n_rv, dtype=np.float64)
    rv = truth_rv_star(t_r, inner, m_in) + truth_rv_star(t_r, outer, m_out) + 12.0 + rng.normal(0, 5.0, n_rv)
    astrom = dict(epoch=t_a, ra=ra, dec=dec, σ_ra=np.full(n_astrom, 10.0), σ_dec=np.full(n_astrom, 10.0))
    rvtab = dict(epoch=t_r, rv=rv, σ_rv=np.full(n_rv, 5.0))
    e1 = draw_walkers(rng, n_walkers, 1.0, 5.0, with_mass=True)
    e2 = draw_walkers(rng, n_walkers, 8.0, 40.0, with_mass=True)
    e2[6] = e1[6]
    e2[7] = e1[7]      # shared system M, plx (merge(θ_system, θ_planet))
    elems = np.concatenate([e1, e2])
    # nuisances in evaluation order: [astrom on planet c (jitter, platescale, northangle), RV (offset, jitter, -)]
    nuis = np.zeros((2 * N_NUIS, n_walkers))
    nuis[0] = rng.uniform(0.0, 5.0, n_walkers)
    nuis[1] = rng.normal(1.0, 0.01, n_walkers)
    nuis[2] = rng.normal(0.0, 0.01, n_walkers)
    nuis[3] = rng.normal(12.0, 3.0, n_walkers)
    nuis[4] = np.exp(rng.uniform(np.log(0.1), np.log(10.0), n_walkers))
    return dict(astrom=astrom, rv=rvtab, elems=elems, nuis=nuis, n_walkers=n_walkers, n_rows=n_astrom + n_rv)


def active_mask(n_planets, n_obs, mass=True, nuis=True):
    m = np.ones(n_planets * N_EL + n_obs * N_NUIS, dtype=np.uint8)
    if not mass:
        m[[p * N_EL + 8 for p in range(n_planets)]] = 0
    if not nuis:
        m[n_planets * N_EL:] = 0
    return m
