"""
make_golden.py — writes tests/golden/fixtures.json from the independent 50-digit oracle
(oracle/mp_oracle.py). Run in the build container:  python oracle/make_golden.py

TEST INFRASTRUCTURE. The tables below that come from the reference's own tests are DATA the
reference ships (cited per case); expected values are computed here, because the reference holds
no golden log-likelihood for this path and cannot be run (no Julia).
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import mpmath as mp
import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import mp_oracle as mpo  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "golden" / "fixtures.json"
C = mpo.DEFAULT_CONSTS
VIS = dict(orbit_kind=0, has_mass=False)
VISM = dict(orbit_kind=0, has_mass=True)
RVO = dict(orbit_kind=1, has_mass=True)


def fl(x):
    return float(x)


def run_case(name, planets, obs, elems, nuis, note):
    """elems: [P*9][W] array; nuis: [n_obs*3][W] or None."""
    elems = np.asarray(elems, dtype=np.float64)
    W = elems.shape[1]
    P = len(planets)
    ll, ge, gn, se, sn = [], [], [], [], []
    for w in range(W):
        el = [[mp.mpf(float(elems[p * 9 + k, w])) for k in range(9)] for p in range(P)]
        nu = None
        if nuis is not None:
            nu = [[mp.mpf(float(nuis[o * 3 + k][w])) for k in range(3)] for o in range(len(obs))]
        f0, g_el, g_nu, s_el, s_nu = mpo.ln_like_and_grad(C, planets, obs, el, nu, with_scale=True)
        ll.append(fl(f0))
        ge.append([fl(g_el[p][k]) for p in range(P) for k in range(9)])
        se.append([fl(s_el[p][k]) for p in range(P) for k in range(9)])
        gn.append([fl(g_nu[o][k]) for o in range(len(obs)) for k in range(3)] if g_nu is not None else None)
        sn.append([fl(s_nu[o][k]) for o in range(len(obs)) for k in range(3)] if s_nu is not None else None)
    case = dict(name=name, note=note, planets=planets,
                obs=[{k: (None if v is None else (v if not isinstance(v, (list, np.ndarray)) else [float(x) for x in v])) for k, v in ob.items()} for ob in obs],
                elems=elems.tolist(), nuis=None if nuis is None else np.asarray(nuis, dtype=np.float64).tolist(),
                ll=ll, g_elems=np.asarray(ge).T.tolist(), s_elems=np.asarray(se).T.tolist(),
                g_nuis=None if nuis is None else np.asarray(gn).T.tolist(),
                s_nuis=None if nuis is None else np.asarray(sn).T.tolist())
    print(f"  {name}: W={W} rows={sum(len(o['epoch']) for o in obs)} ll[0]={ll[0]:.12g}", flush=True)
    return case


def astrom(planet, epoch, y1, y2, s1, s2, cor=None, seppa=False):
    return dict(kind="ASTROM_SEPPA" if seppa else "ASTROM_RADEC", planet=planet, epoch=list(map(float, epoch)),
                y1=list(map(float, y1)), y2=list(map(float, y2)), s1=list(map(float, s1)), s2=list(map(float, s2)),
                cor=None if cor is None else list(map(float, cor)))


def rvtab(kind, planet, epoch, rv, s):
    return dict(kind=kind, planet=planet, epoch=list(map(float, epoch)), y1=list(map(float, rv)), y2=None,
                s1=list(map(float, s)), s2=None, cor=None)


def col(*rows):
    return np.array(rows, dtype=np.float64)


def main():
    rng = np.random.default_rng(20260929)
    cases = []

    # ---- F1: northangle sign regression orbit, /root/reference/test/unit/likelihoods.jl:38-58 -------------
    ep = [50000.0, 50300.0, 50600.0, 50900.0, 51200.0]
    el1 = [15.0, 0.2, 0.6, 0.3, 1.1, 50000.0, 1.2, 50.0, 0.0]
    orb = mpo._orbit(C, 0, [mp.mpf(x) for x in el1])
    sols = [mpo.solve(orb, t) for t in ep]
    ra_m = [s["ra"] for s in sols]
    dec_m = [s["dec"] for s in sols]
    pa_m = [mp.atan2(r, d) for r, d in zip(ra_m, dec_m)]
    sep_m = [mp.sqrt(r * r + d * d) for r, d in zip(ra_m, dec_m)]
    eps = mp.mpf("0.05")
    pa_d = [p + eps for p in pa_m]
    ra_d = [fl(s * mp.sin(p)) for s, p in zip(sep_m, pa_d)]
    dec_d = [fl(s * mp.cos(p)) for s, p in zip(sep_m, pa_d)]
    nu1 = col([0.0, 0.0, 0.0, 0.7], [1.0, 1.0, 1.0, 1.01], [0.0, 0.05, -0.05, -0.03])   # jitter, platescale, northangle
    cases.append(run_case("F1_northangle_seppa", [VIS], [astrom(0, ep, [fl(p) for p in pa_d], [fl(s) for s in sep_m], [0.001] * 5, [1.0] * 5, seppa=True)],
                          np.tile(np.array(el1)[:, None], (1, 4)), nu1, "test/unit/likelihoods.jl:38-58 (sep/PA table)"))
    cases.append(run_case("F1_northangle_radec", [VIS], [astrom(0, ep, ra_d, dec_d, [1.0] * 5, [1.0] * 5)],
                          np.tile(np.array(el1)[:, None], (1, 4)), nu1, "test/unit/likelihoods.jl:38-58 (RA/Dec table)"))

    # ---- F2: tutorial 8-epoch RA/Dec table, /root/reference/test/integration-tests.jl:8-15 -----------------
    ep2 = [50000, 50120, 50240, 50360, 50480, 50600, 50720, 50840]
    ra2 = [-505.7637580573554, -502.570356287689, -498.2089148883798, -492.67768482682357, -485.9770335870402,
           -478.1095526888573, -469.0801731788123, -458.89628893460525]
    dec2 = [-66.92982418533026, -37.47217527025044, -7.927548139010479, 21.63557115669823, 51.147204404903704,
            80.53589069730698, 109.72870493064629, 138.65128697876773]
    W = 16
    a = rng.uniform(8, 20, W); e = rng.uniform(0.0, 0.6, W); inc = np.arccos(rng.uniform(-1, 1, W))
    w_ = rng.uniform(0, 2 * np.pi, W); O = rng.uniform(0, 2 * np.pi, W); M = rng.normal(1.2, 0.1, W); plx = rng.normal(50, 0.02, W)
    tp = 50000 - rng.uniform(0, 1, W) * 365.2568983840419 * np.sqrt(a ** 3 / M)
    el2 = np.stack([a, e, inc, w_, O, tp, M, plx, np.zeros(W)])
    el2[:, 0] = [12.0, 0.11, np.deg2rad(41), np.deg2rad(38), np.deg2rad(16), 41479.14852101943, 1.2000965847634995, 50.0, 0.0]
    cases.append(run_case("F2_tutorial_radec", [VIS], [astrom(0, ep2, ra2, dec2, [10.0] * 8, [10.0] * 8, cor=[0.0] * 8)], el2, None,
                          "test/integration-tests.jl:8-15; walker 0 = the orbit that reproduces the table to 1e-12 mas"))

    # ---- F3: correlated table, /root/reference/test/unit-tests.jl:700-707 ---------------------------------
    ra3 = [-494.4, -495.0, -493.7, -490.4, -485.2, -478.1, -469.1, -458.3]
    dec3 = [-76.7, -44.9, -12.9, 19.1, 51.0, 82.8, 114.3, 145.3]
    s3 = [12.6, 10.4, 9.9, 8.7, 8.0, 6.9, 5.8, 4.2]
    cor3 = [0.2, 0.5, 0.1, -0.8, 0.3, -0.0, 0.1, -0.2]
    nu3 = col([0.0, 0.0, 3.5, 3.5, 0.0, 12.0, 0.5, 0.0], [1.0, 1.0, 1.0, 0.99, 1.02, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.01, -0.02, 0.0, 0.0, 0.0])
    cases.append(run_case("F3_cor_radec", [VIS], [astrom(0, ep2, ra3, dec3, s3, s3, cor=cor3)], el2[:, :8], nu3,
                          "test/unit-tests.jl:700-707 (cor column) with and without jitter/platescale/northangle"))
    cases.append(run_case("F3_cor_radec_nonuis", [VIS], [astrom(0, ep2, ra3, dec3, s3, s3, cor=cor3)], el2[:, :8], None,
                          "same table, no nuisance block (precomputed-Σ branch, relative-astrometry.jl:218-219)"))
    # the same sky positions as a sep/PA table with a correlation column
    pa3 = [fl(mp.atan2(mp.mpf(r), mp.mpf(d))) for r, d in zip(ra3, dec3)]
    sep3 = [fl(mp.sqrt(mp.mpf(r) ** 2 + mp.mpf(d) ** 2)) for r, d in zip(ra3, dec3)]
    cases.append(run_case("F3_cor_seppa", [VIS], [astrom(0, ep2, pa3, sep3, [0.02] * 8, s3, cor=cor3, seppa=True)], el2[:, :8], nu3,
                          "sep/PA form of the F3 table"))

    # ---- F4: jitter sensitivity model, /root/reference/test/unit/distributions.jl:103-131 -------------------
    ep4 = [58000.0, 58200.0, 58400.0]
    el4 = np.tile(col(10.0, 0.2, 0.5, 0.3, 0.4, 58000.0, 1.0, 50.0, 0.0)[:, None], (1, 2))
    nu4 = col([0.001, 300.0], [1.0, 1.0], [0.0, 0.0])
    cases.append(run_case("F4_jitter", [VIS], [astrom(0, ep4, [100.0, 110.0, 120.0], [100.0, 95.0, 90.0], [5.0] * 3, [5.0] * 3)], el4, nu4,
                          "test/unit/distributions.jl:103-131, jitter in {0.001, 300}"))

    # ---- F5: RV, orbit of OctofitterRadialVelocity/test/runtests.jl:173-184 -------------------------------
    ep5 = np.linspace(50000.0, 50200.0, 20)
    a5 = float(np.cbrt(80.0 ** 2 * 1.0))
    W5 = 6
    el5 = np.stack([np.full(W5, a5) * rng.uniform(0.9, 1.1, W5), rng.uniform(0, 0.5, W5), np.zeros(W5), rng.uniform(0, 6.28, W5), np.zeros(W5),
                    50000.0 + rng.uniform(-50, 50, W5), rng.normal(1.0, 0.05, W5), np.zeros(W5), rng.uniform(1, 30, W5)])
    el5[:, 0] = [a5, 0.0, 0.0, 0.0, 0.0, 50000.0, 1.0, 0.0, 5.0]      # the reference's circular orbit (e = 0 early-return path)
    rv5 = 50.0 + 0.1 * rng.normal(0, 1, 20) * 10 + 30 * np.sin(2 * np.pi * (ep5 - 50000) / 1234.5)
    nu5 = col(rng.normal(50, 10, W5), np.exp(rng.uniform(np.log(0.01), np.log(50), W5)), np.zeros(W5))
    cases.append(run_case("F5_rv_relative", [RVO], [rvtab("RV_REL", 0, ep5, rv5, [1.0] * 20)], el5, nu5,
                          "PlanetRelativeRVObs on a RadialVelocityOrbit, runtests.jl:173-184"))
    cases.append(run_case("F5_rv_absolute", [RVO], [rvtab("RV_ABS", -1, ep5, rv5 * 0.01, [0.3] * 20)], el5, nu5 * np.array([[0.01], [0.02], [1.0]]),
                          "StarAbsoluteRVObs, same orbit, reflex of the primary"))
    cases.append(run_case("F5_rv_marginalized", [RVO], [rvtab("RV_ABS_MARG", -1, ep5, rv5 * 0.01, [0.3 + 0.01 * k for k in range(20)])], el5,
                          nu5 * np.array([[0.0], [0.02], [1.0]]), "MarginalizedStarAbsoluteRVObs (rv-absolute-margin.jl:168-181)"))
    elv = el5.copy(); elv[2] = rng.uniform(0.2, 2.9, W5); elv[4] = rng.uniform(0, 6.28, W5); elv[7] = 40.0
    cases.append(run_case("F5_rv_absolute_visual_nonuis", [VISM], [rvtab("RV_ABS", -1, ep5, rv5 * 0.01, [0.3] * 20)], elv, None,
                          "absolute RV on a Visual{KepOrbit} (K carries sin i), no nuisance block"))

    # ---- F6: two planets, inner-barycentre term (relative-astrometry.jl:117-133, rv-relative.jl:145-160) ----
    W6 = 6
    e_in = np.stack([rng.uniform(2.5, 3.5, W6), rng.uniform(0, 0.4, W6), np.arccos(rng.uniform(-1, 1, W6)), rng.uniform(0, 6.28, W6), rng.uniform(0, 6.28, W6),
                     50000 + rng.uniform(0, 1500, W6), np.full(W6, 1.2), np.full(W6, 50.0), np.full(W6, 5.0) * rng.uniform(0.5, 2, W6)])
    e_out = np.stack([rng.uniform(12, 18, W6), rng.uniform(0, 0.5, W6), np.arccos(rng.uniform(-1, 1, W6)), rng.uniform(0, 6.28, W6), rng.uniform(0, 6.28, W6),
                      50000 + rng.uniform(0, 15000, W6), np.full(W6, 1.2), np.full(W6, 50.0), np.full(W6, 10.0) * rng.uniform(0.5, 2, W6)])
    e_out[0, 5] = 2.0      # one walker where the "outer" planet is actually inside: the term must switch off
    el6 = np.concatenate([e_in, e_out])
    ep6 = 50000.0 + 137.0 * np.arange(7)
    ra6 = rng.normal(0, 300, 7); dec6 = rng.normal(0, 300, 7)
    epr = 50010.0 + 91.0 * np.arange(9)
    rv6 = rng.normal(0, 40, 9)
    nu6 = np.concatenate([col(rng.uniform(0, 5, W6), rng.normal(1, 0.01, W6), rng.normal(0, 0.02, W6)),
                          col(rng.normal(0, 20, W6), rng.uniform(0.5, 5, W6), np.zeros(W6)),
                          col(rng.normal(0, 20, W6), rng.uniform(0.5, 5, W6), np.zeros(W6))])
    obs6 = [astrom(1, ep6, ra6, dec6, [8.0] * 7, [9.0] * 7), rvtab("RV_REL", 1, epr, rv6 * 50, [30.0] * 9), rvtab("RV_ABS", -1, epr, rv6, [5.0] * 9)]
    cases.append(run_case("F6_two_planet", [VISM, VISM], obs6, el6, nu6,
                          "astrometry + relative RV on the outer planet, absolute RV on the star; masses ~5 & ~10 Mjup"))
    cases.append(run_case("F6_two_planet_nonuis", [VISM, VISM], obs6, el6, None, "same, no nuisance block"))
    obs6b = [astrom(0, ep6, ra6 * 0.2, dec6 * 0.2, [8.0] * 7, [9.0] * 7), astrom(1, ep6, ra6, dec6, [8.0] * 7, [9.0] * 7, cor=[0.3] * 7),
             rvtab("RV_ABS_MARG", -1, epr, rv6, [5.0] * 9)]
    nu6b = np.concatenate([nu6[:3], nu6[:3] * 0.5 + 0.5, col(np.zeros(W6), rng.uniform(0.5, 5, W6), np.zeros(W6))])
    cases.append(run_case("F6_two_planet_marg", [VISM, VISM], obs6b, el6, nu6b, "astrometry on both planets + marginalised RV"))

    # ---- F7: Kepler edge grid: e in {0,1e-12,.5,.9,.99,.999999} x M in {0,±1e-9,±(π−1e-9),±π,40} -----------
    es = [0.0, 1e-12, 0.5, 0.9, 0.99, 0.999999]
    Ms = [0.0, 1e-9, -1e-9, np.pi - 1e-9, -(np.pi - 1e-9), np.pi, -np.pi, 40.0]
    tp7, a7, M7 = 50000.0, 5.0, 1.0
    P7 = 365.2568983840419 * np.sqrt(a7 ** 3 / M7)
    ep7 = sorted(tp7 + m / (2 * np.pi) * P7 for m in Ms)
    el7 = np.stack([np.full(6, a7), np.array(es), np.full(6, 1.0), np.full(6, 0.5), np.full(6, 2.0), np.full(6, tp7), np.full(6, M7), np.full(6, 50.0), np.zeros(6)])
    ra7 = rng.normal(0, 100, len(ep7)); dec7 = rng.normal(0, 100, len(ep7))
    cases.append(run_case("F7_kepler_edges", [VIS], [astrom(0, ep7, ra7, dec7, [10.0] * len(ep7), [10.0] * len(ep7))], el7, None,
                          "eccentricity / mean-anomaly edge grid incl. t == tp"))

    # ---- F8: ObsPriorAstromONeil2019 (src/likelihoods/prior-observable.jl:78-137) on the F3 / F4 tables ---------
    o8 = astrom(0, ep4, [100.0, 110.0, 120.0], [100.0, 95.0, 90.0], [5.0] * 3, [5.0] * 3); o8["kind"] = "ONEIL_RADEC"
    el8 = el4.copy(); el8[5] = 57990.3    # keep every epoch away from tp: |·| in the O'Neil term has a kink at M = 0
    cases.append(run_case("F8_oneil_jitter", [VIS], [o8], el8, nu4, "test/unit/distributions.jl:103-131 with wrap=true (the wrapper alone)"))
    o8b = astrom(0, ep2, ra3, dec3, s3, s3, cor=cor3); o8b["kind"] = "ONEIL_RADEC"
    cases.append(run_case("F8_oneil_cor_plus_plain", [VIS], [astrom(0, ep2, ra3, dec3, s3, s3, cor=cor3), o8b], el2[:, :8], np.concatenate([nu3, nu3]),
                          "astrometry table and its O'Neil wrapper both attached (test/unit-tests.jl:465-468): wrapped epochs counted twice"))
    o8c = astrom(0, ep2, pa3, sep3, [0.02] * 8, s3, seppa=True); o8c["kind"] = "ONEIL_SEPPA"
    cases.append(run_case("F8_oneil_seppa_nonuis", [VIS], [o8c], el2[:, :8], None, "sep/PA table under the wrapper, no nuisance block"))
    o8d = astrom(1, ep6, ra6, dec6, [8.0] * 7, [9.0] * 7); o8d["kind"] = "ONEIL_RADEC"
    cases.append(run_case("F8_oneil_two_planet", [VISM, VISM], [o8d, rvtab("RV_ABS", -1, epr, rv6, [5.0] * 9)], el6, None,
                          "wrapper on the outer planet of the F6 system (inner-barycentre term active) + absolute RV"))

    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_text(json.dumps(dict(consts=C, cases=cases, generator="oracle/make_golden.py (mpmath %s, dps=%d)" % (mp.__version__, mp.mp.dps)), indent=0))
    print("wrote", OUT, OUT.stat().st_size, "bytes")


def ofti_cases():
    """tests/golden/ofti.json — OFTI marginal likelihood (src/parameterizations.jl:318-405) on the setup of
    examples/ofti_rejection_sampling.jl:26-49,67-82: truth orbit a=10, e=0.3, i=1, ω=0.5, Ω=2, tp=50000, M=1.2, plx=50;
    8 epochs 50000..50840, σ=10 mas, σ_ABFG=1000; walkers drawn from that example's priors."""
    import synth_free as sf
    rng = np.random.default_rng(20260929 + 33)
    out = []
    for name, n_ep, with_cor, sig in (("ofti_example", 8, False, 1000.0), ("ofti_cor_37", 37, True, 300.0)):
        ep = np.linspace(50000.0, 50840.0, n_ep)
        ra, dec = sf.truth_radec(ep)
        ra = ra + 10.0 * rng.normal(size=n_ep); dec = dec + 10.0 * rng.normal(size=n_ep)
        s_ra = np.full(n_ep, 10.0) if not with_cor else rng.uniform(5, 15, n_ep)
        s_dec = np.full(n_ep, 10.0) if not with_cor else rng.uniform(5, 15, n_ep)
        cor = None if not with_cor else rng.uniform(-0.8, 0.8, n_ep)
        W = 12
        M = np.abs(rng.normal(1.2, 0.1, W)) + 0.1; plx = rng.normal(50.0, 0.5, W)
        e = rng.uniform(0, 0.99, W); a = np.exp(rng.uniform(0, np.log(100.0), W)); tau = rng.uniform(0, 1, W)
        e[0], a[0], M[0], plx[0], tau[0] = 0.3, 10.0, 1.2, 50.0, 0.0          # the truth
        e[1] = 0.0                                                            # circular: early-return branch of the solver
        tp = ep[0] + tau * np.sqrt(a ** 3 / M) * C["kepler_year_to_julian_day"]
        nl = np.stack([e, a, tp, M, plx])
        res = [mpo.ofti_linear_solve(C, list(ep), list(ra), list(dec), list(s_ra), list(s_dec), None if cor is None else list(cor), sig, *nl[:, w]) for w in range(W)]
        out.append(dict(name=name, epochs=ep.tolist(), ra=ra.tolist(), dec=dec.tolist(), s_ra=s_ra.tolist(), s_dec=s_dec.tolist(),
                        cor=None if cor is None else cor.tolist(), sigma_abfg=sig, nl=nl.tolist(),
                        abfg=[[fl(r[k]) for r in res] for k in range(4)], logml=[fl(r[4]) for r in res]))
        print(f"  {name}: W={W} logml[0]={out[-1]['logml'][0]:.12g}", flush=True)
    p = ROOT / "tests" / "golden" / "ofti.json"
    p.write_text(json.dumps(dict(consts=C, cases=out, generator="oracle/make_golden.py ofti_cases (mpmath dps=%d)" % mp.mp.dps), indent=0))
    print("wrote", p, p.stat().st_size, "bytes")


def model_cases():
    """tests/golden/model.json — the full callback ℓπcallback/∇ℓπcallback (src/logdensitymodel.jl:110-177) for
    (1) the D = 11 model every reference test uses (test/integration/sampling.jl:29-64: tutorial 8-epoch table,
        a~Uniform(0,100), e~Uniform(0,0.99), i~Sine(), ω,Ω,θ~UniformCircular(), tp=θ_at_epoch_to_tperi(θ,50000;…),
        M~truncated(Normal(1.2,0.1),lower=0.1), plx~truncated(Normal(50,0.02),lower=0.1)), and
    (2) a D = 26 two-planet model with relative astrometry (jitter, northangle priors), absolute RV (offset, jitter
        priors), LogUniform / Normal / two-sided truncated priors and masses."""
    rng = np.random.default_rng(20260929 + 41)
    P = lambda kind, p0=0.0, p1=0.0, lo=None, hi=None: dict(kind=kind, p0=p0, p1=p1, lo=lo, hi=hi)
    S = lambda kind, i0=0, i1=0, flags=0, value=0.0: dict(kind=kind, i0=i0, i1=i1, flags=flags, value=value)
    N01 = P(2, 0.0, 1.0)
    out = []
    # ---- (1)
    ep2 = [50000, 50120, 50240, 50360, 50480, 50600, 50720, 50840]
    ra2 = [-505.76, -502.57, -498.21, -492.68, -485.98, -478.11, -469.08, -458.90]      # sampling.jl:32-33 (rounded table)
    dec2 = [-66.93, -37.47, -7.93, 21.64, 51.15, 80.54, 109.73, 138.65]
    obs = [astrom(0, ep2, ra2, dec2, [10.0] * 8, [10.0] * 8, cor=[0.0] * 8)]
    priors = [P(3, 1.2, 0.1, 0.1, None), P(3, 50.0, 0.02, 0.1, None), P(0, 0.0, 100.0), P(0, 0.0, 0.99), P(4)] + [N01] * 6
    esrc = [S(1, 2), S(1, 3), S(1, 4), S(2, 5, 6, 1, 2 * np.pi), S(2, 7, 8, 1, 2 * np.pi), S(3, 9, 10, 1, 50000.0), S(1, 0), S(1, 1), S(0)]
    W = 10
    th = rng.normal(0, 1, (11, W))
    th[2] = rng.normal(-1.6, 0.4, W)          # a = 100·logistic(·) ~ 10-25 AU
    th[0] = np.log(rng.normal(1.2, 0.05, W) - 0.1); th[1] = np.log(rng.normal(50.0, 0.02, W) - 0.1)
    th[:, 0] = [np.log(1.1), np.log(49.9), -2.0, -2.0, 0.1, 0.78, 0.62, 0.96, 0.28, -0.5, 0.8]
    res = [mpo.model_logpost_and_grad(C, [VIS], obs, priors, esrc, None, list(th[:, w])) for w in range(W)]
    out.append(dict(name="D11_reference_test_model", planets=[VIS], obs=obs, priors=priors, esrc=esrc, nsrc=None, theta_t=th.tolist(),
                    lp=[fl(r[0]) for r in res], grad=np.array([[fl(v) for v in r[1]] for r in res]).T.tolist()))
    print(f"  D11: lp[0]={out[-1]['lp'][0]:.12g}", flush=True)
    # ---- (2)
    ep6 = 50000.0 + 137.0 * np.arange(7)
    ra6 = rng.normal(0, 300, 7); dec6 = rng.normal(0, 300, 7)
    epr = 50010.0 + 91.0 * np.arange(9); rv6 = rng.normal(0, 40, 9)
    obs2 = [astrom(1, ep6, ra6, dec6, [8.0] * 7, [9.0] * 7, cor=[0.3] * 7), rvtab("RV_ABS", -1, epr, rv6, [5.0] * 9)]
    # θ order: system [M, plx], system obs (rv) [offset, jitter], planet b [a, e, i, ωx, ωy, Ωx, Ωy, tp, mass],
    #          planet c [a, e, i, ωx, ωy, Ωx, Ωy, θx, θy, mass], c's astrometry [jitter, northangle]
    priors2 = [P(3, 1.2, 0.1, 0.5, 2.0), P(2, 50.0, 0.5),                     # M two-sided truncated, plx Normal
               P(2, 0.0, 30.0), P(1, 0.1, 50.0),                             # rv offset, jitter
               P(1, 1.0, 5.0), P(0, 0.0, 0.9), P(4), N01, N01, N01, N01, P(0, 49000.0, 51000.0), P(1, 0.5, 50.0),
               P(1, 8.0, 40.0), P(0, 0.0, 0.9), P(4), N01, N01, N01, N01, N01, N01, P(0, 0.0, 30.0),
               P(1, 0.1, 20.0), P(2, 0.0, 0.05)]
    D2 = len(priors2)
    esrc2 = [S(1, 4), S(1, 5), S(1, 6), S(2, 7, 8, 1, 2 * np.pi), S(2, 9, 10, 1, 2 * np.pi), S(1, 11), S(1, 0), S(1, 1), S(1, 12),
             S(1, 13), S(1, 14), S(1, 15), S(2, 16, 17, 1, 2 * np.pi), S(2, 18, 19, 1, 2 * np.pi), S(3, 20, 21, 1, 50000.0), S(1, 0), S(1, 1), S(1, 22)]
    nsrc2 = [S(1, 23), S(0, value=1.0), S(1, 24), S(1, 2), S(1, 3), S(0)]
    W2 = 6
    th2 = rng.normal(0, 1, (D2, W2))
    th2[1] = 50.0 + 0.5 * rng.normal(0, 1, W2)        # plx ~ Normal(50, 0.5): identity link, keep it physical
    th2[24] = 0.05 * rng.normal(0, 1, W2)             # northangle ~ Normal(0, 0.05)
    res2 = [mpo.model_logpost_and_grad(C, [VISM, VISM], obs2, priors2, esrc2, nsrc2, list(th2[:, w])) for w in range(W2)]
    out.append(dict(name="D25_two_planet_rv", planets=[VISM, VISM], obs=obs2, priors=priors2, esrc=esrc2, nsrc=nsrc2, theta_t=th2.tolist(),
                    lp=[fl(r[0]) for r in res2], grad=np.array([[fl(v) for v in r[1]] for r in res2]).T.tolist()))
    print(f"  D{D2}: lp[0]={out[-1]['lp'][0]:.12g}", flush=True)
    p = ROOT / "tests" / "golden" / "model.json"
    p.write_text(json.dumps(dict(consts=C, cases=out, generator="oracle/make_golden.py model_cases (mpmath dps=%d)" % mp.mp.dps), indent=0))
    print("wrote", p, p.stat().st_size, "bytes")


def kep_cases():
    """tests/golden/kep.json — F11: the plain `KepOrbit` basis (a, e, i, ω, Ω, tp, M; src/likelihoods/system.jl:116-118 constructs whatever
    basis the planet declares). It has no parallax, so only radial-velocity tables can be attached: K carries sin i
    (unlike RadialVelocityOrbit), the inclination gets a gradient, Ω and plx do not enter."""
    rng = np.random.default_rng(20260929 + 61)
    KEPM = dict(orbit_kind=3, has_mass=True)
    W = 6
    ep = np.linspace(50000.0, 50900.0, 16)
    def planet(a_lo, a_hi):
        return np.stack([rng.uniform(a_lo, a_hi, W), rng.uniform(0, 0.6, W), rng.uniform(0.1, 3.0, W), rng.uniform(0, 6.28, W), rng.uniform(-7, 7, W),
                         50000 + rng.uniform(-400, 400, W), rng.normal(1.1, 0.05, W), np.full(W, np.nan), rng.uniform(1, 30, W)])
    el = planet(1.0, 3.0)
    rv = rng.normal(0, 40, 16)
    out = []
    nu = col(rng.normal(0, 10, W), np.exp(rng.uniform(np.log(0.1), np.log(20), W)), np.zeros(W))
    el_f = el.copy(); el_f[7] = 0.0       # the parallax row is ignored by this basis: NaN in the GPU/oracle inputs, 0 for the 60-digit arithmetic
    c1 = run_case("F11_kep_rv_absolute", [KEPM], [rvtab("RV_ABS", -1, ep, rv, [2.0 + 0.1 * k for k in range(16)])], el_f, nu,
                  "StarAbsoluteRVObs on a plain KepOrbit planet: K = 2πa sin i/(P√(1−e²))")
    c1["elems"] = np.where(np.isnan(el), None, el).tolist()
    out.append(c1)
    el2 = np.concatenate([planet(0.8, 1.5), planet(3.0, 6.0)])
    el2_f = el2.copy(); el2_f[7] = 0.0; el2_f[16] = 0.0
    nu2 = np.concatenate([col(rng.normal(0, 100, W), np.exp(rng.uniform(np.log(1), np.log(50), W)), np.zeros(W)),
                          col(rng.normal(0, 5, W), np.exp(rng.uniform(np.log(0.1), np.log(5), W)), np.zeros(W))])
    c2 = run_case("F11_kep_two_planets_rel_and_abs", [KEPM, KEPM], [rvtab("RV_REL", 1, ep + 3.0, rng.normal(0, 2000, 16), [60.0] * 16),
                                                                    rvtab("RV_ABS", -1, ep, rv, [2.5] * 16)], el2_f, nu2,
                  "relative RV of the outer planet (inner-companion term active) + absolute RV, both planets plain KepOrbit")
    c2["elems"] = np.where(np.isnan(el2), None, el2).tolist()
    out.append(c2)
    p = ROOT / "tests" / "golden" / "kep.json"
    p.write_text(json.dumps(dict(consts=C, cases=out, generator="oracle/make_golden.py kep_cases (mpmath dps=%d)" % mp.mp.dps), indent=0))
    print("wrote", p, p.stat().st_size, "bytes")


def config1_case():
    """tests/golden/config1.json — BASELINE config 1 as SURVEY.md §8(d) defines it: the D = 11 model of
    test/integration/sampling.jl:29-64 (same priors and derived tp as model_cases (1)) on 50 RA/Dec epochs t_j = 50000 + 17·j,
    σ = 10 mas (truth orbit of examples/ofti_rejection_sampling.jl:26-35 + seeded N(0, 10²) noise), ONE parameter set per call:
    log-posterior and its 11 partials at 60 digits, for 4 θ_t (each evaluated as its own W = 1 call by the tests)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import synth
    rng = np.random.default_rng(20260929 + 1)
    P = lambda kind, p0=0.0, p1=0.0, lo=None, hi=None: dict(kind=kind, p0=p0, p1=p1, lo=lo, hi=hi)
    S = lambda kind, i0=0, i1=0, flags=0, value=0.0: dict(kind=kind, i0=i0, i1=i1, flags=flags, value=value)
    N01 = P(2, 0.0, 1.0)
    t = 50000.0 + 17.0 * np.arange(50)
    ra, dec = synth.truth_radec(t)
    ra = ra + rng.normal(0, 10.0, 50); dec = dec + rng.normal(0, 10.0, 50)
    obs = [astrom(0, t, ra, dec, [10.0] * 50, [10.0] * 50)]
    priors = [P(3, 1.2, 0.1, 0.1, None), P(3, 50.0, 0.02, 0.1, None), P(0, 0.0, 100.0), P(0, 0.0, 0.99), P(4)] + [N01] * 6
    esrc = [S(1, 2), S(1, 3), S(1, 4), S(2, 5, 6, 1, 2 * np.pi), S(2, 7, 8, 1, 2 * np.pi), S(3, 9, 10, 1, 50000.0), S(1, 0), S(1, 1), S(0)]
    W = 4
    th = rng.normal(0, 1, (11, W))
    th[2] = rng.normal(-2.2, 0.3, W)          # a = 100·logistic(·) ~ 7-14 AU
    th[0] = np.log(rng.normal(1.2, 0.05, W) - 0.1); th[1] = np.log(rng.normal(50.0, 0.02, W) - 0.1)
    res = [mpo.model_logpost_and_grad(C, [VIS], obs, priors, esrc, None, list(th[:, w])) for w in range(W)]
    case = dict(name="config1_D11_50_epochs", planets=[VIS], obs=obs, priors=priors, esrc=esrc, nsrc=None, theta_t=th.tolist(),
                lp=[fl(r[0]) for r in res], grad=np.array([[fl(v) for v in r[1]] for r in res]).T.tolist())
    print(f"  config1: lp={case['lp']}", flush=True)
    p = ROOT / "tests" / "golden" / "config1.json"
    p.write_text(json.dumps(dict(consts=C, cases=[case], generator="oracle/make_golden.py config1_case (mpmath dps=%d)" % mp.mp.dps), indent=0))
    print("wrote", p, p.stat().st_size, "bytes")


def hgca_table(hg, N_ave, factor=1.0):
    """Rows and catalogue numbers of an HGCAInstantaneousObs built from a catalogue row (src/likelihoods/hgca.jl:80-145),
    in the C-ABI form (include/octofitter_hip.h: OCTO_HGCA)."""
    mjd = lambda yr: (yr - 2000.0) * 365.25 + 51544.5
    d_hip = [0.0] if N_ave == 1 else np.linspace(-4 * 365.25 / 2, 4 * 365.25 / 2, N_ave)
    d_gaia = [0.0] if N_ave == 1 else np.linspace(-1038.0 / 2, 1038.0 / 2, N_ave)
    rows = []
    for d in d_hip:
        rows += [(mjd(hg["epoch_ra_hip"]) + d, 0, 0), (mjd(hg["epoch_dec_hip"]) + d, 1, 0)]
    for d in d_gaia:
        rows += [(mjd(hg["epoch_ra_gaia"]) + d, 0, 1), (mjd(hg["epoch_dec_gaia"]) + d, 1, 1)]
    rows = np.asarray(rows, dtype=np.float64)
    extra = []
    for tag in ("hip", "hg", "gaia"):
        extra += [hg[f"pmra_{tag}"], hg[f"pmdec_{tag}"], hg[f"pmra_{tag}_error"] * factor, hg[f"pmdec_{tag}_error"] * factor, hg[f"pmra_pmdec_{tag}"]]
    return dict(kind="HGCA", planet=-1, epoch=rows[:, 0].tolist(), y1=rows[:, 1].tolist(), y2=rows[:, 2].tolist(), s1=None, s2=None, cor=None,
                extra=[float(x) for x in extra])


# A catalogue row with the magnitudes of a nearby accelerating star (values made up for the fixture; the HGCA FITS file
# is a download the reference performs at run time and is not available here).
HGCA_ROW = dict(epoch_ra_hip=1991.13, epoch_dec_hip=1991.31, epoch_ra_gaia=2016.05, epoch_dec_gaia=2016.22,
                pmra_hip=4.71, pmdec_hip=-1.86, pmra_hip_error=0.61, pmdec_hip_error=0.49, pmra_pmdec_hip=0.21,
                pmra_hg=4.352, pmdec_hg=-2.013, pmra_hg_error=0.031, pmdec_hg_error=0.024, pmra_pmdec_hg=-0.12,
                pmra_gaia=4.61, pmdec_gaia=-1.72, pmra_gaia_error=0.052, pmdec_gaia_error=0.041, pmra_pmdec_gaia=0.33)


def hgca_cases():
    """F9: HGCAInstantaneousObs (src/likelihoods/hgca.jl:155-400) — alone, averaged, and next to astrometry on two planets."""
    rng = np.random.default_rng(20260929)
    W = 4
    def planet(a_lo, a_hi, m_lo, m_hi):
        return np.stack([rng.uniform(a_lo, a_hi, W), rng.uniform(0, 0.6, W), np.arccos(rng.uniform(-1, 1, W)), rng.uniform(0, 6.28, W), rng.uniform(0, 6.28, W),
                         50000 + rng.uniform(0, 4000, W), np.full(W, 1.2), np.full(W, 50.0), rng.uniform(m_lo, m_hi, W)])
    pm = lambda: col(rng.normal(4.3, 0.2, W), rng.normal(-2.0, 0.2, W), np.zeros(W))
    out = []
    el1 = planet(6, 14, 10, 60)
    out.append(run_case("F9_hgca_instantaneous", [VISM], [hgca_table(HGCA_ROW, 1)], el1, pm(), "one planet, N_ave = 1 (4 rows): hip, hip-gaia and gaia terms"))
    out.append(run_case("F9_hgca_averaged_factor", [VISM], [hgca_table(HGCA_ROW, 5, factor=1.7)], el1, pm(), "N_ave = 5 (20 rows), error inflation factor 1.7"))
    el2 = np.concatenate([planet(2, 4, 2, 12), planet(9, 16, 10, 40)])
    ep = 55000.0 + 211.0 * np.arange(6)
    tab = astrom(1, ep, rng.normal(0, 400, 6), rng.normal(0, 400, 6), [6.0] * 6, [7.0] * 6)
    nu = np.concatenate([col(rng.uniform(0, 3, W), rng.normal(1, 0.01, W), rng.normal(0, 0.02, W)), pm()])
    out.append(run_case("F9_hgca_two_planet_astrom", [VISM, VISM], [tab, hgca_table(HGCA_ROW, 3)], el2, nu,
                        "two massive planets (the reference divides the summed reflex motion by planets x N_ave, hgca.jl:276-308) + astrometry on the outer one"))
    p = ROOT / "tests" / "golden" / "hgca.json"
    p.write_text(json.dumps(dict(consts=C, cases=out, generator="oracle/make_golden.py hgca_cases (mpmath dps=%d)" % mp.mp.dps), indent=0))
    print("wrote", p, p.stat().st_size, "bytes")


TI = dict(orbit_kind=2, has_mass=False)
TIM = dict(orbit_kind=2, has_mass=True)


def campbell_to_ti(el):
    """[9][W] Campbell elements -> the same orbits as ThieleInnesOrbit rows (A, e, B, F, G, tp, M, plx, mass), constants in mas
    (a·plx·rad2as/pc2au times the rotation-matrix entries, src/parameterizations.jl:34-37)."""
    a, e, inc, w, O, tp, M, plx, mass = el
    T = a * plx * C["rad2as"] / C["pc2au"]
    cO, sO, cw, sw, ci = np.cos(O), np.sin(O), np.cos(w), np.sin(w), np.cos(inc)
    return np.stack([T * (cO * cw - sO * sw * ci), e, T * (sO * cw + cO * sw * ci), T * (-cO * sw - sO * cw * ci), T * (-sO * sw + cO * cw * ci), tp, M, plx, mass])


def ti_cases():
    """F10: ThieleInnesOrbit basis (docs/src/thiele-innes.md; constants A, B, F, G in mas, a = α/plx as in
    src/parameterizations.jl:14-19) — astrometry, O'Neil prior, a Thiele-Innes + Campbell pair with the inner-barycentre
    term, HGCA; and the tutorial's D = 9 model through the whole callback."""
    rng = np.random.default_rng(20260929 + 10)
    W = 5
    def planet(a_lo, a_hi, m_lo, m_hi):
        return np.stack([rng.uniform(a_lo, a_hi, W), rng.uniform(0, 0.6, W), np.arccos(rng.uniform(-1, 1, W)), rng.uniform(0, 6.28, W), rng.uniform(0, 6.28, W),
                         50000 + rng.uniform(0, 4000, W), np.full(W, 1.2), np.full(W, 50.0), rng.uniform(m_lo, m_hi, W)])
    out = []
    ep = 50000.0 + 120.0 * np.arange(8)
    tab = astrom(0, ep, rng.normal(0, 400, 8), rng.normal(0, 400, 8), [10.0] * 8, [9.0] * 8, cor=list(rng.uniform(-0.5, 0.5, 8)))
    el1 = campbell_to_ti(planet(6, 14, 0, 0))
    nu = col(rng.uniform(0, 3, W), rng.normal(1, 0.01, W), rng.normal(0, 0.02, W))
    out.append(run_case("F10_ti_astrom_cor_nuis", [TI], [tab], el1, nu, "relative astrometry with cor + nuisances on a Thiele-Innes planet"))
    o = dict(tab); o["kind"] = "ONEIL_RADEC"
    out.append(run_case("F10_ti_oneil_nonuis", [TI], [o], el1, None, "O'Neil observable prior on a Thiele-Innes planet (period from the derived a)"))
    inner, outer = planet(2, 4, 2, 12), planet(9, 16, 10, 40)
    el2 = np.concatenate([campbell_to_ti(inner), outer])
    tab2 = astrom(1, ep, rng.normal(0, 400, 8), rng.normal(0, 400, 8), [6.0] * 8, [7.0] * 8)
    out.append(run_case("F10_ti_inner_campbell_outer", [TIM, VISM], [tab2], el2, None,
                        "massive Thiele-Innes inner planet perturbing astrometry of a Campbell outer planet (ordering by the derived a)"))
    pm = col(rng.normal(4.3, 0.2, W), rng.normal(-2.0, 0.2, W), np.zeros(W))
    el3 = campbell_to_ti(planet(6, 14, 10, 60))
    out.append(run_case("F10_ti_hgca", [TIM], [hgca_table(HGCA_ROW, 2)], el3, pm, "HGCA proper-motion anomaly from a Thiele-Innes planet"))
    p = ROOT / "tests" / "golden" / "ti.json"
    p.write_text(json.dumps(dict(consts=C, cases=out, generator="oracle/make_golden.py ti_cases (mpmath dps=%d)" % mp.mp.dps), indent=0))
    print("wrote", p, p.stat().st_size, "bytes")
    # ---- the tutorial model (docs/src/thiele-innes.md): e~U(0,0.5), A,B,F,G~Normal(0,1000), θ~UniformCircular,
    # tp = θ_at_epoch_to_tperi(θ, 50000; plx, M, e, A, B, F, G), M, plx as in the Campbell tutorial. D = 9.
    P = lambda kind, p0=0.0, p1=0.0, lo=None, hi=None: dict(kind=kind, p0=p0, p1=p1, lo=lo, hi=hi)
    S = lambda kind, i0=0, i1=0, flags=0, value=0.0: dict(kind=kind, i0=i0, i1=i1, flags=flags, value=value)
    ep2 = [50000, 50120, 50240, 50360, 50480, 50600, 50720, 50840]
    ra2 = [-505.76, -502.57, -498.21, -492.68, -485.98, -478.11, -469.08, -458.90]
    dec2 = [-66.93, -37.47, -7.93, 21.64, 51.15, 80.54, 109.73, 138.65]
    obs = [astrom(0, ep2, ra2, dec2, [10.0] * 8, [10.0] * 8, cor=[0.0] * 8)]
    # θ order: system [M, plx], planet b [e, A, B, F, G, θx, θy]
    priors = [P(3, 1.2, 0.1, 0.1, None), P(3, 50.0, 0.02, 0.1, None), P(0, 0.0, 0.5)] + [P(2, 0.0, 1000.0)] * 4 + [P(2, 0.0, 1.0)] * 2
    esrc = [S(1, 3), S(1, 2), S(1, 4), S(1, 5), S(1, 6), S(3, 7, 8, 1 | 2, 50000.0), S(1, 0), S(1, 1), S(0)]
    Wm = 8
    th = rng.normal(0, 1, (9, Wm))
    th[0] = np.log(rng.normal(1.2, 0.05, Wm) - 0.1); th[1] = np.log(rng.normal(50.0, 0.02, Wm) - 0.1)
    th[3:7] = rng.normal(0, 400, (4, Wm))      # A, B, F, G ~ Normal(0, 1000): identity link
    res = [mpo.model_logpost_and_grad(C, [TI], obs, priors, esrc, None, list(th[:, w])) for w in range(Wm)]
    case = dict(name="D9_thiele_innes_tutorial", planets=[TI], obs=obs, priors=priors, esrc=esrc, nsrc=None, theta_t=th.tolist(),
                lp=[fl(r[0]) for r in res], grad=np.array([[fl(v) for v in r[1]] for r in res]).T.tolist())
    print(f"  D9 TI: lp[0]={case['lp'][0]:.12g}", flush=True)
    p = ROOT / "tests" / "golden" / "ti_model.json"
    p.write_text(json.dumps(dict(consts=C, cases=[case], generator="oracle/make_golden.py ti_cases (mpmath dps=%d)" % mp.mp.dps), indent=0))
    print("wrote", p, p.stat().st_size, "bytes")


def trend_cases():
    """tests/golden/trend.json — F13: RV tables with a `trend_function` (OctofitterRadialVelocity/src/rv-absolute.jl:69,143,
    rv-relative.jl:64,131, rv-absolute-margin.jl:52,111). The model is the reference's own "PlanetRelativeRV with offset and trend"
    test (OctofitterRadialVelocity/test/runtests.jl:168-231): P = 80, M = 1, a = ∛(P²·M), circular RadialVelocityOrbit, tp = 50000,
    20 epochs 50000…50200, σ_rv = 1 m/s, observed = planet + offset 50 + slope 0.1·(epoch − 50000) + noise, trend_function =
    (θ_obs, epoch) -> θ_obs.trend_slope * (epoch - ref_epoch). On the C ABI the trend is the θ_obs variable it is linear in
    (nuisance row OCTO_NU_RV_TREND) times the closure evaluated at that variable = 1, one value per row (`extra`)."""
    rng = np.random.default_rng(20260929 + 71)
    out = []
    ref_epoch, P_true, M_true = 50000.0, 80.0, 1.0
    a_true = float(np.cbrt(P_true ** 2 * M_true))
    ep = np.linspace(50000.0, 50200.0, 20)
    orb = mpo._orbit(C, 1, [mp.mpf(x) for x in [a_true, 0.0, 0.0, 0.0, 0.0, ref_epoch, M_true, 0.0, 0.0]])
    planet_rv = np.array([fl(mpo.solve(orb, t)["rv"]) for t in ep])
    observed = planet_rv + 50.0 + 0.1 * (ep - ref_epoch) + rng.normal(0, 1.0, 20)
    basis = ep - ref_epoch                              # trend_function(θ_obs with trend_slope = 1, epoch)
    def rvt(kind, planet, epoch, rv, s, extra):
        d = rvtab(kind, planet, epoch, rv, s)
        d["extra"] = None if extra is None else [float(x) for x in extra]
        return d
    W = 6
    el = np.stack([np.full(W, a_true) * rng.uniform(0.9, 1.1, W), rng.uniform(0, 0.5, W), np.zeros(W), rng.uniform(0, 6.28, W), np.zeros(W),
                   ref_epoch + rng.uniform(-20, 20, W), rng.normal(1.0, 0.05, W), np.zeros(W), np.zeros(W)])
    el[:, 0] = [a_true, 0.0, 0.0, 0.0, 0.0, ref_epoch, M_true, 0.0, 0.0]          # the test's fixed orbit (mass = 0.0)
    nu = col(rng.normal(50, 10, W), np.exp(rng.uniform(np.log(0.01), np.log(50), W)), rng.normal(0.1, 0.05, W))   # offset, jitter, trend_slope
    nu[:, 0] = [50.0, 1.0, 0.1]                                                    # the truth
    out.append(run_case("F13_trend_relative_reference_test", [RVO], [rvt("RV_REL", 0, ep, observed, [1.0] * 20, basis)], el, nu,
                        "OctofitterRadialVelocity/test/runtests.jl:168-231: PlanetRelativeRVObs, offset + linear trend"))
    # StarAbsoluteRVObs with the documented trend θ_obs.trend_slope * (epoch - 57000) (rv-absolute.jl:26), on a Visual{KepOrbit} with a mass
    W2 = 5
    elv = np.stack([rng.uniform(2, 5, W2), rng.uniform(0, 0.6, W2), rng.uniform(0.2, 2.9, W2), rng.uniform(0, 6.28, W2), rng.uniform(0, 6.28, W2),
                    57000 + rng.uniform(-400, 400, W2), rng.normal(1.1, 0.05, W2), np.full(W2, 40.0), rng.uniform(1, 30, W2)])
    ep2 = 56800.0 + 23.0 * np.arange(24) + rng.uniform(0, 5, 24)
    rv2 = rng.normal(0, 30, 24) + 0.02 * (ep2 - 57000.0)
    nu2 = col(rng.normal(0, 10, W2), np.exp(rng.uniform(np.log(0.1), np.log(20), W2)), rng.normal(0.02, 0.02, W2))
    out.append(run_case("F13_trend_absolute", [VISM], [rvt("RV_ABS", -1, ep2, rv2, [2.0 + 0.1 * k for k in range(24)], ep2 - 57000.0)], elv, nu2,
                        "StarAbsoluteRVObs, trend θ_obs.trend_slope * (epoch - 57000) (rv-absolute.jl:26)"))
    # the marginalised kind: trend, no offset (rv-absolute-margin.jl:111) — and a quadratic basis (any closure linear in one variable)
    nu3 = nu2.copy(); nu3[0] = 0.0; nu3[2] = rng.normal(1e-4, 1e-4, W2)
    out.append(run_case("F13_trend_marginalized_quadratic", [VISM], [rvt("RV_ABS_MARG", -1, ep2, rv2, [2.0 + 0.1 * k for k in range(24)], (ep2 - 57000.0) ** 2)], elv, nu3,
                        "MarginalizedStarAbsoluteRVObs, trend θ_obs.curv * (epoch - 57000)^2"))
    # two planets: astrometry on the outer one, relative RV WITH a trend on the outer one, absolute RV WITHOUT a basis column whose
    # third nuisance row is non-zero all the same: ignored, zero gradient
    W3 = 5
    e_in = np.stack([rng.uniform(2.5, 3.5, W3), rng.uniform(0, 0.4, W3), np.arccos(rng.uniform(-1, 1, W3)), rng.uniform(0, 6.28, W3), rng.uniform(0, 6.28, W3),
                     50000 + rng.uniform(0, 1500, W3), np.full(W3, 1.2), np.full(W3, 50.0), 5.0 * rng.uniform(0.5, 2, W3)])
    e_out = np.stack([rng.uniform(12, 18, W3), rng.uniform(0, 0.5, W3), np.arccos(rng.uniform(-1, 1, W3)), rng.uniform(0, 6.28, W3), rng.uniform(0, 6.28, W3),
                      50000 + rng.uniform(0, 15000, W3), np.full(W3, 1.2), np.full(W3, 50.0), 10.0 * rng.uniform(0.5, 2, W3)])
    el6 = np.concatenate([e_in, e_out])
    ep6 = 50000.0 + 137.0 * np.arange(7)
    epr = 50010.0 + 91.0 * np.arange(9)
    rv6 = rng.normal(0, 40, 9)
    obs6 = [astrom(1, ep6, rng.normal(0, 300, 7), rng.normal(0, 300, 7), [8.0] * 7, [9.0] * 7),
            rvt("RV_REL", 1, epr, rv6 * 50, [30.0] * 9, epr - 50400.0), rvt("RV_ABS", -1, epr, rv6, [5.0] * 9, None)]
    nu6 = np.concatenate([col(rng.uniform(0, 5, W3), rng.normal(1, 0.01, W3), rng.normal(0, 0.02, W3)),
                          col(rng.normal(0, 20, W3), rng.uniform(0.5, 5, W3), rng.normal(0, 0.5, W3)),
                          col(rng.normal(0, 20, W3), rng.uniform(0.5, 5, W3), rng.normal(0, 0.5, W3))])
    out.append(run_case("F13_trend_two_planet_mixed", [VISM, VISM], obs6, el6, nu6,
                        "astrometry + relative RV with a trend on the outer planet + absolute RV without a basis column (its third nuisance row is ignored)"))
    p = ROOT / "tests" / "golden" / "trend.json"
    p.write_text(json.dumps(dict(consts=C, cases=out, generator="oracle/make_golden.py trend_cases (mpmath dps=%d)" % mp.mp.dps), indent=0))
    print("wrote", p, p.stat().st_size, "bytes")
    # ---- the same model through the whole callback (D = 3: offset ~ Normal(0, 200), jitter ~ LogUniform(0.01, 50), trend_slope ~ Normal(0, 1);
    # the orbit is fixed, runtests.jl:190-220)
    P = lambda kind, p0=0.0, p1=0.0, lo=None, hi=None: dict(kind=kind, p0=p0, p1=p1, lo=lo, hi=hi)
    S = lambda kind, i0=0, i1=0, flags=0, value=0.0: dict(kind=kind, i0=i0, i1=i1, flags=flags, value=value)
    priors = [P(2, 0.0, 200.0), P(1, 0.01, 50.0), P(2, 0.0, 1.0)]
    esrc = [S(0, value=a_true), S(0), S(0), S(0), S(0), S(0, value=ref_epoch), S(0, value=M_true), S(0), S(0)]
    nsrc = [S(1, 0), S(1, 1), S(1, 2)]
    Wm = 6
    th = np.stack([rng.normal(50, 20, Wm), rng.normal(0, 1.5, Wm), rng.normal(0.1, 0.1, Wm)])
    th[:, 0] = [50.0, 0.2, 0.1]
    obs = [rvt("RV_REL", 0, ep, observed, [1.0] * 20, basis)]
    res = [mpo.model_logpost_and_grad(C, [RVO], obs, priors, esrc, nsrc, list(th[:, w])) for w in range(Wm)]
    case = dict(name="D3_relative_rv_offset_trend", planets=[RVO], obs=obs, priors=priors, esrc=esrc, nsrc=nsrc, theta_t=th.tolist(),
                lp=[fl(r[0]) for r in res], grad=np.array([[fl(v) for v in r[1]] for r in res]).T.tolist())
    print(f"  D3 trend: lp[0]={case['lp'][0]:.12g}", flush=True)
    p = ROOT / "tests" / "golden" / "trend_model.json"
    p.write_text(json.dumps(dict(consts=C, cases=[case], generator="oracle/make_golden.py trend_cases (mpmath dps=%d)" % mp.mp.dps), indent=0))
    print("wrote", p, p.stat().st_size, "bytes")


def dense_cases():
    """tests/golden/dense.json — round 5's two new code paths against 60 digits.
    F14: DENSE tables (half-day and one-day cadence: rows sorted by epoch at construction, src/likelihoods/relative-astrometry.jl:46-47,
    rv-relative.jl:85-86) — what k_main's warm-started row loop runs on: each Kepler solve starts from the previous row's solution, with a
    wave-uniform fallback to the Markley starter. Walkers whose periods still pass the wave's entry test, eccentricities to 0.93, periastron
    passages inside the table (where the fallback fires). The root is unique, so the 60-digit values are those of ANY correct solver.
    F15: a SIX-planet system (the reference unrolls over any number of planets, src/likelihoods/system.jl:116-118, 156-170): RA/Dec, sep/PA and cor
    tables on four of the planets, relative RV on one, absolute RV, per-walker nuisances — the planet-per-wave kernels (k_mainp, k_finishp)."""
    rng = np.random.default_rng(20260929 + 71)
    out = []
    # ---- F14a: RA/Dec, half-day cadence
    W = 6
    n = 120
    t = 50000.0 + 0.5 * np.arange(n)
    def walkers(a_lo, a_hi, mass=False):
        a = np.exp(rng.uniform(np.log(a_lo), np.log(a_hi), W)); e = rng.uniform(0.0, 0.93, W)
        el = np.stack([a, e, np.arccos(rng.uniform(-1, 1, W)), rng.uniform(0, 6.28, W), rng.uniform(0, 6.28, W),
                       50000.0 + rng.uniform(5.0, 55.0, W), rng.normal(1.2, 0.05, W), rng.normal(50.0, 0.5, W), rng.uniform(1, 20, W) if mass else np.zeros(W)])
        el[1, 0] = 0.9; el[0, 0] = a_lo * 1.05      # the fastest orbit, eccentric, periastron inside the table
        el[1, 1] = 0.0
        return el
    el = walkers(0.75, 20.0)
    ra, dec = rng.normal(0, 200, n), rng.normal(0, 200, n)
    out.append(run_case("F14_dense_radec_half_day", [VIS], [astrom(0, t, ra, dec, [5.0] * n, [7.0] * n)], el, None,
                        "RA/Dec table at half-day cadence, no nuisances: k_main<1, ·, false, RADEC> warm loop"))
    # ---- F14b: RA/Dec + cor with per-walker jitter / platescale / northangle, one-day cadence
    n = 90
    t = 50000.0 + 1.0 * np.arange(n)
    el = walkers(1.2, 25.0)
    nu = col(rng.uniform(0.5, 4, W), rng.normal(1, 0.01, W), rng.normal(0, 0.02, W))
    nu[0, 2] = 0.0
    out.append(run_case("F14_dense_radec_cor_nuisances", [VIS], [astrom(0, t, rng.normal(0, 200, n), rng.normal(0, 200, n), rng.uniform(3, 9, n), rng.uniform(3, 9, n),
                                                                          cor=rng.uniform(-0.6, 0.6, n))], el, nu,
                        "RA/Dec + cor at one-day cadence with jitter / platescale / northangle: k_main<1, ·, true, RADEC|COR> warm loop"))
    # ---- F14c: sep/PA + relative RV + absolute RV, no nuisances (the widest kind set with a warm loop), one-day cadence
    n = 70
    t = 50000.0 + 1.0 * np.arange(n)
    el = walkers(1.2, 25.0, mass=True)
    ra, dec = rng.normal(0, 300, n), rng.normal(0, 300, n)
    obs = [astrom(0, t, np.arctan2(ra, dec), np.hypot(ra, dec), [0.02] * n, [6.0] * n, seppa=True),
           rvtab("RV_REL", 0, t[::2], rng.normal(0, 800, n // 2), [50.0] * (n // 2)),
           rvtab("RV_ABS", -1, t, rng.normal(0, 30, n), [3.0] * n)]
    out.append(run_case("F14_dense_seppa_rv", [VISM], obs, el, None, "sep/PA + relative RV + absolute RV at one-day cadence, no nuisances: warm loops of both row bodies"))
    # ---- F15: six planets
    P = 6
    W = 4
    planets = [VISM] * P
    els = []
    for p in range(P):
        a = rng.uniform(1.5 + 4 * p, 4.5 + 4 * p, W)
        els.append(np.stack([a, rng.uniform(0, 0.7, W), np.arccos(rng.uniform(-1, 1, W)), rng.uniform(-7, 7, W), rng.uniform(-7, 7, W),
                             50000 + rng.uniform(-3000, 3000, W), np.zeros(W), np.zeros(W), rng.uniform(1, 30, W)]))
    el6 = np.concatenate(els)
    Mt, plx = rng.uniform(0.9, 1.5, W), rng.uniform(20, 60, W)
    for p in range(P):
        el6[p * 9 + 6] = Mt; el6[p * 9 + 7] = plx
    el6[0, 0] = el6[9, 0] * 1.5      # planet 0 outside planet 1 for one walker (the strictly-inner rule)
    obs = []
    for p, kind in ((0, "radec"), (1, "seppa"), (3, "cor"), (5, "radec")):
        m = 9
        ep = np.sort(50000 + rng.uniform(0, 4000, m))
        ra, dec = rng.normal(0, 300, m), rng.normal(0, 300, m)
        if kind == "seppa": obs.append(astrom(p, ep, np.arctan2(ra, dec), np.hypot(ra, dec), [0.03] * m, rng.uniform(3, 12, m), seppa=True))
        else: obs.append(astrom(p, ep, ra, dec, rng.uniform(3, 12, m), rng.uniform(3, 12, m), cor=rng.uniform(-0.7, 0.7, m) if kind == "cor" else None))
    ep = np.sort(50000 + rng.uniform(0, 4000, 11))
    obs.append(rvtab("RV_REL", 4, ep, rng.normal(0, 500, 11), rng.uniform(20, 80, 11)))
    ep = np.sort(50000 + rng.uniform(0, 4000, 13))
    obs.append(rvtab("RV_ABS", -1, ep, rng.normal(0, 30, 13), rng.uniform(1, 8, 13)))
    nu = np.zeros((len(obs) * 3, W))
    for io in range(4):
        nu[io * 3] = rng.uniform(0, 4, W); nu[io * 3 + 1] = rng.normal(1, 0.01, W); nu[io * 3 + 2] = rng.normal(0, 0.02, W)
    for io in (4, 5):
        nu[io * 3] = rng.normal(0, 10, W); nu[io * 3 + 1] = np.exp(rng.uniform(np.log(0.1), np.log(10), W))
    out.append(run_case("F15_six_planets", planets, obs, el6, nu, "six planets: RA/Dec, sep/PA, cor tables, relative and absolute RV, per-walker nuisances (k_mainp, k_finishp)"))
    p = ROOT / "tests" / "golden" / "dense.json"
    p.write_text(json.dumps(dict(consts=C, cases=out, generator="oracle/make_golden.py dense_cases (mpmath dps=%d)" % mp.mp.dps), indent=0))
    print("wrote", p, p.stat().st_size, "bytes")


def gappy_cases():
    """tests/golden/gappy.json — F16 (round 6): tables with GAPS, at 60 digits. Nightly runs with seasonal gaps (steps of 0.02 d,
    ~1 d and ~300 d): k_main's warm-started row loop now takes its step bound per wave from a ladder of the table's step quantiles and sends the
    rows beyond it to the cold starter behind a scalar branch (round 5 took the table's largest step for every row: such a table ran cold).
    F16a: absolute RV, no nuisances (the prefetching warm loop with the per-row test); F16b: the same table with per-walker offset / jitter / trend
    (the nuisance kernels' warm loop, plain scalar row loads); F16c: RA/Dec at the same epochs with one walker so fast that the wave takes the
    intra-night rung of the ladder (its night-to-night rows go cold). The root of Kepler's equation is unique: the 60-digit values know nothing of
    starters. Epochs sorted, as the reference's constructors leave them (src/likelihoods/relative-astrometry.jl:46-47, rv-absolute.jl:98-99)."""
    rng = np.random.default_rng(20260929 + 81)
    out = []
    W = 6
    t = []
    for season in range(3):
        for night in range(14):
            start = 50000.0 + 330.0 * season + night + rng.uniform(-0.05, 0.05)
            t.extend(start + 0.02 * np.arange(3))
    t = np.sort(np.asarray(t))
    n = t.size
    def walkers(a_lo, a_hi, mass=False):
        a = np.exp(rng.uniform(np.log(a_lo), np.log(a_hi), W)); e = rng.uniform(0.0, 0.9, W)
        el = np.stack([a, e, np.arccos(rng.uniform(-1, 1, W)), rng.uniform(0, 6.28, W), rng.uniform(0, 6.28, W),
                       50000.0 + rng.uniform(2.0, 12.0, W), rng.normal(1.2, 0.05, W), rng.normal(50.0, 0.5, W), rng.uniform(1, 20, W) if mass else np.zeros(W)])
        el[1, 0] = 0.85; el[0, 0] = a_lo * 1.05      # the fastest orbit, eccentric, periastron inside the first run
        el[1, 1] = 0.0
        return el
    el = walkers(1.1, 20.0, mass=True)
    rv = rng.normal(0, 30, n)
    out.append(run_case("F16_gappy_rv", [VISM], [rvtab("RV_ABS", -1, t, rv, [3.0] * n)], el, None,
                        "absolute RV in nightly runs with seasonal gaps, no nuisances: k_main<1, ·, false, RADEC|RVABS> warm loop with the per-row test"))
    nu = col(rng.normal(0, 5, W), np.exp(rng.uniform(np.log(0.2), np.log(6), W)), rng.normal(0, 2, W))
    tab = rvtab("RV_ABS", -1, t, rv, rng.uniform(2, 5, n))
    tab["extra"] = list(map(float, (t - 50300.0) / 100.0))
    out.append(run_case("F16_gappy_rv_nuisances", [VISM], [tab], el, nu,
                        "the same epochs with per-walker offset / jitter / trend: the nuisance kernels' warm loop (plain scalar row loads)"))
    el = walkers(1.1, 20.0)
    el[0, 2] = 0.2; el[1, 2] = 0.3      # P ~ 30 days: only the intra-night steps pass the wave's veto — the wave takes that rung
    out.append(run_case("F16_gappy_radec_fast_walker", [VIS], [astrom(0, t, rng.normal(0, 200, n), rng.normal(0, 200, n), [5.0] * n, [7.0] * n)], el, None,
                        "RA/Dec at the same epochs, one walker with a 30-day period: the wave takes the intra-night rung, night-to-night rows go cold"))
    p = ROOT / "tests" / "golden" / "gappy.json"
    p.write_text(json.dumps(dict(consts=C, cases=out, generator="oracle/make_golden.py gappy_cases (mpmath dps=%d)" % mp.mp.dps), indent=0))
    print("wrote", p, p.stat().st_size, "bytes")


if __name__ == "__main__":
    sys.path.insert(0, str(ROOT / "oracle"))
    only = [f for f in ("--ofti-only", "--model-only", "--hgca-only", "--ti-only", "--config1-only", "--kep-only", "--trend-only", "--dense-only", "--gappy-only") if f in sys.argv]
    if not only:
        main()
    if not only or "--ofti-only" in only:
        ofti_cases()
    if not only or "--model-only" in only:
        model_cases()
    if not only or "--hgca-only" in only:
        hgca_cases()
    if not only or "--ti-only" in only:
        ti_cases()
    if not only or "--config1-only" in only:
        config1_case()
    if not only or "--kep-only" in only:
        kep_cases()
    if not only or "--trend-only" in only:
        trend_cases()
    if not only or "--dense-only" in only:
        dense_cases()
    if not only or "--gappy-only" in only:
        gappy_cases()
