"""
mp_oracle.py — independent 50-digit oracle for the epoch-loop likelihood path.

TEST INFRASTRUCTURE (only tests/ and oracle/make_golden.py import this). PARITY UNPINNED:
see the header of oracle/octo_oracle.c — the reference is Julia-only, cannot run here and
ships no golden value for this path, so this file pins the *mathematics* instead:

  * Kepler's equation is solved by Newton iteration at 50 digits (no Markley formula);
  * the sky-plane position uses the textbook rotation of the orbital-plane vector
    (r cos ν, r sin ν) by ω, i, Ω with the axis conventions the reference documents in
    docs/src/faq.md:106-143 (+x East/RA, +y North/Dec, +z away; Ω from North through East),
    i.e. neither the reference-order `2atan(ν_fact tan(E/2))` path of oracle/octo_oracle.c nor
    the Thiele-Innes path of the HIP kernels;
  * the likelihood terms are the closed-form Gaussian densities the reference evaluates through
    Distributions.jl (src/likelihoods/relative-astrometry.jl:166-253,
    OctofitterRadialVelocity/src/rv-absolute.jl:172-204, rv-absolute-margin.jl:140-185,
    rv-relative.jl:177-211);
  * gradients are central differences at 60 digits (error ~1e-40), so they test the
    analytic adjoints rather than restate them.

Inputs use the C-ABI layout of include/octofitter_hip.h.
"""
from __future__ import annotations

import mpmath as mp

mp.mp.dps = 60

KINDS = {"ASTROM_RADEC": 0, "ASTROM_SEPPA": 1, "RV_ABS": 2, "RV_ABS_MARG": 3, "RV_REL": 4, "ONEIL_RADEC": 5, "ONEIL_SEPPA": 6, "HGCA": 7}
ORBIT_VISUAL_KEP, ORBIT_RADVEL, ORBIT_THIELE_INNES, ORBIT_KEP = 0, 1, 2, 3
EL = ["a", "e", "i", "w", "O", "tp", "M", "plx", "mass"]
N_EL, N_NUIS = 9, 3

DEFAULT_CONSTS = dict(
    kepler_year_to_julian_day=365.2568983840419,
    year2day_julian=365.25,
    au2m=1.495978707e11,
    sec2year_julian=3.168808781402895e-8,
    pc2au=206265.0,
    rad2as=206265.0,
    mjup2msol=0.0009545942339693249,
)


def kepler_newton(MA, e):
    """Eccentric anomaly in [-π, π] congruent to the root of E - e sin E = MA."""
    twopi = 2 * mp.pi
    M = MA - twopi * mp.nint(MA / twopi)
    E = M if e < mp.mpf("0.8") else mp.pi * mp.sign(M) if M != 0 else mp.mpf(0)
    for _ in range(200):
        f = E - e * mp.sin(E) - M
        fp = 1 - e * mp.cos(E)
        dE = f / fp
        E -= dE
        if abs(dE) < mp.mpf(10) ** (-(mp.mp.dps - 5)):
            break
    return E


def _ti_sma(A, B, F, G, plx):
    """Semi-major axis [AU] of a Thiele-Innes orbit with constants in mas: the constants are a·plx times a rotation
    matrix [[A, F], [B, G]] = R(Ω)·diag(1, cos i)·R(ω) whose singular values are 1 and |cos i|, so a·plx is the larger
    singular value of the 2x2 matrix — computed here from its SVD invariants, not from the reference's u, v formula."""
    fro2 = A * A + B * B + F * F + G * G          # σ1² + σ2²
    det = A * G - B * F                           # ±σ1 σ2
    s1 = mp.sqrt((fro2 + mp.sqrt(fro2 * fro2 - 4 * det * det)) / 2)
    return s1 / plx


def _orbit(c, kind, el):
    if kind == ORBIT_THIELE_INNES:
        A, e, B, F, G, tp, M, plx, _mass = el      # rows a, i, ω, Ω carry A, B, F, G [mas]
        a = _ti_sma(A, B, F, G, plx)
        P_d = mp.mpf(c["kepler_year_to_julian_day"]) * mp.sqrt(a ** 3 / M)
        return dict(ti=(A, B, F, G), a=a, e=e, tp=tp, M=M, P_d=P_d, K=mp.nan, w=mp.mpf(0), i=mp.mpf(0), O=mp.mpf(0), mas_per_au=mp.mpf(1))
    a, e, inc, w, O, tp, M, plx, _mass = el
    P_d = mp.mpf(c["kepler_year_to_julian_day"]) * mp.sqrt(a ** 3 / M)
    P_yr = P_d / mp.mpf(c["year2day_julian"])
    o = dict(a=a, e=e, w=w, tp=tp, M=M, P_d=P_d)
    if kind == ORBIT_KEP:       # plain KepOrbit: inclined (K carries sin i) but at no distance: RV tables only
        o.update(i=inc, O=O, mas_per_au=mp.mpf(0),
                 K=2 * mp.pi * a * mp.sin(inc) / (P_yr * mp.sqrt(1 - e * e)) * mp.mpf(c["au2m"]) * mp.mpf(c["sec2year_julian"]))
    elif kind == ORBIT_VISUAL_KEP:
        o.update(i=inc, O=O, mas_per_au=plx * mp.mpf(c["rad2as"]) / mp.mpf(c["pc2au"]),
                 K=2 * mp.pi * a * mp.sin(inc) / (P_yr * mp.sqrt(1 - e * e)) * mp.mpf(c["au2m"]) * mp.mpf(c["sec2year_julian"]))
    else:
        o.update(i=mp.pi / 2, O=mp.mpf(0), mas_per_au=mp.mpf(0),
                 K=2 * mp.pi * a / (P_yr * mp.sqrt(1 - e * e)) * mp.mpf(c["au2m"]) * mp.mpf(c["sec2year_julian"]))
    return o


def solve(o, t):
    """-> dict(E, nu, r, ra, dec, rv) for the companion relative to the primary."""
    MA = 2 * mp.pi * (mp.mpf(t) - o["tp"]) / o["P_d"]
    E = kepler_newton(MA, o["e"])
    e = o["e"]
    if "ti" in o:
        # sky offsets are linear in the orbital-plane coordinates in units of a: (ΔDec, ΔRA) = [[A, F], [B, G]] · (X, Y)
        A, B, F, G = o["ti"]
        X, Y = mp.cos(E) - e, mp.sqrt(1 - e * e) * mp.sin(E)
        return dict(E=E, nu=mp.atan2(Y, X), r=o["a"] * mp.sqrt(X * X + Y * Y), ra=B * X + G * Y, dec=A * X + F * Y, rv=mp.nan)
    Xo = o["a"] * (mp.cos(E) - e)                       # orbital-plane coordinates, periapsis on +X
    Yo = o["a"] * mp.sqrt(1 - e * e) * mp.sin(E)
    r = mp.sqrt(Xo * Xo + Yo * Yo)
    nu = mp.atan2(Yo, Xo)
    u = nu + o["w"]
    # rotate by ω (in-plane), tilt by i about the line of nodes, rotate node to PA Ω (from North through East)
    north = r * (mp.cos(u) * mp.cos(o["O"]) - mp.sin(u) * mp.cos(o["i"]) * mp.sin(o["O"]))
    east = r * (mp.cos(u) * mp.sin(o["O"]) + mp.sin(u) * mp.cos(o["i"]) * mp.cos(o["O"]))
    rv = o["K"] * (mp.cos(u) + e * mp.cos(o["w"]))
    return dict(E=E, nu=nu, r=r, ra=east * o["mas_per_au"], dec=north * o["mas_per_au"], rv=rv)


def ln_like(c, planets, obs, elems, nuis):
    """One walker: total log-likelihood."""
    return mp.fsum(ln_like_terms(c, planets, obs, elems, nuis))


def ln_like_terms(c, planets, obs, elems, nuis):
    """One walker. elems: [n_planets][9] mpf; nuis: [n_obs][3] mpf or None.
    Returns the additive terms of the log-likelihood: one per table row, one per marginalised-RV table."""
    n_pl = len(planets)
    orbs = [_orbit(c, planets[p]["orbit_kind"], elems[p]) for p in range(n_pl)]
    m_sol = [elems[p][8] * mp.mpf(c["mjup2msol"]) if planets[p]["has_mass"] else mp.mpf(0) for p in range(n_pl)]
    log2pi = mp.log(2 * mp.pi)
    terms = []
    for io, ob in enumerate(obs):
        kind = KINDS[ob["kind"]] if isinstance(ob["kind"], str) else ob["kind"]
        nz = nuis[io] if nuis is not None else None
        ip = ob["planet"]
        ll = mp.mpf(0)
        if kind == 7:
            # HGCAInstantaneousObs (hgca.jl:155-400), formulated independently of the C restatement: the primary's
            # reflex position is −m/M_tot × the companion's offset, its proper motion is the TIME DERIVATIVE of that
            # position (numerical, 60 digits) — no velocity formula is assumed here.
            pm_sys = [nz[0], nz[1]] if nz is not None else [mp.mpf(0), mp.mpf(0)]
            yr = mp.mpf(c["year2day_julian"])
            pos = [[mp.mpf(0)] * 2 for _ in range(2)]; pmv = [[mp.mpf(0)] * 2 for _ in range(2)]
            ep = [[mp.mpf(0)] * 2 for _ in range(2)]; cnt = [[0, 0], [0, 0]]
            for p in range(n_pl):
                if planets[p]["orbit_kind"] == ORBIT_RADVEL:
                    continue
                fac = -m_sol[p] / orbs[p]["M"]
                for j, t in enumerate(ob["epoch"]):
                    ax, m = int(ob["y1"][j]), int(ob["y2"][j])
                    key = "ra" if ax == 0 else "dec"
                    f = lambda tt, o=orbs[p], key=key: solve(o, tt)[key]
                    cnt[m][ax] += 1              # once per (planet, row), as the reference counts (hgca.jl:276-278)
                    ep[m][ax] += mp.mpf(t)
                    pos[m][ax] += fac * f(mp.mpf(t))
                    pmv[m][ax] += fac * mp.diff(f, mp.mpf(t)) * yr      # mas/day -> mas/yr (mp.diff raises the working precision itself)
            for m in range(2):
                for ax in range(2):
                    pos[m][ax] /= cnt[m][ax]; ep[m][ax] /= cnt[m][ax]
                    pmv[m][ax] = pmv[m][ax] / cnt[m][ax] + pm_sys[ax]
            model = [[pmv[0][0], pmv[0][1]],
                     [(pos[1][ax] - pos[0][ax]) / (ep[1][ax] - ep[0][ax]) * yr + pm_sys[ax] for ax in range(2)],
                     [pmv[1][0], pmv[1][1]]]
            x = [mp.mpf(v) for v in ob["extra"]]
            for k in range(3):
                r1, r2 = model[k][0] - x[5 * k], model[k][1] - x[5 * k + 1]
                s1, s2, rho = x[5 * k + 2], x[5 * k + 3], x[5 * k + 4]
                det = (s1 * s2) ** 2 * (1 - rho * rho)
                q = ((r1 / s1) ** 2 - 2 * rho * (r1 / s1) * (r2 / s2) + (r2 / s2) ** 2) / (1 - rho * rho)
                terms.append(-log2pi - mp.log(det) / 2 - q / 2)
            continue
        oneil = kind in (5, 6)
        if oneil:
            kind -= 5
            # O'Neil et al. 2019 observable-based prior, as the reference evaluates it (prior-observable.jl:96-139):
            # 2 log( Σ_j |3M(e + cos E) + 2(−2 + e² + e cos E) sin E| · ∛P / √(1−e²) ), P in Julian years, M = E − e sin E
            o_ = orbs[ip]
            jac = mp.mpf(0)
            for t in ob["epoch"]:
                E = solve(o_, t)["E"]
                Mm = E - o_["e"] * mp.sin(E)
                jac += abs(3 * Mm * (o_["e"] + mp.cos(E)) + 2 * (-2 + o_["e"] ** 2 + o_["e"] * mp.cos(E)) * mp.sin(E))
            if len(ob["epoch"]) > 0:
                terms.append(2 * mp.log(jac * mp.cbrt(o_["P_d"] / mp.mpf("365.25")) / mp.sqrt(1 - o_["e"] ** 2)))
        if kind in (0, 1):
            jitter = nz[0] if nz is not None else mp.mpf(0)
            plate = nz[1] if nz is not None else mp.mpf(1)
            north = nz[2] if nz is not None else mp.mpf(0)
            for j, t in enumerate(ob["epoch"]):
                sols = [solve(o, t) for o in orbs]
                ra = sols[ip]["ra"]
                dec = sols[ip]["dec"]
                for p in range(n_pl):
                    if p != ip and orbs[p]["a"] < orbs[ip]["a"] and planets[p]["has_mass"]:
                        f = m_sol[p] / orbs[p]["M"]          # star about the inner barycentre: -(−m/M · offset)
                        ra += f * sols[p]["ra"]
                        dec += f * sols[p]["dec"]
                if kind == 1:
                    rho = mp.sqrt(ra * ra + dec * dec)
                    pa = mp.atan2(ra, dec)
                    d = (mp.mpf(ob["y1"][j]) + north) - pa
                    d = d - 2 * mp.pi * mp.nint(d / (2 * mp.pi))     # wrapped to [-π, π]
                    r1 = d
                    r2 = mp.mpf(ob["y2"][j]) * plate - rho
                else:
                    x, y = mp.mpf(ob["y1"][j]), mp.mpf(ob["y2"][j])
                    cn, sn = mp.cos(north), mp.sin(north)
                    # data rotated by −northangle (angle measured East through North), scaled by platescale
                    r1 = plate * (x * cn + y * sn) - ra
                    r2 = plate * (y * cn - x * sn) - dec
                v1 = mp.mpf(ob["s1"][j]) ** 2 + jitter ** 2
                v2 = mp.mpf(ob["s2"][j]) ** 2 + jitter ** 2
                rho_c = mp.mpf(ob["cor"][j]) if ob.get("cor") is not None else mp.mpf(0)
                det = v1 * v2 * (1 - rho_c ** 2)
                quad = (r1 * r1 / v1 - 2 * rho_c * r1 * r2 / mp.sqrt(v1 * v2) + r2 * r2 / v2) / (1 - rho_c ** 2)
                terms.append(-log2pi - mp.log(det) / 2 - quad / 2)
        else:
            offset = nz[0] if (nz is not None and kind != 3) else mp.mpf(0)
            jitter = nz[1] if nz is not None else mp.mpf(0)
            # trend_function(θ_obs, epoch_j) = (one θ_obs variable) × (the user's closure at that variable = 1): ob["extra"][j]
            basis = ob.get("extra") if nz is not None else None
            A = B = C = mp.mpf(0)
            for j, t in enumerate(ob["epoch"]):
                sols = [solve(o, t) for o in orbs]
                model = offset
                if basis is not None:
                    model += nz[2] * mp.mpf(basis[j])
                if kind == 4:
                    model += sols[ip]["rv"]
                    for p in range(n_pl):
                        if p != ip and orbs[p]["a"] < orbs[ip]["a"] and planets[p]["has_mass"]:
                            model += -m_sol[p] / orbs[p]["M"] * sols[p]["rv"]
                else:
                    for p in range(n_pl):
                        model += -m_sol[p] / orbs[p]["M"] * sols[p]["rv"]
                resid = mp.mpf(ob["y1"][j]) - model
                var = mp.mpf(ob["s1"][j]) ** 2 + jitter ** 2
                if kind == 3:
                    A += 1 / var
                    B -= 2 * resid / var
                    C += resid ** 2 / var
                    ll -= mp.log(2 * mp.pi * var)
                else:
                    terms.append(-(log2pi + mp.log(var)) / 2 - resid ** 2 / var / 2)
            if kind == 3:
                ll -= -B ** 2 / (4 * A) + C + mp.log(A)
                terms.append(ll)
    return terms


def ln_like_and_grad(c, planets, obs, elems, nuis, h_rel=mp.mpf(10) ** -25, with_scale=False):
    """Central-difference gradient at working precision 60 digits. Returns (ll, g_elems, g_nuis) and, with
    with_scale, also (s_elems, s_nuis) = Σ_terms |∂term/∂θ|: the size of what a gradient component sums over,
    i.e. the scale its rounding error is measured against when the terms cancel."""
    elems = [[mp.mpf(x) for x in row] for row in elems]
    nuis_m = [[mp.mpf(x) for x in row] for row in nuis] if nuis is not None else None
    f0 = ln_like(c, planets, obs, elems, nuis_m)
    g_el = [[mp.mpf(0)] * N_EL for _ in elems]
    s_el = [[mp.mpf(0)] * N_EL for _ in elems]
    for p in range(len(elems)):
        for k in range(N_EL):
            if planets[p]["orbit_kind"] == ORBIT_RADVEL and k in (2, 4, 7):
                continue
            if planets[p]["orbit_kind"] == ORBIT_KEP and k in (4, 7):      # Ω does not enter a radial velocity; no parallax
                continue
            if k == 8 and not planets[p]["has_mass"]:
                continue
            x = elems[p][k]
            h = h_rel * max(abs(x), mp.mpf(1))
            elems[p][k] = x + h
            fp = ln_like_terms(c, planets, obs, elems, nuis_m)
            elems[p][k] = x - h
            fm = ln_like_terms(c, planets, obs, elems, nuis_m)
            elems[p][k] = x
            d = [(a - b) / (2 * h) for a, b in zip(fp, fm)]
            g_el[p][k] = mp.fsum(d)
            s_el[p][k] = mp.fsum(abs(v) for v in d)
    g_nu = s_nu = None
    if nuis_m is not None:
        g_nu = [[mp.mpf(0)] * N_NUIS for _ in nuis_m]
        s_nu = [[mp.mpf(0)] * N_NUIS for _ in nuis_m]
        for io in range(len(nuis_m)):
            for k in range(N_NUIS):
                x = nuis_m[io][k]
                h = h_rel * max(abs(x), mp.mpf(1))
                nuis_m[io][k] = x + h
                fp = ln_like_terms(c, planets, obs, elems, nuis_m)
                nuis_m[io][k] = x - h
                fm = ln_like_terms(c, planets, obs, elems, nuis_m)
                nuis_m[io][k] = x
                d = [(a - b) / (2 * h) for a, b in zip(fp, fm)]
                g_nu[io][k] = mp.fsum(d)
                s_nu[io][k] = mp.fsum(abs(v) for v in d)
    if with_scale:
        return f0, g_el, g_nu, s_el, s_nu
    return f0, g_el, g_nu


def ofti_linear_solve(c, epochs, ra, dec, s_ra, s_dec, cor, sigma_abfg, e, a, tp, M, plx):
    """Independent 60-digit evaluation of the OFTI marginal likelihood (src/parameterizations.jl:288-316 docstring):
    Gaussian integral over (A, B, F, G) ~ N(0, σ²I) of the astrometry likelihood, by completing the square with
    mpmath matrices; Kepler by Newton. Returns (A, B, F, G, log_marginal)."""
    e, a, tp, M = mp.mpf(e), mp.mpf(a), mp.mpf(tp), mp.mpf(M)
    P_d = mp.mpf(c["kepler_year_to_julian_day"]) * mp.sqrt(a ** 3 / M)
    N = len(epochs)
    S = mp.matrix(4, 4)
    b = mp.matrix(4, 1)
    dq = mp.mpf(0); ldc = mp.mpf(0)
    for j in range(N):
        E = kepler_newton(2 * mp.pi * (mp.mpf(epochs[j]) - tp) / P_d, e)
        x = mp.cos(E) - e
        y = mp.sqrt(1 - e * e) * mp.sin(E)
        sr, sd = mp.mpf(s_ra[j]), mp.mpf(s_dec[j])
        rho = mp.mpf(cor[j]) if cor is not None else mp.mpf(0)
        Sig = mp.matrix([[sr * sr, rho * sr * sd], [rho * sr * sd, sd * sd]])
        Wj = Sig ** -1
        Dj = mp.matrix([[0, x, 0, y], [x, 0, y, 0]])
        dj = mp.matrix([mp.mpf(ra[j]), mp.mpf(dec[j])])
        S += Dj.T * Wj * Dj
        b += Dj.T * Wj * dj
        dq += (dj.T * Wj * dj)[0]
        ldc += mp.log(mp.det(Sig))
    lam = 1 / mp.mpf(sigma_abfg) ** 2
    for i in range(4):
        S[i, i] += lam
    mu = mp.lu_solve(S, b)
    pq = (mu.T * b)[0]
    lm = -(dq - pq + mp.log(mp.det(S)) - 4 * mp.log(lam) + ldc) / 2 - N * mp.log(2 * mp.pi)
    return mu[0], mu[1], mu[2], mu[3], lm


# ------------------------------------------------------------------------------------------- standard parameterisation
PRIOR_KINDS = {"UNIFORM": 0, "LOGUNIFORM": 1, "NORMAL": 2, "TRUNCNORMAL": 3, "SINE": 4}
_EPS = mp.mpf(2) ** -52


def _prior(pr, y):
    """(x, logpdf_with_trans) for one prior dict(kind, p0, p1, lo, hi) at unconstrained y — closed forms, 60 digits."""
    kind = pr["kind"]
    inf = mp.inf
    if kind in (0, 1):
        a, b = mp.mpf(pr["p0"]), mp.mpf(pr["p1"])
    elif kind == 3:
        a = -inf if pr["lo"] is None else mp.mpf(pr["lo"])
        b = inf if pr["hi"] is None else mp.mpf(pr["hi"])
    elif kind == 4:
        a, b = _EPS, mp.pi - _EPS          # Float64 eps: minimum/maximum of Sine() in the reference (distributions.jl:31-32)
        a, b = mp.mpf(2.220446049250313e-16), mp.mpf(3.141592653589793) - mp.mpf(2.220446049250313e-16)
    else:
        a, b = -inf, inf
    if a > -inf and b < inf:
        sg = 1 / (1 + mp.exp(-y))
        x = (b - a) * sg + a
        ladj = mp.log(b - a) + mp.log(sg) + mp.log(1 - sg)          # log |dx/dy|
    elif a > -inf:
        x = mp.exp(y) + a
        ladj = y
    elif b < inf:
        x = b - mp.exp(y)
        ladj = y
    else:
        x, ladj = y, mp.mpf(0)
    if kind == 0:
        lp = -mp.log(b - a)
    elif kind == 1:
        lp = -mp.log(x) - mp.log(mp.log(b / a))
    elif kind in (2, 3):
        mu, sg_ = mp.mpf(pr["p0"]), mp.mpf(pr["p1"])
        lp = -((x - mu) / sg_) ** 2 / 2 - mp.log(sg_) - mp.log(2 * mp.pi) / 2
        if kind == 3:
            lo = mp.ncdf((a - mu) / sg_) if a > -inf else mp.mpf(0)
            hi = mp.ncdf((b - mu) / sg_) if b < inf else mp.mpf(1)
            lp -= mp.log(hi - lo)
    else:
        lp = mp.log(mp.sin(x) / 2)
    return x, lp + ladj


def _tperi_ti(c, th, epoch, M, e, A, B, F, G, plx):
    """Thiele-Innes variant: the sky direction at position angle th is (ΔDec, ΔRA) ∝ (cos th, sin th) = [[A, F], [B, G]]·(X, Y);
    invert for the in-plane direction, which is the true anomaly (periapsis on +X)."""
    north, east = mp.cos(th), mp.sin(th)
    det = A * G - F * B
    X = (G * north - F * east) / det
    Y = (A * east - B * north) / det
    nu = mp.atan2(Y, X)
    E = 2 * mp.atan(mp.sqrt((1 - e) / (1 + e)) * mp.tan(nu / 2))
    MA = E - e * mp.sin(E)
    MA = MA - 2 * mp.pi * mp.floor(MA / (2 * mp.pi))
    a = _ti_sma(A, B, F, G, plx)
    P_d = mp.mpf(c["kepler_year_to_julian_day"]) * mp.sqrt(a ** 3 / M)
    return mp.mpf(epoch) - MA / (2 * mp.pi) * P_d


def _tperi(c, th, epoch, M, e, a, inc, w, O):
    """Epoch of periastron such that the position angle at `epoch` is th — textbook route: PA -> true anomaly in the
    orbital plane -> eccentric -> mean anomaly -> tp (independent of the reference's matrix-solve formulation)."""
    # direction on the sky at PA th: (east, north) = (sin th, cos th); invert the rotation for the in-plane angle u = ν + ω
    # east = r (cos u sinΩ + sin u cos i cosΩ), north = r (cos u cosΩ − sin u cos i sinΩ)
    # => cos u ∝ north cosΩ + east sinΩ ; sin u cos i ∝ east cosΩ − north sinΩ
    east, north = mp.sin(th), mp.cos(th)
    cu = north * mp.cos(O) + east * mp.sin(O)
    su = (east * mp.cos(O) - north * mp.sin(O)) / mp.cos(inc)
    nu = mp.atan2(su, cu) - w
    E = 2 * mp.atan(mp.sqrt((1 - e) / (1 + e)) * mp.tan(nu / 2))
    MA = E - e * mp.sin(E)
    MA = MA - 2 * mp.pi * mp.floor(MA / (2 * mp.pi))          # the reference's formula yields MA in [0, 2π)
    P_d = mp.mpf(c["kepler_year_to_julian_day"]) * mp.sqrt(a ** 3 / M)
    return mp.mpf(epoch) - MA / (2 * mp.pi) * P_d


def model_logpost(c, planets, obs, priors, esrc, nsrc, theta_t):
    """One walker. priors: list of dicts; esrc/nsrc: lists of dict(kind, i0, i1, flags, value). Returns lp (mpf)."""
    x = []
    lp = mp.mpf(0)
    for pr, y in zip(priors, theta_t):
        xi, l = _prior(pr, mp.mpf(y))
        x.append(xi)
        lp += l
    n_pl = len(planets)
    elems = [[None] * N_EL for _ in range(n_pl)]
    ul = mp.mpf(0)

    def unit_len(i0, i1):
        r = mp.sqrt(x[i0] ** 2 + x[i1] ** 2)
        return -mp.log(r) - mp.log(mp.mpf("0.1") * mp.sqrt(2 * mp.pi)) - mp.log(r) ** 2 / (2 * mp.mpf("0.01"))

    def resolve(sc, p):
        nonlocal ul
        if sc["kind"] == 0:
            return mp.mpf(sc["value"])
        if sc["kind"] == 1:
            return x[sc["i0"]]
        ang = mp.atan2(x[sc["i1"]], x[sc["i0"]])
        if sc["flags"] & 1:
            ul += unit_len(sc["i0"], sc["i1"])
        if sc["kind"] == 2:
            return ang / (2 * mp.pi) * mp.mpf(sc["value"])
        e_ = elems[p]
        if sc["flags"] & 2:      # OCTO_SRC_FLAG_TI
            return _tperi_ti(c, ang, sc["value"], e_[6], e_[1], e_[0], e_[2], e_[3], e_[4], e_[7])
        return _tperi(c, ang, sc["value"], e_[6], e_[1], e_[0], e_[2], e_[3], e_[4])
    for want in (False, True):
        for k, sc in enumerate(esrc):
            if (sc["kind"] == 3) == want:
                elems[k // N_EL][k % N_EL] = resolve(sc, k // N_EL)
    nuis = None
    if nsrc is not None:
        flat = [resolve(sc, 0) for sc in nsrc]
        nuis = [flat[i * N_NUIS:(i + 1) * N_NUIS] for i in range(len(obs))]
    else:
        nuis = [[mp.mpf(0), mp.mpf(1), mp.mpf(0)] if (KINDS[o["kind"]] if isinstance(o["kind"], str) else o["kind"]) in (0, 1, 5, 6) else [mp.mpf(0)] * 3 for o in obs]
    return lp + ul + ln_like(c, planets, obs, elems, nuis)


def model_logpost_and_grad(c, planets, obs, priors, esrc, nsrc, theta_t, h=mp.mpf(10) ** -25):
    th = [mp.mpf(v) for v in theta_t]
    f0 = model_logpost(c, planets, obs, priors, esrc, nsrc, th)
    g = []
    for k in range(len(th)):
        t0 = th[k]
        th[k] = t0 + h
        fp = model_logpost(c, planets, obs, priors, esrc, nsrc, th)
        th[k] = t0 - h
        fm = model_logpost(c, planets, obs, priors, esrc, nsrc, th)
        th[k] = t0
        g.append((fp - fm) / (2 * h))
    return f0, g
