"""Noise-free sky track of the reference's example orbit, for fixture generation (oracle/make_golden.py).
Same construction as tests/synth.py::truth_radec, kept here so oracle/ does not import from tests/."""
import numpy as np

K_YR = 365.2568983840419
TRUTH = dict(a=10.0, e=0.3, i=1.0, w=0.5, O=2.0, tp=50000.0, M=1.2, plx=50.0)   # examples/ofti_rejection_sampling.jl:26-35


def truth_radec(t, el=TRUTH):
    P_d = K_YR * np.sqrt(el["a"] ** 3 / el["M"])
    MA = 2 * np.pi * (t - el["tp"]) / P_d
    M = MA - 2 * np.pi * np.round(MA / (2 * np.pi))
    E = M + el["e"] * np.sin(M)
    for _ in range(60):
        E = E - (E - el["e"] * np.sin(E) - M) / (1 - el["e"] * np.cos(E))
    X = el["a"] * (np.cos(E) - el["e"])
    Y = el["a"] * np.sqrt(1 - el["e"] ** 2) * np.sin(E)
    cw, sw, ci = np.cos(el["w"]), np.sin(el["w"]), np.cos(el["i"])
    cO, sO = np.cos(el["O"]), np.sin(el["O"])
    east = X * (cw * sO + sw * ci * cO) + Y * (-sw * sO + cw * ci * cO)
    north = X * (cw * cO - sw * ci * sO) + Y * (-sw * cO - cw * ci * sO)
    return east * el["plx"], north * el["plx"]
