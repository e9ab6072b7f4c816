import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding as ob
    ob.load_oracle()
    return ob


KIND_IDS = {"ASTROM_RADEC": 0, "ASTROM_SEPPA": 1, "RV_ABS": 2, "RV_ABS_MARG": 3, "RV_REL": 4, "ONEIL_RADEC": 5, "ONEIL_SEPPA": 6, "HGCA": 7}


def case_tables(case):
    """golden JSON case -> (obs_tables for capi.pack_obs, planets, elems, nuis)."""
    obs = []
    for ob in case["obs"]:
        obs.append(dict(kind=KIND_IDS[ob["kind"]], planet=ob["planet"],
                        **{k: (None if ob[k] is None else np.asarray(ob[k], dtype=np.float64)) for k in ("epoch", "y1", "y2", "s1", "s2", "cor")},
                        extra=None if ob.get("extra") is None else np.asarray(ob["extra"], dtype=np.float64)))
    elems = np.asarray(case["elems"], dtype=np.float64)
    nuis = None if case["nuis"] is None else np.asarray(case["nuis"], dtype=np.float64)
    return obs, case["planets"], elems, nuis


@pytest.fixture(scope="session")
def golden():
    with open(ROOT / "tests" / "golden" / "fixtures.json") as f:
        g = json.load(f)
    with open(ROOT / "tests" / "golden" / "hgca.json") as f:      # F9: HGCAInstantaneousObs (oracle/make_golden.py --hgca-only)
        g["cases"] = g["cases"] + json.load(f)["cases"]
    with open(ROOT / "tests" / "golden" / "ti.json") as f:        # F10: ThieleInnesOrbit basis (oracle/make_golden.py --ti-only)
        g["cases"] = g["cases"] + json.load(f)["cases"]
    with open(ROOT / "tests" / "golden" / "kep.json") as f:       # F11: plain KepOrbit basis, RV tables (oracle/make_golden.py --kep-only)
        g["cases"] = g["cases"] + json.load(f)["cases"]
    with open(ROOT / "tests" / "golden" / "trend.json") as f:     # F13: RV tables with a trend_function (oracle/make_golden.py --trend-only)
        g["cases"] = g["cases"] + json.load(f)["cases"]
    with open(ROOT / "tests" / "golden" / "dense.json") as f:     # F14: dense tables (k_main's warm-started row loop); F15: six planets (k_mainp) (--dense-only)
        g["cases"] = g["cases"] + json.load(f)["cases"]
    with open(ROOT / "tests" / "golden" / "gappy.json") as f:     # F16: tables with gaps (the warm loop's per-wave bound and per-row test) (--gappy-only)
        g["cases"] = g["cases"] + json.load(f)["cases"]
    return g


def rel_err(x, ref, scale=None):
    """Per-component error relative to max(|ref|, scale)."""
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    den = np.maximum(np.abs(ref), 0.0 if scale is None else scale)
    den = np.where(den == 0, 1.0, den)
    return np.abs(x - ref) / den
