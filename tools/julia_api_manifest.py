"""
julia_api_manifest.py — which names of the REFERENCE does julia/OctofitterHIP.jl rely on, and where does the reference define them?

The shim cannot be executed in the build image (no Julia). tests/test_abi.py pins its structs, ccall arities and constants against
the header; this script pins the other half — every `Octofitter.<name>` it calls or extends and every field it reads from one of
the reference's structs (`obs.trend_function`, `obs.gaussian_process`, `obs.wrapped_like`, `pl.observations`, `system.priors`, …) —
against the reference's sources, so that API drift in never-executed Julia is caught by the CPU suite.

    python tools/julia_api_manifest.py            # rewrites tests/golden/julia_api_names.json from /root/reference

The manifest is DATA (names and the file:line that defines each); tests/test_abi.py::test_julia_shim_names_exist_in_the_reference
checks the shim against it everywhere and, where /root/reference is present, the manifest against the sources.
"""
from __future__ import annotations

import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
JULIA = ROOT / "octofitter.jl_amd" / "julia" / "OctofitterHIP.jl"
OUT = ROOT / "tests" / "golden" / "julia_api_names.json"
REF = Path("/root/reference")
REF_DIRS = ("src", "OctofitterRadialVelocity/src", "ext")

# Field reads that are NOT on a reference struct: fields of the shim's own structs are parsed from the shim; these are the model's
# variable names inside θ (NamedTuples built by arr2nt: src/variables.jl:1372-1431), ForwardDiff / Base names and local NamedTuples.
THETA_FIELDS = {"planets", "observations", "pmra", "pmdec", "jitter", "platescale", "northangle", "offset", "mass"}
OTHER_FIELDS = {"x", "y", "value", "partials", "Σ", "untruncated", "lower", "upper",      # Distributions.Truncated / MvNormal, ForwardDiff
                "jl", "so", "md"}                                                          # file extensions inside comments / strings
# names of OTHER modules reached with a dot (Random.AbstractRNG, ForwardDiff.Dual, LogDensityProblems.LogDensityOrder, PlanetOrbits.<const>):
# the PlanetOrbits constants are checked where the reference itself names them; au2m, sec2year_julian, rad2as it never names ([PO])
EXTERNAL = {"AbstractRNG", "Dual", "LogDensityOrder", "Xoshiro", "kepler_year_to_julian_day_conversion_factor", "year2day_julian", "au2m", "sec2year_julian",
            "pc2au", "rad2as", "default_rng", "jacobian", "logdensity", "logdensity_and_gradient", "dimension", "capabilities", "nthreads"}
# columns of the reference's observation tables (TypedTables: `obs.table.<col>`) and the fields of the `hgca` NamedTuple an HGCAInstantaneousObs
# carries: not struct fields, but names the reference's constructors fix — looked up as tokens in the file that fixes them
TABLE_COLS = {"epoch": "src/likelihoods/relative-astrometry.jl", "ra": "src/likelihoods/relative-astrometry.jl", "dec": "src/likelihoods/relative-astrometry.jl",
              "pa": "src/likelihoods/relative-astrometry.jl", "sep": "src/likelihoods/relative-astrometry.jl", "σ_ra": "src/likelihoods/relative-astrometry.jl",
              "σ_dec": "src/likelihoods/relative-astrometry.jl", "σ_pa": "src/likelihoods/relative-astrometry.jl", "σ_sep": "src/likelihoods/relative-astrometry.jl",
              "cor": "src/likelihoods/relative-astrometry.jl", "rv": "OctofitterRadialVelocity/src/rv-absolute.jl", "σ_rv": "OctofitterRadialVelocity/src/rv-absolute.jl",
              "meas": "src/likelihoods/hgca.jl", "inst": "src/likelihoods/hgca.jl",
              **{f"{q}_{tag}": "src/likelihoods/hgca.jl" for q in ("pmra", "pmdec", "dist") for tag in ("hip", "hg", "gaia")}}


# fields the shim reads from REFERENCE structs although its own structs (HIPObs, HIPLogDensityModel, …) have fields of the same name,
# which the own-field filter below would otherwise hide: always required
REQUIRED_REFERENCE_FIELDS = ("wrapped_like", "priors", "derived", "observations", "planets", "name", "system", "arr2nt", "sample_priors", "link",
                             "invlink", "D", "starting_points", "ℓπcallback", "∇ℓπcallback", "table", "trend_function", "gaussian_process", "hgca")


def shim_names(txt: str):
    code = re.sub(r'"""(?:.|\n)*?"""', "", txt)                 # docstrings
    code = "\n".join(line.split("#")[0] for line in code.splitlines())
    code = re.sub(r'"(?:\\.|[^"\\])*"', '""', code)             # string literals
    qualified = set(re.findall(r"\bOctofitter\.([A-Za-z_θ]\w*!?)", code))
    m = re.search(r"using Octofitter:\s*([^\n]+)", code)
    imported = {n.strip() for n in m.group(1).split(",")} if m else set()
    own_fields = set()
    for sm in re.finditer(r"^(?:mutable )?struct (\w+)[^\n]*\n(.*?)^end", code, flags=re.S | re.M):
        for line in sm.group(2).splitlines():
            fm = re.match(r"\s*(?:const\s+)?(\w+)\s*(?:::|$)", line)
            if fm:
                own_fields.add(fm.group(1))
    fields = set(re.findall(r"(?<=[\w\]\)])\.([A-Za-z_σθ]\w*)(?![\w(!])", code))
    fields -= {"jl"}
    return qualified, imported, own_fields, fields


def scan_reference():
    defs, struct_fields = {}, {}
    for d in REF_DIRS:
        for f in sorted((REF / d).rglob("*.jl")):
            rel = str(f.relative_to(REF))
            lines = f.read_text(errors="replace").splitlines()
            in_struct, sname = False, None
            for i, line in enumerate(lines, 1):
                for pat in (r"^\s*(?:@inline\s+|@generated\s+)?function\s+(?:Octofitter\.)?([A-Za-z_θ]\w*!?)\s*[({]", r"^\s*(?:Octofitter\.)?([A-Za-z_θ]\w*!?)\([^=]*\)\s*(?:where[^=]*)?=(?!=)",
                            r"^\s*(?:mutable\s+)?struct\s+([A-Za-z_]\w*)", r"^\s*const\s+([A-Za-z_]\w*)\s*=", r"^\s*abstract type\s+([A-Za-z_]\w*)",
                            r"^\s*macro\s+([A-Za-z_]\w*)"):
                    m = re.match(pat, line)
                    if m:
                        defs.setdefault(m.group(1), f"{rel}:{i}")
                m = re.match(r"^\s*(?:mutable\s+)?struct\s+([A-Za-z_]\w*)", line)
                if m:
                    in_struct, sname = True, m.group(1)
                    continue
                if in_struct:
                    if re.match(r"^\s*end\b", line):
                        in_struct = False
                        continue
                    if re.match(r"^\s*function\b", line):      # inner constructor: the fields are over
                        in_struct = False
                        continue
                    fm = re.match(r"^\s*(?:const\s+)?([A-Za-z_σθℓ∇][\wℓπ∇]*)\s*(?:::[^=]*)?$", line.split("#")[0].rstrip())
                    if fm and fm.group(1) not in ("end", "new"):
                        struct_fields.setdefault(fm.group(1), f"{rel}:{i} ({sname})")
    return defs, struct_fields


def build():
    qualified, imported, own_fields, fields = shim_names(JULIA.read_text().replace('include("OctofitterHIP_capi.jl")', (JULIA.parent / 'OctofitterHIP_capi.jl').read_text()))
    defs, struct_fields = scan_reference()
    names = {}
    missing = []
    for n in sorted(qualified | imported):
        if n in defs:
            names[n] = defs[n]
        else:
            missing.append(n)
    fmap = {}
    for n in sorted((fields - own_fields - THETA_FIELDS - OTHER_FIELDS - EXTERNAL) | set(REQUIRED_REFERENCE_FIELDS)):
        if n in struct_fields:
            fmap[n] = struct_fields[n]
        elif n in TABLE_COLS:
            src = (REF / TABLE_COLS[n]).read_text(errors="replace").splitlines()
            hit = next((i for i, line in enumerate(src, 1) if re.search(r"(?<![\w])" + re.escape(n) + r"(?![\w])", line)), None)
            if hit is None:
                missing.append("." + n)
            else:
                fmap[n] = f"{TABLE_COLS[n]}:{hit} (table column / hgca field)"
        elif n in defs:
            fmap[n] = defs[n]                       # a module-qualified function reached through `X.name` (e.g. LogDensityProblems.logdensity)
        else:
            missing.append("." + n)
    return dict(generator="tools/julia_api_manifest.py", names=names, fields=fmap), missing


if __name__ == "__main__":
    if not REF.exists():
        sys.exit("/root/reference not present: the manifest can only be rebuilt in the build container")
    man, missing = build()
    if missing:
        print("NOT FOUND in the reference:", missing)
    OUT.write_text(json.dumps(man, indent=1, ensure_ascii=False) + "\n")
    print(f"wrote {OUT}: {len(man['names'])} names, {len(man['fields'])} fields")
    sys.exit(1 if missing else 0)
