"""Build a variant of the HIP library with extra compiler flags for a same-box A/B (boxes differ by a few % between gpurun calls, so both
builds run in ONE call, selected with OCTOFITTER_HIP_LIB):   python tools/build_variant.py <tag> [-DOCTO_...=...] ...
-> octofitter.jl_amd/lib/variants/liboctofitter_hip_<tag>.so (git-ignored, travels with gpurun). Development aid."""
import os, shutil, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import CSRC, HIPCC_FLAGS, PKG_DIR

tag, extra = sys.argv[1], sys.argv[2:]
hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
objdir = CSRC / "build" / f"variant_{tag}"
objdir.mkdir(parents=True, exist_ok=True)
srcs = sorted(CSRC.glob("*.hip"))
only = os.environ.get("OCTO_VARIANT_ONLY")      # e.g. "octo_inst_p1,octo_api": recompile these with the flags, reuse the default objects for the rest


def cc(src):
    obj = objdir / (src.stem + ".o")
    if only and src.stem not in only.split(","):
        return CSRC / "build" / (src.stem + ".o")
    subprocess.run([hipcc, *HIPCC_FLAGS, *extra, f"-I{ROOT / 'include'}", f"-I{CSRC}", "-c", "-o", str(obj), str(src)], check=True)
    return obj


with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
    objs = list(ex.map(cc, srcs))
out = PKG_DIR / "lib" / "variants" / f"liboctofitter_hip_{tag}.so"
out.parent.mkdir(parents=True, exist_ok=True)
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", str(out), *map(str, objs)], check=True)
print(out)
