"""Rebuild the library, then: scalar-load hazards of every k_main variant, registers / spills / code size of the kernels whose demangled name contains any
of the given filters. Development aid (round 6):   python tools/r6_check.py "k_main<1, true, false, 1, true" ..."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import __graft_entry__ as g
import kernel_resources as kr
g.build_hip()
bad = kr.scalar_load_hazards()
names = sorted(set(b[0] for b in bad))
d = kr.demangle(names)
for n in names:
    print("HAZARD", d[n], sum(1 for b in bad if b[0] == n))
print("hazards:", len(bad))
rows = kr.resources()
print("code MB:", sum(r["code_bytes"] for r in rows) / 1e6, "lib MB:", g.LIB.stat().st_size / 1e6, "kernels:", len(rows))
for f in sys.argv[1:]:
    for r in rows:
        if f in r["name"]:
            print(r["name"], "vgpr", r["vgpr_count"], "sgpr", r["sgpr_count"], "spill", r["vgpr_spill_count"], "scratch", r["private_segment_fixed_size"], "bytes", r["code_bytes"])
