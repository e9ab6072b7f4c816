#!/usr/bin/env python
"""Instruction mix of k_main's row loop, straight from the ISA hipcc emits for ONE instantiation ("account for the
instructions"): compiles a one-kernel translation unit with -S, cuts the innermost loop that holds the prefetching s_load_dwordx8 pair
(two rows per trip) and prints the per-row histogram by class.
    python tools/loop_histogram.py "1, true, false, 1"        # config 3      (P, GRAD, NUIS, KM)
    python tools/loop_histogram.py "2, true, true, 5"         # config 4"""
import collections, re, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import CSRC, HIPCC_FLAGS
import os
CSRC = Path(os.environ.get("OCTO_CSRC", CSRC))      # (experiments: another copy of the kernel headers)

targs = sys.argv[1] if len(sys.argv) > 1 else "1, true, false, 1"
src = f'''#include "octo_host.h"
namespace octo {{ void f(EvalArgs a) {{ hipLaunchKernelGGL((k_main<{targs}, true>), dim3(1), dim3(256), (fused_lds_bytes<{targs}>()), 0, a); }} }}
'''
with tempfile.TemporaryDirectory() as td:
    (Path(td) / "t.hip").write_text(src)
    subprocess.run(["hipcc", *[f for f in HIPCC_FLAGS if f != "-fPIC"], f"-I{ROOT / 'include'}", f"-I{CSRC}", "-S", "--offload-device-only",
                    "-o", f"{td}/t.s", f"{td}/t.hip"], check=True, capture_output=True)
    txt = (Path(td) / "t.s").read_text()
m = re.search(r"^(_ZN4octoL6k_mainI\w+):.*?\.end_amdhsa_kernel", txt, re.S | re.M)
lines = m.group(0).split("\n")
# innermost loops = label .. backward s_branch / s_cbranch to it; keep those with two s_load_dwordx8 (the prefetch pair)
labels = {mm.group(1): i for i, l in enumerate(lines) if (mm := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i, l in enumerate(lines):
    mm = re.match(r"\s+s_(?:c)?branch\w* (\.LBB\d+_\d+)", l)
    if mm and labels.get(mm.group(1), 10**9) < i:
        body = [x.split()[0] for x in lines[labels[mm.group(1)]:i + 1] if x.startswith("\t") and not x.strip().startswith((";", "."))]
        if body.count("s_load_dwordx8") == 2:
            loops.append(body)
if not loops:
    raise SystemExit("row loop not found")
for body in loops:
    rows = 2.0 * max(1, len(loops) // 1) / len(loops) * 1      # two rows per trip
    cls = collections.Counter()
    for ins in body:
        if ins.startswith("v_"):
            if "f64" in ins and ("fma" in ins or "fmac" in ins): cls["v_fma/fmac_f64"] += 1
            elif "mul_f64" in ins: cls["v_mul_f64"] += 1
            elif "add_f64" in ins: cls["v_add_f64"] += 1
            elif ins.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): cls["v_rcp/rsq_f64"] += 1
            elif "f64" in ins: cls["other f64-rate (cvt, rndne, frexp, cmp)"] += 1
            elif ins.startswith(("v_sqrt_f32", "v_rcp_f32", "v_log_f32", "v_exp_f32", "v_rsq_f32")): cls["f32 transcendental"] += 1
            elif "f32" in ins: cls["f32 fma/mul/add (Markley starter, table index)"] += 1
            else: cls["integer / move / select"] += 1
        elif ins.startswith("ds_"): cls["LDS"] += 1
        elif ins.startswith("s_load"): cls["scalar loads"] += 1
        elif ins.startswith("s_"): cls["SALU / branch / waitcnt"] += 1
        else: cls["vector memory"] += 1
    valu = sum(v for k, v in cls.items() if k.startswith(("v_", "other f64", "f32", "integer")))
    print(f"k_main<{targs}, true>: row loop of {len(body)} instructions per trip (2 rows); per ROW: {valu / 2:.1f} VALU")
    for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
        print(f"   {k:52s} {v / 2:6.1f} per row")
