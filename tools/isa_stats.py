#!/usr/bin/env python
"""Per-kernel static ISA statistics from a hipcc -save-temps .s file: VGPR/SGPR counts, LDS, scratch, instruction counts by
class, and (for the epoch-loop kernels) the instruction count of the innermost loop body. Used to check that a refactor of the
hot kernel leaves its code unchanged and to see what a new variant costs.   python tools/isa_stats.py file.s [filter]"""
import re, sys, subprocess, collections

def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}

def main():
    path = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
    txt = open(path).read()
    kernels = re.findall(r"^(_Z\w+):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel", txt, re.S | re.M)
    dm = demangle([k for k, _ in kernels])
    for name, body in kernels:
        d = dm[name]
        if filt and filt not in d: continue
        code = body.split(".section")[0]
        ins = [l.strip().split()[0] for l in code.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
        c = collections.Counter()
        for i in ins:
            if i.startswith("v_") and "f64" in i: c["valu_f64"] += 1
            elif i.startswith("v_"): c["valu_other"] += 1
            elif i.startswith("s_load") or i.startswith("s_buffer"): c["smem"] += 1
            elif i.startswith("s_"): c["salu"] += 1
            elif i.startswith("ds_"): c["lds"] += 1
            elif i.startswith(("global_", "flat_", "buffer_", "scratch_")): c["vmem"] += 1
        m = lambda k: (re.search(rf"\.amdhsa_{k} (\d+)", body) or re.search(rf"; {k}: (\d+)", body))
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body); sg = re.search(r"\.amdhsa_next_free_sgpr (\d+)", body)
        lds = re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body); scr = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        # innermost loops: backward branches
        labels = {}
        lines = code.split("\n")
        pos = 0
        for idx, l in enumerate(lines):
            mm = re.match(r"^(\.LBB\d+_\d+):", l)
            if mm: labels[mm.group(1)] = idx
        loops = []
        for idx, l in enumerate(lines):
            mm = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < idx:
                seg = [x.strip().split()[0] for x in lines[labels[mm.group(1)]:idx] if x.startswith("\t") and x.strip() and not x.strip().startswith((".", ";"))]
                loops.append((len(seg), sum(1 for s in seg if s.startswith("v_")), sum(1 for s in seg if s.startswith("v_") and "f64" in s)))
        loops.sort(reverse=True)
        print(f"{d[:110]}\n   vgpr {vg.group(1) if vg else '?'} sgpr {sg.group(1) if sg else '?'} lds {lds.group(1) if lds else '?'} scratch {scr.group(1) if scr else '?'} | total {len(ins)} {dict(c)} | loops(total,valu,f64) {loops[:3]}")

main()
