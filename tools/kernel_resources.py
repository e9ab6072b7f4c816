#!/usr/bin/env python
"""Scratch memory, register spills and code size of every compiled kernel, read from the code objects' own metadata
(the .note section of the gfx950 ELF inside each csrc/build/*.o — what the loader sees, not a compiler remark):

    python tools/kernel_resources.py                 # table of every kernel that uses scratch or spills + totals
    python tools/kernel_resources.py --all [filter]  # every kernel (optionally: demangled name contains `filter`)
    python tools/kernel_resources.py --json out.json

`resources()` is imported by tests/test_kernel_resources.py, which holds the build to the budget below: the latency kernels
(k_small, k_hgca) carry NO scratch allocation unless they spill, and the spill counts stay under the per-family limits.
"""
from __future__ import annotations

import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "octofitter.jl_amd" / "csrc" / "build"
LLVM = Path("/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "kernarg_segment_size", "max_flat_workgroup_size")


def code_object(obj: Path, tmp: Path) -> Path | None:
    """The gfx950 ELF bundled in a host object's .hip_fatbin section."""
    fat = tmp / (obj.stem + ".fatbin")
    co = tmp / (obj.stem + ".co")
    r = subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(obj)], capture_output=True, text=True)
    if r.returncode != 0 or not fat.exists() or fat.stat().st_size == 0:
        return None
    r = subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}"],
                       capture_output=True, text=True)
    return co if r.returncode == 0 and co.exists() else None


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def kernel_text_sizes(co: Path):
    """bytes of machine code per kernel symbol (FUNC symbols of the code object)."""
    out = subprocess.run([str(LLVM / "llvm-readelf"), "-s", "--wide", str(co)], capture_output=True, text=True).stdout
    sizes = {}
    for line in out.splitlines():
        p = line.split()
        if len(p) >= 8 and p[3] == "FUNC":
            sizes[p[7]] = int(p[2])
    return sizes


def scratch_instruction_counts(co: Path):
    """scratch_load / scratch_store instructions per kernel symbol, from the disassembly: a kernel may carry a private segment without
    touching it (a dead SGPR spill slot the compiler forgot to drop: every such spill went to VGPR lanes)."""
    out = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True).stdout
    counts, cur = {}, None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1); counts.setdefault(cur, 0)
        elif cur is not None and "scratch_" in line:
            counts[cur] += 1
    return counts


def resources(build_dir: Path = BUILD, disassemble: bool = True):
    """[{name (demangled), object, vgpr_count, ..., code_bytes, scratch_instructions}] for every kernel of every csrc/build/*.o"""
    rows = []
    with tempfile.TemporaryDirectory() as td:
        tmp = Path(td)
        for obj in sorted(build_dir.glob("*.o")):
            co = code_object(obj, tmp)
            if co is None:
                continue
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
            sizes = kernel_text_sizes(co)
            scr = scratch_instruction_counts(co) if disassemble else {}
            for blk in re.split(r"\n  - \.agpr_count:", "\n" + notes)[1:]:
                blk = ".agpr_count:" + blk
                m = re.search(r"^\s*\.name:\s+(\S+)", blk, re.M)
                if not m:
                    continue
                row = {"symbol": m.group(1), "object": obj.name}
                for f in FIELDS:
                    mm = re.search(rf"^\s*\.{f}:\s+(\d+)", blk, re.M)
                    row[f] = int(mm.group(1)) if mm else 0
                row["code_bytes"] = sizes.get(row["symbol"], 0)
                row["scratch_instructions"] = scr.get(row["symbol"], 0) if disassemble else None
                rows.append(row)
    dm = demangle([r["symbol"] for r in rows])
    for r in rows:
        d = dm.get(r["symbol"], r["symbol"])
        r["name"] = re.sub(r"^void octo::|\(octo::.*$|\(anonymous namespace\)::", "", d).strip()
    return rows


def family(name: str) -> str:
    return re.match(r"[\w:]+", name).group(0)


def main():
    args = sys.argv[1:]
    rows = resources()
    if not rows:
        raise SystemExit("no code objects under csrc/build/: run `python __graft_entry__.py` first")
    if "--json" in args:
        Path(args[args.index("--json") + 1]).write_text(json.dumps(rows, indent=1))
        return
    show_all = "--all" in args
    filt = [a for a in args if not a.startswith("--")]
    filt = filt[0] if filt else ""
    print(f"{'kernel':78s} {'vgpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch B':>9s} {'scr.ins':>7s} {'LDS B':>7s} {'code B':>8s}")
    for r in sorted(rows, key=lambda r: (family(r["name"]), r["name"])):
        interesting = r["private_segment_fixed_size"] or r["vgpr_spill_count"]      # (SGPR spills go to VGPR lanes: --all shows them)
        if (show_all or interesting) and filt in r["name"]:
            print(f"{r['name'][:78]:78s} {r['vgpr_count']:5d} {r['sgpr_count']:5d} {r['vgpr_spill_count']:6d} {r['sgpr_spill_count']:6d} "
                  f"{r['private_segment_fixed_size']:9d} {r['scratch_instructions']:7d} {r['group_segment_fixed_size']:7d} {r['code_bytes']:8d}")
    fams = {}
    for r in rows:
        f = fams.setdefault(family(r["name"]), {"n": 0, "code": 0, "scratch": 0, "used": 0, "spill": 0})
        f["n"] += 1; f["code"] += r["code_bytes"]
        f["scratch"] += 1 if r["private_segment_fixed_size"] else 0
        f["used"] += 1 if r["scratch_instructions"] else 0
        f["spill"] += 1 if r["vgpr_spill_count"] else 0
    print(f"\n{'family':18s} {'kernels':>7s} {'code MB':>8s} {'private segment':>15s} {'... accessed':>12s} {'VGPR spills':>11s}")
    for k, f in sorted(fams.items()):
        print(f"{k:18s} {f['n']:7d} {f['code'] / 1e6:8.2f} {f['scratch']:15d} {f['used']:12d} {f['spill']:11d}")
    print(f"{'total':18s} {len(rows):7d} {sum(r['code_bytes'] for r in rows) / 1e6:8.2f}")




# ---- scalar-load hazard check: k_main's row prefetch issues s_load_dwordx8/x4 from inline assembly and waits for them in a LATER
# asm block, which the compiler's own s_waitcnt insertion does not track. The contract is checked on the final ISA instead: between a
# scalar load and the next `s_waitcnt … lgkmcnt(0)` no instruction may name a register of the load's destination.
def _sregs(tok):
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def kernel_hazards(sym, base, ins):
    """The analysis of ONE kernel: ins = [(address, instruction text, branch-target offset from `base` or None)] in address order."""
    bad = []
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    succ = []
    for i, (a, txt, tgt) in enumerate(ins):
        op = txt.split()[0]
        nxt = [i + 1] if i + 1 < len(ins) else []
        if op == "s_endpgm":
            succ.append([])
        elif op == "s_branch":
            succ.append([addr_index[base + tgt]] if tgt is not None and base + tgt in addr_index else [])
        elif op.startswith("s_cbranch"):
            succ.append(nxt + ([addr_index[base + tgt]] if tgt is not None and base + tgt in addr_index else []))
        else:
            succ.append(nxt)
    # Forward data-flow with a little path sensitivity (round 6): the compiler sometimes puts the early exit's wait on one arm of a
    # diamond and records which arm ran in an SGPR pair (s_mov_b64 s[a:b], -1 / 0) that a later `s_andn2_b64 vcc, exec, s[a:b]` +
    # s_cbranch_vcc(n)z tests again. A path-insensitive merge then sees "load pending" on the path that in fact waited. So a state is
    # (pending registers, the SGPR pairs known to hold -1 or 0, what is known of vcc), an instruction keeps a SET of states, and a
    # branch on a vcc derived from a known pair follows only the arm that can be taken (exec != 0 inside these uniform loops).
    state = [set() for _ in ins]
    state[0].add((frozenset(), frozenset(), None))
    work = [0]
    flagged = set()
    n_states = 0
    while work:
        i = work.pop()
        a, txt, _ = ins[i]
        parts = re.split(r"[ ,]+", txt)
        op, args = parts[0], parts[1:]
        outs = {}      # successor index -> set of states
        for (pend0, consts0, vcc0) in list(state[i]):
            pend, consts, vcc = set(pend0), dict(consts0), vcc0
            take = None      # None: both arms; True: branch taken only; False: fall through only
            if op.startswith("s_waitcnt"):
                if "lgkmcnt(0)" in txt:
                    pend = set()
            else:
                used = set().union(*[_sregs(t) for t in args]) if args else set()
                if pend & used and i not in flagged:
                    flagged.add(i); bad.append((sym, a, txt))
                if op.startswith(("s_load_", "s_buffer_load_")) and args:
                    pend |= _sregs(args[0])
                # what the instruction writes: its first operand (SALU / VOP3 with an SGPR destination), vcc for e32 compares
                dst = _sregs(args[0]) if args and not op.startswith(("s_cmp", "s_cbranch", "s_branch", "s_bitcmp", "s_setpc")) else set()
                for lo in [k for k in consts if k in dst or k + 1 in dst]:
                    del consts[lo]
                writes_vcc = (args and args[0] == "vcc") or op.startswith(("v_cmp", "v_div_scale")) and not (args and args[0].startswith("s"))
                if writes_vcc:
                    vcc = None
                m2 = re.fullmatch(r"s\[(\d+):(\d+)\]", args[0]) if args else None
                if op == "s_mov_b64" and m2 and len(args) == 2 and args[1] in ("-1", "0"):
                    consts[int(m2.group(1))] = int(args[1])
                if op in ("s_andn2_b64", "s_and_b64") and len(args) == 3 and args[0] == "vcc" and args[1] == "exec":
                    m3 = re.fullmatch(r"s\[(\d+):(\d+)\]", args[2])
                    if m3 and int(m3.group(1)) in consts:
                        v = consts[int(m3.group(1))]
                        ones = (v == -1) if op == "s_and_b64" else (v == 0)
                        vcc = "nz" if ones else "z"
                if op == "s_cbranch_vccnz" and vcc is not None:
                    take = vcc == "nz"
                if op == "s_cbranch_vccz" and vcc is not None:
                    take = vcc == "z"
            st = (frozenset(pend), frozenset(consts.items()), vcc)
            ss = succ[i]
            if take is not None and op.startswith("s_cbranch") and len(ss) == 2:
                ss = [ss[1]] if take else [ss[0]]
            for j in ss:
                outs.setdefault(j, set()).add(st)
        for j, sts in outs.items():
            new = sts - state[j]
            if new:
                state[j] |= new; n_states += len(new); work.append(j)
                if n_states > 2_000_000:
                    raise RuntimeError(f"scalar_load_hazards: state explosion in {sym}")
    return bad


def scalar_load_hazards(build_dir: Path = BUILD, name_filter: str = "k_main"):
    """[(kernel symbol, address, instruction text)] for every instruction that names a register of a scalar load's destination while that load
    may still be in flight — a forward data-flow over the kernel's control-flow graph (blocks cut at branches and branch targets; a load is
    pending from its issue to the next `s_waitcnt … lgkmcnt(0)` on every path)."""
    bad = []
    with tempfile.TemporaryDirectory() as td:
        for obj in sorted(build_dir.glob("*.o")):
            co = code_object(obj, Path(td))
            if co is None:
                continue
            out = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True).stdout
            kernels, cur = {}, None
            for line in out.splitlines():
                m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
                if m:
                    cur = m.group(2) if name_filter in m.group(2) else None
                    if cur:
                        kernels[cur] = (int(m.group(1), 16), [])
                    continue
                if cur is None or "//" not in line:
                    continue
                txt, com = line.split("//", 1)
                txt = txt.strip()
                ma = re.match(r"\s*([0-9A-Fa-f]+):", com)
                if not txt or not ma:
                    continue
                tgt = re.search(r"<\S+?\+0x([0-9a-fA-F]+)>", com)
                kernels[cur][1].append((int(ma.group(1), 16), txt, int(tgt.group(1), 16) if tgt else None))
            for sym, (base, ins) in kernels.items():
                bad += kernel_hazards(sym, base, ins)
    return bad

if __name__ == "__main__":
    main()
