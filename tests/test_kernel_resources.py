"""
Register spills, scratch memory and code size of the compiled kernels, read from the code objects themselves (tools/kernel_resources.py:
the gfx950 ELF's metadata notes + its disassembly). VERDICT r3 items 1, 3, 7: the multi-planet latency kernels carried a 40·P-byte stack
array for a table most calls do not have, and the 3-/4-planet finish and model kernels spilled hundreds of VGPRs. This holds the build
to what round 4 left:
  * no kernel EXECUTES a scratch instruction except the listed 4-planet corner variants, and those stay within 16 spilled VGPRs;
  * k_small / k_hgca / k_main: no VGPR spill at all; a private segment appears only as the compiler's dead SGPR-spill slot
    (<= 36 bytes, never accessed: every SGPR spill goes to VGPR lanes) — checked by counting scratch_* instructions;
  * the library stays under the code-size budget.
CPU suite: hipcc cross-compiles, no GPU needed.
"""
import re
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


@pytest.fixture(scope="module")
def rows():
    from __graft_entry__ import build_hip
    import kernel_resources as kr
    build_hip()      # no-op when csrc/build/ is up to date
    r = kr.resources()
    assert len(r) > 100, "no code objects found under csrc/build/"
    return r


def _planets(name):
    m = re.match(r"k_\w+<(\d+)", name)
    return int(m.group(1)) if m else 0


def test_latency_kernels_have_no_scratch_and_no_vgpr_spills(rows):
    bad = []
    for r in rows:
        if not (r["name"].startswith("k_small<") or r["name"].startswith("k_hgca<")):
            continue
        if r["vgpr_spill_count"] or r["scratch_instructions"] or r["private_segment_fixed_size"] > 36:
            bad.append((r["name"], r["vgpr_spill_count"], r["scratch_instructions"], r["private_segment_fixed_size"]))
    assert not bad, bad


# The nuisance-free single-planet RA/Dec gradient kernels: rounds 2-4 held them to 72 VGPRs (seven waves per SIMD) with 12-16 bytes parked outside
# the row loop; since round 5's warm-started loop they keep the 79 registers they want (six waves, 2-3 % faster: octo_kernels.h: main_min_waves) and the
# budget below — nothing in the loop, at most a few dwords outside it — now simply holds them to no spills worth the name.
SEVEN_WAVES = re.compile(r"k_main<1, true, false, (1|33), (true|false), (4|8), (true|false)>")


# The four-planet gradient kernels are held to 256 VGPRs = two waves per SIMD (main_min_waves): left alone they take 312-339 registers, ONE wave
# per SIMD, and run at 0.21 of the FP64 peak. Held, they park 2-23 doubles per row in scratch memory and are 18 % faster (same-box A/B,
# profiles/r4_p4_waves_ab.txt). Deliberate; the budget below keeps the parking from growing unnoticed.
FOUR_PLANETS_TWO_WAVES = re.compile(r"k_main<4, true, (true|false), \d+, (true|false), 4, false>")


# The planet-per-wave kernels (octo_mainp.h, round 5) are held to three (4-6 planets) or four (7-8) waves per SIMD: the nuisance gradient variants
# then park 13-40 VGPRs around the density phase (the per-table θ_obs and its sin/cos, reloaded once per owned row), the rest nothing. Deliberate:
# at the next lower occupancy they do not spill and are slower (profiles/r5_mainp_ab.txt); the budget keeps the parking from growing unnoticed.
PLANET_PER_WAVE = re.compile(r"k_mainp<(true|false), (true|false), \d+, \d+, \d+>")


def test_throughput_kernels_do_not_spill(rows):
    bad = []
    for r in rows:
        n = r["name"]
        if n.startswith("k_main<"):
            if SEVEN_WAVES.fullmatch(n):
                # (round 5: with the warm-started row loop next to the cold one a third dword is parked, still outside both loops: 16 bytes, four instructions)
                # (round 6: the eight-wave block's kernel, held to 80 registers, keeps the warm solve's DIAMOND: as a triangle it parked a double in each cold
                # block — two scratch round trips per rejected row, +2 µs per step of the 1 250-walker shard)
                if r["vgpr_spill_count"] > 3 or r["private_segment_fixed_size"] > 16 or r["scratch_instructions"] > 4:
                    bad.append((n, r["vgpr_spill_count"], r["private_segment_fixed_size"], r["scratch_instructions"]))
            elif FOUR_PLANETS_TWO_WAVES.fullmatch(n):
                if r["vgpr_count"] > 256 or r["vgpr_spill_count"] > 140 or r["private_segment_fixed_size"] > 400:
                    bad.append((n, r["vgpr_count"], r["vgpr_spill_count"], r["private_segment_fixed_size"]))
            elif r["vgpr_spill_count"] or r["private_segment_fixed_size"]:
                bad.append((n, r["vgpr_spill_count"], r["private_segment_fixed_size"]))
        if PLANET_PER_WAVE.fullmatch(n):
            # (the others: up to four dwords parked outside the row loops — the tile-in-block index and its LDS offsets joined the 128-VGPR shape's prologue)
            limit = 48 if ", true, true, " in n[:24] or n.startswith("k_mainp<true, true") else 4
            seg = 128
            if ", 127, " in n:      # (round 6: the kind set with marginalised RV and the O'Neil prior, more than four planets only — three + four more sums, μ̂ and 1/A per lane
                                    # in the density phase, the prior's term in the solve phase of the attached planet's wave)
                limit, seg = (80, 200) if n.startswith("k_mainp<true, true") else (16, 64)
            if r["vgpr_spill_count"] > limit or r["private_segment_fixed_size"] > seg:
                bad.append((n, r["vgpr_spill_count"], r["private_segment_fixed_size"]))
        if n.startswith("k_finishp<") and (r["vgpr_spill_count"] or r["private_segment_fixed_size"]):
            bad.append((n, r["vgpr_spill_count"], r["private_segment_fixed_size"]))
        if n.startswith("k_finish<"):
            limit = 0 if _planets(n) <= 3 else 16
            if r["vgpr_spill_count"] > limit:
                bad.append((n, r["vgpr_spill_count"], r["private_segment_fixed_size"]))
    assert not bad, bad


def test_private_segments_are_dead_spill_slots_or_listed(rows):
    """Whatever still declares a private segment either never touches it (dead slot) or is one of the listed, deliberate cases."""
    for r in rows:
        if r["private_segment_fixed_size"] and r["scratch_instructions"]:
            assert r["name"].startswith("k_finish<4") or SEVEN_WAVES.fullmatch(r["name"]) or FOUR_PLANETS_TWO_WAVES.fullmatch(r["name"]) or \
                PLANET_PER_WAVE.fullmatch(r["name"]), \
                (r["name"], r["private_segment_fixed_size"], r["scratch_instructions"])


def test_code_size_budget(rows):
    total = sum(r["code_bytes"] for r in rows)
    lib = ROOT / "octofitter.jl_amd" / "lib" / "liboctofitter_hip.so"
    assert total < 12.0e6, f"device code {total / 1e6:.1f} MB"
    assert lib.stat().st_size < 15.0e6, f"liboctofitter_hip.so {lib.stat().st_size / 1e6:.1f} MB (VERDICT r3 item 7: < 15 MB)"


def test_no_scalar_load_is_read_before_its_wait():
    """k_main's row prefetch issues s_load_dwordx8 / x4 from inline assembly and waits in a later asm block — the compiler's
    own s_waitcnt insertion does not see them. The contract is checked on the ISA of EVERY k_main variant: on no path of the kernel's
    control-flow graph does an instruction name a destination register of a scalar load before an `s_waitcnt … lgkmcnt(0)`.
    (Round 3's loop failed this check on its early exit: copies of the in-flight tuple ahead of the wait, values that happened to be dead.)"""
    import kernel_resources as kr
    bad = kr.scalar_load_hazards()
    assert not bad, bad[:10]


def _ins(lines):
    """[(address, text, branch-target offset)] from 'text' or ('text', target index) items, 4 bytes per instruction."""
    out = []
    for k, it in enumerate(lines):
        txt, tgt = (it, None) if isinstance(it, str) else it
        out.append((4 * k, txt, None if tgt is None else 4 * tgt))
    return out


def test_hazard_checker_itself():
    """The checker of the test above, on hand-written instruction lists: (a) a copy of an in-flight tuple ahead of the wait is flagged (what a
    restructured warm loop compiled to in round 6: a loop-carried tuple copied at the loop head); (b) the early exit's wait on one arm of a
    diamond whose arm is recorded in an SGPR pair and tested again by the later branch is NOT (the path that skips the wait never reaches the
    code that reuses the registers); (c) the same diamond with the flag pair overwritten in between IS flagged (nothing known: both arms)."""
    import kernel_resources as kr
    a = _ins(["s_load_dwordx8 s[4:11], s[0:1], 0x0", "s_mov_b64 s[20:21], s[4:5]", "s_waitcnt lgkmcnt(0)", "s_endpgm"])
    assert [b[1] for b in kr.kernel_hazards("a", 0, a)] == [4]
    diamond = ["s_load_dwordx8 s[20:27], s[0:1], 0x0",          # 0  B in flight
               "s_andn2_b64 vcc, exec, s[58:59]",                # 1  s[58:59]: early exit?
               "s_mov_b64 s[58:59], -1",                         # 2
               ("s_cbranch_vccnz 2", 6),                         # 3  not the early exit: skip the wait
               "s_mov_b64 s[58:59], 0",                          # 4
               "s_waitcnt lgkmcnt(0)",                           # 5
               "v_add_f64 v[0:1], v[0:1], v[2:3]",               # 6  (the sunk tail of the row body)
               "s_andn2_b64 vcc, exec, s[58:59]",                # 7
               ("s_cbranch_vccnz 3", 11),                        # 8  early exit -> the code that reuses B's registers
               "s_waitcnt lgkmcnt(0)",                           # 9  normal path: wait for B, use it
               "s_endpgm",                                       # 10
               "s_cselect_b64 s[20:21], -1, 0",                  # 11
               "s_endpgm"]
    assert kr.kernel_hazards("b", 0, _ins(diamond)) == []
    clobbered = list(diamond)
    clobbered[6] = "s_mov_b64 s[58:59], s[60:61]"
    assert [b[2] for b in kr.kernel_hazards("c", 0, _ins(clobbered))] == ["s_cselect_b64 s[20:21], -1, 0"]
