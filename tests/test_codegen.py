"""Codegen guard (no GPU: hipcc cross-compiles): what the hot kernels of the default arithmetic compile to, against a committed ratchet.

Round 4 found its regressions by reading ISA by hand (LAB_NOTEBOOK 'Round 4, second half': a scratch reload or a compiler-inserted
`s_waitcnt vmcnt(0)` inside a prefetching K loop costs 10-30 % of a kernel and changes no result, so no parity test sees it).  This test
compiles every source of `tools/isa_scan.HOT_SOURCES` with `-S --cuda-device-only` and checks, for every kernel above 1 % of the image step
(`isa_scan.HOT_KERNELS`):

  * HARD: no scratch instruction inside a K loop (a leaf loop holding >= 4 MFMAs) -- a spill there is a VGPR-destination load, i.e. vmcnt(0);
  * RATCHET (tests/golden/codegen_budget.json, refreshed by `python tools/isa_scan.py budget --write`): `.vgpr_spill_count`, the number of
    `s_waitcnt vmcnt(0)` per K loop and the LDS size may not rise above the recorded value; kernels recorded at 0 spills stay at 0.

A number that goes DOWN is reported (refresh the budget in the same commit); a kernel that disappears from the build fails.
"""
import json
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

BUDGET = os.path.join(ROOT, "tests", "golden", "codegen_budget.json")


@pytest.fixture(scope="module")
def report():
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc in this environment")
    import isa_scan
    return isa_scan.codegen_report(jobs=min(8, os.cpu_count() or 2))


def test_every_hot_kernel_is_still_built(report):
    import isa_scan
    missing = [k for k in isa_scan.HOT_KERNELS if k not in report]
    assert not missing, f"hot kernels no longer instantiated (update isa_scan.HOT_KERNELS with the profile that justifies it): {missing}"
    assert set(json.load(open(BUDGET))) == set(isa_scan.HOT_KERNELS)


def test_no_scratch_access_inside_a_k_loop(report):
    bad = {k: [l for l in r["k_loops"] if l["scratch"]] for k, r in report.items()}
    bad = {k: v for k, v in bad.items() if v}
    assert not bad, f"scratch (spill) instructions inside MFMA K loops: {bad}"
    assert all(r["k_loops"] for k, r in report.items() if k != "conv3x3_patch_kernel<4, 1, 1, 1, 2>"), "a hot kernel without an MFMA loop?"


def test_spills_and_k_loop_waits_do_not_exceed_the_recorded_budget(report):
    budget = json.load(open(BUDGET))
    worse, better = [], []
    for k, b in budget.items():
        r = report[k]
        for key in ("vgpr_spill", "lds"):
            if r[key] > b[key]:
                worse.append(f"{k}: {key} {b[key]} -> {r[key]}")
            elif r[key] < b[key]:
                better.append(f"{k}: {key} {b[key]} -> {r[key]}")
        if len(r["k_loops"]) != len(b["k_loops"]):
            worse.append(f"{k}: {len(b['k_loops'])} K loops -> {len(r['k_loops'])} (re-record with a reason)")
            continue
        for i, (lr, lb) in enumerate(zip(r["k_loops"], b["k_loops"])):
            if lr["vm0"] > lb["vm0"]:
                worse.append(f"{k}: K loop {i}: vmcnt(0) {lb['vm0']} -> {lr['vm0']}")
            elif lr["vm0"] < lb["vm0"]:
                better.append(f"{k}: K loop {i}: vmcnt(0) {lb['vm0']} -> {lr['vm0']}")
    if better:
        print("improved (refresh tests/golden/codegen_budget.json):", *better, sep="\n  ")
    assert not worse, "codegen regressions against tests/golden/codegen_budget.json:\n  " + "\n  ".join(worse)


def test_the_budget_itself_only_tolerates_known_spills():
    """the ratchet's current debt, spelled out: every other hot kernel is recorded at zero spilled VGPRs"""
    budget = json.load(open(BUDGET))
    spilling = {k: b["vgpr_spill"] for k, b in budget.items() if b["vgpr_spill"]}
    assert set(spilling) <= KNOWN_SPILLS.keys(), spilling
    for k, v in spilling.items():
        assert v <= KNOWN_SPILLS[k], (k, v)


# Kernels allowed to spill at all, with a ceiling each (none inside a K loop).  Round 5 had six entries of 32 - 98 registers, all TN = 3 tiles
# (96 accumulators per lane).  Round 6 removed four: gemm_pl<3> 67 -> 0 and gemm1x1_pc<3, *, 2, 2, *> 92 - 97 -> 0 by (a) one store path for K-slice
# partial sums and final outputs instead of two inlined copies, (b) building those register tiles without the tanh epilogue (refused by their
# launchers: test_wide_register_tiles_refuse_tanh), (c) passing the thread id of the GRN prologue through an opaque copy so that its per-thread
# addresses are not hoisted out of the persistent tile loop.  What is left: the dominant 3x3 kernel's 32 (20 saved once before the K loop and
# restored after the last store, 12 around its unrolled first taps) and the 128 x 192 patch kernel's 90 (halo staging before the first chunk).
KNOWN_SPILLS = {"conv3x3_pl_kernel<3>": 32, "conv3x3_patch_pc_kernel<3, 8, 2>": 90}


def test_untracked_lds_dma_is_m0_neutral():
    """csrc/lds_dma.h: the hand-written `global_load_lds_dwordx4` takes its LDS base from M0, which belongs to the compiler
    (an "m0" clobber is ignored: 'clobber list contains reserved registers').  Every inline-asm block of the built kernels that touches M0 must
    save it into a scalar register first and restore it last -- then hipcc's own idea of M0 (the builtin form's base, a v_readlane index) stays
    true across the statement and both DMA forms may be mixed in one kernel.  Checked on the emitted ISA of every source that uses the asm form."""
    import re
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc in this environment")
    import isa_scan
    csrc = os.path.join(ROOT, "videoseal_amd", "csrc")
    users = [f for f in sorted(os.listdir(csrc)) if f.endswith(".hip") and "vs_lds_dma16_untracked" in open(os.path.join(csrc, f)).read() or
             f.endswith(".hip") and re.search(r"asm[^;]*\bm0\b", open(os.path.join(csrc, f)).read())]
    assert {"convnext_fused.hip", "gemm1x1_pc.hip"} <= set(users)
    # no source writes M0 in its own asm: the helper of lds_dma.h is the one place
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")) and f != "lds_dma.h":
            assert not re.search(r"asm[^;]*\bm0\b", open(os.path.join(csrc, f)).read()), f"{f}: inline asm that names m0 outside lds_dma.h"
    for f in users:
        text = isa_scan.compile_s(os.path.join(csrc, f))
        blocks = re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", text, re.S)
        dma = [[l.strip() for l in b.strip().splitlines()] for b in blocks if "m0" in b]
        assert dma, f"{f}: no hand-written LDS-DMA in the ISA?"
        for b in dma:
            m = re.fullmatch(r"s_mov_b32 (s\d+), m0", b[0])
            assert m, (f, b)
            assert b[-1] == f"s_mov_b32 m0, {m.group(1)}", (f, b)
            body = b[1:-1]
            assert re.fullmatch(r"s_mov_b32 m0, \S+", body[0]) and body[-1].startswith("global_load_lds_dwordx4"), (f, b)
            assert not any(l.startswith("s_mov_b32 m0") for l in body[1:]), (f, b)
