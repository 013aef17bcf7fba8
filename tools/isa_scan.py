#!/usr/bin/env python
"""ISA-level checks of the hot kernels (no GPU needed: hipcc cross-compiles).  Three views of `hipcc -S --cuda-device-only` output:

  tools/isa_scan.py loops  <file.hip> [name-substring]     per loop that contains MFMAs: how many `s_waitcnt vmcnt(0)`, counted vm waits, scratch
                                                          operations, LDS-DMAs, global loads and barriers it holds.  A `vmcnt(0)` inside a K loop that
                                                          prefetches is a wait for the data just requested; on gfx9 it also waits for every store.
  tools/isa_scan.py order  <file.hip> <name-substring>     the order of loads / stores / LDS-DMAs / waits / barriers / scratch in the kernel text
                                                          (run-length compressed): `LD vmcnt(0) ST LD vmcnt(0) ST ...` in an epilogue is a chain of
                                                          dependent round trips.
  tools/isa_scan.py mix    <file.hip> <name-substring>     instruction mix of the loop that holds the MFMAs, in order (M mfma, v VALU, r ds_read,
                                                          w ds_write, D LDS-DMA, G global, a accvgpr move, n s_nop, W(..) waits, B barrier, | branch):
                                                          shows whether MFMAs and VALU work are interleaved or run as separate phases.

What the round-4 pass over the kernels found with these: LAB_NOTEBOOK.md "Round 4, second half"."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_s(src):
    if src.endswith(".s"):
        return open(src).read()
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/videoseal_amd/csrc", "-S",
           "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(text, sub):
    for m in re.finditer(r"^(_Z\w+):", text, re.M):
        if sub and sub not in m.group(1):
            continue
        end = text.find("s_endpgm", m.end())
        if end > 0:
            yield m.group(1), text[m.end():end].splitlines()


def rle(ev, limit):
    out = []
    for e in ev:
        if out and out[-1][0] == e:
            out[-1][1] += 1
        else:
            out.append([e, 1])
    return " ".join(f"{e}x{c}" if c > 1 else e for e, c in out)[:limit]


def loops(text, sub):
    for name, lines in kernels(text, sub):
        cur, stats = None, collections.OrderedDict()
        for l in lines:
            if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
                mm = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", l)
                if "Loop Header" in l:
                    cur = (re.match(r"^\.L(BB\d+_\d+)", l).group(1), re.search(r"Depth=(\d+)", l).group(1))
                elif mm:
                    cur = (mm.group(1), mm.group(2))
                else:
                    cur = None
                continue
            if cur is None:
                continue
            s = l.strip()
            c = stats.setdefault(cur, collections.Counter())
            for key, pre in (("mfma", "v_mfma"), ("scratch", "scratch_"), ("dma", "global_load_lds"), ("bar", "s_barrier"), ("gst", "global_store")):
                if s.startswith(pre):
                    c[key] += 1
            if s.startswith("global_load") and not s.startswith("global_load_lds"):
                c["gld"] += 1
            if s.startswith("s_waitcnt") and "vmcnt" in s:
                c["vm0" if "vmcnt(0)" in s else "vmN"] += 1
        for (hdr, depth), c in stats.items():
            if c["mfma"] >= 4:
                print(f"{name[:80]:80s} loop {hdr} depth {depth}: {dict(c)}")


def order(text, sub, limit=4000):
    for name, lines in kernels(text, sub):
        ev = []
        for l in lines:
            s = l.strip()
            if "Loop Header" in l: ev.append("LOOP")
            elif s.startswith("s_barrier"): ev.append("BAR")
            elif s.startswith("s_waitcnt") and "vmcnt" in s: ev.append(s.replace("s_waitcnt ", "").replace(" ", ""))
            elif s.startswith("global_load_lds"): ev.append("DMA")
            elif s.startswith("global_load"): ev.append("LD")
            elif s.startswith("global_store"): ev.append("ST")
            elif s.startswith("scratch_"): ev.append("SCR")
            elif s.startswith("v_mfma"): ev.append("M")
        print(name[:100], len(lines), "lines")
        print(rle(ev, limit))


def mix(text, sub, limit=4000):
    for name, lines in kernels(text, sub):
        hdrs = [n for n, l in enumerate(lines) if "Loop Header" in l]
        for h in hdrs:
            seg = lines[h:h + 2500]
            if sum("v_mfma" in l for l in seg[:900]) < 10:
                continue
            ev = []
            for l in seg:
                s = l.strip()
                if not s or s.startswith(";") or s.startswith("."):
                    continue
                op = s.split()[0]
                if op.startswith("v_mfma"): ev.append("M")
                elif op.startswith("ds_read"): ev.append("r")
                elif op.startswith("ds_write"): ev.append("w")
                elif op.startswith("v_accvgpr"): ev.append("a")
                elif op.startswith("v_"): ev.append("v")
                elif op.startswith("s_waitcnt"): ev.append("W(" + s.split(None, 1)[1].replace(" ", "") + ")")
                elif op.startswith("s_barrier"): ev.append("B")
                elif op.startswith("global_load_lds"): ev.append("D")
                elif op.startswith("global_"): ev.append("G")
                elif op.startswith("s_cbranch"): ev.append("|")
                elif op.startswith("s_nop"): ev.append("n")
            print(name[:100], "loop at line", h)
            print(rle(ev, limit))
            break


if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in ("loops", "order", "mix"):
        sys.exit(__doc__)
    txt = compile_s(sys.argv[2])
    sub = sys.argv[3] if len(sys.argv) > 3 else ""
    {"loops": loops, "order": order, "mix": mix}[sys.argv[1]](txt, sub)
