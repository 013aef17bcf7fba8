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

  tools/isa_scan.py budget [--write]                       registers / spills / K-loop wait counts of the hot kernels (HOT_KERNELS), the table behind
                                                          tests/test_codegen.py; --write refreshes tests/golden/codegen_budget.json (a ratchet: do
                                                          that only when a number went DOWN, or with a reason in the commit message).

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


def loop_stats(text, sub):
    """{kernel: {(header, depth): Counter}} with every basic block counted in its INNERMOST loop.  LLVM annotates a block either on its label
    line (`; in Loop: Header=BBx Depth=d`, `; =>This Loop Header: Depth=d`) or, for an inner header, on the label line plus continuation comment
    lines (`; Parent Loop BBx Depth=1` / `; => This Inner Loop Header: Depth=2`)."""
    out = collections.OrderedDict()
    for name, lines in kernels(text, sub):
        cur, stats, n, parents = None, collections.OrderedDict(), 0, set()
        while n < len(lines):
            l = lines[n]
            lab = re.match(r"^\.L(BB\d+_\d+):", l)
            if lab or l.startswith("; %bb."):
                note = l
                while n + 1 < len(lines) and re.match(r"^\s*;", lines[n + 1]) and not lines[n + 1].startswith("; %bb."):
                    n += 1
                    note += lines[n]
                if lab and "Child Loop" in note:
                    parents.add(lab.group(1))
                hd = re.search(r"Loop Header: Depth=(\d+)", note)
                inl = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", note)
                if hd and lab:
                    cur = (lab.group(1), hd.group(1))
                elif inl:
                    cur = (inl.group(1), inl.group(2))
                else:
                    cur = None
                n += 1
                continue
            n += 1
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
            c["leaf"] = 0 if hdr in parents else 1
        out[name] = stats
    return out


def resources(text):
    """{kernel: {vgpr, agpr, vgpr_spill, sgpr_spill, lds, scratch}} from the code-object metadata at the end of the assembly"""
    out = {}
    for b in text.split("- .agpr_count:")[1:]:
        nm = re.search(r"\.name:\s+(_Z\S+)", b)
        if not nm:
            continue
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))
        out[nm.group(1)] = {"vgpr": g("vgpr_count"), "agpr": int(b.split()[0]), "vgpr_spill": g("vgpr_spill_count"), "sgpr_spill": g("sgpr_spill_count"),
                            "lds": g("group_segment_fixed_size"), "scratch": g("private_segment_fixed_size")}
    return out


def loops(text, sub):
    for name, stats in loop_stats(text, sub).items():
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


# ---- the codegen budget (tests/test_codegen.py): what the hot kernels of the default arithmetic compile to, as a ratchet ----
HOT_SOURCES = ["conv3x3_pl", "gemm_pl", "gemm1x1_pc", "conv3x3_patch_pc", "conv3x3_patch", "conv3x3_small", "conv_gemm", "convnext_fused",
               "resblock_thin", "upconv_fused"]
# kernels above 1 % of the 32 x 768 x 768 image step (profiles/r04x_bench_image_b32_768_kernel_stats.csv), default 2 x f16 arithmetic
HOT_KERNELS = ["conv3x3_pl_kernel<3>", "conv3x3_pl_kernel<2>", "gemm_pl_kernel<3>", "gemm_pl_kernel<2>", "gemm1x1_pc_kernel<3, true, 2, 2, 1536>",
               "gemm1x1_pc_kernel<3, true, 2, 2, 3072>", "gemm1x1_pc_kernel<3, false, 2, 2, 3072>", "gemm1x1_pc_kernel<2, true, 2, 2, 1536>",
               "gemm1x1_pc_kernel<2, false, 2, 2, 3072>", "gemm1x1_pc_kernel<3, true, 2, 1, 1536>", "gemm1x1_pc_kernel<3, true, 2, 1, 3072>",
               "conv3x3_patch_pc_kernel<3, 8, 2>", "conv3x3_patch_pc_kernel<2, 8, 2>", "conv3x3_patch_pc_kernel<1, 8, 2>",
               "conv3x3_patch_pc_kernel<2, 16, 2>", "conv3x3_patch_kernel<2, 2, 2, 1, 2>", "conv3x3_patch_kernel<4, 1, 1, 1, 2>",
               "conv3x3_patch_kernel<2, 2, 2, 2, 2>", "conv3x3_small_kernel<true, 2>", "conv3x3_small_kernel<false, 2>",
               "conv_gemm_kernel<2, 2, 1, 1, 2>", "conv_gemm_kernel<4, 1, 1, 3, 2>", "conv_gemm_kernel<4, 1, 2, 1, 2>",
               "conv_gemm_kernel<2, 2, 2, 1, 2>", "conv_gemm_kernel<2, 2, 1, 2, 2>", "cnx_pipe_kernel<6, 1, false>", "cnx_pipe_kernel<12, 1, false>",
               "cnx_pipe_kernel<6, 2, true>", "cnx_pipe_kernel<12, 1, true>", "cnx_pipe_kernel<6, 1, true>", "cnx_pipe_kernel<6, 2, false>",
               "resblock_thin_kernel<2, 2, 2>", "resblock_thin_kernel<3, 1, 2>", "resblock_thin32_kernel", "upconv_fused_kernel<5, 4, 2>",
               "upconv_fused_kernel<9, 8, 2>"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    clean = [re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0].replace("void ", "").strip() for d in out]
    return dict(zip(names, clean))


def codegen_report(jobs=8):
    """{kernel (demangled, no arguments): {vgpr_spill, sgpr_spill, vgpr, agpr, lds, k_loops: [{mfma, scratch, vm0, vmN, bar, dma}]}} for HOT_KERNELS.
    A K loop = a loop without child loops that holds >= 4 MFMAs (the persistent tile loop of gemm1x1_pc, which wraps K loop + epilogue, is not one)."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(ROOT, "videoseal_amd", "csrc", f + ".hip") for f in HOT_SOURCES]
    with ThreadPoolExecutor(jobs) as ex:
        texts = list(ex.map(compile_s, srcs))
    report = {}
    for text in texts:
        res = resources(text)
        names = demangle(list(res))
        stats = loop_stats(text, "")
        for mangled, r in res.items():
            k = names[mangled]
            if k not in HOT_KERNELS:
                continue
            kl = [{key: c.get(key, 0) for key in ("mfma", "scratch", "vm0", "vmN", "bar", "dma")}
                  for (hdr, depth), c in stats.get(mangled, {}).items() if c["leaf"] and c["mfma"] >= 4]
            report[k] = dict(r, k_loops=kl)
    return report


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "budget":
        import json
        rep = codegen_report()
        path = os.path.join(ROOT, "tests", "golden", "codegen_budget.json")
        if "--write" in sys.argv:
            json.dump(rep, open(path, "w"), indent=1, sort_keys=True)
            print("wrote", path)
        for k in HOT_KERNELS:
            r = rep.get(k)
            print(f"{k:42s}", "MISSING" if r is None else f"vgpr {r['vgpr']:3d}+{r['agpr']:3d}a spill {r['vgpr_spill']:3d} sgpr_spill {r['sgpr_spill']:3d} "
                  f"lds {r['lds']:6d}  K loops: " + "; ".join(f"{l['mfma']} mfma, {l['scratch']} scratch, {l['vm0']} vmcnt(0)" for l in r["k_loops"]))
        sys.exit(0)
    if len(sys.argv) < 3 or sys.argv[1] not in ("loops", "order", "mix"):
        sys.exit(__doc__)
    txt = compile_s(sys.argv[2])
    sub = sys.argv[3] if len(sys.argv) > 3 else ""
    {"loops": loops, "order": order, "mix": mix}[sys.argv[1]](txt, sub)
