#!/usr/bin/env python3
"""How far behind is the producer of each vector instruction's inputs?  No GPU needed.

A lone wave pays 3.6-3.9 ns for a VALU instruction that waits for the one before it and 2.0-2.3 ns for an independent one
(tools/valu_ilp_microbench.hip), so for the launches with one stepping wave per SIMD the ORDER of the instructions matters, not only
their count.  For every VALU instruction of a kernel this prints the histogram of the distance (in VALU instructions) to the nearest
earlier VALU instruction that writes one of its source registers: 1 = it waits for its predecessor.

    python tools/isa_dep_distance.py ELF-or-.so-codeobject SUBSTRING-of-the-mangled-kernel-name [first last]   (instruction index range)
"""
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin"


def regs(tok):
    r = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        r.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        r.add(int(m.group(1)))
    return r


def kernel(elf, pat):
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", elf], capture_output=True, text=True, check=True).stdout
    out, on = [], False
    for l in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", l)
        if m:
            on = pat in m.group(1)
            continue
        if on:
            mm = re.match(r"^\s*(\S.*?)\s+//\s*([0-9A-F]+):", l)
            if mm:
                out.append(mm.group(1))
    return out


def main():
    elf, pat = sys.argv[1], sys.argv[2]
    ins = kernel(elf, pat)
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(ins))
    hist, prev, nv = {}, [], 0
    for i, t in enumerate(ins):
        if not t.startswith("v_"):
            continue
        ops = t.split(None, 1)[1] if " " in t else ""
        parts = [p.strip() for p in ops.split(",")]
        dst = regs(parts[0]) if parts else set()
        src = set()
        for p_ in parts[1:]:
            src |= regs(p_)
        if t.startswith(("v_fmac", "v_mac")):
            src |= dst
        d = None
        for k, pd in enumerate(reversed(prev[-8:])):
            if pd & src:
                d = k + 1
                break
        prev.append(dst)
        if lo <= i < hi:
            nv += 1
            hist[d] = hist.get(d, 0) + 1
    print(f"{pat}: {len(ins)} instructions, {nv} VALU in [{lo}, {hi})")
    for k in sorted(hist, key=lambda x: (x is None, x)):
        print(f"  producer {k if k is not None else '> 8 (or none)'} back: {hist[k]:5d}  {100.0 * hist[k] / nv:5.1f} %")


if __name__ == "__main__":
    main()
