#!/usr/bin/env python3
"""Which kernels' machine code differs between two builds of the library?  CPU only (no GPU needed).

    python tools/codeobj_diff.py A.so B.so [--list] [--strip-mag]

Extracts the gfx950 code object of each library (llvm-objdump --offloading), disassembles it and compares every kernel's
instruction stream symbol by symbol (addresses and branch targets normalised).  Prints the kernels that exist in one build
only and those whose code differs; exit code 0 when the device code of every common kernel is identical.  What it is for:
a host-side change (launch rule, C-ABI) must leave every kernel untouched; a kernel change shows exactly which of the 126
step_kernel instantiations it reached (tests/test_gpu_digest.py then tells whether their RESULTS changed).
--strip-mag: compare a build from before the 11th template argument (MAG, round 6) with one after it — the MAG = false kernels of
the newer build are matched with the older build's kernels of the same first ten arguments.
"""
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib: str) -> dict:
    """{demangled kernel name: sha1 of its normalised disassembly, instruction count}"""
    tmp = tempfile.mkdtemp(prefix="codeobj_")
    try:
        dst = os.path.join(tmp, "lib.so")
        shutil.copy(lib, dst)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", dst], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not co:
            raise SystemExit(f"{lib}: no gfx950 code object")
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "-C", os.path.join(tmp, co[0])],
                             check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out, name, body = {}, None, []

    def flush():
        if name is not None:
            text = "\n".join(body)
            out[name] = (hashlib.sha1(text.encode()).hexdigest(), len(body))

    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            flush()
            name, body = m.group(1), []
            continue
        if name is None or not line.strip():
            continue
        ins = line.split("//")[0].strip()                       # drop the address comment
        if ins == "...":                                        # zero fill between symbols
            continue
        ins = re.sub(r"^[0-9a-f]+:\s*", "", ins)
        ins = re.sub(r"<[^>]*\+0x[0-9a-f]+>", "<L>", ins)      # branch targets: symbol + offset
        ins = re.sub(r"\b(s_c?branch\S*|s_call\S*)\s+\S+", r"\1 L", ins)
        body.append(ins)
    flush()
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if len(args) != 2:
        raise SystemExit(__doc__)
    a, b = kernels(args[0]), kernels(args[1])
    if "--strip-mag" in sys.argv:
        def strip(d):
            return {re.sub(r"(step_kernel<(?:[^,<>]+, ){9}[^,<>]+), false>", r"\1>", k): v for k, v in d.items()}
        a, b = strip(a), strip(b)
    only_a, only_b = sorted(set(a) - set(b)), sorted(set(b) - set(a))
    diff = sorted(k for k in set(a) & set(b) if a[k][0] != b[k][0])
    print(f"{args[0]}: {len(a)} kernels; {args[1]}: {len(b)} kernels; common {len(set(a) & set(b))}, identical {len(set(a) & set(b)) - len(diff)}")
    for k in only_a:
        print(f"  only in A ({a[k][1]} insts): {k[:160]}")
    for k in only_b:
        print(f"  only in B ({b[k][1]} insts): {k[:160]}")
    for k in diff:
        print(f"  DIFFERENT ({a[k][1]} -> {b[k][1]} insts): {k[:160]}")
    if "--list" in sys.argv:
        for k in sorted(set(a) & set(b)):
            print(f"  {a[k][1]:6d} {k[:160]}")
    sys.exit(1 if diff else 0)


if __name__ == "__main__":
    main()
