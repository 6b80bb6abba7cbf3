#!/usr/bin/env python3
"""Where do a step_kernel instantiation's instructions go?  No GPU needed.

Builds the device code with full debug info (one env kind, default layout: ~10 s), disassembles ONE instantiation and asks
llvm-symbolizer for the inline stack of every instruction address.  Each instruction is attributed to
  * the SECTION of step_kernel it was inlined into: the `//@sec <name>` marker comment nearest above the step_kernel line of its
    outermost frame (quadrotor_kernels.hip), and
  * the function called directly from step_kernel at that point (integrate, action_map, quad_done, store_state, ...).
Static counts; the per-env-step loop of a multi-step instantiation and the substep loop are marked by their sections.

    python tools/isa_breakdown.py [--kind 0|1|2] [--flags TRAJ,ADAPT,POLICY,SINGLE,HELP] [--by-callee] [--elf /tmp/x.elf]
    e.g. the plain one-step Quad-v0 kernel:   python tools/isa_breakdown.py --kind 0 --flags 0,0,0,1,0
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
SRC = os.path.join(ROOT, "gym_rotor_amd", "csrc", "quadrotor_kernels.hip")

p = argparse.ArgumentParser()
p.add_argument("--kind", type=int, default=0)
p.add_argument("--flags", default="0,0,0,1,0", help="TRAJ,ADAPT,POLICY,SINGLE,HELP[,HREW[,MAG]] of the instantiation (HREW = 1, MAG = 0 by default)")
p.add_argument("--elf", default="")
p.add_argument("--by-callee", action="store_true")
p.add_argument("--extra", default="", help="extra -D flags for the build")
a = p.parse_args()
traj, adapt, policy, single, helpw, hrew, mag = ([int(x) for x in a.flags.split(",")] + [1, 0])[:7] if len(a.flags.split(",")) < 7 else [int(x) for x in a.flags.split(",")]
if len(a.flags.split(",")) == 5:
    hrew, mag = 1, 0

elf = a.elf
if not elf:
    obj, elf = f"/tmp/qr_dbg_k{a.kind}.o", f"/tmp/qr_dbg_k{a.kind}.elf"
    cmd = [f"/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{ROOT}/include", "-ffp-contract=fast", "-fno-slp-vectorize",
           "-mllvm", "-amdgpu-kernarg-preload-count=16", "-Wno-everything", f"-DQR_ONLY_KIND={a.kind}", "-DQR_ONLY_LAYOUT=0", "-g",
           "--cuda-device-only", "-c", "-o", obj, SRC] + a.extra.split()
    subprocess.run(cmd, check=True, cwd="/tmp")
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={obj}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={elf}"], check=True)

# the instantiation's symbol
want = f"step_kernelILi{a.kind}EfdLi64ELi{traj}ELb{adapt}ELi{policy}ELb{single}ELb{helpw}ELb{hrew}ELb{mag}EE"
syms = subprocess.run([f"{LLVM}/llvm-objdump", "-t", elf], capture_output=True, text=True, check=True).stdout
sym = None
for l in syms.splitlines():
    f = l.split()
    if len(f) >= 5 and want in f[-1] and "F" in f and ".text" in f:
        sym = (int(f[0], 16), int(f[f.index(".text") + 1], 16), f[-1])
if not sym:
    sys.exit(f"no symbol matching {want}")
start, size, name = sym
dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f"--start-address={start}", f"--stop-address={start + size}", elf],
                     capture_output=True, text=True, check=True).stdout
insts = []   # (addr, mnemonic)
for l in dis.splitlines():
    m = re.match(r"\s+(\S+)\s.*//\s*([0-9A-Fa-f]+):", l)
    if m:
        insts.append((int(m.group(2), 16), m.group(1)))
if not insts:
    sys.exit("no instructions disassembled")
sy = subprocess.run([f"{LLVM}/llvm-symbolizer", "--inlines", f"--obj={elf}", "--output-style=LLVM"], input="\n".join(hex(x) for x, _ in insts) + "\n",
                    capture_output=True, text=True, check=True).stdout
blocks = [b for b in sy.split("\n\n") if b.strip()]
assert len(blocks) == len(insts), (len(blocks), len(insts))

# section markers of step_kernel
src = open(SRC).read().splitlines()
sec_at, cur = {}, "prologue"
for n, l in enumerate(src, 1):
    m = re.search(r"//@sec\s+(\S+)", l)
    if m:
        cur = m.group(1)
    sec_at[n] = cur


def kind_of(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    return "vmem"


tab = collections.OrderedDict()
for (addr, op), b in zip(insts, blocks):
    lines = b.strip().splitlines()
    frames = [(lines[k], lines[k + 1]) for k in range(0, len(lines) - 1, 2)]   # innermost first: (function, file:line:col)
    sec, callee = "?", "(step_kernel)"
    for k in range(len(frames) - 1, -1, -1):
        fn, loc = frames[k]
        if "step_kernel" in fn and "quadrotor_kernels.hip" in loc:
            ln = int(loc.split(":")[-2])
            sec = sec_at.get(ln, "?")
            if k > 0:
                callee = re.sub(r"<.*", "", frames[k - 1][0]).replace("qr::", "").split("(")[0]
                callee = re.sub(r".*::(operator\(\))", r"lambda", callee)
            break
    key = (sec, callee) if a.by_callee else (sec,)
    row = tab.setdefault(key, collections.Counter())
    row[kind_of(op)] += 1
    if op.startswith("v_") and "_f64" in op: row["f64"] += 1
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op): row["trans"] += 1
    if op.startswith("v_readlane") or op.startswith("v_writelane"): row["lane_spill"] += 1

print(f"{name}: {len(insts)} instructions")
cols = ["valu", "f64", "trans", "lane_spill", "mfma", "salu", "smem", "lds", "vmem", "wait"]
print("%-44s" % "section" + "".join("%8s" % c for c in cols) + "%8s" % "all")
tot = collections.Counter()
for key, row in tab.items():
    print("%-44s" % " / ".join(key)[:44] + "".join("%8d" % row[c] for c in cols) + "%8d" % sum(row[c] for c in ("valu", "mfma", "salu", "smem", "lds", "vmem", "wait")))
    tot.update(row)
print("%-44s" % "TOTAL" + "".join("%8d" % tot[c] for c in cols) + "%8d" % len(insts))
