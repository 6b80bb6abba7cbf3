#!/usr/bin/env python3
"""Do two builds of the library compute the same bits?  Each build (QR_LIB) steps the same Quad-v0 batch with in-launch
resets in its own subprocess and prints a digest of state, parameters, rewards, dones, episode and tile counters.

    python tools/ab_equal.py build/ab/A.so build/ab/B.so        (GPU box)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, sys, torch
sys.path.insert(0, %r)
from gym_rotor_amd import QuadVecEnv
env = QuadVecEnv("quad", 65536, device="cuda", seed=3, auto_reset=True)
env.reset("train")
g = torch.Generator(device="cuda"); g.manual_seed(1)
h = hashlib.sha256()
for t in range(300):
    _, r, d, _, _ = env.step(torch.rand(65536, 4, device="cuda", generator=g) * 2 - 1)
    if t %% 50 == 49:
        for x in (env.get_current_state(), env._params, r, d, env._episode, env._reset_count):
            h.update(x.cpu().numpy().tobytes())
print(h.hexdigest(), int(env._episode.sum()))
''' % ROOT
out = []
for lib in sys.argv[1:3]:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, QR_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    out.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 else "FAILED " + r.stderr[-300:])
    print(os.path.basename(lib), out[-1])
print("IDENTICAL" if out[0] == out[1] and not out[0].startswith("FAILED") else "DIFFERENT")
