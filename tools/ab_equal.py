#!/usr/bin/env python3
"""Do two builds of the library compute the same bits?  Each build (QR_LIB) steps the same batch (QR_AB_KIND, default quad; QR_AB_ENVS, default 65 536
envs) with in-launch resets in its own subprocess and prints a digest of state, integrators, parameters, observation rows, rewards,
dones, terminal observations, episode and tile counters — then the same for qr_rollout and (wrappers) qr_rollout_actor with a PPO and an SAC actor.

    [QR_AB_KIND=coupled] python tools/ab_equal.py build/ab/A.so build/ab/B.so        (GPU box)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, sys, torch
sys.path.insert(0, %r)
from gym_rotor_amd import QuadVecEnv
import os
kind = os.environ.get("QR_AB_KIND", "quad")
NE = int(os.environ.get("QR_AB_ENVS", "65536"))
env = QuadVecEnv(kind, NE, device="cuda", seed=3, auto_reset=True, obs_rows=True, final_obs=True, want_raw_reward=True)
env.reset("train")
if kind != "quad":
    env.get_norm_error_state()
g = torch.Generator(device="cuda"); g.manual_seed(1)
h = hashlib.sha256()
for t in range(300):
    o, r, d, _, _ = env.step(torch.rand(NE, env.action_dim, device="cuda", generator=g) * 2 - 1)
    if t %% 10 == 9:
        obs = [o] if isinstance(o, torch.Tensor) else list(o)
        fin = env.final_observation()
        fin = [fin] if isinstance(fin, torch.Tensor) else list(fin)
        rows = d.reshape(NE, -1).any(dim=1)
        for x in [env.get_current_state(), env._params, r, env._reward_raw, d, env._episode, env._reset_count] + obs + [f[rows] for f in fin] + ([env._integ] if env._integ is not None else []):
            h.update(x.cpu().numpy().tobytes())
step_digest = h.hexdigest()
# the multi-step instantiations: qr_rollout (T = 24, twice) and, for the wrappers, qr_rollout_actor (PPO and SAC forms, T = 8)
h = hashlib.sha256()
def upd(d):
    for k in sorted(d):
        v = d[k]
        if k == "obs" or v is None:
            continue
        for x in (v if isinstance(v, (tuple, list)) else [v]):
            h.update(x.cpu().numpy().tobytes())
for rep in range(2):
    upd(env.rollout(torch.rand(24, NE, env.action_dim, device="cuda", generator=g) * 2 - 1))
h.update(env.get_current_state().cpu().numpy().tobytes()); h.update(env._reset_count.cpu().numpy().tobytes())
if kind != "quad":
    from gym_rotor_amd import random_actors
    for algo in ("ppo", "sac"):
        actors = random_actors(kind, "cuda", generator=torch.Generator(device="cuda").manual_seed(5), log_std=-0.5, algo=algo)
        env.get_norm_error_state()
        upd(env.rollout_actor(actors, 8))
    h.update(env.get_current_state().cpu().numpy().tobytes()); h.update(env._integ.cpu().numpy().tobytes())
print(step_digest, h.hexdigest(), int(env._episode.sum()))
''' % ROOT
out = []
for lib in sys.argv[1:3]:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, QR_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    out.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 else "FAILED " + r.stderr[-300:])
    print(os.path.basename(lib), out[-1])
print("IDENTICAL" if out[0] == out[1] and not out[0].startswith("FAILED") else "DIFFERENT")
