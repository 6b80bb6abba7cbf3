"""SHA-256 digests of what the shipped library computes — the bits behind every "bit-identical" claim in DESIGN.md, as data.

One digest per (env kind, launch family): 300 qr_step launches with in-launch resets at 65 536 envs (helper-wave instantiation), the
same at 300 001 envs x 20 steps (plain instantiation, ragged last tile), two qr_rollout horizons, and (wrappers) qr_rollout_actor with a
PPO-form and an SAC-form actor; with 4 / 10 / 2 substeps (the Magnus-substep instantiations, MAG = 1) a helper-wave and a plain
qr_step series and a rollout — state, parameters, integrators, rewards, raw rewards, dones, observation rows, terminal observations,
episode and tile counters all go in.  tests/test_gpu_digest.py compares them with tests/golden/digest_gfx950.json;
tools/make_digest.py (re)writes that file on a GPU box.
"""
import hashlib
import subprocess

import torch

KINDS = ("quad", "coupled", "decoupled")


def compiler_id() -> str:
    """First line of `hipcc --version` + the clang line: the code generator the digests belong to."""
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout.splitlines()
        return " | ".join(l.strip() for l in out if l.startswith("HIP version") or "clang version" in l)
    except Exception as ex:  # pragma: no cover
        return f"unavailable ({type(ex).__name__})"


def _upd(h, tensors):
    for x in tensors:
        if x is not None:
            h.update(x.detach().contiguous().cpu().numpy().tobytes())


def _rows(x):
    return [] if x is None else ([x] if isinstance(x, torch.Tensor) else list(x))


def digests(kind: str, n_small: int = 65536, n_large: int = 300001) -> dict:
    from gym_rotor_amd import QuadVecEnv, random_actors
    out = {}
    g = torch.Generator(device="cuda"); g.manual_seed(1)

    def make(n, substeps=1):
        env = QuadVecEnv(kind, n, device="cuda", seed=3, auto_reset=True, obs_rows=True, final_obs=True, want_raw_reward=True, autotune=False,
                         substeps=substeps)
        env.reset("train")
        if kind != "quad":
            env.get_norm_error_state()
        return env

    def everything(env, o, r, d):
        fin = _rows(env.final_observation())
        rows = d.reshape(env.num_envs, -1).any(dim=1)
        return [env.get_current_state(), env._params, r, env._reward_raw, d, env._episode, env._reset_count] + _rows(o) + [f[rows] for f in fin] + [env._integ]

    def step_series(tag, n, steps, every, sub, gen):
        env = make(n, sub)
        h = hashlib.sha256()
        for t in range(steps):
            o, r, d, _, _ = env.step(torch.rand(n, env.action_dim, device="cuda", generator=gen) * 2 - 1)
            if t % every == every - 1:
                _upd(h, everything(env, o, r, d))
        out[tag] = h.hexdigest()
        out[tag + "_kernel"] = env.launch_plan()["name"]
        return env

    env = step_series("step_helper", n_small, 300, 10, 1, g)
    step_series("step_plain", n_large, 20, 5, 1, g)

    def upd_dict(h, d):
        for k in sorted(d):
            if k != "obs":
                _upd(h, _rows(d[k]))

    h = hashlib.sha256()
    for rep in range(2):
        upd_dict(h, env.rollout(torch.rand(24, n_small, env.action_dim, device="cuda", generator=g) * 2 - 1))
    _upd(h, [env.get_current_state(), env._reset_count, env._integ])
    out["rollout"] = h.hexdigest()
    out["rollout_kernel"] = env.launch_plan(24)["name"]
    if kind != "quad":
        for algo in ("ppo", "sac"):
            h = hashlib.sha256()
            actors = random_actors(kind, "cuda", generator=torch.Generator(device="cuda").manual_seed(5), log_std=-0.5, algo=algo)
            env.get_norm_error_state()
            upd_dict(h, env.rollout_actor(actors, 8))
            _upd(h, [env.get_current_state(), env._integ, env._reset_count])
            out["rollout_actor_" + algo] = h.hexdigest()
            out[f"rollout_actor_{algo}_kernel"] = env.launch_plan(8, actor=algo)["name"]
    out["episodes"] = int(env._episode.sum())
    # two or more substeps: the Magnus-substep instantiations (their own random stream, after everything above: the one-substep
    # digests do not move when this part changes)
    g2 = torch.Generator(device="cuda"); g2.manual_seed(2)
    step_series("step_helper_x4", n_small, 100, 10, 4, g2)
    step_series("step_plain_x10", n_large, 10, 5, 10, g2)
    env2 = make(n_small, 2)
    h = hashlib.sha256()
    upd_dict(h, env2.rollout(torch.rand(24, n_small, env2.action_dim, device="cuda", generator=g2) * 2 - 1))
    _upd(h, [env2.get_current_state(), env2._reset_count, env2._integ])
    out["rollout_x2"] = h.hexdigest()
    out["rollout_x2_kernel"] = env2.launch_plan(24)["name"]
    return out
