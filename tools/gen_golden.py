#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (fdcl-gwu/gym-rotor, read-only
at /root/reference) in the build container.  The reference never travels to the GPU
box: only the vectors written here do.  Re-run with `python tools/gen_golden.py`.

Requires: /root/reference, numpy, scipy.  `gymnasium` is not installed in this image,
so a ~40-line stand-in (tools/_gymnasium_shim) is put on sys.path first; `sys.argv` is
set before every constructor because the reference re-parses it (quad.py:24-25).

Inputs (x, v, W, actions, parameters, goals, integrator terms) are float32-representable so
the GPU path, whose I/O is float32, sees bit-identical inputs; input rotations are exact
float64 rotations (see state_in).

Files written (see tests/golden/README.md for the field lists):
  kat_units.npz            hat / ensure_SO3 / angle / euler / interp / mixing known answers
  onestep_{kind}.npz       512 single-step transitions per env kind
  traj_free_{kind}.npz     1000-step free-run (no reset) trajectories, 4 envs per kind
  traj_reset_{kind}.npz    1000-step trajectories with reset-on-done to injected states
  errobs_formats.npz       (`python tools/gen_golden.py errobs`) get_norm_error_state("MONO" | "MODUL") on BOTH wrapper classes: the argument selects the format
  flightlog_modul.npz      all 3600 rows of results/MODUL_log_20250303_120200.dat
  gae.npz                  the reference's own GAE + normalisation lines (ppo.py:134-147) on synthetic data
  trajgoal_m{0,1,6}_{kind}.npz  closed loop env + TrajectoryGenerator (modes 0/1/6) as main.py drives them
  actor_ppo.npz            the reference's MLP_Actor_PPO (torch): weights, obs -> mean, injected-noise action, log_prob
  actor_td3.npz            the reference's MLP_Actor_TD3 + explicit-noise choose_action
  actor_sac.npz            the reference's MLP_Actor_SAC forward + explicit-noise sample
  actorloop_{kind}.npz     closed loop: reference wrapper env stepped by the reference's actor(s), 4 envs x 200 steps
  closedloop_td3_{mono,modul}.npz   (`python tools/gen_golden.py td3`) the reference's eval loop (main.py:270-345) with the SHIPPED
                           TD3-EMLP actors (models/*.pth) in TrajectoryGenerator modes 0 / 1 / 6: realistic, non-random actions
                           incl. eight-shaped-curve tracking.  Needs tools/_plum_shim (see there).
  closedloop_td3_{mono,modul}_modes2345.npz   (`python tools/gen_golden.py td3modes`) the same loop in TrajectoryGenerator modes 2 (take-off), 3 (landing),
                           4 (stay) and 5 (circle), from start positions that reach every branch of the generator
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(HERE, "_gymnasium_shim"))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

ARGV = sys.argv[1:]
sys.argv = ["gen_golden"]
import gym_rotor.envs.quad as refquad  # noqa: E402
from gym_rotor.envs import quad_utils as refutils  # noqa: E402
from gym_rotor.envs.quad import QuadEnv  # noqa: E402
from gym_rotor.wrappers.coupled_yaw_wrapper import CoupledWrapper  # noqa: E402
from gym_rotor.wrappers.decoupled_yaw_wrapper import DecoupledWrapper  # noqa: E402

from oracle import quad_oracle as orc  # noqa: E402  (only for the reset-state sampler)

f32r = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)  # float32-representable f64


def state_in(s):
    """Input states: x, v, W float32-representable; R an exact (float64) rotation, as every
    state the reference itself produces is (reset builds R from Euler angles in float64 and
    the ODE flow keeps it orthonormal to ~3e-15)."""
    s = np.array(s, dtype=np.float64)
    lead = s.shape[:-1]
    R = np.swapaxes(s[..., 6:15].reshape(-1, 3, 3), 1, 2)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    out = f32r(s)
    out[..., 6:15] = np.swapaxes(R, 1, 2).reshape(lead + (9,))
    return out


def make_env(kind):
    if kind == "decoupled":
        sys.argv = ["x", "--framework", "MODUL"]
        env = DecoupledWrapper()
    elif kind == "coupled":
        sys.argv = ["x", "--framework", "MONO"]
        env = CoupledWrapper()
    else:
        sys.argv = ["x", "--framework", "MONO"]
        env = QuadEnv()
        env.alpha, env.beta, env.eIx_lim, env.eIb1_lim = 0.01, 0.05, 3.0, 3.0
    env.reset(env_type="eval")
    return env


def inject_params(env, p):
    """Let the reference itself derive hover_force/max_force/mixing from prescribed draws:
    `set_random_parameters('train')` calls `uniform(low=, high=)` six times in the order
    m, d, J1, J3, c_tf, c_tw (quad.py:380-387)."""
    queue = list(p)
    saved = refquad.uniform

    def fake_uniform(low=0.0, high=1.0, size=None):
        v = queue.pop(0)
        assert low - 1e-12 <= v <= high + 1e-12, (low, v, high)
        return v

    refquad.uniform = fake_uniform
    try:
        env.set_random_parameters("train")
    finally:
        refquad.uniform = saved
    assert not queue


def inject(env, state, goal, integ):
    env.state = np.array(state, dtype=np.float64)
    env.set_goal_state(goal[0:3].copy(), goal[3:6].copy(), goal[6:9].copy(), np.zeros(3), goal[9:12].copy())
    env.eIx.error, env.eIx.integrand = integ[0:3].copy(), integ[3:6].copy()
    env.eIb1.error, env.eIb1.integrand = float(integ[6]), float(integ[7])


def read_integ(env):
    return np.concatenate([env.eIx.error, env.eIx.integrand, [env.eIb1.error, env.eIb1.integrand]])


def ref_step(env, kind, action):
    """One reference step.  Returns next_state, f, M, obs(list), reward_raw, reward, done."""
    if kind == "quad":  # QuadEnv.step raises at HEAD (reward[0] on a scalar): call the hooks
        env.action_wrapper(np.array(action, dtype=np.float64))
        obs = env.observation_wrapper(env.state.copy())
        raw = [float(env.reward_wrapper(obs))]
        done = [bool(env.done_wrapper(obs))]
        reward = [float(np.interp(raw[0], [env.reward_min, 0.0], [0.0, 1.0]))]
        if done[0]:
            reward[0] = env.reward_crash
        obs = [np.array(obs, dtype=np.float64)]
    else:
        # run the hooks exactly as QuadEnv.step does, but keep the raw reward too
        env.action_wrapper(np.array(action, dtype=np.float64))
        obs = env.observation_wrapper(env.state.copy())
        raw = [float(r) for r in env.reward_wrapper(obs)]
        mins = [env.reward_min_1, env.reward_min_2] if kind == "decoupled" else [env.reward_min]
        reward = [float(np.interp(r, [mn, 0.0], [0.0, 1.0])) for r, mn in zip(raw, mins)]
        done = [bool(d) for d in env.done_wrapper(obs)]
        for i, d in enumerate(done):
            if d:
                reward[i] = env.reward_crash
    M = np.array(env.fM[1:4]).ravel() if kind == "decoupled" else np.array(env.M, dtype=np.float64).ravel()
    return env.state.copy(), float(np.ravel(env.f)[0]), M, obs, raw, reward, done


def check_step_template(kind):
    """Confirm the hook sequence above reproduces env.step() for the two wrappers."""
    rng = np.random.default_rng(5)
    env_a, env_b = make_env(kind), make_env(kind)
    st = state_in(orc.sample_reset_state(rng, 1)[0])
    for e in (env_a, env_b):
        inject(e, st, orc.DEFAULT_GOAL, np.zeros(8))
    for _ in range(20):
        a = f32r(rng.uniform(-1, 1, orc.ACTION_DIM[kind]))
        obs, rwd, done, _, _ = env_a.step(a.copy())
        _, _, _, obs_b, _, rwd_b, done_b = ref_step(env_b, kind, a)
        assert all(np.array_equal(x, y) for x, y in zip(obs, obs_b))
        assert list(map(float, rwd)) == rwd_b and list(done) == done_b


# ------------------------------------------------------------------------------------
def random_params(rng, n):
    return f32r(orc.sample_params(rng, n, "train"))


def random_goal(rng, n):
    g = np.tile(orc.DEFAULT_GOAL, (n, 1))
    psi = rng.uniform(-np.pi, np.pi, n)
    g[:, 0:3] = rng.uniform(-0.3, 0.3, (n, 3))
    g[:, 3:6] = rng.uniform(-0.5, 0.5, (n, 3))
    g[:, 6], g[:, 7], g[:, 8] = np.cos(psi), np.sin(psi), 0.0
    g[:, 9:12] = rng.uniform(-0.5, 0.5, (n, 3))
    return f32r(g)


def boundary_states(rng, n):
    """States straddling each termination threshold (|x|~1, |v|~4, |W|~2pi, roll/pitch~85deg)."""
    s = orc.sample_reset_state(rng, n, "train")
    for i in range(n):
        which = i % 5
        eps = rng.uniform(-0.02, 0.02)
        j = rng.integers(0, 3)
        sgn = rng.choice([-1.0, 1.0])
        if which == 0:
            s[i, 0 + j] = sgn * (1.0 + eps)
        elif which == 1:
            s[i, 3 + j] = sgn * (4.0 + 4 * eps)
        elif which == 2:
            s[i, 15 + j] = sgn * (2 * np.pi + 6 * eps)
        else:
            ang = np.deg2rad(85.0 + 100 * eps)
            roll, pitch = (sgn * ang, rng.uniform(-0.3, 0.3)) if which == 3 else (rng.uniform(-0.3, 0.3), sgn * ang)
            s[i, 6:15] = orc.euler_xyz_to_R(roll, pitch, rng.uniform(-np.pi, np.pi)).reshape(9, order="F")
            s[i, 15:18] *= 0.05
    return s


def gen_onestep(kind, n=512, seed=0):
    rng = np.random.default_rng(1000 + seed + 17 * orc.KINDS.index(kind))
    A = orc.ACTION_DIM[kind]
    nb = n // 4
    state = state_in(np.concatenate([orc.sample_reset_state(rng, n - nb, "train"), boundary_states(rng, nb)]))
    action = f32r(rng.uniform(-1, 1, (n, A)))
    action[::7] = f32r(np.sign(action[::7]))  # saturated commands
    params = np.tile(orc.NOMINAL_PARAMS, (n, 1)); params[1::2] = random_params(rng, n)[1::2]
    params = f32r(params)
    goal = np.tile(orc.DEFAULT_GOAL, (n, 1)); goal[n // 2:] = random_goal(rng, n)[n // 2:]
    integ = np.zeros((n, 8))
    integ[:, 0:3] = rng.uniform(-1.0, 1.0, (n, 3)); integ[::5, 0:3] = rng.uniform(-4.0, 4.0, (len(integ[::5]), 3))
    integ[:, 3:6] = rng.uniform(-1.0, 1.0, (n, 3))
    integ[:, 6] = rng.uniform(-2.0, 2.0, n); integ[::9, 6] = rng.uniform(-4.0, 4.0, len(integ[::9]))
    integ[:, 7] = rng.uniform(-3.0, 3.0, n)
    integ = f32r(integ)
    if kind == "quad":
        integ[:] = 0.0
    env = make_env(kind)
    obs_dims = {"quad": [18], "coupled": [23], "decoupled": [15, 3]}[kind]
    nag = orc.N_AGENTS[kind]
    out = dict(state=state, action=action, params=params, goal=goal, integ=integ,
               next_state=np.zeros((n, 18)), f=np.zeros(n), M=np.zeros((n, 3)),
               reward_raw=np.zeros((n, nag)), reward=np.zeros((n, nag)), done=np.zeros((n, nag), bool),
               next_integ=np.zeros((n, 8)))
    obs_out = [np.zeros((n, d), np.float64 if kind == "quad" else np.float32) for d in obs_dims]
    for i in range(n):
        inject_params(env, params[i])
        inject(env, state[i], goal[i], integ[i])
        ns, f, M, obs, raw, rwd, done = ref_step(env, kind, action[i])
        out["next_state"][i], out["f"][i], out["M"][i] = ns, f, M
        out["reward_raw"][i], out["reward"][i], out["done"][i] = raw, rwd, done
        out["next_integ"][i] = read_integ(env)
        for k, o in enumerate(obs):
            obs_out[k][i] = o
    for k, o in enumerate(obs_out):
        out[f"obs{k}"] = o
    np.savez_compressed(os.path.join(OUT, f"onestep_{kind}.npz"), **out)
    print(f"onestep_{kind}: n={n} done-rate={out['done'].mean():.3f}")


def gen_traj(kind, mode, n_env=4, T=1000, seed=0):
    """mode 'free': never reset.  mode 'reset': when any agent's done fires, the caller
    resets that env to the next injected state, zeroes the integrators and calls
    get_norm_error_state() once (main.py:226-230) before the next step."""
    rng = np.random.default_rng(2000 + seed + 31 * orc.KINDS.index(kind) + (7 if mode == "reset" else 0))
    A = orc.ACTION_DIM[kind]
    nag = orc.N_AGENTS[kind]
    obs_dims = {"quad": [18], "coupled": [23], "decoupled": [15, 3]}[kind]
    params = np.tile(orc.NOMINAL_PARAMS, (n_env, 1)); params[n_env // 2:] = random_params(rng, n_env)[n_env // 2:]
    params = f32r(params)
    goal = np.tile(orc.DEFAULT_GOAL, (n_env, 1)); goal[1::2] = random_goal(rng, n_env)[1::2]
    init = state_in(orc.sample_reset_state(rng, n_env, "train"))
    pool = state_in(orc.sample_reset_state(rng, 64 * n_env, "train")).reshape(n_env, 64, 18)
    actions = f32r(rng.uniform(-1, 1, (T, n_env, A)))
    states = np.zeros((T + 1, n_env, 18)); integs = np.zeros((T + 1, n_env, 8))
    rewards = np.zeros((T, n_env, nag)); raws = np.zeros((T, n_env, nag)); dones = np.zeros((T, n_env, nag), bool)
    obs_out = [np.zeros((T, n_env, d), np.float64 if kind == "quad" else np.float32) for d in obs_dims]
    reset_at = np.zeros((T + 1, n_env), bool)   # reset applied BEFORE step t
    n_resets = np.zeros(n_env, int)
    for e in range(n_env):
        env = make_env(kind)
        inject_params(env, params[e])
        inject(env, init[e], goal[e], np.zeros(8))
        if kind != "quad":
            env.get_norm_error_state(env.framework)  # first obs after reset (main.py:129)
        for t in range(T):
            states[t, e], integs[t, e] = env.state, read_integ(env)
            ns, f, M, obs, raw, rwd, done = ref_step(env, kind, actions[t, e])
            rewards[t, e], raws[t, e], dones[t, e] = rwd, raw, done
            for k, o in enumerate(obs):
                obs_out[k][t, e] = o
            if mode == "reset" and any(done):
                inject(env, pool[e, n_resets[e] % 64], goal[e], np.zeros(8))
                n_resets[e] += 1
                reset_at[t + 1, e] = True
                if kind != "quad":
                    env.get_norm_error_state(env.framework)
        states[T, e], integs[T, e] = env.state, read_integ(env)
    out = dict(params=params, goal=goal, init_state=init, reset_pool=pool, actions=actions.astype(np.float32),
               states=states, integs=integs, rewards=rewards, rewards_raw=raws, dones=dones, reset_at=reset_at)
    for k, o in enumerate(obs_out):
        out[f"obs{k}"] = o
    np.savez_compressed(os.path.join(OUT, f"traj_{mode}_{kind}.npz"), **out)
    print(f"traj_{mode}_{kind}: resets={n_resets.tolist()} max|x|={np.abs(states[:, :, 0:3]).max():.2f} "
          f"max|W|={np.abs(states[:, :, 15:18]).max():.2f}")


def gen_kats():
    rng = np.random.default_rng(7)
    k = {}
    v = rng.normal(size=(16, 3))
    k["hat_in"], k["hat_out"] = v, np.stack([refutils.hat(x) for x in v])
    # ensure_SO3: both branches
    Rs, outs = [], []
    for i, eps in enumerate([0.0, 1e-7, 1e-6, 5e-6, 9e-6, 2e-5, 1e-4, 1e-3, 1e-2, 5e-2] * 3):
        R = orc.euler_xyz_to_R(*rng.uniform(-1.2, 1.2, 3)) + eps * rng.normal(size=(3, 3))
        Rs.append(R); outs.append(refutils.ensure_SO3(R.copy()))
    k["so3_in"], k["so3_out"] = np.stack(Rs), np.stack(outs)
    # heading angle helpers
    a = rng.normal(size=(64, 3)); b = rng.normal(size=(64, 3)); a[:, 2] = 0; b[:, 2] = 0
    k["ang_a"], k["ang_b"] = a, b
    k["ang_out"] = np.array([refutils.norm_ang_btw_two_vectors(x, y) for x, y in zip(a, b)])
    Rr = np.stack([orc.euler_xyz_to_R(*rng.uniform(-np.pi / 2.2, np.pi / 2.2, 2), rng.uniform(-np.pi, np.pi))
                   for _ in range(64)])
    k["b1_R"] = Rr
    k["b1_out"] = np.stack([refutils.get_current_b1(R) for R in Rr])
    # euler conventions (scipy)
    from scipy.spatial.transform import Rotation
    eul = rng.uniform(-1.4, 1.4, (64, 3)); eul[:, 2] = rng.uniform(-np.pi, np.pi, 64)
    k["euler_in"] = eul
    k["euler_R"] = np.stack([Rotation.from_euler("xyz", e).as_matrix() for e in eul])
    k["euler_back_deg"] = np.stack([Rotation.from_matrix(R).as_euler("xyz", degrees=True) for R in k["euler_R"]])
    # np.interp reward normalisation incl. both clip ends
    r = np.concatenate([np.linspace(-20, 2, 45), [-14.0, -8.0, -7.0, 0.0]])
    k["interp_in"] = r
    for mn in (-14.0, -8.0, -7.0):
        k[f"interp_out_{int(-mn)}"] = np.array([np.interp(x, [mn, 0.0], [0.0, 1.0]) for x in r])
    # motor mixing / action maps for the three kinds, nominal + random params
    for kind in orc.KINDS:
        env = make_env(kind)
        n = 64
        P = np.tile(orc.NOMINAL_PARAMS, (n, 1)); P[1::2] = random_params(rng, n)[1::2]
        P = f32r(P)
        Aa = f32r(rng.uniform(-1.3, 1.3, (n, orc.ACTION_DIM[kind])))
        S = state_in(orc.sample_reset_state(rng, n))
        F, Mo, der = np.zeros(n), np.zeros((n, 3)), np.zeros((n, 4))
        for i in range(n):
            inject_params(env, P[i]); inject(env, S[i], orc.DEFAULT_GOAL, np.zeros(8))
            env.action_wrapper(Aa[i].copy())
            if kind == "decoupled":  # M1, M2 are formed inside observation_wrapper (decoupled:68-73)
                env.observation_wrapper(env.state.copy())
                Mo[i] = np.array(env.fM[1:4]).ravel()
            else:
                Mo[i] = np.ravel(env.M)
            F[i] = float(np.ravel(env.f)[0])
            der[i] = [env.hover_force, env.max_force, env.avrg_act, env.scale_act]
        k[f"act_{kind}_params"], k[f"act_{kind}_a"], k[f"act_{kind}_state"] = P, Aa, S
        k[f"act_{kind}_f"], k[f"act_{kind}_M"], k[f"act_{kind}_derived"] = F, Mo, der
    k["constants"] = np.array([env.reward_min, make_env("decoupled").reward_min_1, make_env("decoupled").reward_min_2,
                               env.dt, env.x_lim, env.v_lim, env.W_lim, env.euler_lim])
    np.savez_compressed(os.path.join(OUT, "kat_units.npz"), **k)
    print("kat_units written")


def gen_trajgoal(kind, mode, n_env=4, T=400, seed=0):
    """Closed loop exactly as main.py drives it (eval loop :310-331, train loop :145-165,212-230):
    reset -> mark_traj_start(state) -> get_desired -> set_goal_state -> first obs; then every step
    get_desired(current state) -> set_goal_state -> step(action); reset-on-done to injected states.
    The generator's np.random.uniform draws are injected and recorded."""
    import utils.trajectory_generator as tg_mod
    rng = np.random.default_rng(3000 + seed + 31 * orc.KINDS.index(kind) + 7 * mode)
    A, nag = orc.ACTION_DIM[kind], orc.N_AGENTS[kind]
    obs_dims = {"coupled": [23], "decoupled": [15, 3]}[kind]
    params = np.tile(orc.NOMINAL_PARAMS, (n_env, 1)); params[n_env // 2:] = random_params(rng, n_env)[n_env // 2:]
    params = f32r(params)
    init = state_in(orc.sample_reset_state(rng, n_env, "train"))
    pool = state_in(orc.sample_reset_state(rng, 32 * n_env, "train")).reshape(n_env, 32, 18)
    actions = f32r(0.5 * rng.uniform(-1, 1, (T, n_env, A)))  # gentler actions: longer episodes
    draws = np.zeros((n_env, 33, 3))  # per episode: theta_b1d, t_traj, w_b1d
    draws[:, :, 0] = f32r(rng.uniform(-np.deg2rad(25), np.deg2rad(25), (n_env, 33)))
    draws[:, :, 1] = f32r(rng.uniform(2.0, 5.0, (n_env, 33)))
    draws[:, :, 2] = f32r(rng.uniform(-0.15 * np.pi, 0.15 * np.pi, (n_env, 33)))
    states = np.zeros((T + 1, n_env, 18)); goals = np.zeros((T, n_env, 15)); first_goal = np.zeros((T + 1, n_env, 15))
    rewards = np.zeros((T, n_env, nag)); dones = np.zeros((T, n_env, nag), bool)
    obs_out = [np.zeros((T, n_env, d), np.float32) for d in obs_dims]
    first_obs = [np.zeros((T + 1, n_env, d), np.float32) for d in obs_dims]
    reset_at = np.zeros((T + 1, n_env), bool); episode_of = np.zeros((T + 1, n_env), int)
    saved_uniform = np.random.uniform
    for e in range(n_env):
        env = make_env(kind)
        sys.argv = ["x", "--framework", env.framework]
        gen = tg_mod.TrajectoryGenerator(env)
        queue = []

        def fake_uniform(size=None, low=0.0, high=1.0):
            v = queue.pop(0)
            assert low - 1e-9 <= v <= high + 1e-9, (low, v, high)
            return np.array([v])

        def start_episode(state, ep, t):
            inject(env, state, orc.DEFAULT_GOAL, np.zeros(8))
            gen.mark_traj_start(env.state)
            queue[:] = {0: [draws[e, ep, 0]], 1: [draws[e, ep, 1], draws[e, ep, 2]]}.get(mode, [])
            xd, vd, b1d, b1d_dot, Wd = gen.get_desired(env.state, mode)
            assert not queue
            env.set_goal_state(xd, vd, b1d, b1d_dot, Wd)
            first_goal[t, e] = np.concatenate([xd, vd, b1d, b1d_dot, Wd])
            ob = env.get_norm_error_state(env.framework)
            for k, o in enumerate(ob):
                first_obs[k][t, e] = o
            reset_at[t, e] = True

        np.random.uniform = fake_uniform
        try:
            inject_params(env, params[e])
            ep = 0
            start_episode(init[e], ep, 0)
            for t in range(T):
                states[t, e], episode_of[t, e] = env.state, ep
                xd, vd, b1d, b1d_dot, Wd = gen.get_desired(env.state, mode)
                env.set_goal_state(np.copy(xd), np.copy(vd), np.copy(b1d), np.copy(b1d_dot), np.copy(Wd))
                goals[t, e] = np.concatenate([xd, vd, b1d, b1d_dot, Wd])
                ns, f, M, obs, raw, rwd, done = ref_step(env, kind, actions[t, e])
                rewards[t, e], dones[t, e] = rwd, done
                for k, o in enumerate(obs):
                    obs_out[k][t, e] = o
                if any(done):
                    ep += 1
                    start_episode(pool[e, (ep - 1) % 32], ep, t + 1)
            states[T, e] = env.state
        finally:
            np.random.uniform = saved_uniform
    out = dict(params=params, init_state=init, reset_pool=pool, actions=actions.astype(np.float32), draws=draws,
               states=states, goals=goals, first_goal=first_goal, rewards=rewards, dones=dones, reset_at=reset_at,
               episode_of=episode_of)
    for k in range(len(obs_dims)):
        out[f"obs{k}"] = obs_out[k]; out[f"first_obs{k}"] = first_obs[k]
    np.savez_compressed(os.path.join(OUT, f"trajgoal_m{mode}_{kind}.npz"), **out)
    print(f"trajgoal_m{mode}_{kind}: resets/env={reset_at.sum(0).tolist()} max|Wd3|={np.abs(goals[..., 14]).max():.3f}")


def gen_gae(T=64, M=48, seed=0):
    """Run the reference's OWN GAE lines (algos/ppo/ppo.py: from `td_errors = ...` to the advantage
    normalisation) on synthetic critic outputs: the source text is read from /root/reference at
    generation time and executed here; only its inputs and outputs are stored."""
    import copy, textwrap, types
    import torch
    src = open(os.path.join(REF, "algos", "ppo", "ppo.py")).read().split("\n")
    i0 = next(i for i, l in enumerate(src) if "td_errors = batch_rwd" in l)
    i1 = next(i for i, l in enumerate(src) if "advantages = (advantages - advantages.mean())" in l)
    block = textwrap.dedent("\n".join(src[i0:i1 + 1]))
    rng = np.random.default_rng(4000 + seed)
    out = {}
    for name, (gamma, lam) in {"a": (0.99, 0.9), "b": (0.95, 0.97)}.items():
        rwd = rng.uniform(-1, 1, (T, M)).astype(np.float32)
        rwd[rng.uniform(size=(T, M)) < 0.05] = -1.0
        done = (rng.uniform(size=(T, M)) < 0.06)
        val = rng.normal(0, 2, (T + 1, M)).astype(np.float32)
        adv = np.zeros((T, M), np.float32); tgt = np.zeros((T, M), np.float32); nrm = np.zeros((T, M), np.float32)
        for c in range(M):  # the reference processes ONE env's horizon per call
            ns = {"torch": torch, "copy": copy, "self": types.SimpleNamespace(discount=gamma, GAE_lambda=lam, device="cpu"),
                  "batch_rwd": torch.tensor(rwd[:, c:c + 1]), "next_V": torch.tensor(val[1:, c:c + 1]),
                  "batch_done": torch.tensor(done[:, c:c + 1], dtype=torch.float), "current_V": torch.tensor(val[:-1, c:c + 1])}
            exec(block, ns)
            nrm[:, c] = ns["advantages"].numpy()[:, 0]
            tgt[:, c] = ns["td_targets"].numpy()[:, 0]
            adv[:, c] = tgt[:, c] - val[:-1, c]
        out.update({f"{name}_reward": rwd, f"{name}_done": done, f"{name}_value": val, f"{name}_gamma_lam": np.array([gamma, lam]),
                    f"{name}_td_target": tgt, f"{name}_advantage": adv, f"{name}_normalized_per_column": nrm})
    np.savez_compressed(os.path.join(OUT, "gae.npz"), **out)
    print("gae golden written")


ACTOR_DIMS = {"coupled": [(23, 16, 4)], "decoupled": [(15, 16, 4), (3, 4, 1)]}  # main.py:68-73, args_parse.py:40


def make_actors(kind, seed):
    """The reference's own actor modules (algos/ppo/ppo_mlp.py), default init, then log_std and the
    mean bias moved off their zero defaults so that every term is exercised."""
    import types
    import torch
    from algos.ppo.ppo_mlp import MLP_Actor_PPO
    dims = ACTOR_DIMS[kind]
    args = types.SimpleNamespace(obs_dim_n=[d[0] for d in dims], actor_hidden_dim=[d[1] for d in dims],
                                 action_dim_n=[d[2] for d in dims])
    torch.manual_seed(seed)
    actors = []
    for k in range(len(dims)):
        a = MLP_Actor_PPO(args, k)
        with torch.no_grad():
            a.log_std.copy_(torch.linspace(-1.6, -0.9, dims[k][2]).reshape(1, -1))
            a.mean_linear.bias.uniform_(-0.2, 0.2)
            a.mean_linear.weight.mul_(8.0)  # default x0.1 keeps tanh in its linear part; reach |mean| ~ 0.8 too
        actors.append(a)
    return actors


def actor_weights(a):
    return {"fc1_w": a.fc1.weight, "fc1_b": a.fc1.bias, "fc2_w": a.fc2.weight, "fc2_b": a.fc2.bias,
            "mean_w": a.mean_linear.weight, "mean_b": a.mean_linear.bias, "log_std": a.log_std.reshape(-1)}


def ref_choose_action(actor, obs, eps, max_action=1.0):
    """PPO.choose_action (ppo.py:93-99) with the noise made explicit: Normal.sample() is
    mean + std * eps; everything else is the reference's module."""
    import torch
    with torch.no_grad():
        dist = actor.get_dist(torch.as_tensor(obs, dtype=torch.float32))
        action = torch.clamp(dist.mean + dist.stddev * torch.as_tensor(eps, dtype=torch.float32), -max_action, max_action)
        return action.numpy(), dist.log_prob(action).numpy(), dist.mean.numpy()


def gen_actor(seed=0, n=256):
    rng = np.random.default_rng(6000 + seed)
    out = {}
    for kind in ("coupled", "decoupled"):
        for k, a in enumerate(make_actors(kind, 100 + seed)):
            D, H, A = ACTOR_DIMS[kind][k]
            obs = rng.uniform(-1.5, 1.5, (n, D)).astype(np.float32)
            eps = rng.standard_normal((n, A)).astype(np.float32)
            eps[: n // 8] *= 4.0  # push some samples into the clamp
            action, logprob, mean = ref_choose_action(a, obs, eps)
            tag = f"{kind}{k}"
            for name, w in actor_weights(a).items():
                out[f"{tag}_{name}"] = w.detach().numpy().copy()
            out.update({f"{tag}_obs": obs, f"{tag}_eps": eps, f"{tag}_mean": mean, f"{tag}_action": action, f"{tag}_logprob": logprob})
            print(f"actor {tag}: |mean| max {np.abs(mean).max():.3f}, clamped {np.mean(np.abs(action) == 1.0):.3f}")
    np.savez_compressed(os.path.join(OUT, "actor_ppo.npz"), **out)


def gen_actor_td3(seed=0, n=256, sigma=0.1):
    """The reference's TD3 actor (algos/td3/td3_mlp.py) and TD3.choose_action (td3.py:93-96) with the
    exploration noise made explicit: clip(actor(obs) + sigma * eps)."""
    import types
    import torch
    from algos.td3.td3_mlp import MLP_Actor_TD3
    rng = np.random.default_rng(6500 + seed)
    out = {"sigma": np.float32(sigma)}
    for kind in ("coupled", "decoupled"):
        dims = ACTOR_DIMS[kind]
        args = types.SimpleNamespace(obs_dim_n=[d[0] for d in dims], actor_hidden_dim=[d[1] for d in dims], action_dim_n=[d[2] for d in dims])
        torch.manual_seed(300 + seed)
        for k, (D, H, A) in enumerate(dims):
            a = MLP_Actor_TD3(args, k)
            obs = rng.uniform(-1.5, 1.5, (n, D)).astype(np.float32)
            eps = rng.standard_normal((n, A)).astype(np.float32)
            with torch.no_grad():
                mean = a(torch.from_numpy(obs)).numpy()
            action = np.clip(mean + np.float32(sigma) * eps, -1.0, 1.0).astype(np.float32)  # td3.py:95-96 with explicit noise
            tag = f"{kind}{k}"
            out.update({f"{tag}_fc1_w": a.fc1.weight.detach().numpy().copy(), f"{tag}_fc1_b": a.fc1.bias.detach().numpy().copy(),
                        f"{tag}_fc2_w": a.fc2.weight.detach().numpy().copy(), f"{tag}_fc2_b": a.fc2.bias.detach().numpy().copy(),
                        f"{tag}_fc3_w": a.fc3.weight.detach().numpy().copy(), f"{tag}_fc3_b": a.fc3.bias.detach().numpy().copy(),
                        f"{tag}_obs": obs, f"{tag}_eps": eps, f"{tag}_mean": mean, f"{tag}_action": action})
    np.savez_compressed(os.path.join(OUT, "actor_td3.npz"), **out)
    print("actor_td3 golden written")


def gen_actor_sac(seed=0, n=256):
    """The reference's SAC actor (algos/sac/sac_mlp.py): forward() for mean / log_std, and sample() with the
    reparameterisation noise made explicit (x_t = mean + std * eps), per-component log-probs before the sum."""
    import types
    import torch
    from torch.distributions import Normal
    from algos.sac.sac_mlp import MLP_Actor_SAC, epsilon
    rng = np.random.default_rng(6800 + seed)
    out = {}
    for kind in ("coupled", "decoupled"):
        dims = ACTOR_DIMS[kind]
        args = types.SimpleNamespace(obs_dim_n=[d[0] for d in dims], actor_hidden_dim=[d[1] for d in dims], action_dim_n=[d[2] for d in dims])
        torch.manual_seed(400 + seed)
        for k, (D, H, A) in enumerate(dims):
            a = MLP_Actor_SAC(args, k)
            with torch.no_grad():   # xavier init with zero bias: move the biases so that every term is exercised
                a.mean_linear.bias.uniform_(-0.2, 0.2); a.log_std_linear.bias.uniform_(-2.0, -0.5)
                a.fc1.bias.uniform_(-0.2, 0.2); a.fc2.bias.uniform_(-0.2, 0.2)
            obs = rng.uniform(-1.5, 1.5, (n, D)).astype(np.float32)
            obs[: n // 16] *= 40.0   # a few rows far out: log_std hits its clamp
            eps = rng.standard_normal((n, A)).astype(np.float32)
            with torch.no_grad():
                mean, log_std = a(torch.from_numpy(obs))
                normal = Normal(mean, log_std.exp())
                x_t = mean + log_std.exp() * torch.from_numpy(eps)
                action = torch.tanh(x_t)
                logp = normal.log_prob(x_t) - torch.log((1 - action.pow(2)) + epsilon)
            tag = f"{kind}{k}"
            out.update({f"{tag}_fc1_w": a.fc1.weight.detach().numpy().copy(), f"{tag}_fc1_b": a.fc1.bias.detach().numpy().copy(),
                        f"{tag}_fc2_w": a.fc2.weight.detach().numpy().copy(), f"{tag}_fc2_b": a.fc2.bias.detach().numpy().copy(),
                        f"{tag}_mean_w": a.mean_linear.weight.detach().numpy().copy(), f"{tag}_mean_b": a.mean_linear.bias.detach().numpy().copy(),
                        f"{tag}_log_std_w": a.log_std_linear.weight.detach().numpy().copy(),
                        f"{tag}_log_std_b": a.log_std_linear.bias.detach().numpy().copy(),
                        f"{tag}_obs": obs, f"{tag}_eps": eps, f"{tag}_mean": mean.numpy(), f"{tag}_log_std": log_std.numpy(),
                        f"{tag}_action": action.numpy(), f"{tag}_logprob": logp.numpy()})
            print(f"actor_sac {tag}: log_std range {float(log_std.min()):.2f}..{float(log_std.max()):.2f}, |action| max {float(action.abs().max()):.4f}")
    np.savez_compressed(os.path.join(OUT, "actor_sac.npz"), **out)


def gen_actorloop(kind, n_env=4, T=200, seed=0):
    """The collection loop of main.py:141-166 with the reference's env AND the reference's actor(s):
    obs_t -> choose_action (injected noise) -> concatenate -> env.step -> obs_{t+1}.  Free run."""
    rng = np.random.default_rng(7000 + seed + 13 * orc.KINDS.index(kind))
    dims = ACTOR_DIMS[kind]
    A = sum(d[2] for d in dims)
    nag = len(dims)
    actors = make_actors(kind, 200 + seed)
    params = f32r(random_params(rng, n_env)); params[0] = orc.NOMINAL_PARAMS
    params = f32r(params)
    goal = np.tile(orc.DEFAULT_GOAL, (n_env, 1))
    init = state_in(orc.sample_reset_state(rng, n_env, "train"))
    eps = rng.standard_normal((T, n_env, A)).astype(np.float32)
    states = np.zeros((T + 1, n_env, 18)); integs = np.zeros((T + 1, n_env, 8))
    obs_out = [np.zeros((T + 1, n_env, d[0]), np.float32) for d in dims]
    actions = np.zeros((T, n_env, A), np.float32); logprobs = np.zeros((T, n_env, A), np.float32)
    rewards = np.zeros((T, n_env, nag)); dones = np.zeros((T, n_env, nag), bool)
    for e in range(n_env):
        env = make_env(kind)
        inject_params(env, params[e])
        inject(env, init[e], goal[e], np.zeros(8))
        obs = env.get_norm_error_state(env.framework)  # first obs after reset (main.py:129)
        for t in range(T):
            states[t, e], integs[t, e] = env.state, read_integ(env)
            col = 0
            for k, a in enumerate(actors):
                obs_out[k][t, e] = obs[k]
                act, lp, _ = ref_choose_action(a, np.asarray(obs[k])[None], eps[t, e, col:col + dims[k][2]][None])
                actions[t, e, col:col + dims[k][2]], logprobs[t, e, col:col + dims[k][2]] = act[0], lp[0]
                col += dims[k][2]
            _, _, _, obs, _, rwd, done = ref_step(env, kind, actions[t, e].astype(np.float64))
            rewards[t, e], dones[t, e] = rwd, done
        states[T, e], integs[T, e] = env.state, read_integ(env)
        for k in range(nag):
            obs_out[k][T, e] = obs[k]
    out = dict(params=params, goal=goal, init_state=init, eps=eps, states=states, integs=integs, actions=actions,
               logprobs=logprobs, rewards=rewards, dones=dones)
    for k, a in enumerate(actors):
        out[f"obs{k}"] = obs_out[k]
        for name, w in actor_weights(a).items():
            out[f"actor{k}_{name}"] = w.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"actorloop_{kind}.npz"), **out)
    print(f"actorloop_{kind}: first done at {[int(np.argmax(dones[:, e].any(-1))) if dones[:, e].any() else -1 for e in range(n_env)]}, "
          f"max|x| {np.abs(states[..., 0:3]).max():.2f}, |action| mean {np.abs(actions).mean():.3f}")


# ------------------------------------------------------------------------------------------------------------
# f3: closed loop with the shipped TD3-EMLP actors
# ------------------------------------------------------------------------------------------------------------
def _td3_args(framework):
    sys.argv = ["x", "--framework", framework, "--test_model", "True"]
    import torch
    import args_parse
    args = args_parse.create_parser().parse_args()
    args.device = torch.device("cpu")
    return args


def _install_deterministic_tiebreak():
    """The reference orders representations of equal size by `hash(self) < hash(other)` (representation.py:171-188), and
    its hashes are `hash((type(self), self.G))` / `hash(repr(group))`: the addresses of type objects and salted string
    hashes, i.e. different in every interpreter.  The order decides which input slots a BiLinear layer's torch.randint
    picks (representation.py:374-376) — it is part of the function the checkpoint's weights belong to, and it is not
    in the checkpoint.  Here `hash` is replaced, inside those two modules only, by a deterministic salted hash, and the
    salt is chosen so that the actors reproduce the reference-owned flight log (build_shipped_actors)."""
    import zlib
    import algos.emlp_torch.reps.representation as rep_mod
    import algos.emlp_torch.groups as grp_mod
    salt = [0]

    def det_hash(o):
        if isinstance(o, tuple):
            s = "(" + ",".join(str(det_hash(x)) for x in o) + ")"
        elif isinstance(o, type):
            s = o.__module__ + "." + o.__qualname__
        elif isinstance(o, (int, np.integer)):
            return int(o)
        elif isinstance(o, str):
            s = o
        else:
            return o.__hash__()
        return zlib.crc32((str(salt[0]) + "|" + s).encode())

    rep_mod.hash = det_hash
    grp_mod.hash = det_hash
    return salt


def _flightlog_obs():
    """Observations (MODUL) rebuilt from the flight log: row i's action was computed from the observation returned by
    step i-1, i.e. from state i and the goal of row i-1 (quad.py:421-466 uses the goal set BEFORE the step); the
    integral terms and eb1 are logged directly (main.py:343-352, utils.py:21-39)."""
    log = np.loadtxt(os.path.join(REF, "results", "MODUL_log_20250303_120200.dat"))
    act, st, eIx, eb1, eIb1, cmd = log[:, 0:5], log[:, 5:23], log[:, 23:26], log[:, 26], log[:, 27], log[:, 28:40]
    x, v, W, b1, b2, b3 = st[:, 0:3], st[:, 3:6], st[:, 15:18], st[:, 6:9], st[:, 9:12], st[:, 12:15]
    prev = np.concatenate([cmd[:1], cmd[:-1]])
    ex, ev, eW = (x - prev[:, 0:3]) / 1.0, (v - prev[:, 3:6]) / 4.0, (W - prev[:, 9:12]) / (2 * np.pi)
    o1 = np.concatenate([ex, eIx / 3.0, ev, b3, eW[:, 0:1] * b1 + eW[:, 1:2] * b2], 1).astype(np.float32)
    o2 = np.stack([eb1 / np.pi, eIb1 / 3.0, eW[:, 2]], 1).astype(np.float32)
    return o1, o2, act


def build_shipped_actors():
    """The three shipped TD3-EMLP actors, constructed as main.py constructs them in test mode (set_seed(1992), then the agents in
    order: main.py:64,82-83; only actors: td3.py:35-47) and validated against the reference-owned flight log: fed the log's own
    observations, the MODUL pair must reproduce the logged action columns to 1e-6 on rows 1..3599 (row 0's observation
    was formed with a goal the log does not hold).  Returns ({'MODUL': [a0, a1], 'MONO': [a]}, report dict)."""
    import random
    import torch
    sys.path.insert(0, os.path.join(HERE, "_plum_shim"))
    args = _td3_args("MODUL")
    salt = _install_deterministic_tiebreak()
    from algos.td3.td3_emlp import EMLP_MODUL1_Actor_TD3, EMLP_MODUL2_Actor_TD3, EMLP_MONO_Actor_TD3
    # (weights_only: the checkpoints come from a public, untrusted tree — tensors are loaded, nothing is unpickled into code)
    sd = {k: torch.load(os.path.join(REF, "models", f), map_location="cpu", weights_only=True) for k, f in
          (("m0", "TD3_MODUL_564.0k_steps_agent_0_1992.pth"), ("m1", "TD3_MODUL_850.0k_steps_agent_1_1992.pth"),
           ("mono", "TD3_MONO_700.0k_steps_agent_0_1992.pth"))}
    o1, o2, act = _flightlog_obs()

    def seeded():
        random.seed(1992); np.random.seed(1992); torch.manual_seed(1992)

    report = {}
    for s_ in range(64):
        salt[0] = s_
        seeded()
        a0, a1 = EMLP_MODUL1_Actor_TD3(args, 0), EMLP_MODUL2_Actor_TD3(args, 1)
        a0.load_state_dict(sd["m0"]); a1.load_state_dict(sd["m1"])
        with torch.no_grad():
            pred = np.concatenate([a0(torch.from_numpy(o1)).numpy(), a1(torch.from_numpy(o2)).numpy()], 1)
        err = np.abs(pred - act)[1:]
        if err.max() <= 1e-6:
            report = {"tiebreak_salt": s_, "flightlog_action_max_err_rows_1_3599": float(err.max()),
                      "flightlog_action_err_row_0": np.abs(pred - act)[0].tolist()}
            break
    else:
        raise SystemExit("f3: no tie-break order reproduces the flight log's actions; last worst error %g" % err.max())
    # MONO: no reference-owned log.  Its representations hold no equal-size ties inside one group, so its function must
    # not depend on the tie-break at all: checked over salts on random observations.
    margs = _td3_args("MONO")
    outs = []
    probe = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (64, 23)).astype(np.float32))
    for s_ in (report["tiebreak_salt"], 0, 1, 2, 3, 4, 5):
        salt[0] = s_
        seeded()
        am = EMLP_MONO_Actor_TD3(margs, 0)
        am.load_state_dict(sd["mono"])
        with torch.no_grad():
            outs.append(am(probe).numpy())
    report["mono_max_spread_over_tiebreaks"] = float(max(np.abs(o - outs[0]).max() for o in outs))
    # the MONO actor has no independent log to be validated against: its only guard is that no tie-break can matter
    assert report["mono_max_spread_over_tiebreaks"] <= 1e-6, report
    salt[0] = report["tiebreak_salt"]
    seeded()
    am = EMLP_MONO_Actor_TD3(margs, 0)
    am.load_state_dict(sd["mono"])
    print("shipped actors:", report)
    return {"MODUL": [a0, a1], "MONO": [am]}, report


def gen_closedloop_td3(framework, actors, report, modes=((0, 600), (1, 1000), (6, 1800)), suffix="", init_x=None):
    """main.py's eval loop (:290-345) with the shipped actor(s): eval reset, mark_traj_start, per step get_desired(current
    state) -> set_goal_state -> actor(obs) -> step.  One env per mode; the generator's draws are injected and recorded."""
    import torch
    import utils.trajectory_generator as tg_mod
    kind = {"MODUL": "decoupled", "MONO": "coupled"}[framework]
    A, nag = orc.ACTION_DIM[kind], orc.N_AGENTS[kind]
    obs_dims = {"coupled": [23], "decoupled": [15, 3]}[kind]
    rng = np.random.default_rng(9100 + len(framework))
    out = {"params": f32r(orc.NOMINAL_PARAMS)[None], "modes": np.array([m for m, _ in modes]), "steps": np.array([T for _, T in modes]),
           "tiebreak_salt": np.array(report["tiebreak_salt"]),
           "flightlog_action_max_err": np.array(report["flightlog_action_max_err_rows_1_3599"])}
    saved_uniform = np.random.uniform
    for mode, T in modes:
        env = make_env(kind)
        sys.argv = ["x", "--framework", framework]
        gen = tg_mod.TrajectoryGenerator(env)
        draws = np.array([f32r(rng.uniform(-np.deg2rad(25), np.deg2rad(25))), f32r(rng.uniform(2.0, 5.0)),
                          f32r(rng.uniform(-0.15 * np.pi, 0.15 * np.pi))])
        queue = {0: [draws[0]], 1: [draws[1], draws[2]]}.get(mode, [])

        def fake_uniform(size=None, low=0.0, high=1.0):
            return np.array([queue.pop(0)])

        init = state_in(orc.sample_reset_state(rng, 1, "eval"))[0]       # eval reset: |x| <= 0.4, yaw only (quad.py:352-356)
        if init_x is not None and mode in init_x:                        # (modes 2-5: a start position from which every branch is flown)
            init[0:3] = f32r(init_x[mode])
        states = np.zeros((T + 1, 18)); goals = np.zeros((T, 15)); actions = np.zeros((T, A), np.float32)
        rewards = np.zeros((T, nag)); dones = np.zeros((T, nag), bool)
        obs_in = [np.zeros((T, d), np.float32) for d in obs_dims]; obs_out = [np.zeros((T, d), np.float32) for d in obs_dims]
        np.random.uniform = fake_uniform
        try:
            inject_params(env, out["params"][0])
            inject(env, init, orc.DEFAULT_GOAL, np.zeros(8))
            gen.mark_traj_start(env.state)
            xd, vd, b1d, b1d_dot, Wd = gen.get_desired(env.state, mode)
            env.set_goal_state(xd, vd, b1d, b1d_dot, Wd)
            first_goal = np.concatenate([xd, vd, b1d, b1d_dot, Wd])
            obs = env.get_norm_error_state(framework)
            first_obs = [np.asarray(o, np.float32).copy() for o in obs]
            n_done = T
            for t in range(T):
                states[t] = env.state
                xd, vd, b1d, b1d_dot, Wd = gen.get_desired(env.state, mode)
                env.set_goal_state(np.copy(xd), np.copy(vd), np.copy(b1d), np.copy(b1d_dot), np.copy(Wd))
                goals[t] = np.concatenate([xd, vd, b1d, b1d_dot, Wd])
                with torch.no_grad():   # TD3.choose_action with explor_noise_std = 0 (td3.py:82-96, main.py:335)
                    act = np.concatenate([a(torch.tensor(np.asarray(o), dtype=torch.float)[None]).numpy().flatten().clip(-1.0, 1.0)
                                          for a, o in zip(actors, obs)])
                for k, o in enumerate(obs):
                    obs_in[k][t] = o
                actions[t] = act
                _, _, _, obs, _, rwd, done = ref_step(env, kind, actions[t].astype(np.float64))
                rewards[t], dones[t] = rwd, done
                for k, o in enumerate(obs):
                    obs_out[k][t] = o
                if any(done):
                    n_done = t + 1
                    break
            states[n_done] = env.state
        finally:
            np.random.uniform = saved_uniform
        assert n_done == T, f"{framework} mode {mode}: the shipped policy lost the vehicle at step {n_done}"
        tag = f"m{mode}_"
        out.update({tag + "init_state": init, tag + "draws": draws, tag + "first_goal": first_goal, tag + "states": states,
                    tag + "goals": goals, tag + "actions": actions, tag + "rewards": rewards, tag + "dones": dones})
        for k in range(len(obs_dims)):
            out[tag + f"first_obs{k}"] = first_obs[k]; out[tag + f"obs_in{k}"] = obs_in[k]; out[tag + f"obs{k}"] = obs_out[k]
        ex = np.abs(states[-200:, 0:3] - goals[-200:, 0:3].mean(0)).max() if mode in (0, 1) else np.abs(states[:T, 0:3] - goals[:, 0:3]).max()
        print(f"closedloop_td3_{framework.lower()} mode {mode}: {T} steps, max|x - xd| {'over the flight' if mode == 6 else 'last 200 steps'} {ex:.3f} m, "
              f"|x| max {np.abs(states[:, 0:3]).max():.2f}, reward mean {rewards.mean(0)}")
    np.savez_compressed(os.path.join(OUT, f"closedloop_td3_{framework.lower()}{suffix}.npz"), **out)


def gen_errobs_formats(n=192, seed=0):
    """get_norm_error_state(framework) with EITHER framework on each wrapper class (quad.py:421-466: the argument, not the class,
    selects the format) — the same injected state / goal / integrator terms through CoupledWrapper and DecoupledWrapper, asked for
    MONO and for MODUL: observation rows and the advanced integrator terms.  A bare QuadEnv raises AttributeError here (it has no
    alpha / beta / eIx_lim): recorded as `quad_raises`."""
    rng = np.random.default_rng(4100 + seed)
    state = state_in(np.concatenate([orc.sample_reset_state(rng, n - n // 4, "train"), boundary_states(rng, n // 4)]))
    goal = np.tile(orc.DEFAULT_GOAL, (n, 1)); goal[n // 3:] = random_goal(rng, n)[n // 3:]
    integ = np.zeros((n, 8))
    integ[:, 0:3] = rng.uniform(-1.0, 1.0, (n, 3)); integ[::5, 0:3] = rng.uniform(-4.0, 4.0, (len(integ[::5]), 3))
    integ[:, 3:6] = rng.uniform(-1.0, 1.0, (n, 3))
    integ[:, 6] = rng.uniform(-2.0, 2.0, n); integ[::9, 6] = rng.uniform(-4.0, 4.0, len(integ[::9]))
    integ[:, 7] = rng.uniform(-3.0, 3.0, n)
    integ = f32r(integ)
    out = dict(state=state, goal=goal, integ=integ)
    for kind in ("coupled", "decoupled"):
        env = make_env(kind)
        for fw, dims in (("MONO", [23]), ("MODUL", [15, 3])):
            obs = [np.zeros((n, d), np.float32) for d in dims]
            nxt = np.zeros((n, 8))
            for i in range(n):
                inject(env, state[i], goal[i], integ[i])
                o = env.get_norm_error_state(fw)
                assert [x.dtype for x in o] == [np.float32] * len(dims)
                for k in range(len(dims)):
                    obs[k][i] = o[k]
                nxt[i] = read_integ(env)
            for k in range(len(dims)):
                out[f"{kind}_{fw}_obs{k}"] = obs[k]
            out[f"{kind}_{fw}_next_integ"] = nxt
    # the class does not matter: both wrappers inherit the method unchanged
    for fw, nk in (("MONO", 1), ("MODUL", 2)):
        for k in range(nk):
            assert np.array_equal(out[f"coupled_{fw}_obs{k}"], out[f"decoupled_{fw}_obs{k}"])
    sys.argv = ["x", "--framework", "MONO"]
    bare = QuadEnv(); bare.reset(env_type="eval")
    try:
        bare.get_norm_error_state("MONO")
        raises = ""
    except AttributeError as ex:
        raises = type(ex).__name__
    out["quad_raises"] = np.array([ord(ch) for ch in raises], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "errobs_formats.npz"), **out)
    print(f"errobs_formats: n={n} bare QuadEnv raises {raises!r}")


def gen_flightlog(rows=3600):
    log = np.loadtxt(os.path.join(REF, "results", "MODUL_log_20250303_120200.dat"))
    np.savez_compressed(os.path.join(OUT, "flightlog_modul.npz"), log=log[:rows])
    print("flightlog rows", rows, "of", log.shape)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if ARGV[:1] == ["td3"]:  # f3: the shipped TD3-EMLP actors in the loop (needs tools/_plum_shim)
        shipped, rep = build_shipped_actors()
        for fw in ("MODUL", "MONO"):
            gen_closedloop_td3(fw, shipped[fw], rep)
        sys.exit(0)
    if ARGV[:1] == ["td3modes"]:  # TrajectoryGenerator modes 2-5 (take-off, landing, stay, circle) flown by the shipped actors
        # start positions chosen so that every branch of the generator is reached inside the arena (|x| < 1): take-off from 0.2 m
        # below the take-off height (t_traj = 4 s, then the way-point test, then manual mode); landing from 0.35 m above the motor
        # cut-off height (ramp 0.1 s, then the descent branch until x3 > -0.25, then "landed"); circle centred 0.1 m off the origin
        # (1.75 s run-up, two circles of 0.7 m radius: 33.2 s, then manual mode)
        shipped, rep = build_shipped_actors()
        init_x = {2: [0.1, -0.15, -0.3], 3: [-0.2, 0.1, -0.6], 4: [0.3, 0.2, -0.1], 5: [-0.1, 0.05, -0.2]}
        for fw in ("MODUL", "MONO"):
            gen_closedloop_td3(fw, shipped[fw], rep, modes=((2, 1300), (3, 500), (4, 300), (5, 6900)), suffix="_modes2345", init_x=init_x)
        sys.exit(0)
    if ARGV[:1] == ["errobs"]:  # only get_norm_error_state(framework) in both formats on both wrapper classes
        gen_errobs_formats()
        sys.exit(0)
    if ARGV[:1] == ["actor"]:  # only the actor files
        gen_actor()
        gen_actor_td3()
        gen_actor_sac()
        for kind in ("coupled", "decoupled"):
            gen_actorloop(kind)
        sys.exit(0)
    for kind in ("coupled", "decoupled"):
        check_step_template(kind)
    gen_kats()
    gen_flightlog()
    gen_gae()
    for kind in orc.KINDS:
        gen_onestep(kind)
    for kind in orc.KINDS:
        gen_traj(kind, "free")
        gen_traj(kind, "reset")
    for kind in ("coupled", "decoupled"):
        for mode in (0, 1, 6):
            gen_trajgoal(kind, mode)
    gen_errobs_formats()
    gen_actor()
    gen_actor_td3()
    gen_actor_sac()
    for kind in ("coupled", "decoupled"):
        gen_actorloop(kind)
