"""Goal generation: TrajectoryGenerator modes 0/1 — oracle pinned on the
reference (CPU), then the HIP path (standalone qr_get_desired and fused into qr_step) against
the same closed-loop golden runs (GPU)."""
import numpy as np
import pytest
import torch

from conftest import grouped_rel_err
from oracle import quad_oracle as orc
from oracle import traj_oracle as trj

def _np(t):
    return t.detach().cpu().numpy()


CASES = [(k, m) for k in ("coupled", "decoupled") for m in (0, 1, 6)]


def _draws(d, e, ep):
    return d["draws"][e, ep]


@pytest.mark.parametrize("kind,mode", CASES)
def test_traj_oracle_matches_reference(kind, mode, golden):
    d = golden(f"trajgoal_m{mode}_{kind}")
    T, n = d["actions"].shape[:2]
    for e in range(n):
        tr = None
        for t in range(T):
            if d["reset_at"][t, e]:
                ep = d["episode_of"][t, e]
                th, tt, w = _draws(d, e, ep)
                tr = trj.traj_start_batch(d["states"][t, e], mode, theta_b1d=th, t_traj=tt, w_b1d=w)
                g = np.concatenate(trj.get_desired_batch(tr, d["states"][t, e]), 1)[0]
                assert np.abs(g - d["first_goal"][t, e]).max() <= 1e-13
            g = np.concatenate(trj.get_desired_batch(tr, d["states"][t, e]), 1)[0]
            assert np.abs(g - d["goals"][t, e]).max() <= 1e-12


@pytest.mark.parametrize("kind,mode", CASES)
def test_oracle_closed_loop_with_goals(kind, mode, golden):
    """quad_oracle.step_batch fed with the golden goals reproduces the closed-loop run."""
    d = golden(f"trajgoal_m{mode}_{kind}")
    T, n = d["actions"].shape[:2]
    state, integ = d["init_state"].copy(), np.zeros((n, 8))
    for t in range(T):
        ra = d["reset_at"][t]
        if ra.any():
            state[ra] = d["states"][t][ra]; integ[ra] = 0.0
            g = d["first_goal"][t][ra]
            goal12 = np.concatenate([g[:, 0:9], g[:, 12:15]], 1)
            o = _first_obs(kind, state[ra], goal12, integ[ra])
            integ[ra] = o
        g = d["goals"][t]
        goal12 = np.concatenate([g[:, 0:9], g[:, 12:15]], 1)
        out = orc.step_batch(kind, state, d["actions"][t].astype(np.float64), d["params"], goal12, integ)
        state, integ = out["state"], out["integ"]
        for k, ob in enumerate(out["obs"]):
            assert np.abs(ob.astype(np.float64) - d[f"obs{k}"][t]).max() <= 2e-7
        assert np.array_equal(out["done"], d["dones"][t])
        nxt = d["reset_at"][t + 1]
        assert grouped_rel_err(state[~nxt], d["states"][t + 1][~nxt]) <= 1e-10 if (~nxt).any() else True


def _first_obs(kind, state, goal12, integ):
    from test_oracle_golden import _advance
    return _advance(kind, state, goal12, integ)


def _log_goal_inputs(golden):
    """The reference-owned flight log is an eight-shaped-curve flight (mode 6) recorded with
    eight_T = 18 s (HEAD has 9 s): its command columns (xd, vd, b1c, Wd) are a pure function of its
    state columns.  Row i was produced by the (i+2)-th get_desired call (main.py:310-331)."""
    log = golden("flightlog_modul")["log"]
    state, cmd = log[:, 5:23], log[:, 28:40]
    return state, cmd


def test_flightlog_commands_oracle(golden):
    state, cmd = _log_goal_inputs(golden)
    tr = trj.traj_start_batch(state[0], 6)
    trj.get_desired_batch(tr, state[0], eight={"T": 18.0})            # call 1: first obs after reset
    for i in range(len(state)):
        xd, vd, b1d, b1d_dot, Wd = trj.get_desired_batch(tr, state[i], eight={"T": 18.0})
        b3 = state[i, 12:15]
        b1c = b1d[0] - np.dot(b1d[0], b3) * b3                        # what main.py:349-351 logs
        # the generator was started from the float32 state reset() returns: centre/theta_init ~1e-7 off
        assert np.abs(xd[0] - cmd[i, 0:3]).max() <= 3e-7 and np.abs(vd[0] - cmd[i, 3:6]).max() <= 1e-7
        assert np.abs(b1c - cmd[i, 6:9]).max() <= 2e-7 and np.abs(Wd[0] - cmd[i, 9:12]).max() <= 2e-8


@pytest.mark.gpu
def test_flightlog_commands_gpu(golden):
    """Device goal generator (mode 6, eight_T = 18) on the logged states vs the logged commands."""
    from gym_rotor_amd import QuadConstants, QuadVecEnv
    state, cmd = _log_goal_inputs(golden)
    n = 1
    env = QuadVecEnv("decoupled", n, device="cuda", goal_mode=6, layout="f64", use_UDM=False,
                     constants=QuadConstants(eight_T=18.0))
    env.set_state(state[0:1])
    env.mark_traj_start()
    env.get_desired()
    worst = np.zeros(4)
    for i in range(400):
        env.set_state(state[i:i + 1])
        xd, vd, b1d, _, Wd = (_np(g)[0].astype(np.float64) for g in env.get_desired())
        b3 = state[i, 12:15]
        b1c = b1d - np.dot(b1d, b3) * b3
        for k, (a, b) in enumerate(((xd, cmd[i, 0:3]), (vd, cmd[i, 3:6]), (b1c, cmd[i, 6:9]), (Wd, cmd[i, 9:12]))):
            worst[k] = max(worst[k], np.abs(a - b).max())
    print("flight-log commands (xd, vd, b1c, Wd):", worst)
    assert (worst <= [1e-6, 1e-6, 1e-6, 1e-6]).all()


@pytest.mark.gpu
def test_flightlog_closed_loop_replay(golden):
    """The whole reference-owned flight (3599 steps = 18 s) on the device: start from the first
    logged state, feed the logged actions, let the fused goal generator (mode 6, eight_T = 18) and
    the integrators run, and compare with the log all the way.
    The state evolves open-loop in the actions, so the log's own print rounding (1e-10 per step) is
    amplified by the unstable double integrator (~x4 per 600 steps): the float64 DOP853 oracle
    replays the log with the same 2.7e-8 after 1200 steps and 7.5e-6 after 3599 — the bounds below."""
    from gym_rotor_amd import QuadConstants, QuadVecEnv
    log = golden("flightlog_modul")["log"]
    act, state = log[:, 0:5], log[:, 5:23]
    env = QuadVecEnv("decoupled", 1, device="cuda", goal_mode=6, layout="f64", use_UDM=False,
                     constants=QuadConstants(eight_T=18.0))
    env.set_state(state[0:1], integ=np.zeros((1, 8)))
    env.mark_traj_start()
    env.get_desired(store_goal=True)
    env.get_norm_error_state()                     # first obs after reset (main.py:312-314)
    a = torch.from_numpy(act.astype(np.float32)).cuda()
    worst_s = worst_i = worst_b = 0.0
    for t in range(len(log) - 1):
        (o1, o2), _, done, _, _ = env.step(a[t:t + 1])
        assert not bool(done.any())
        got = _np(env.get_current_state())[0]
        worst_s = max(worst_s, np.abs(got - state[t + 1]).max())
        eIx = _np(o1)[0, 3:6].astype(np.float64) * 3.0
        eb1, eIb1 = float(o2[0, 0]) * np.pi, float(o2[0, 1]) * 3.0
        worst_i = max(worst_i, np.abs(eIx - log[t + 1, 23:26]).max(), abs(eIb1 - log[t + 1, 27]))
        worst_b = max(worst_b, abs(eb1 - log[t + 1, 26]))
        if t == 1198:
            print(f"closed-loop flight replay, 1199 steps: state {worst_s:.2e} integral terms {worst_i:.2e} eb1 {worst_b:.2e}")
            assert worst_s <= 1e-7 and worst_i <= 2e-6 and worst_b <= 2e-6
    print(f"closed-loop flight replay, 3599 steps: state {worst_s:.2e} integral terms {worst_i:.2e} eb1 {worst_b:.2e}")
    assert worst_s <= 2e-5 and worst_i <= 3e-5 and worst_b <= 1e-5


# ---------------------------------------------------------------- GPU ----------------
def _mk(kind, n, mode, **kw):
    from gym_rotor_amd import QuadVecEnv
    return QuadVecEnv(kind, n, device="cuda", goal_mode=mode, layout="f64", **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,mode", CASES)
def test_get_desired_standalone(kind, mode, golden):
    """qr_traj_start + qr_get_desired on the golden states: (xd, vd, b1d, b1d_dot, Wd) per call."""
    d = golden(f"trajgoal_m{mode}_{kind}")
    T, n = d["actions"].shape[:2]
    env = _mk(kind, n, mode)
    worst = 0.0
    for t in range(T):
        env.set_state(d["states"][t])
        ra = d["reset_at"][t]
        if ra.any():
            ep = d["episode_of"][t]
            dr = np.stack([d["draws"][e, ep[e]] for e in range(n)])
            m = torch.from_numpy(ra).cuda()
            env.mark_traj_start(mask=m, theta_b1d=dr[:, 0], t_traj=dr[:, 1], w_b1d=dr[:, 2])
            g = torch.cat(env.get_desired(mask=m), 1)
            assert np.abs(_np(g)[ra] - d["first_goal"][t][ra]).max() <= 3e-6
        g = _np(torch.cat(env.get_desired(), 1))
        err = np.abs(g - d["goals"][t]) / np.maximum(np.abs(d["goals"][t]), 1.0)
        worst = max(worst, err.max())
    print(f"get_desired {kind} mode {mode}: worst {worst:.2e}")
    assert worst <= 3e-6


@pytest.mark.gpu
@pytest.mark.parametrize("kind,mode", CASES)
def test_fused_goal_closed_loop(kind, mode, golden):
    """qr_step with the goal generator fused (goal_mode) against the reference closed loop:
    obs / reward / done / state of every step, first observation of every episode."""
    d = golden(f"trajgoal_m{mode}_{kind}")
    T, n = d["actions"].shape[:2]
    env = _mk(kind, n, mode)
    env.set_state(d["init_state"], integ=np.zeros((n, 8)), params=d["params"])
    acts = torch.from_numpy(d["actions"]).cuda()
    worst = worst_obs = 0.0
    ndiff = 0
    for t in range(T):
        ra = d["reset_at"][t]
        if ra.any():
            ep = d["episode_of"][t]
            dr = np.stack([d["draws"][e, ep[e]] for e in range(n)])
            m = torch.from_numpy(ra).cuda()
            if t > 0:
                env.set_state(d["states"][t], mask=m)
                env._integ[:, m] = 0.0
            env.mark_traj_start(mask=m, theta_b1d=dr[:, 0], t_traj=dr[:, 1], w_b1d=dr[:, 2])
            env.get_desired(store_goal=True, mask=m)          # main.py:227-229
            keep = env._integ.clone()
            first = env.get_norm_error_state()
            env._integ.copy_(torch.where(m[None, :], env._integ, keep))
            for k, o in enumerate(first):
                assert np.abs(_np(o)[ra] - d[f"first_obs{k}"][t][ra]).max() <= 3e-6
        obs, rwd, done, _, _ = env.step(acts[t])
        obs = [obs] if isinstance(obs, torch.Tensor) else list(obs)
        for k, o in enumerate(obs):
            worst_obs = max(worst_obs, float(np.abs(_np(o).astype(np.float64) - d[f"obs{k}"][t]).max()))
        dd = _np(done) != d["dones"][t]
        ndiff += int(dd.sum())
        assert np.abs(_np(rwd).astype(np.float64) - d["rewards"][t])[~dd].max() <= 1e-5
        got = _np(env.get_current_state())
        nxt = d["reset_at"][t + 1]
        if (~nxt).any():
            worst = max(worst, grouped_rel_err(got[~nxt], d["states"][t + 1][~nxt]))
    print(f"fused goal {kind} mode {mode}: state {worst:.2e} obs {worst_obs:.2e} done-diffs {ndiff}")
    assert worst <= 1e-6 and worst_obs <= 5e-6 and ndiff == 0


@pytest.mark.gpu
def test_fused_goal_auto_reset_runs_and_matches_manual():
    """auto_reset + goal_mode: the in-launch episode start (reset draw -> mark_traj_start ->
    first get_desired -> first obs) equals doing the same by hand from the post-reset state."""
    n = 2048
    for mode in (0, 1, 2, 3, 4, 5):
        env = _mk("coupled", n, mode, seed=4, auto_reset=True)
        env.reset("train")
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        hits = 0
        for _ in range(80):
            obs, rwd, done, _, _ = env.step(torch.rand(n, 4, device="cuda", generator=g) * 2 - 1)
            hit = _np(done[:, 0])
            if hit.any():
                hits += hit.sum()
                chk = _mk("coupled", n, mode, seed=0)
                chk.load_state_dict(env.state_dict())         # state, params, traj state after the reset
                if mode >= 2:                                 # the stateful modes (no draws): mark_traj_start again — flags, persistent goal —
                    chk.mark_traj_start(mask=torch.from_numpy(hit).cuda())   # and redo the episode's first call from there
                else:
                    chk._traj[0].fill_(0.0)                   # calls = 0: redo the episode's first call (same draws)
                chk._integ.zero_()
                chk.get_desired(store_goal=True)
                first = chk.get_norm_error_state()[0]
                assert np.abs(_np(first)[hit] - _np(obs)[hit]).max() <= 1e-6
                assert (_np(env._traj[0])[hit] == 1.0).all()
        assert hits > 0 and torch.isfinite(env.get_current_state()).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 6, 2, 3, 4, 5])
def test_rollout_equals_steps_with_fused_goal(mode):
    """K-step rollout with the goal generator fused is bit-identical to K single steps — also in the generator's stateful modes
    (2-5), whose persistent fields then live in registers across the steps instead of going through the goal buffer."""
    from gym_rotor_amd import QuadVecEnv
    n, T = 900, 9 if mode in (0, 1) else 70
    envs = [QuadVecEnv("decoupled", n, device="cuda", goal_mode=mode, seed=8, auto_reset=True) for _ in range(2)]
    for e in envs:
        e.reset("train")
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    acts = torch.rand(T, n, 5, device="cuda", generator=g) * 2 - 1
    outs = []
    for t in range(T):
        (o1, o2), r, d, _, _ = envs[0].step(acts[t])
        outs.append((o1.clone(), o2.clone(), r.clone(), d.clone()))
    ro = envs[1].rollout(acts)
    for t in range(T):
        assert torch.equal(outs[t][0], ro["obs0"][t]) and torch.equal(outs[t][1], ro["obs1"][t])
        assert torch.equal(outs[t][2], ro["reward"][t]) and torch.equal(outs[t][3], ro["terminated"][t])
    assert torch.equal(envs[0].get_current_state(), envs[1].get_current_state())
    assert torch.equal(envs[0]._traj, envs[1]._traj)
    if mode in (2, 3, 4, 5):
        assert torch.equal(envs[0]._goal, envs[1]._goal)
        flags = envs[0]._traj[3].to(torch.int32)
        assert bool((flags & 1).all())                         # every env's trajectory has started ...
        if mode == 4:
            assert bool(((flags & 4) != 0).all())              # ... "stay" is in manual mode from its second call on
        if mode == 3:
            assert bool(((flags & 16) != 0).any())             # ... some landings have reached the cut-off height
    with pytest.raises(RuntimeError):
        envs[0].set_goal_state(np.zeros(3), np.zeros(3), np.array([1.0, 0, 0]))


@pytest.mark.gpu
@pytest.mark.parametrize("mode,z0s,vz", [(2, [-0.3, -0.4, -0.25, -0.31234], 0.05), (3, [-0.75, -1.25, -0.5, -0.61357], 1.0)])
def test_stateful_phase_boundaries_within_one_call(mode, z0s, vz):
    """quadrotor_hip.h, deviation (2) of the stateful goal modes: the device generator clocks its phases with t = calls * dt in
    float32, the reference accumulates float64 t += dt.  Start heights that put the take-off / landing phase boundary on an EXACT
    multiple of dt (z0 = -0.3: t_traj = 4 s = 800 dt; landing from -0.75: 0.5 s = 100 dt) and arbitrary ones: at every call the
    goal agrees with the oracle's (the reference's accumulation, restated) up to at most ONE call's worth of motion, |v| dt; a start
    height whose boundary is not near a multiple of dt switches at the same call as the reference: exact (float32 round-off)."""
    from oracle import traj_oracle as tor
    n = len(z0s)
    state = np.zeros((n, 18)); state[:, 6] = state[:, 10] = state[:, 14] = 1.0
    state[:, 0:2] = [[0.1, -0.2]] * n
    state[:, 2] = z0s
    env = _mk("decoupled", n, mode)
    env.set_state(state)
    env.mark_traj_start()
    tr = tor.traj_start_batch(state, mode)
    dt, worst = 0.005, np.zeros(n)
    T = int(max(abs((-0.5 if mode == 2 else -0.25) - z) / vz for z in z0s) / dt) + 40
    for t in range(T):
        got = _np(torch.cat(env.get_desired(), 1))                     # xd, vd, b1d, b1d_dot, Wd
        want = np.concatenate(tor.get_desired_batch(tr, state), 1)
        err = np.abs(got[:, 0:3] - want[:, 0:3]).max(1)                 # position goal
        assert (err <= vz * dt + 2e-6).all(), (t, err)                  # never more than one call's worth of motion apart
        worst = np.maximum(worst, err)
    # (the state is held still here, so a goal that froze one call early / late at the switch stays |v| dt off: the worst case of the
    #  bound, visible at every later call).  Boundaries that are NOT within ~1e-6 s of a multiple of dt switch at the same call: exact.
    assert worst[3] <= 2e-6, worst
    print(f"stateful mode {mode}: worst goal difference per start height {np.array2string(worst, precision=2)} m over {T} calls "
          f"(bound |v| dt = {vz * dt:.1e} m)")
