"""GPU parity tests proper: the HIP path (through the C-ABI) against the committed golden
vectors produced by the reference, and against the float64 oracle on seeded inputs.

Stated tolerances (`grouped` = max over groups x,v,R,W of
||got-ref||_inf / max(||ref||_inf, 1)):
  one-step state      layout f64: grouped <= 2e-8 / 2e-9 / 2e-10 at 1 / 2 / 4 RK4 substeps (4th-order
                      convergence to the reference's DOP853 solution; the set contains saturated
                      torques at |W| ~ 2 pi); layout mixed: <= 2e-7 (x, v stored as float32)
  1000-step free-run  grouped <= 1e-5  — the north-star bar (measured: f64 ~3e-7, mixed ~2e-6)
  obs                 <= 2e-6 abs one-step, <= 1e-5 relative over trajectories (float32 rows)
  reward              <= 1e-5 abs
  done                identical, except where the deciding quantity is within 1e-6 of its threshold
"""
import numpy as np
import pytest
import torch

from conftest import grouped_rel_err
from oracle import quad_oracle as orc

pytestmark = pytest.mark.gpu
KINDS = orc.KINDS
ONESTEP_TOL = {("f64", 1): 2e-8, ("f64", 2): 2e-9, ("f64", 4): 2e-10, ("mixed", 1): 2e-7}


def _env(kind, n, **kw):
    from gym_rotor_amd import QuadVecEnv
    return QuadVecEnv(kind, n, device="cuda", want_raw_reward=True, **kw)


def _set_goal(env, goal):
    g = torch.as_tensor(goal, dtype=torch.float32, device=env.device)
    env.set_goal_state(g[:, 0:3], g[:, 3:6], g[:, 6:9], None, g[:, 9:12])


def _obs_list(obs):
    return [obs] if isinstance(obs, torch.Tensor) else list(obs)


def _np(t):
    return t.detach().cpu().numpy()


def _done_mismatch_ok(kind, d, got_done):
    """Envs whose done flag differs must sit within 1e-6 of a threshold."""
    bad = np.argwhere(got_done != d["done"])
    for i, _ in bad:
        s = d["next_state"][i]
        margins = [np.abs(np.abs(s[0:3]) - 1.0).min(), np.abs(np.abs(s[3:6]) - 4.0).min(),
                   np.abs(np.abs(s[15:18]) - 2 * np.pi).min()]
        if kind != "quad":
            o = [d[k][i] for k in ("obs0", "obs1") if k in d]
            margins.append(min(np.abs(np.abs(x.astype(np.float64)) - 1.0).min() for x in o))
        assert min(margins) < 1e-6, (kind, i, margins)
    return len(bad)


@pytest.mark.parametrize("layout,substeps", list(ONESTEP_TOL))
@pytest.mark.parametrize("kind", KINDS)
def test_onestep_golden(kind, layout, substeps, golden):
    d = golden(f"onestep_{kind}")
    n = d["state"].shape[0]
    env = _env(kind, n, layout=layout, substeps=substeps, obs_rows=True)
    env.set_state(d["state"], integ=d["integ"], params=d["params"])
    _set_goal(env, d["goal"])
    obs, rwd, done, _, _ = env.step(torch.from_numpy(d["action"].astype(np.float32)).cuda())
    torch.cuda.synchronize()
    got = _np(env.get_current_state())
    err = grouped_rel_err(got, d["next_state"])
    print(f"onestep {kind}/{layout}/S={substeps}: {err:.2e}")
    assert err <= ONESTEP_TOL[(layout, substeps)]
    for k, o in enumerate(_obs_list(obs)):
        ref = d[f"obs{k}"].astype(np.float64)
        assert (np.abs(_np(o).astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)).max() <= 2e-6
    raw = _np(env._reward_raw).astype(np.float64)
    assert np.abs(raw - d["reward_raw"]).max() <= 1e-5 * max(1.0, np.abs(d["reward_raw"]).max())
    nbad = _done_mismatch_ok(kind, d, _np(done))
    same = _np(done) == d["done"]
    assert np.abs(_np(rwd).astype(np.float64) - d["reward"])[same].max() <= 1e-5
    if kind != "quad":
        assert np.abs(_np(env.integ).astype(np.float64) - d["next_integ"]).max() <= 2e-6
    assert nbad <= 2



@pytest.mark.parametrize("substeps", [2, 3, 4, 10])
@pytest.mark.parametrize("kind", KINDS)
def test_onestep_golden_magnus_substeps(kind, substeps, golden):
    """Two or more substeps in the default layout: the Magnus-substep instantiations (MAG = 1; qr_dynamics.h: integrate_magnus)
    against the reference's one-step vectors — the set holds saturated torques at |W| ~ 2 pi.  The plain (non-adaptive) kernel
    is reached with the caller's reset promise; same bar as the default layout's RK4 step (x, v are stored as float32)."""
    d = golden(f"onestep_{kind}")
    n = d["state"].shape[0]
    env = _env(kind, n, layout="mixed", substeps=substeps, obs_rows=True, reset_on_done=True)
    plan = env.launch_plan()
    assert plan["mag"] == 1 and plan["adapt"] == 0 and plan["name"].endswith(",1>")
    env.set_state(d["state"], integ=d["integ"], params=d["params"])
    _set_goal(env, d["goal"])
    obs, rwd, done, _, _ = env.step(torch.from_numpy(d["action"].astype(np.float32)).cuda())
    torch.cuda.synchronize()
    err = grouped_rel_err(_np(env.get_current_state()), d["next_state"])
    print(f"onestep {kind}/mixed/Magnus S={substeps}: {err:.2e}")
    assert err <= 2e-7
    for k, o in enumerate(_obs_list(obs)):
        ref = d[f"obs{k}"].astype(np.float64)
        assert (np.abs(_np(o).astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)).max() <= 2e-6
    assert np.abs(_np(rwd).astype(np.float64) - d["reward"]).max() <= 1e-5
    assert _done_mismatch_ok(kind, d, _np(done)) <= 2


def test_magnus_substeps_same_bits_in_every_launch_family():
    """The integrator rides on `substeps` alone: with 2 substeps a rollout equals the per-step launches, the helper-wave launch equals
    the plain one and a shard equals its slice of the global batch — bit for bit, as with one substep."""
    n, T = 64 * 40 + 9, 6
    g = torch.Generator(device="cuda"); g.manual_seed(12)
    acts = torch.rand(T, n, 4, device="cuda", generator=g) * 2 - 1

    def run(mode, **kw):
        env = _env("coupled", n if "env_offset" not in kw else 64 * 10, seed=5, auto_reset=True, obs_rows=True, substeps=2, autotune=False, **kw)
        env.reset("train")
        env.get_norm_error_state()
        a = acts if "env_offset" not in kw else acts[:, kw["env_offset"]:kw["env_offset"] + 64 * 10]
        if mode == "rollout":
            out = env.rollout(a.contiguous())
            rw = out["reward"]
        else:
            rw = torch.stack([env.step(a[t].contiguous())[1].clone() for t in range(T)])
        return _np(env.get_current_state()), _np(rw), env

    s0, r0, e0 = run("step")
    assert e0.launch_plan()["mag"] == 1 and e0.launch_plan()["help"] == 1
    s1, r1, e1 = run("step", helper=False)
    assert e1.launch_plan()["help"] == 0 and e1.launch_plan()["mag"] == 1
    s2, r2, e2 = run("rollout")
    assert e2.launch_plan(T)["mag"] == 1 and e2.launch_plan(T)["single"] == 0
    s3, r3, _ = run("step", env_offset=64 * 7)
    assert np.array_equal(s0, s1) and np.array_equal(r0, r1)
    assert np.array_equal(s0, s2) and np.array_equal(r0.reshape(r2.shape), r2)
    assert np.array_equal(s0[64 * 7:64 * 17], s3) and np.array_equal(r0[:, 64 * 7:64 * 17], r3)

@pytest.mark.parametrize("layout,mode", [("mixed", "free"), ("mixed", "reset"), ("f64", "free")])
@pytest.mark.parametrize("kind", KINDS)
def test_trajectory_golden_1000_steps(kind, layout, mode, golden):
    """State-for-state over 1000 steps against the reference's own trajectories (the
    north-star parity bar: 1e-5 relative over 1000 steps)."""
    d = golden(f"traj_{mode}_{kind}")
    T, n = d["actions"].shape[:2]
    env = _env(kind, n, layout=layout, obs_rows=True)
    env.set_state(d["init_state"], integ=np.zeros((n, 8)), params=d["params"])
    _set_goal(env, d["goal"])
    if kind != "quad":
        env.get_norm_error_state()  # first obs after reset (main.py:129)
    acts = torch.from_numpy(d["actions"]).cuda()
    worst = worst_obs = worst_rwd = 0.0
    n_done_diff = 0
    for t in range(T):
        ra = d["reset_at"][t]
        if mode == "reset" and ra.any():
            # the caller resets exactly these envs to the injected states, zeroes their integral
            # terms and asks for their first observation (main.py:226-230)
            m = torch.from_numpy(ra).cuda()
            env.set_state(d["states"][t], mask=m)
            if kind != "quad":
                env._integ[:, m] = 0.0
                keep = env._integ.clone()
                env.get_norm_error_state()
                env._integ.copy_(torch.where(m[None, :], env._integ, keep))
        obs, rwd, done, _, _ = env.step(acts[t])
        got = _np(env.get_current_state())
        ref = d["states"][t + 1].copy()
        nxt = d["reset_at"][t + 1]
        ref[nxt] = got[nxt]  # rows replaced by an injected reset state are not post-step states
        worst = max(worst, grouped_rel_err(got, ref))
        for k, o in enumerate(_obs_list(obs)):
            r = d[f"obs{k}"][t].astype(np.float64)
            worst_obs = max(worst_obs, float((np.abs(_np(o).astype(np.float64) - r) / np.maximum(np.abs(r), 1.0)).max()))
        dd = _np(done) != d["dones"][t]
        n_done_diff += int(dd.sum())
        worst_rwd = max(worst_rwd, float(np.abs(_np(rwd).astype(np.float64) - d["rewards"][t])[~dd].max()))
    print(f"{kind}/{layout}/{mode}: state {worst:.2e} obs {worst_obs:.2e} reward {worst_rwd:.2e} done-diffs {n_done_diff}")
    # The north-star bar is 1e-5; the GUARD is 8e-6: the default layout's thinnest margin is here (coupled free run, observation
    # words of x, v at the float32 ulp: 7.25e-6 measured on three boxes in rounds 3-4) and an instruction-saving edit of the
    # stage arithmetic must not eat what is left of it unnoticed.
    assert worst <= 8e-6
    assert worst_obs <= 8e-6
    assert worst_rwd <= 1e-5
    assert n_done_diff <= 2


def test_flightlog_replay(golden):
    """One-step replay of the reference-owned flight log results/MODUL_log_20250303_120200.dat
    (DecoupledWrapper eval flight, nominal parameters; column map draw_plot.py:24-47):
    state[t] + action[t] -> state[t+1] to the log's print precision (1.1e-10)."""
    log = golden("flightlog_modul")["log"]
    act, state, cmd = log[:-1, 0:5], log[:-1, 5:23], log[:-1, 28:40]
    nxt = log[1:, 5:23]
    n = act.shape[0]
    env = _env("decoupled", n, layout="f64", use_UDM=False)
    env.set_state(state)
    # b1d from the logged b1c = b1d - (b1d.b3) b3 with b1d_z = 0:  b1d = b1c - (b1c_z / b3_z) b3
    b3 = state[:, 12:15]
    b1c = cmd[:, 6:9]
    b1d = b1c - (b1c[:, 2:3] / b3[:, 2:3]) * b3
    goal = np.concatenate([cmd[:, 0:6], b1d, cmd[:, 9:12]], 1)
    _set_goal(env, goal)
    env.step(torch.from_numpy(act.astype(np.float32)).cuda())
    got = _np(env.get_current_state())
    err = np.abs(got - nxt).max()
    print(f"flight-log replay: max|dstate| = {err:.3e} over {n} transitions")
    assert err <= 5e-9  # actions are float32-rounded here (float32 I/O); the log prints 1e-10


@pytest.mark.parametrize("kind", KINDS)
def test_rollout_equals_steps(kind):
    """qr_rollout(T) is bit-identical to T x qr_step (same maths, state kept in registers)."""
    n, T = 1000, 7
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    e1, e2 = _env(kind, n, seed=5, obs_rows=True), _env(kind, n, seed=5, obs_rows=True)
    for e in (e1, e2):
        e.reset("train")
    acts = torch.rand(T, n, e1.action_dim, device="cuda", generator=g) * 2 - 1
    outs = []
    for t in range(T):
        obs, r, d, _, _ = e1.step(acts[t])
        outs.append(([o.clone() for o in _obs_list(obs)], r.clone(), d.clone()))
    ro = e2.rollout(acts)
    assert torch.equal(e1.get_current_state(), e2.get_current_state())
    for t in range(T):
        for k, o in enumerate(outs[t][0]):
            assert torch.equal(o, _obs_list(ro["obs"])[k][t])
        assert torch.equal(outs[t][1], ro["reward"][t]) and torch.equal(outs[t][2], ro["terminated"][t])


@pytest.mark.parametrize("n", [1, 63, 64, 65, 300, 262144 + 77])
def test_ragged_and_large_batches_vs_oracle(n):
    """Ragged tails of both workgroup sizes (64 / 256 lanes) and N=1 against the oracle."""
    rng = np.random.default_rng(n)
    kind = "coupled"
    m = min(n, 512)  # oracle on a sample of envs, incl. the last rows
    idx = np.unique(np.concatenate([rng.integers(0, n, m), np.arange(max(0, n - 70), n)]))
    state = orc.sample_reset_state(rng, len(idx)).astype(np.float32).astype(np.float64)
    full = np.tile(state[:1], (n, 1)); full[idx] = state
    action = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    env = _env(kind, n, layout="f64", substeps=2, use_UDM=False)
    env.set_state(full, integ=np.zeros((n, 8)))
    obs, rwd, done, _, _ = env.step(torch.from_numpy(action).cuda())
    ref = orc.step_batch(kind, _state_roundtrip(full[idx]), action[idx].astype(np.float64))
    got = _np(env.get_current_state())[idx]
    assert grouped_rel_err(got, ref["state"]) <= 1e-9
    assert np.abs(_np(obs)[idx].astype(np.float64) - ref["obs"][0].astype(np.float64)).max() <= 2e-6
    assert (np.abs(_np(rwd)[idx] - ref["reward"]) <= 1e-5).all()


def _state_roundtrip(s):
    """set_state projects R onto SO(3); do the same for the oracle input."""
    s = s.copy()
    R = np.swapaxes(s[:, 6:15].reshape(-1, 3, 3), 1, 2)
    U, _, Vt = np.linalg.svd(R)
    s[:, 6:15] = np.swapaxes(U @ Vt, 1, 2).reshape(-1, 9)
    return s


def test_set_get_state_and_ensure_SO3(golden):
    """qr_set_state applies the reference's ensure_SO3 projection (nearest rotation = U V^T,
    quad_utils.py:123-142); qr_get_state returns the 18-vector.  KAT from the reference."""
    k = golden("kat_units")
    Rin, Rout = k["so3_in"], k["so3_out"]
    n = Rin.shape[0]
    s = np.zeros((n, 18)); s[:, 0:3] = 0.25; s[:, 3:6] = -1.5; s[:, 15:18] = 2.0
    s[:, 6:15] = np.swapaxes(Rin, 1, 2).reshape(n, 9)
    env = _env("quad", n, layout="f64")
    env.set_state(s)
    got = _np(env.get_current_state())
    Rgot = np.swapaxes(got[:, 6:15].reshape(n, 3, 3), 1, 2)
    # where the reference projected, it returned U V^T; where it kept R (error < 1e-5) the
    # nearest rotation differs from R by less than that error
    proj = np.abs(Rin - Rout).max(axis=(1, 2)) > 0
    U, _, Vt = np.linalg.svd(Rin)
    assert np.abs(Rgot - U @ Vt).max() <= 1e-12
    assert np.abs(Rgot[proj] - Rout[proj]).max() <= 1e-12
    assert np.abs(Rgot[~proj] - Rout[~proj]).max() <= 2e-5
    assert np.array_equal(got[:, 0:6], s[:, 0:6]) and np.array_equal(got[:, 15:18], s[:, 15:18])


def test_reset_distribution_and_determinism():
    """reset() follows quad.py:338-404 distributionally; draws depend only on
    (seed, global env id, episode): two half-size shards reproduce one full-size env."""
    n = 100000
    env = _env("coupled", n, seed=11)
    s = _np(env.reset("train"))
    p = _np(env.params)
    nom = orc.NOMINAL_PARAMS
    width = np.array([0.1, 0.1, 0.1, 0.1, 0.1, 0.05])
    rel = p / nom - 1.0
    assert (np.abs(rel) <= width * (1 + 1e-6)).all()
    assert np.allclose(rel.mean(0), 0, atol=3e-3 * 1) and np.allclose(rel.std(0), width / np.sqrt(3), rtol=0.03)
    zero = (np.abs(s[:, 0:6]).max(1) == 0)
    assert abs(zero.mean() - 0.2) < 0.01  # quad.py:342
    nz = ~zero
    assert np.abs(s[nz, 0:3]).max() <= 0.6 and np.abs(s[nz, 3:6]).max() <= 2.0 and np.abs(s[nz, 15:18]).max() <= np.pi + 1e-6
    assert np.abs(s[nz, 0:3]).max() > 0.59 and np.abs(s[nz, 15:18]).max() > 3.1
    R = np.swapaxes(s[:, 6:15].astype(np.float64).reshape(n, 3, 3), 1, 2)
    full = _np(env.get_current_state())
    Rf = np.swapaxes(full[:, 6:15].reshape(n, 3, 3), 1, 2)
    assert np.abs(np.swapaxes(Rf, 1, 2) @ Rf - np.eye(3)).max() < 1e-13 and np.abs(np.linalg.det(Rf) - 1).max() < 1e-13
    roll = np.arctan2(Rf[:, 2, 1], Rf[:, 2, 2]); pitch = -np.arcsin(Rf[:, 2, 0]); yaw = np.arctan2(Rf[:, 1, 0], Rf[:, 0, 0])
    lim = np.deg2rad(50.0)
    assert np.abs(roll).max() <= lim + 1e-6 and np.abs(pitch).max() <= lim + 1e-6
    assert np.abs(roll[zero]).max() < 1e-7 and np.abs(pitch[zero]).max() < 1e-7
    assert abs(yaw.mean()) < 0.03 and abs(yaw.std() - np.pi / np.sqrt(3)) < 0.03
    assert np.abs(_np(env.integ)).max() == 0
    # eval reset: nominal parameters, x ~ U(+-0.4), everything else zero error (quad.py:352-356)
    se = _np(env.reset("eval"))
    assert np.abs(se[:, 0:3]).max() <= 0.4 and np.abs(se[:, 3:6]).max() == 0 and np.abs(se[:, 15:18]).max() == 0
    assert np.allclose(_np(env.params), nom.astype(np.float32)[None])
    # sharding independence
    a = _env("coupled", 1000, seed=7, env_offset=0)
    b0 = _env("coupled", 600, seed=7, env_offset=0)
    b1 = _env("coupled", 400, seed=7, env_offset=600)
    sa = _np(a.reset("train"))
    sb = np.concatenate([_np(b0.reset("train")), _np(b1.reset("train"))])
    assert np.array_equal(sa, sb)
    # a second reset draws a new episode
    assert not np.array_equal(_np(a.reset("train")), sa)


def test_masked_reset_only_touches_masked_envs():
    env = _env("decoupled", 500, seed=3)
    env.reset("train")
    before = _np(env.get_current_state())
    mask = torch.zeros(500, dtype=torch.bool, device="cuda"); mask[::3] = True
    env.reset("train", mask=mask)
    after = _np(env.get_current_state())
    m = _np(mask)
    assert np.array_equal(before[~m], after[~m]) and not np.array_equal(before[m], after[m])


@pytest.mark.parametrize("kind", KINDS)
def test_auto_reset_semantics(kind):
    """With auto_reset the env that terminates is re-sampled inside the launch: reward/done
    belong to the terminated step, the observation is the first one of the new episode
    (integral terms advanced once, main.py:226-230), and the step counter restarts."""
    n = 4096
    rng = np.random.default_rng(0)
    env = _env(kind, n, seed=9, auto_reset=True, max_episode_steps=50, obs_rows=True)
    ref = _env(kind, n, seed=9, auto_reset=False, max_episode_steps=50, obs_rows=True, w_adapt=0.0)   # (the plain instantiation: same arithmetic)
    for e in (env, ref):
        e.reset("train")
    a = torch.from_numpy(rng.uniform(-1, 1, (n, env.action_dim)).astype(np.float32)).cuda()
    seen = 0
    for t in range(60):
        ref.load_state_dict(env.state_dict())
        obs, rwd, done, trunc, _ = env.step(a)
        o2, r2, d2, t2, _ = ref.step(a)
        assert torch.equal(rwd, r2) and torch.equal(done, d2) and torch.equal(trunc, t2)  # same step outcome
        hit = _np(done.any(1) | trunc)
        seen += hit.sum()
        st, st_ref = _np(env.get_current_state()), _np(ref.get_current_state())
        assert np.array_equal(st[~hit], st_ref[~hit])
        if hit.any():
            assert not np.array_equal(st[hit], st_ref[hit])
            assert (_np(env.episode_steps)[hit] == 0).all()
            if kind != "quad":  # first obs of the new episode == get_norm_error_state on the new state
                chk = _env(kind, n, seed=0, obs_rows=True)
                chk.set_state(st, integ=np.zeros((n, 8)), params=_np(env.params))
                first = chk.get_norm_error_state()
                for k, o in enumerate(_obs_list(obs)):
                    assert np.abs(_np(o)[hit] - _np(first[k])[hit]).max() <= 1e-6
    assert seen > 0
    assert (_np(env.episode_steps) < 50).all()


def test_properties_at_full_size():
    """Size-independent properties at BASELINE.json's batch (65 536 envs): R(q) stays in SO(3),
    crashes give reward -1, rewards in [0,1] otherwise, hover equilibrium is a fixed point,
    MONO observations are equivariant under a yaw rotation about e3."""
    n = 65536
    env = _env("coupled", n, seed=1, obs_rows=True)
    env.reset("train")
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for _ in range(200):
        a = torch.rand(n, 4, device="cuda", generator=g) * 2 - 1
        obs, rwd, done, _, _ = env.step(a)
    s = _np(env.get_current_state())
    assert np.isfinite(s).all()
    R = np.swapaxes(s[:, 6:15].reshape(n, 3, 3), 1, 2)
    assert np.abs(np.swapaxes(R, 1, 2) @ R - np.eye(3)).max() < 1e-12
    r, d = _np(rwd)[:, 0], _np(done)[:, 0]
    assert (r[d] == -1.0).all() and ((r[~d] >= 0) & (r[~d] <= 1)).all()
    # hover: f = m g (a0 such that 4(scale a0 + avrg) = m g), M = 0, from rest -> stays at rest
    h = _env("coupled", 1024, use_UDM=False, layout="f64")
    s0 = np.zeros((1024, 18)); s0[:, 6] = s0[:, 10] = s0[:, 14] = 1.0
    h.set_state(s0)
    a0 = (h.m_nominal * h.g / 4.0 - h.avrg_act) / h.scale_act
    act = torch.zeros(1024, 4, device="cuda"); act[:, 0] = a0
    for _ in range(100):
        h.step(act)
    drift = np.abs(_np(h.get_current_state()) - s0).max()
    assert drift < 1e-6  # a0 is float32-rounded: residual thrust ~1e-7 * g
    # yaw equivariance: rotate state and goal by Rz(psi) -> identical ex/ev/eb1/eW observation
    rng = np.random.default_rng(2)
    m = 2048
    st = orc.sample_reset_state(rng, m)
    psi = rng.uniform(-np.pi, np.pi, m)
    c, sn = np.cos(psi), np.sin(psi)
    Rz = np.zeros((m, 3, 3)); Rz[:, 0, 0] = c; Rz[:, 0, 1] = -sn; Rz[:, 1, 0] = sn; Rz[:, 1, 1] = c; Rz[:, 2, 2] = 1
    rot = lambda v: np.einsum("nij,nj->ni", Rz, v)
    st2 = st.copy()
    st2[:, 0:3], st2[:, 3:6] = rot(st[:, 0:3]), rot(st[:, 3:6])
    R0 = np.swapaxes(st[:, 6:15].reshape(m, 3, 3), 1, 2)
    st2[:, 6:15] = np.swapaxes(Rz @ R0, 1, 2).reshape(m, 9)
    goal = np.tile(orc.DEFAULT_GOAL, (m, 1)); goal2 = goal.copy(); goal2[:, 6:9] = rot(goal[:, 6:9])
    act = torch.from_numpy(rng.uniform(-1, 1, (m, 4)).astype(np.float32)).cuda()
    outs = []
    for s_, g_ in ((st, goal), (st2, goal2)):
        e = _env("coupled", m, use_UDM=False, layout="f64")
        e.set_state(s_, integ=np.zeros((m, 8))); _set_goal(e, g_)
        o, _, _, _, _ = e.step(act)
        outs.append(_np(o).astype(np.float64))
    o1, o2 = outs
    assert np.abs(rot(o1[:, 0:3]) - o2[:, 0:3]).max() < 2e-6 and np.abs(rot(o1[:, 6:9]) - o2[:, 6:9]).max() < 2e-6
    assert np.abs(o1[:, 18] - o2[:, 18]).max() < 2e-6 and np.abs(o1[:, 20:23] - o2[:, 20:23]).max() < 2e-6


def test_f32_layout_runs_close():
    """layout='f32' is the fast approximate mode: same step within float32 accuracy."""
    n = 2048
    rng = np.random.default_rng(4)
    st = orc.sample_reset_state(rng, n)
    act = torch.from_numpy(rng.uniform(-1, 1, (n, 5)).astype(np.float32)).cuda()
    res = []
    for layout in ("f64", "f32"):
        e = _env("decoupled", n, layout=layout, use_UDM=False)
        e.set_state(st, integ=np.zeros((n, 8)))
        e.step(act)
        res.append(_np(e.get_current_state()))
    assert grouped_rel_err(res[1], res[0]) < 5e-6


def test_api_errors():
    env = _env("coupled", 8)
    with pytest.raises(ValueError):
        env.step(torch.zeros(8, 5, device="cuda"))
    with pytest.raises(TypeError):
        env.step(torch.zeros(8, 4, device="cuda", dtype=torch.float64))
    with pytest.raises(ValueError):
        env.step(torch.zeros(8, 4))
    with pytest.raises(ValueError):
        env.get_norm_error_state("BOTH")
    with pytest.raises(AttributeError):                      # a bare QuadEnv raises AttributeError in the reference (no alpha / eIx_lim)
        _env("quad", 8).get_norm_error_state("MONO")
    with pytest.raises(ValueError):
        env.set_state(np.zeros((7, 18)))


def test_compat_single_env_matches_reference_trajectory(golden):
    """The num_envs=1 adapters reproduce the reference's return types and values
    (list of float32 arrays, list of floats, list of bools, False, {})."""
    from gym_rotor_amd import CoupledWrapper, DecoupledWrapper
    for cls, kind in ((CoupledWrapper, "coupled"), (DecoupledWrapper, "decoupled")):
        d = golden(f"traj_free_{kind}")
        env = cls(layout="f64")
        env.vec.set_state(d["init_state"][:1], integ=np.zeros((1, 8)), params=d["params"][:1])
        g = d["goal"][0]
        env.set_goal_state(g[0:3], g[3:6], g[6:9], np.zeros(3), g[9:12])
        first = env.get_norm_error_state(env.framework)
        assert isinstance(first, list) and first[0].dtype == np.float32
        for t in range(50):
            obs, rwd, done, trunc, info = env.step(d["actions"][t, 0])
            assert isinstance(obs, list) and isinstance(rwd, list) and isinstance(done, list)
            assert trunc is False and info == {} and isinstance(done[0], bool) and isinstance(rwd[0], float)
            for k, o in enumerate(obs):
                assert o.dtype == np.float32 and np.abs(o - d[f"obs{k}"][t, 0]).max() <= 2e-6
            assert np.abs(np.array(rwd) - d["rewards"][t, 0]).max() <= 1e-5
            assert done == list(d["dones"][t, 0])
        assert np.abs(env.get_current_state() - d["states"][50, 0]).max() <= 1e-7


@pytest.mark.parametrize("layout,tol_state,tol_obs", [("mixed", 1e-5, 1e-5), ("f64", 2e-6, 1e-5)])
@pytest.mark.parametrize("kind", KINDS)
def test_trajectory_vs_oracle_256_envs_1000_steps(kind, layout, tol_state, tol_obs):
    """The north-star bar on a larger sample than the reference goldens: 256 envs x 1000 free-run
    random-action steps, 1 substep, against the float64 DOP853 oracle (itself pinned to the
    reference at 1e-13 per step).  Default layout (`mixed`: x, v stored as float32; the launches without in-launch
    resets form their quaternion stages in delta form): the worst env sits at 1.3-2.4e-6 — in this far-out-of-regime
    free run |x| reaches 150 m (float32 ulp 1.5e-5) — and every observation row is within 1e-5 of the oracle's, relative
    to the row's own scale.  The all-float64 layout: 2e-7..1e-6."""
    n, T = 256, 1000
    rng = np.random.default_rng(77 + KINDS.index(kind))
    A = orc.ACTION_DIM[kind]
    state = _state_roundtrip(orc.sample_reset_state(rng, n).astype(np.float32).astype(np.float64))
    params = orc.sample_params(rng, n).astype(np.float32).astype(np.float64)
    acts = rng.uniform(-1, 1, (T, n, A)).astype(np.float32)
    env = _env(kind, n, layout=layout, obs_rows=True)
    env.set_state(state, integ=np.zeros((n, 8)), params=params)
    ro = env.rollout(torch.from_numpy(acts).cuda())      # one launch, bit-identical to 1000 steps
    got = _np(env.get_current_state())
    s, integ = state, np.zeros((n, 8))
    worst_obs, where, worst_elem = 0.0, None, 0.0
    for t in range(T):
        o = orc.step_batch(kind, s, acts[t].astype(np.float64), params, None, integ)
        s, integ = o["state"], o["integ"]
        if t % 100 == 99:
            for k, ob in enumerate(o["obs"]):
                g = _np(_obs_list(ro["obs"])[k][t]).astype(np.float64)
                r = np.asarray(ob, dtype=np.float64)
                # (relative to the row's own scale: far out of regime |x| reaches 150 m, where ONE float32 ulp of an
                # observation word is 1.5e-5 — SURVEY.md 8(d)'s "obs <= 1e-5 abs" is a statement about in-regime rows, |.| <= 1)
                e = np.abs(g - r) / np.maximum(np.abs(r).max(-1, keepdims=True), 1.0)
                # (... and every element against ITS OWN magnitude at the looser 1e-4: the row-scale figure alone would let the
                # small words of a far-out row — attitude, eW — drift by 1e-3 unnoticed)
                worst_elem = max(worst_elem, float((np.abs(g - r) / np.maximum(np.abs(r), 1.0)).max()))
                if e.max() > worst_obs:
                    worst_obs, where = float(e.max()), (t, k) + tuple(int(i) for i in np.unravel_index(e.argmax(), e.shape))
    err = grouped_rel_err(got, s)
    print(f"256 envs x 1000 steps {kind}/{layout}: final-state grouped error {err:.2e}, obs {worst_obs:.2e} at (t, obs, env, col) = {where}, "
          f"per element {worst_elem:.2e}")
    assert err <= tol_state and worst_obs <= tol_obs and worst_elem <= 1e-4


@pytest.mark.parametrize("kind", KINDS)
def test_free_run_tail_2048_envs_1000_steps(kind):
    """Default configuration on a population large enough to have a tail: 2048 envs stepped on for
    1000 random-action steps with no reset — far beyond termination, the fastest spin 35-50 rad/s
    (the termination bound is 2 pi).  With a fixed substep count the RK4 truncation error of those
    envs reaches 2e-5..6e-5; the rate-adaptive substep count (w_adapt = 16 rad/s, default) keeps
    EVERY env inside the 1e-5 bar against the float64 DOP853 oracle."""
    n, T = 2048, 1000
    rng = np.random.default_rng(500 + KINDS.index(kind))
    A = orc.ACTION_DIM[kind]
    state = _state_roundtrip(orc.sample_reset_state(rng, n).astype(np.float32).astype(np.float64))
    params = orc.sample_params(rng, n).astype(np.float32).astype(np.float64)
    acts = rng.uniform(-1, 1, (T, n, A)).astype(np.float32)
    s, integ = state, np.zeros((n, 8))
    for t in range(T):
        o = orc.step_batch(kind, s, acts[t].astype(np.float64), params, None, integ)
        s, integ = o["state"], o["integ"]
    wmax = np.abs(s[:, 15:18]).max()
    acts_d = torch.from_numpy(acts).cuda()
    errs = {}
    for w_adapt in (16.0, 0.0):
        env = _env(kind, n, obs_rows=True, w_adapt=w_adapt)
        assert env.layout == "mixed" and env.substeps == 1
        env.set_state(state, integ=np.zeros((n, 8)), params=params)
        env.rollout(acts_d)
        errs[w_adapt] = grouped_rel_err(_np(env.get_current_state()), s)
    print(f"2048-env free run {kind}: max|W| {wmax:.1f} rad/s, grouped error adaptive {errs[16.0]:.2e}, fixed {errs[0.0]:.2e}")
    assert errs[16.0] <= 1e-5


@pytest.mark.parametrize("kind", KINDS)
def test_adaptive_kernel_in_regime(kind):
    """In regime (|W| below w_adapt) the rate-adaptive kernel takes exactly `substeps` substeps.  With in-launch resets the
    launcher itself picks the plain kernel (no env can leave the regime): bit-identical with and without w_adapt, 300 steps.
    Without them (free run from reset, 40 steps) the rate-adaptive instantiation runs, whose quaternion stages are in delta
    form (qr_dynamics.h: integrate_delta): the same trajectory to ~1e-9 per step, not the same bits."""
    n = 1000
    A = orc.ACTION_DIM[kind]
    for auto_reset, T in ((False, 40), (True, 300)):
        acts = (torch.rand(T, n, A, device="cuda", generator=torch.Generator("cuda").manual_seed(9)) * 2 - 1) * 0.3
        out = []
        for w_adapt in (16.0, 0.0):
            env = _env(kind, n, seed=4, auto_reset=auto_reset, w_adapt=w_adapt)
            env.reset("train")
            ro = env.rollout(acts)
            out.append((_np(env.get_current_state()), _np(ro["reward"]), _np(ro["terminated"])))
        assert np.abs(out[0][0][:, 15:18]).max() < 16.0
        if auto_reset:
            for x, y in zip(*out):
                assert np.array_equal(x, y)
        else:
            assert grouped_rel_err(out[0][0], out[1][0]) <= 1e-6 and not np.array_equal(out[0][0], out[1][0])   # (measured 2.3e-7)
            assert np.abs(out[0][1] - out[1][1]).max() <= 1e-5 and (out[0][2] != out[1][2]).mean() <= 1e-3


@pytest.mark.parametrize("kind", KINDS)
def test_reset_on_done_promise_takes_the_plain_kernel(kind):
    """QuadVecEnv(reset_on_done=True) (QR_FLAG_CALLER_RESETS) is the caller's promise of the reference's own loop — reset every env
    that step() reports done before stepping it again (main.py:183-186) — under which no env leaves the regime and the launcher
    runs the kernel compiled WITHOUT rate adaptivity: the same bits as an env with w_adapt = 0 driven by the same loop (and hence
    as the in-launch-reset mode's arithmetic), through ~1000 masked resets; the default (free-run) env beside them runs the
    rate-adaptive delta-form kernel: the same flight to ~1e-9 per step, not the same bits."""
    n, T = 2000, 150
    A = orc.ACTION_DIM[kind]
    g = torch.Generator(device="cuda"); g.manual_seed(12)
    envs = [_env(kind, n, seed=4, obs_rows=True, reset_on_done=True), _env(kind, n, seed=4, obs_rows=True, w_adapt=0.0), _env(kind, n, seed=4, obs_rows=True)]
    for e in envs:
        e.reset("train")
        if kind != "quad":
            e.get_norm_error_state()
    resets = 0
    for t in range(T):
        act = torch.rand(n, A, device="cuda", generator=g) * 2 - 1
        outs = [e.step(act) for e in envs]
        (oa, ra, da, _, _), (ob, rb, db, _, _), (oc, rc, dc, _, _) = outs
        assert torch.equal(ra, rb) and torch.equal(da, db)
        for x, y in zip(_obs_list(oa), _obs_list(ob)):
            assert torch.equal(x, y)
        mask = da.reshape(n, -1).any(dim=1)
        resets += int(mask.sum())
        if bool(mask.any()):
            for e in envs:          # the promise, kept: done envs start a new episode before the next step
                e.reset("train", mask=mask | dc.reshape(n, -1).any(dim=1))
    assert resets > 500
    sa, sb, sc = (_np(e.get_current_state()) for e in envs)
    assert np.array_equal(sa, sb)
    assert grouped_rel_err(sa, sc) <= 1e-6 and not np.array_equal(sa, sc)
    with pytest.raises(ValueError):
        _env(kind, 8, auto_reset=True, reset_on_done=True)


def test_one_million_envs_ten_substeps_and_shard_equivalence():
    """BASELINE.json configs[4] at full size on one GPU: Quad-v0, 1 048 576 envs, 10 substeps,
    auto-reset.  Size-independent properties (finite, R in SO(3), crash => -1, rewards in [0, 1]),
    and the multi-GPU contract: the envs a rank would own (here the 131 072 of rank 5 of 8) stepped
    as their own shard (env_offset) produce the same bits as inside the global batch — no result
    depends on how the batch is partitioned."""
    n, shard, rank, T = 1 << 20, 1 << 17, 5, 12
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    acts = torch.rand(T, n, 4, device="cuda", generator=g) * 2 - 1
    env = _env("quad", n, seed=9, substeps=10, auto_reset=True, obs_rows=True)
    env.reset("train")
    part = _env("quad", shard, seed=9, substeps=10, auto_reset=True, obs_rows=True, env_offset=rank * shard)
    part.reset("train")
    lo, hi = rank * shard, (rank + 1) * shard
    dones = 0
    for t in range(T):
        obs, rwd, done, _, _ = env.step(acts[t])
        o2, r2, d2, _, _ = part.step(acts[t, lo:hi].contiguous())
        assert torch.equal(obs[lo:hi], o2) and torch.equal(rwd[lo:hi], r2) and torch.equal(done[lo:hi], d2)
        dones += int(done.sum())
    assert dones > 0                                    # resets happened inside the launches
    s = env.get_current_state()
    assert torch.isfinite(s).all() and torch.equal(s[lo:hi], part.get_current_state())
    idx = torch.randint(0, n, (8192,), device="cuda", generator=g)
    R = s[idx, 6:15].reshape(-1, 3, 3).transpose(1, 2)
    assert (R.transpose(1, 2) @ R - torch.eye(3, device="cuda", dtype=R.dtype)).abs().max() < 1e-12
    r, d = rwd[:, 0], done[:, 0]
    assert (r[d] == -1.0).all() and ((r[~d] >= 0) & (r[~d] <= 1)).all()


@pytest.mark.parametrize("kind", KINDS)
def test_helper_wave_launch_equals_the_plain_one(kind):
    """Small grids run the one-step kernel with a helper wavefront per tile (reset pool, Quad-v0 reward, observation
    rows formed / carried out by the second wave of the workgroup), large ones without.  The same envs stepped as a
    shard small enough for the helper launch and inside a batch too large for it give the same bits — state,
    observation rows, rewards (also the raw ones), dones, terminal observations, episode counters — and so does a
    rollout, whose helper wave hands over one pool per env-step."""
    n, shard, rank, T = 294912, 32768, 3, 160       # 4608 tiles: beyond every helper-wave limit (3328 Quad-v0 / 2560 wrappers)
    adim = 5 if kind == "decoupled" else 4
    g = torch.Generator(device="cuda"); g.manual_seed(17)
    acts = torch.rand(T, n, adim, device="cuda", generator=g) * 2 - 1
    big = _env(kind, n, seed=21, auto_reset=True, obs_rows=True, final_obs=True)
    part = _env(kind, shard, seed=21, auto_reset=True, obs_rows=True, final_obs=True, env_offset=rank * shard)
    assert big.kernel_info()[2] == 64 and part.kernel_info()[2] == 128      # threads per workgroup: plain / with helper
    lo, hi = rank * shard, (rank + 1) * shard
    for e in (big, part):
        e.reset("train")
        if kind != "quad":
            e.get_norm_error_state()
    dones = 0
    for t in range(T // 2):
        o, r, d, _, info = big.step(acts[t])
        o2, r2, d2, _, info2 = part.step(acts[t, lo:hi].contiguous())
        for a_, b_ in zip(_obs_list(o), _obs_list(o2)):
            assert torch.equal(a_[lo:hi], b_)
        assert torch.equal(r[lo:hi], r2) and torch.equal(d[lo:hi], d2)
        fo, fo2 = big.final_observation(), part.final_observation()
        reset_rows = d2.reshape(shard, -1).any(dim=1)
        for a_, b_ in zip(_obs_list(fo), _obs_list(fo2)):
            assert torch.equal(a_[lo:hi][reset_rows], b_[reset_rows])
        dones += int(reset_rows.sum())
    assert dones > 1000
    assert torch.equal(big.get_current_state()[lo:hi], part.get_current_state())
    assert torch.equal(big._episode[lo:hi], part._episode)
    # the second half as ONE rollout launch on the shard (helper wave, a pool per step) against steps on the batch
    ro = part.rollout(acts[T // 2:, lo:hi].contiguous())
    for t in range(T // 2, T):
        o, r, d, _, _ = big.step(acts[t])
        assert torch.equal(r[lo:hi], ro["reward"][t - T // 2]) and torch.equal(d[lo:hi], ro["terminated"][t - T // 2])
    assert torch.equal(big.get_current_state()[lo:hi], part.get_current_state())


@pytest.mark.parametrize("kind", KINDS)
def test_launch_rule_overrides_change_no_bit(kind):
    """QuadVecEnv(helper=True / False) (QR_FLAG_FORCE_HELPER / QR_FLAG_NO_HELPER) pins the launch rule's choice per env, and
    autotune_launch() picks it by timing: the SAME batch — a ragged one, per-env goals, a time limit — stepped under either choice
    gives the same bits (state, rows, rewards, dones, truncations, terminal observations, parameters, every counter), tuning
    leaves no trace in the env, and the forced choice reaches beyond the rule's own limits in both directions."""
    adim = 5 if kind == "decoupled" else 4
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    n, T = 64 * 700 + 17, 40
    kw = dict(seed=11, auto_reset=True, obs_rows=True, max_episode_steps=25, final_obs=True)
    a, b = _env(kind, n, helper=True, **kw), _env(kind, n, helper=False, **kw)
    c = _env(kind, n, **kw)
    assert a.kernel_info()[1:] == (701, 128) and b.kernel_info()[1:] == (701, 64) and c.kernel_info()[2] == 128
    big = _env(kind, 64 * 5000, helper=True, auto_reset=True)
    assert big.kernel_info()[1:] == (5000, 128) and _env(kind, 64 * 5000, auto_reset=True).kernel_info()[2] == 64
    assert _env(kind, n, helper=True).kernel_info()[2] == 64      # no in-launch resets: no such instantiation, the bit is ignored
    goal = torch.zeros(n, 12, device="cuda")
    goal[:, 0:6] = (torch.rand(n, 6, device="cuda", generator=g) - 0.5) * 0.2
    ang = torch.rand(n, device="cuda", generator=g) * 6.28
    goal[:, 6], goal[:, 7] = torch.cos(ang), torch.sin(ang)
    for e in (a, b, c):
        e.reset("train")
        _set_goal(e, goal)
        if kind != "quad":
            e.get_norm_error_state()
    resets = 0
    for t in range(T):
        act = torch.rand(n, adim, device="cuda", generator=g) * 2 - 1
        if t == 10:     # tuning in mid-flight: times both launches on this env's own buffers, then puts everything back
            rep = c.autotune_launch(launches=50, repeats=2)
            assert rep["picked"] in ("default", "helper", "no_helper") and rep["helper"] > 0 and rep["no_helper"] > 0
        oa, ra, da, ta, _ = a.step(act)
        ob, rb, db, tb, _ = b.step(act)
        oc, rc, dc, tc, _ = c.step(act)
        for x, y, z in zip(_obs_list(oa), _obs_list(ob), _obs_list(oc)):
            assert torch.equal(x, y) and torch.equal(x, z), t
        assert torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(ta, tb), t
        assert torch.equal(ra, rc) and torch.equal(da, dc) and torch.equal(ta, tc), t
        assert torch.equal(a._reward_raw, b._reward_raw)
        rows = da.reshape(n, -1).any(dim=1) | ta
        resets += int(rows.sum())
        for x, y in zip(_obs_list(a.final_observation()), _obs_list(b.final_observation())):
            assert torch.equal(x[rows], y[rows])
    assert resets >= n                                   # (the time limit alone ended every episode once)
    for e in (b, c):
        assert torch.equal(a.get_current_state(), e.get_current_state())
        for name in ("_params", "_episode", "_steps", "_reset_count", "_integ"):
            x, y = getattr(a, name), getattr(e, name)
            assert (x is None and y is None) or torch.equal(x, y), name


def _kill(env, kill):
    """Make the envs in `kill` terminate in the next step: x far outside the arena."""
    st = _np(env.get_current_state())
    st[_np(kill), 0] = 5.0
    env.set_state(st, mask=kill)


@pytest.mark.parametrize("kind", KINDS)
def test_in_launch_reset_pool_keying(kind):
    """In-launch resets take episode starts from the wavefront's pool (qr_rng.h): the stream is keyed by
    (seed, global id of the 64-env tile's first env, the tile's reset counter, slot), slot = rank of the
    lane among the tile's lanes that reset in that step.  Checked with 1, a few, 12, 13..63 and all 64
    lanes of a tile resetting at once (more than 12 = further pools drawn on demand):
      * the r-th resetting lane of a tile receives the same start whichever lane it is;
      * slots, tiles and counter values give different starts; a tile's counter advances by one per
        env-step whether or not anything reset;
      * a shard that starts at a multiple of 64 envs reproduces the global batch bit for bit;
      * episode counters of the re-sampled envs advance by one."""
    counts = [1, 2, 3, 11, 12, 13, 24, 25, 37, 63, 64, 5]
    n = 64 * 40 + 17
    g = torch.Generator(device="cuda"); g.manual_seed(123)

    def make(offset=0, n_=n):
        e = _env(kind, n_, seed=77, auto_reset=True, obs_rows=True, env_offset=offset)
        e.reset("train")
        if kind != "quad":
            e.get_norm_error_state()
        return e

    def kill_sets():
        k = torch.zeros(n, dtype=torch.bool, device="cuda")
        for w_, cnt in enumerate(counts):
            k[torch.randperm(64, device="cuda", generator=g)[:cnt] + 64 * w_] = True
        k[n - 3:] = True                                            # ragged tail wave
        return k

    A, B = make(), make()
    ka, kb = kill_sets(), kill_sets()
    assert not torch.equal(ka, kb)
    zero = torch.zeros(n, A.action_dim, device="cuda")
    ep0 = A._episode.clone()
    for e, k in ((A, ka), (B, kb)):
        _kill(e, k)
        _, _, done, _, _ = e.step(zero)
        assert bool(done[k].any(1).all())
    assert bool((A._episode == ep0 + ka.int()).all())
    assert bool((A._reset_count == 1).all())                         # every tile, with or without a reset
    sa, sb = _np(A.get_current_state()), _np(B.get_current_state())
    pa, pb = _np(A.params), _np(B.params)
    starts = []
    for w_, cnt in enumerate(counts):
        la = np.flatnonzero(_np(ka)[64 * w_:64 * w_ + 64]) + 64 * w_   # lanes in rank order
        lb = np.flatnonzero(_np(kb)[64 * w_:64 * w_ + 64]) + 64 * w_
        assert np.array_equal(sa[la], sb[lb]) and np.array_equal(pa[la], pb[lb])   # slot = rank, not lane
        starts.append(sa[la])
    allst = np.concatenate(starts)
    assert len(np.unique(allst[:, 15:18].round(12), axis=0)) >= 0.75 * len(allst)   # (20 % start at rest: W = 0)
    nz = np.abs(allst[:, 0:3]).max(1) > 0
    assert len(np.unique(allst[nz, 0:3], axis=0)) == nz.sum()        # distinct slots / tiles: distinct starts
    # envs that did not reset are untouched by the pool
    keep = ~_np(ka | kb)
    assert np.array_equal(sa[keep], sb[keep])
    # next step, same lanes: the counter moved on, so the same slots hold new starts
    _kill(A, ka)
    A.step(zero)
    assert bool((A._reset_count == 2).all())
    s2 = _np(A.get_current_state())
    la = np.flatnonzero(_np(ka))
    assert not np.array_equal(s2[la][:, 15:18], sa[la][:, 15:18])
    # a 64-aligned shard reproduces its part of the global batch
    lo, hi = 64 * 5, 64 * 15
    C = make(offset=lo, n_=hi - lo)
    assert np.array_equal(_np(C.get_current_state()), _np(make().get_current_state())[lo:hi])
    _kill(C, ka[lo:hi].clone())
    C.step(zero[lo:hi].contiguous())
    assert np.array_equal(_np(C.get_current_state()), sa[lo:hi])
    assert np.array_equal(_np(C.params), pa[lo:hi])


def test_in_launch_reset_pool_distribution():
    """Starts drawn from the pools follow quad.py:338-404 like qr_reset's: all 64 lanes of 2000 tiles
    re-sampled in one step (six pools per tile)."""
    n = 64 * 2000
    env = _env("coupled", n, seed=5, auto_reset=True, obs_rows=True)
    env.reset("train")
    env.get_norm_error_state()
    _kill(env, torch.ones(n, dtype=torch.bool, device="cuda"))
    _, _, done, _, _ = env.step(torch.zeros(n, 4, device="cuda"))
    assert bool(done.all())
    s, p = _np(env.get_current_state()), _np(env.params)
    rel = p / orc.NOMINAL_PARAMS - 1.0
    width = np.array([0.1, 0.1, 0.1, 0.1, 0.1, 0.05])
    assert (np.abs(rel) <= width * (1 + 1e-6)).all()
    assert np.allclose(rel.mean(0), 0, atol=3e-3) and np.allclose(rel.std(0), width / np.sqrt(3), rtol=0.03)
    zero = np.abs(s[:, 0:6]).max(1) == 0
    assert abs(zero.mean() - 0.2) < 0.01
    nz = ~zero
    assert np.abs(s[nz, 0:3]).max() <= 0.6 and np.abs(s[nz, 3:6]).max() <= 2.0 and np.abs(s[nz, 15:18]).max() <= np.pi + 1e-6
    assert np.abs(s[nz, 0:3]).max() > 0.59 and np.abs(s[nz, 15:18]).max() > 3.1
    for col in (0, 4, 16):                                            # uniform: mean 0, variance range^2 / 3
        rng_ = {0: 0.6, 4: 2.0, 16: np.pi}[col]
        assert abs(s[nz, col].mean()) < 0.02 * rng_ and abs(s[nz, col].std() - rng_ / np.sqrt(3)) < 0.02 * rng_
    R = np.swapaxes(s[:, 6:15].reshape(n, 3, 3), 1, 2)
    assert np.abs(np.swapaxes(R, 1, 2) @ R - np.eye(3)).max() < 1e-13
    roll = np.arctan2(R[:, 2, 1], R[:, 2, 2]); pitch = -np.arcsin(R[:, 2, 0]); yaw = np.arctan2(R[:, 1, 0], R[:, 0, 0])
    lim = np.deg2rad(50.0)
    assert np.abs(roll).max() <= lim + 1e-6 and np.abs(pitch).max() <= lim + 1e-6 and np.abs(roll[zero]).max() < 1e-7
    assert abs(yaw.mean()) < 0.03 and abs(yaw.std() - np.pi / np.sqrt(3)) < 0.03
    # lanes of one tile are independent of each other (slots of six different pools)
    c = np.corrcoef(np.stack([s[nz, 0][:-1], s[nz, 0][1:], s[nz, 16][:-1], s[nz, 16][1:]]))
    assert np.abs(c - np.eye(4)).max() < 0.02
    assert np.abs(_np(env.integ)[:, [0, 1, 2, 6]]).max() < 0.05         # integrators restarted (advanced once by the first obs)


@pytest.mark.parametrize("kind,substeps", [(k, 1) for k in KINDS] + [("quad", 10), ("decoupled", 4), ("coupled", 2), ("quad", 3)])
def test_production_mode_1000_steps_every_episode_vs_oracle(kind, substeps):
    """The mode a training loop runs (default layout, 1 substep, in-launch auto-reset, random
    actions): 512 envs x 1000 steps, ~5000 complete episodes.  The oracle steps every env from its
    own state and only adopts the GPU's freshly sampled state when an episode ends, so the error of
    every env is followed through every whole episode; rewards and done flags are compared at
    every step including the terminal one.  Also with 10 substeps (BASELINE.json configs[4]'s mode), 4, 2 and 3 — the
    Magnus-substep instantiations (MAG = 1): within one env-step the substeps' increments are summed in float32 and the
    float64 state takes them once — the error must not grow with the substep count."""
    n, T = 512, 1000
    rng = np.random.default_rng(4242 + KINDS.index(kind))
    A = orc.ACTION_DIM[kind]
    env = _env(kind, n, seed=31, auto_reset=True, obs_rows=True, substeps=substeps)
    assert env.layout == "mixed" and env.substeps == substeps and env.launch_plan()["mag"] == int(substeps >= 2)
    env.reset("train")
    if kind != "quad":
        env.get_norm_error_state()
    s = _np(env.get_current_state())
    params = _np(env.params).astype(np.float64)
    integ = _np(env.integ).astype(np.float64) if kind != "quad" else np.zeros((n, 8))
    worst_state = worst_rwd = 0.0
    episodes = ties = 0
    for t in range(T):
        act = rng.uniform(-1, 1, (n, A)).astype(np.float32)
        obs, rwd, done, _, _ = env.step(torch.from_numpy(act).cuda())
        o = orc.step_batch(kind, s, act.astype(np.float64), params, None, integ)
        g_state, g_done, g_rwd = _np(env.get_current_state()), _np(done), _np(rwd)
        mism = (g_done != o["done"]).any(1)
        ties += int(mism.sum())                      # deciding quantity within round-off of its threshold
        ended = g_done.any(1)
        keep = ~ended & ~mism
        worst_state = max(worst_state, grouped_rel_err(g_state[keep], o["state"][keep]))
        worst_rwd = max(worst_rwd, np.abs(g_rwd[~mism] - o["reward"][~mism]).max())
        # continue from the oracle's own state; adopt the GPU's new episode where one ended
        s, integ = o["state"], o["integ"]
        adopt = ended | mism
        if adopt.any():
            s[adopt] = g_state[adopt]
            params[adopt] = _np(env.params)[adopt]
            if kind != "quad":
                integ[adopt] = _np(env.integ)[adopt]
            episodes += int(ended.sum())
    print(f"production mode {kind} x{substeps}: {episodes} episodes, worst in-episode state error {worst_state:.2e}, reward {worst_rwd:.2e}, threshold ties {ties}")
    assert episodes > 2000 and ties <= 3
    assert worst_state <= 2e-6 and worst_rwd <= 2e-5


@pytest.mark.parametrize("kind,n", [("quad", 65536), ("decoupled", 32768)])
def test_headline_size_auto_reset_sampled_envs_vs_oracle(kind, n):
    """The headline instantiation at the headline size (BASELINE.json configs[1]: Quad-v0, 65 536 envs; configs[3]'s
    per-GPU share: DecoupledWrapper, 32 768 envs) in the mode the bench times: default layout, 1 substep, in-launch
    auto-reset, randomised parameters, 200 steps.  512 sampled envs are followed by the oracle through every
    episode (it steps from its own state and adopts the GPU's only where a new episode starts); size-independent
    properties are checked on the whole batch."""
    T, m = 200, 512
    rng = np.random.default_rng(99 + n)
    A = orc.ACTION_DIM[kind]
    env = _env(kind, n, seed=17, auto_reset=True, obs_rows=True)
    assert env.layout == "mixed" and env.substeps == 1 and env.use_UDM
    env.reset("train")
    if kind != "quad":
        env.get_norm_error_state()
    idx = np.sort(rng.choice(n, m, replace=False))
    idx_t = torch.from_numpy(idx).cuda()
    s = _np(env.get_current_state())[idx]
    params = _np(env.params).astype(np.float64)[idx]
    integ = _np(env.integ).astype(np.float64)[idx] if kind != "quad" else np.zeros((m, 8))
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    worst_state = worst_rwd = 0.0
    episodes = ties = total_done = 0
    for t in range(T):
        act = torch.rand(n, A, device="cuda", generator=g) * 2 - 1
        obs, rwd, done, _, _ = env.step(act)
        total_done += int(done.any(1).sum())
        a_s = _np(act[idx_t]).astype(np.float64)
        o = orc.step_batch(kind, s, a_s, params, None, integ)
        g_state = _np(env.get_current_state()[idx_t])
        g_done, g_rwd = _np(done[idx_t]), _np(rwd[idx_t])
        mism = (g_done != o["done"]).any(1)
        ties += int(mism.sum())
        ended = g_done.any(1)
        keep = ~ended & ~mism
        worst_state = max(worst_state, grouped_rel_err(g_state[keep], o["state"][keep]))
        worst_rwd = max(worst_rwd, np.abs(g_rwd[~mism] - o["reward"][~mism]).max())
        s, integ = o["state"], o["integ"]
        adopt = ended | mism
        if adopt.any():
            s[adopt] = g_state[adopt]
            params[adopt] = _np(env.params[idx_t]).astype(np.float64)[adopt]
            if kind != "quad":
                integ[adopt] = _np(env.integ[idx_t]).astype(np.float64)[adopt]
            episodes += int(ended.sum())
    print(f"{kind} {n} envs, auto-reset, {T} steps: {episodes} sampled episodes, worst in-episode state error {worst_state:.2e}, "
          f"reward {worst_rwd:.2e}, threshold ties {ties}, done rate {total_done / (n * T):.4f}")
    assert episodes > 500 and ties <= 3
    assert worst_state <= 2e-6 and worst_rwd <= 2e-5
    # whole batch
    full = _np(env.get_current_state())
    assert np.isfinite(full).all()
    R = np.swapaxes(full[:, 6:15].reshape(n, 3, 3), 1, 2)
    assert np.abs(np.swapaxes(R, 1, 2) @ R - np.eye(3)).max() < 1e-12
    r, d = _np(rwd), _np(done)
    assert (r[d] == -1.0).all() and ((r[~d] >= 0) & (r[~d] <= 1)).all()
    assert 0.003 < total_done / (n * T) < 0.05                                   # random actions: ~1 % of the envs end per step
    assert int(env._episode.sum()) == n + total_done                              # one explicit reset + every in-launch one
    assert bool((env._reset_count == T).all())
    p = _np(env.params) / orc.NOMINAL_PARAMS - 1.0
    assert (np.abs(p) <= np.array([0.1, 0.1, 0.1, 0.1, 0.1, 0.05]) * (1 + 1e-6)).all() and p.std(0).min() > 0.02
    # in regime: an env that left the arena was re-sampled in that very step
    assert np.abs(full[:, 0:3]).max() < 1.0 and np.abs(full[:, 3:6]).max() < 4.0


def test_long_episode_integrators_into_the_clip():
    """A 4000-step CoupledWrapper episode (the reference's episode cap, args_parse.py:16) with a goal offset that
    drives both integral terms far into their +-3 clip: the device holds the trapezoid integrators
    (quad_utils.py:38-63) as float32 words, the reference in float64 — after 4000 steps they still agree to 1e-5."""
    n, T = 16, 4000
    rng = np.random.default_rng(8)
    env = _env("coupled", n, use_UDM=False, obs_rows=True)
    s0 = np.zeros((n, 18)); s0[:, 6] = s0[:, 10] = s0[:, 14] = 1.0
    s0[:, 0:3] = rng.uniform(-0.2, 0.2, (n, 3))
    env.set_state(s0, integ=np.zeros((n, 8)))
    goal = np.tile(orc.DEFAULT_GOAL, (n, 1))
    goal[:, 0:3] = s0[:, 0:3] - rng.uniform(0.15, 0.35, (n, 3)) * np.sign(rng.uniform(-1, 1, (n, 3)))
    psi = rng.uniform(0.3, 0.8, n) * np.sign(rng.uniform(-1, 1, n))
    goal[:, 6], goal[:, 7] = np.cos(psi), np.sin(psi)
    goal = goal.astype(np.float32).astype(np.float64)
    _set_goal(env, goal)
    env.get_norm_error_state()
    a0 = (env.m_nominal * env.g / 4.0 - env.avrg_act) / env.scale_act        # hover thrust, no torque: stays aloft
    acts = np.zeros((T, n, 4), np.float32)
    acts[:, :, 0] = a0 + 0.01 * rng.uniform(-1, 1, (T, n))
    ro = env.rollout(torch.from_numpy(acts).cuda())
    s, integ = s0.copy(), np.zeros((n, 8))
    o = orc.error_obs_batch("coupled", s, goal, integ)
    integ = o["integ"]
    clipped = 0
    for t in range(T):
        o = orc.step_batch("coupled", s, acts[t].astype(np.float64), None, goal, integ)
        s, integ = o["state"], o["integ"]
        if t % 500 == 499 or t == T - 1:
            got = _np(ro["obs"][t]).astype(np.float64)
            ref = o["obs"][0].astype(np.float64)
            assert np.abs(got - ref).max() <= 2e-5
            clipped = max(clipped, int((np.abs(ref[:, 3:6]) == 1.0).sum() + (np.abs(ref[:, 19]) == 1.0).sum()))
    gi = _np(env.integ).astype(np.float64)
    err = np.abs(gi - integ) / np.maximum(np.abs(integ), 1.0)
    print(f"4000-step episode: integrators up to {np.abs(integ[:, [0, 1, 2, 6]]).max():.2f} (clip 3), worst relative error {err.max():.2e}, "
          f"{clipped} clipped observation entries at the end")
    assert np.abs(integ[:, 0:3]).max() > 4.0 and np.abs(integ[:, 6]).max() > 4.0   # far beyond the clip
    assert clipped >= n
    assert err.max() <= 1e-5
    assert grouped_rel_err(_np(env.get_current_state()), s) <= 1e-5


def test_set_state_rejects_rows_without_a_nearest_rotation():
    """det R <= 0 (a reflection) or non-finite attitude entries have no nearest rotation.  The C-ABI's qr_set_state leaves those
    envs untouched, counts them and updates every other env; qr_check_state makes the same count and writes nothing; the host
    wrapper validates with it first and raises before ANY env has changed."""
    n = 200
    env = _env("quad", n, layout="f64")
    env.reset("train")
    before = _np(env.get_current_state())
    s = before.copy()
    s[:, 0] += 0.125
    bad = np.zeros(n, bool)
    bad[[3, 77, 199]] = True
    s[3, 6:15] = np.diag([1.0, 1.0, -1.0]).T.reshape(9)          # reflection: det = -1
    s[77, 6:15] = 0.0                                             # singular
    s[199, 8] = np.nan
    with pytest.raises(ValueError, match="3 row"):
        env.set_state(s)
    assert np.array_equal(_np(env.get_current_state()), before)   # validated first (qr_check_state): nothing changed
    # the C-ABI entry points themselves: the dry run counts and writes nothing, the real call skips exactly the counted rows
    import ctypes as C
    rows, rej = torch.from_numpy(s).cuda(), torch.zeros(1, dtype=torch.int32, device="cuda")
    assert env._lib.qr_check_state(C.byref(env._cenv), rows.data_ptr(), None, rej.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int(rej.item()) == 3 and np.array_equal(_np(env.get_current_state()), before)
    rej.zero_()
    assert env._lib.qr_set_state(C.byref(env._cenv), rows.data_ptr(), None, rej.data_ptr(), None) == 0
    torch.cuda.synchronize()
    after = _np(env.get_current_state())
    assert int(rej.item()) == 3 and np.array_equal(after[bad], before[bad])                # rejected rows: previous state kept
    assert np.allclose(after[~bad, 0], s[~bad, 0]) and np.isfinite(after).all()
    env.set_state(s, mask=torch.from_numpy(~bad).cuda())          # masked out: not looked at


@pytest.mark.parametrize("kind", ["coupled", "decoupled"])
@pytest.mark.parametrize("layout", ["mixed", "f64"])
def test_error_obs_in_either_format_on_either_wrapper(kind, layout, golden):
    """get_norm_error_state(framework) (quad.py:421-466): the ARGUMENT selects the format — MONO [N,23] or MODUL [N,15] + [N,3] —
    on whichever wrapper class it is called; the integral terms advance the same way.  Golden: the reference's own method called
    with both frameworks on both wrapper classes (tests/golden/errobs_formats.npz, tools/gen_golden.py errobs); oracle beside it."""
    gd = golden("errobs_formats")
    n = gd["state"].shape[0]
    assert bytes(gd["quad_raises"].astype(np.uint8)).decode() == "AttributeError"
    for fw, okind in (("MONO", "coupled"), ("MODUL", "decoupled")):
        env = _env(kind, n, use_UDM=False, layout=layout)
        env.set_state(gd["state"], integ=gd["integ"])
        _set_goal(env, torch.from_numpy(gd["goal"]).float().cuda())
        rows = env.get_norm_error_state(fw)
        want = [gd[f"{kind}_{fw}_obs{k}"] for k in range(len(rows))]
        orc_out = orc.error_obs_batch(okind, gd["state"], gd["goal"], gd["integ"])
        assert [tuple(r.shape) for r in rows] == [w.shape for w in want]
        for r, w, o in zip(rows, want, orc_out["obs"]):
            assert np.abs(_np(r) - w).max() <= 2e-6 and np.abs(o - w).max() <= 1e-6
        assert np.abs(_np(env.integ) - gd[f"{kind}_{fw}_next_integ"]).max() <= 4e-6
        # the env's own format lands in its output buffers (and is what the next rollout_actor starts from); the other one does not
        own = env.framework == fw
        assert (env._last_obs is not None) == own
        if own:
            assert rows[0].data_ptr() == env._obs0.data_ptr()
        # the same through the torch custom op (one op for both entry points): identical rows from an identical start
        from gym_rotor_amd import torch_ops as ops
        twin = _env(kind, n, use_UDM=False, layout=layout)
        twin.set_state(gd["state"], integ=gd["integ"])
        _set_goal(twin, torch.from_numpy(gd["goal"]).float().cuda())
        out = [torch.empty_like(r) for r in rows]
        ops.error_obs(twin, fw, out if not own else None)
        got = out if not own else [twin._obs0] + ([twin._obs1] if twin._obs1 is not None else [])
        assert all(torch.equal(a, b) for a, b in zip(got, rows)) and torch.equal(twin._integ, env._integ)


@pytest.mark.parametrize("kind", KINDS)
def test_rollout_ignores_the_callers_reset_promise(kind):
    """QR_FLAG_CALLER_RESETS (QuadVecEnv(reset_on_done=True)) is a promise about what happens BETWEEN one-step launches: nobody can
    reset an env between two steps of a fused rollout, so a multi-step launch must keep the rate-adaptive kernel — envs that
    terminate mid-launch fly on with the accuracy guard in place.  A 300-step random-action rollout (the population tumbles far
    beyond |W| = w_adapt) with the flag set equals the free-running env's rollout bit for bit, and differs from what fixed
    substeps would give; the one-step launch still honours the flag."""
    n, T = 2048, 300
    adim = 5 if kind == "decoupled" else 4
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    acts = torch.rand(T, n, adim, device="cuda", generator=g) * 2 - 1
    if kind != "quad":   # the wrappers' torque commands are unscaled Nm: hold a per-env sign pattern so that the envs do spin up
        acts[:, :, 1:] = torch.sign(acts[0, :, 1:])[None]
    outs = []
    for kw in (dict(reset_on_done=True), dict(), dict(w_adapt=0.0)):
        e = _env(kind, n, seed=4, **kw)
        e.reset("train")
        if kind != "quad":
            e.get_norm_error_state()
        ro = e.rollout(acts)
        outs.append((_np(e.get_current_state()), _np(ro["reward"]), _np(ro["terminated"])))
    promised, free, fixed = outs
    assert np.abs(free[0][:, 15:18]).max() > 16.0                       # the population did leave the regime
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(promised, free))
    assert not np.array_equal(free[0], fixed[0], equal_nan=True)        # (adaptivity mattered on this rollout)
    # one-step launches: the promise selects the plain kernel = the arithmetic of the in-launch-reset mode
    a, b = _env(kind, n, seed=4, reset_on_done=True), _env(kind, n, seed=4, w_adapt=0.0)
    for e in (a, b):
        e.reset("train")
        if kind != "quad":
            e.get_norm_error_state()
        for t in range(40):
            e.step(acts[t])
    assert torch.equal(a.get_current_state(), b.get_current_state())


def test_default_autotune_reads_the_cache_and_times_only_on_request(tmp_path, monkeypatch):
    """The constructor's default (autotune=None) launches nothing: away from the launch rule's threshold nothing is read either; within
    +-25 % of it the choice RECORDED in the launch cache for (device, library, kind, size, substeps, ...) is taken — and without a
    record the compiled rule stays in force.  Timing is on request: autotune=True (or QR_AUTOTUNE=time) times both step()
    instantiations once (~0.2 s) and records the choice.  Whatever is picked, no bit changes; QR_AUTOTUNE=0 and an explicit
    `helper=` switch all of it off."""
    import json
    cache = tmp_path / "launch.json"
    monkeypatch.setenv("QR_LAUNCH_CACHE", str(cache))
    monkeypatch.delenv("QR_AUTOTUNE", raising=False)
    far = _env("quad", 64 * 700, seed=2, auto_reset=True)
    assert far.autotune_report is None and not cache.exists()                  # 700 tiles against 3328: the rule is unambiguous
    n = 64 * 3400                                                              # 3400 tiles: 2 % above Quad-v0's threshold
    cold = _env("quad", n, seed=2, auto_reset=True)
    assert cold.autotune_report is None and not cache.exists() and cold.kernel_info()[2] == 64   # no record: the compiled rule, nothing timed
    a = _env("quad", n, seed=2, auto_reset=True, autotune=True)
    assert a.autotune_report["helper"] > 0 and a.autotune_report["no_helper"] > 0
    entries = json.loads(cache.read_text())["entries"]
    assert len(entries) == 1 and list(entries.values())[0]["picked"] == a.autotune_report["picked"] and list(entries)[0].endswith("|s1")
    assert int(a._reset_count.abs().sum()) == 0 and int(a._episode.sum()) == 0  # tuning left no trace in the env
    b = _env("quad", n, seed=2, auto_reset=True)
    assert b.autotune_report["source"] == "cache" and b.autotune_report["picked"] == a.autotune_report["picked"]
    assert b.kernel_info() == a.kernel_info()
    # another instantiation family (two substeps: its own effective threshold, 2560 tiles, and its own cache entry)
    assert _env("quad", n, seed=2, auto_reset=True, substeps=2).autotune_report is None            # 3400 tiles are > 1.25 x 2560: not near
    monkeypatch.setenv("QR_AUTOTUNE", "time")
    s2 = _env("quad", 64 * 2600, seed=2, auto_reset=True, substeps=2)
    assert s2.autotune_report["source"] == "timed" and len(json.loads(cache.read_text())["entries"]) == 2
    assert s2.launch_plan()["help"] == (0 if s2.autotune_report["picked"] == "no_helper" else 1) or s2.autotune_report["picked"] == "default"
    monkeypatch.delenv("QR_AUTOTUNE")
    c = _env("quad", n, seed=2, auto_reset=True, helper=(a.kernel_info()[2] != 128))   # the OTHER instantiation, pinned
    assert c.autotune_report is None and c.kernel_info()[2] != a.kernel_info()[2]
    monkeypatch.setenv("QR_AUTOTUNE", "0")
    d = _env("quad", n, seed=2, auto_reset=True)
    assert d.autotune_report is None and d.kernel_info()[2] == 64              # the compiled rule: 3400 > 3328 tiles -> plain launch
    g = torch.Generator(device="cuda"); g.manual_seed(8)
    for e in (a, b, c, d):
        e.reset("train")
    for t in range(12):
        act = torch.rand(n, 4, device="cuda", generator=g) * 2 - 1
        outs = [e.step(act) for e in (a, b, c, d)]
        for o in outs[1:]:
            assert torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])
    for e in (b, c, d):
        assert torch.equal(e.get_current_state(), a.get_current_state()) and torch.equal(e._reset_count, a._reset_count)
