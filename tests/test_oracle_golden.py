"""CPU: pin the oracle (oracle/quad_oracle.py) against vectors produced by the reference itself
(tools/gen_golden.py, run in the build container) and against the reference-owned flight log.
These are the checks that justify using the oracle as the checker on the GPU box."""
import numpy as np
import pytest

from conftest import grouped_rel_err
from oracle import quad_oracle as orc

KINDS = orc.KINDS


@pytest.mark.parametrize("kind", KINDS)
def test_step_batch_matches_reference_onestep(kind, golden):
    d = golden(f"onestep_{kind}")
    o = orc.step_batch(kind, d["state"], d["action"], d["params"], d["goal"], d["integ"])
    assert np.abs(o["state"] - d["next_state"]).max() <= 1e-13      # fixed DOP853 step == adaptive solve_ivp
    assert np.abs(o["f"] - d["f"]).max() <= 1e-13 and np.abs(o["M"] - d["M"]).max() <= 1e-13
    for k, ob in enumerate(o["obs"]):
        ref = d[f"obs{k}"]
        if kind == "quad":
            assert np.abs(ob - ref).max() <= 1e-13
        else:
            assert ob.dtype == np.float32 and np.array_equal(ob, ref)  # float32 rows bit-identical
    assert np.abs(o["reward_raw"] - d["reward_raw"]).max() <= 4e-6   # float32 summation order (NEP 50)
    assert np.abs(o["reward"] - d["reward"]).max() <= 3e-7
    assert np.array_equal(o["done"], d["done"])
    assert np.abs(o["integ"] - d["next_integ"]).max() <= 1e-14


@pytest.mark.parametrize("kind", KINDS)
def test_refenv_single_matches_reference_onestep(kind, golden):
    """The reference-faithful single-env path (scipy DOP853 + ensure_SO3 in the RHS)."""
    d = golden(f"onestep_{kind}")
    for i in range(0, 512, 23):
        e = orc.RefEnv(kind, d["params"][i])
        e.state = d["state"][i].copy(); e.set_goal(d["goal"][i]); e.set_integ(d["integ"][i])
        obs, r, dn, trunc, info = e.step(d["action"][i])
        assert np.abs(e.state - d["next_state"][i]).max() <= 1e-14
        assert np.allclose(r, d["reward"][i], atol=1e-15) and list(dn) == list(d["done"][i])
        assert np.allclose(e.last_raw, d["reward_raw"][i], atol=0, rtol=0)
        assert trunc is False and info == {}
        for k, ob in enumerate([obs] if kind == "quad" else obs):
            assert np.array_equal(np.asarray(ob, dtype=np.float64), d[f"obs{k}"][i].astype(np.float64))
        assert np.abs(e.get_integ() - d["next_integ"][i]).max() <= 1e-15


@pytest.mark.parametrize("mode", ["free", "reset"])
@pytest.mark.parametrize("kind", KINDS)
def test_step_batch_trajectory_1000_steps(kind, mode, golden):
    d = golden(f"traj_{mode}_{kind}")
    T, n = d["actions"].shape[:2]
    state, integ = d["init_state"].copy(), np.zeros((n, 8))
    goal = d["goal"]
    if kind != "quad":  # first get_norm_error_state after reset (main.py:129) advances the integrators
        integ = _advance(kind, state, goal, integ)
    worst = 0.0
    for t in range(T):
        ra = d["reset_at"][t]
        if ra.any():
            state[ra] = d["states"][t][ra]
            integ[ra] = 0.0
            if kind != "quad":
                integ[ra] = _advance(kind, state[ra], goal[ra], integ[ra])
        assert np.abs(integ - d["integs"][t]).max() <= 1e-9
        o = orc.step_batch(kind, state, d["actions"][t].astype(np.float64), d["params"], goal, integ)
        state, integ = o["state"], o["integ"]
        nxt = d["reset_at"][t + 1]
        ref = d["states"][t + 1].copy(); ref[nxt] = state[nxt]
        worst = max(worst, grouped_rel_err(state, ref))
        assert np.array_equal(o["done"], d["dones"][t])
        assert np.abs(o["reward"] - d["rewards"][t]).max() <= 3e-7
    assert worst <= 1e-9


def _advance(kind, state, goal, integ):
    """get_norm_error_state's side effect on the integrators, via a zero-length 'step' of the
    observation part: reuse step_batch's obs code on the given state."""
    n = state.shape[0]
    x, W = state[:, 0:3], state[:, 15:18]
    R = np.swapaxes(state[:, 6:15].reshape(n, 3, 3), 1, 2)
    b1, b2, b3 = R[:, :, 0], R[:, :, 1], R[:, :, 2]
    b1d = goal[:, 6:9]
    ex = x / orc.X_LIM - goal[:, 0:3] / orc.X_LIM
    b1c = b1d - (b1d * b3).sum(1, keepdims=True) * b3
    eb1 = np.arctan2(-(b1c * b2).sum(1), (b1c * b1).sum(1)) / np.pi
    out = integ.copy()
    gx = -orc.ALPHA * integ[:, 0:3] + ex * orc.X_LIM
    out[:, 0:3] = integ[:, 0:3] + (integ[:, 3:6] + gx) * orc.DT / 2
    out[:, 3:6] = gx
    gb = -orc.BETA * integ[:, 6] + eb1 * np.pi
    out[:, 6] = integ[:, 6] + (integ[:, 7] + gb) * orc.DT / 2
    out[:, 7] = gb
    return out


def test_refenv_trajectory_prefix(golden):
    """Reference-faithful single env over the first 150 steps of a golden trajectory."""
    for kind in KINDS:
        d = golden(f"traj_free_{kind}")
        e = orc.RefEnv(kind, d["params"][1])
        e.state = d["init_state"][1].copy(); e.set_goal(d["goal"][1])
        if kind != "quad":
            e.get_norm_error_state()
        for t in range(150):
            obs, r, dn, _, _ = e.step(d["actions"][t, 1])
            assert np.abs(e.state - d["states"][t + 1, 1]).max() <= 1e-12
            assert np.allclose(r, d["rewards"][t, 1], atol=1e-15) and list(dn) == list(d["dones"][t, 1])


def test_flightlog_replay_oracle(golden):
    """results/MODUL_log_20250303_120200.dat: state[t], action[t] -> state[t+1] to the log's
    print precision (%.10f), eb1 / integral terms to float32-obs precision."""
    log = golden("flightlog_modul")["log"]
    act, state, cmd, nxt = log[:-1, 0:5], log[:-1, 5:23], log[:-1, 28:40], log[1:, 5:23]
    b3, b1c = state[:, 12:15], cmd[:, 6:9]
    b1d = b1c - (b1c[:, 2:3] / b3[:, 2:3]) * b3
    goal = np.concatenate([cmd[:, 0:6], b1d, cmd[:, 9:12]], 1)
    integ = np.zeros((len(act), 8)); integ[:, 0:3] = log[:-1, 23:26]; integ[:, 6] = log[:-1, 27]
    o = orc.step_batch("decoupled", state, act, None, goal, integ)
    assert np.abs(o["state"] - nxt).max() <= 1.2e-10
    eb1_next = o["obs"][1][:, 0].astype(np.float64) * np.pi
    assert np.abs(eb1_next - log[1:, 26]).max() <= 2e-7
    assert o["reward"].min() > 0.99 and not o["done"].any()  # smooth regime only


def test_unit_kats(golden):
    k = golden("kat_units")
    for v, H in zip(k["hat_in"], k["hat_out"]):
        assert np.array_equal(orc.hat(v), H)
    for Rin, Rout in zip(k["so3_in"], k["so3_out"]):
        assert np.abs(orc.ensure_SO3(Rin.copy()) - Rout).max() <= 1e-15
    assert np.abs(orc.ensure_SO3_batch(k["so3_in"]) - k["so3_out"]).max() <= 1e-14
    kept = np.abs(k["so3_in"] - k["so3_out"]).max(axis=(1, 2)) == 0
    assert kept.any() and (~kept).any()  # both branches exercised
    got = np.array([orc.norm_ang_btw_two_vectors(a, b) for a, b in zip(k["ang_a"], k["ang_b"])])
    assert np.abs(got - k["ang_out"]).max() <= 1e-15
    assert np.abs(np.stack([orc.get_current_b1(R) for R in k["b1_R"]]) - k["b1_out"]).max() <= 1e-15
    Re = np.stack([orc.euler_xyz_to_R(*e) for e in k["euler_in"]])
    assert np.abs(Re - k["euler_R"]).max() <= 1e-15  # scipy 'xyz' == Rz Ry Rx
    roll = np.degrees(np.arctan2(Re[:, 2, 1], Re[:, 2, 2])); pitch = np.degrees(-np.arcsin(Re[:, 2, 0]))
    assert np.abs(roll - k["euler_back_deg"][:, 0]).max() <= 1e-10 and np.abs(pitch - k["euler_back_deg"][:, 1]).max() <= 1e-10
    for mn in (14, 8, 7):
        assert np.array_equal(orc.interp01(k["interp_in"], -float(mn)), k[f"interp_out_{mn}"])
        assert np.abs(np.clip((k["interp_in"] + mn) / mn, 0, 1) - k[f"interp_out_{mn}"]).max() <= 1e-15
    c = k["constants"]
    assert (orc.REWARD_MIN, orc.REWARD_MIN_1, orc.REWARD_MIN_2) == (c[0], c[1], c[2]) == (-14, -8, -7)
    assert (orc.DT, orc.X_LIM, orc.V_LIM, orc.W_LIM, orc.EULER_LIM_DEG) == tuple(c[3:8])


@pytest.mark.parametrize("kind", KINDS)
def test_action_maps(kind, golden):
    k = golden("kat_units")
    P, A, S = k[f"act_{kind}_params"], k[f"act_{kind}_a"], k[f"act_{kind}_state"]
    dv = orc.derive(P)
    f, M = orc.action_map_batch(kind, A, S, dv)
    assert np.abs(f - k[f"act_{kind}_f"]).max() <= 1e-13 and np.abs(M - k[f"act_{kind}_M"]).max() <= 1e-13
    der = np.stack([dv.hover_force, dv.max_force, dv.avrg_act, dv.scale_act], 1)
    assert np.abs(der - k[f"act_{kind}_derived"]).max() <= 1e-14


def test_reset_sampler_distribution():
    rng = np.random.default_rng(0)
    s = orc.sample_reset_state(rng, 20000, "train")
    zero = np.abs(s[:, 0:6]).max(1) == 0
    assert abs(zero.mean() - 0.2) < 0.01
    assert np.abs(s[:, 0:3]).max() <= 0.6 and np.abs(s[:, 3:6]).max() <= 2.0 and np.abs(s[:, 15:18]).max() <= np.pi
    R = np.swapaxes(s[:, 6:15].reshape(-1, 3, 3), 1, 2)
    assert np.abs(np.swapaxes(R, 1, 2) @ R - np.eye(3)).max() < 1e-14
    p = orc.sample_params(rng, 20000)
    assert (np.abs(p / orc.NOMINAL_PARAMS - 1) <= np.array([.1, .1, .1, .1, .1, .05]) + 1e-12).all()
    assert np.array_equal(orc.sample_params(rng, 3, "eval"), np.tile(orc.NOMINAL_PARAMS, (3, 1)))


# ---------------------------------------------------------------------------------------
# PPO actor (caller side of the collection loop fused by qr_rollout_actor)
# ---------------------------------------------------------------------------------------
ACTOR_TAGS = {"coupled": ["coupled0"], "decoupled": ["decoupled0", "decoupled1"]}
ACTOR_FIELDS = ("fc1_w", "fc1_b", "fc2_w", "fc2_b", "mean_w", "mean_b", "log_std")


@pytest.mark.parametrize("tag", ["coupled0", "decoupled0", "decoupled1"])
def test_actor_oracle_vs_reference_module(golden, tag):
    """oracle/actor_oracle.py against the reference's MLP_Actor_PPO + Normal (torch float32)."""
    from oracle import actor_oracle as ao
    d = golden("actor_ppo")
    p = {n: d[f"{tag}_{n}"] for n in ACTOR_FIELDS}
    action, logprob, mean = ao.choose_action(p, d[f"{tag}_obs"], d[f"{tag}_eps"])
    assert np.abs(mean - d[f"{tag}_mean"]).max() <= 5e-7
    assert np.abs(action - d[f"{tag}_action"]).max() <= 5e-7
    assert (np.abs(action) == 1.0).sum() == (np.abs(d[f"{tag}_action"]) == 1.0).sum() > 0   # the clamp is exercised
    assert np.abs(logprob - d[f"{tag}_logprob"]).max() <= 2e-5   # (a - mean)^2 / (2 std^2) in float32, |z| up to 16
    det, none, _ = ao.choose_action(p, d[f"{tag}_obs"], None)
    assert none is None and np.abs(det - np.clip(d[f"{tag}_mean"], -1, 1)).max() <= 5e-7


@pytest.mark.parametrize("kind", ["coupled", "decoupled"])
def test_actor_loop_oracle_vs_reference(golden, kind):
    """Closed loop: oracle env + oracle actor against the reference env stepped by the reference's
    actor modules (main.py:141-166), 4 envs x 200 steps, injected action noise."""
    from oracle import actor_oracle as ao
    d = golden(f"actorloop_{kind}")
    nag = orc.N_AGENTS[kind]
    actors = [{n: d[f"actor{k}_{n}"] for n in ACTOR_FIELDS} for k in range(nag)]
    adims = [a["mean_w"].shape[0] for a in actors]
    T, n = d["eps"].shape[:2]
    s, integ = d["init_state"].copy(), np.zeros((n, 8))
    o = orc.error_obs_batch(kind, s, d["goal"], integ)
    obs, integ = o["obs"], o["integ"]
    worst = {"obs": 0.0, "act": 0.0, "logp": 0.0, "rwd": 0.0, "state": 0.0}
    for t in range(T):
        col, act, logp = 0, [], []
        for k in range(nag):
            worst["obs"] = max(worst["obs"], np.abs(np.asarray(obs[k], np.float64) - d[f"obs{k}"][t]).max())
            a_, l_, _ = ao.choose_action(actors[k], np.asarray(obs[k], np.float32), d["eps"][t, :, col:col + adims[k]])
            act.append(a_); logp.append(l_); col += adims[k]
        act, logp = np.concatenate(act, -1), np.concatenate(logp, -1)
        worst["act"] = max(worst["act"], np.abs(act - d["actions"][t]).max())
        worst["logp"] = max(worst["logp"], np.abs(logp - d["logprobs"][t]).max())
        # feed the REFERENCE's float32 action (what its env saw) so that the env parity is not
        # polluted by float32-vs-float64 actor rounding
        o = orc.step_batch(kind, s, d["actions"][t].astype(np.float64), d["params"], d["goal"], integ)
        s, integ, obs = o["state"], o["integ"], o["obs"]
        worst["rwd"] = max(worst["rwd"], np.abs(o["reward"] - d["rewards"][t]).max())
        worst["state"] = max(worst["state"], np.abs(s - d["states"][t + 1]).max())
        assert np.array_equal(o["done"], d["dones"][t])
    assert worst["state"] <= 1e-9 and worst["obs"] <= 1e-6 and worst["rwd"] <= 1e-6, worst
    assert worst["act"] <= 2e-6 and worst["logp"] <= 5e-5, worst


@pytest.mark.parametrize("tag", ["coupled0", "decoupled0", "decoupled1"])
def test_sac_actor_oracle_vs_reference_module(golden, tag):
    """oracle sac_sample against the reference's MLP_Actor_SAC forward / sample (torch float32)."""
    from oracle import actor_oracle as ao
    d = golden("actor_sac")
    p = {n: d[f"{tag}_{n}"] for n in ("fc1_w", "fc1_b", "fc2_w", "fc2_b", "mean_w", "mean_b", "log_std_w", "log_std_b")}
    action, logprob, mean, log_std = ao.sac_sample(p, d[f"{tag}_obs"], d[f"{tag}_eps"])
    scale = np.maximum(np.abs(d[f"{tag}_mean"]), 1.0)
    assert (np.abs(mean - d[f"{tag}_mean"]) / scale).max() <= 1e-5   # float32 torch sums over inputs up to 60 in magnitude
    assert np.abs(log_std - d[f"{tag}_log_std"]).max() <= 2e-5 and log_std.min() == -20.0
    ok = np.abs(d[f"{tag}_log_std"]) < 1.99        # away from the clamp edges a 1e-6 shift of the head moves nothing else
    assert np.abs(action - d[f"{tag}_action"])[ok].max() <= 5e-5
    inner = ok & (np.abs(d[f"{tag}_action"]) < 0.99)   # 1 - a^2 cancels in float32 near saturation
    assert np.abs(logprob - d[f"{tag}_logprob"])[inner].max() <= 5e-4
    det, none, _, _ = ao.sac_sample(p, d[f"{tag}_obs"], None)
    assert none is None and np.abs(det - np.tanh(d[f"{tag}_mean"].astype(np.float64))).max() <= 2e-6


def test_error_obs_formats_oracle_vs_reference(golden):
    """get_norm_error_state(framework): the argument, not the wrapper class, selects MONO / MODUL (quad.py:452-466).  The oracle's
    batch form and its single-env RefEnv against the reference's own outputs in both formats (tests/golden/errobs_formats.npz)."""
    gd = golden("errobs_formats")
    for fw, okind in (("MONO", "coupled"), ("MODUL", "decoupled")):
        out = orc.error_obs_batch(okind, gd["state"], gd["goal"], gd["integ"])
        for cls in ("coupled", "decoupled"):
            for k, o in enumerate(out["obs"]):
                assert o.dtype == np.float32 and np.array_equal(o, gd[f"{cls}_{fw}_obs{k}"]) or np.abs(o - gd[f"{cls}_{fw}_obs{k}"]).max() <= 1.2e-7
            assert np.abs(out["integ"] - gd[f"{cls}_{fw}_next_integ"]).max() <= 1e-12
    env = orc.RefEnv("coupled")
    for i in (0, 7, 100):
        for fw in ("MONO", "MODUL"):
            env.state = gd["state"][i].copy(); env.set_goal(gd["goal"][i]); env.set_integ(gd["integ"][i])
            rows = env.get_norm_error_state(fw)
            for k, r in enumerate(rows):
                assert np.abs(r - gd[f"coupled_{fw}_obs{k}"][i]).max() <= 1.2e-7
