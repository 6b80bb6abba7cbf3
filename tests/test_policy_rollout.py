"""qr_rollout_actor: the PPO collection loop (env + actor) in one launch, against the reference's
own actor modules and env (tests/golden/actor_ppo.npz, actorloop_*.npz) and against the oracle."""
import numpy as np
import pytest
import torch

from conftest import grouped_rel_err
from oracle import actor_oracle as ao
from oracle import quad_oracle as orc

pytestmark = pytest.mark.gpu
FIELDS = ("fc1_w", "fc1_b", "fc2_w", "fc2_b", "mean_w", "mean_b", "log_std")


def _np(t):
    return t.detach().cpu().numpy()


def _env(kind, n, **kw):
    from gym_rotor_amd import QuadVecEnv
    return QuadVecEnv(kind, n, device="cuda", obs_rows=True, **kw)


def _actor(d, prefix):
    from gym_rotor_amd import ActorParams
    return ActorParams(*[torch.from_numpy(np.ascontiguousarray(d[f"{prefix}_{n}"], dtype=np.float32)).cuda() for n in FIELDS])


def _round_trip(env, state):
    env.set_state(state)
    return _np(env.get_current_state())


@pytest.mark.parametrize("kind", ["coupled", "decoupled"])
def test_actor_in_kernel_vs_reference_module(golden, kind):
    """One policy step on the reference's recorded (obs, eps): the in-kernel MLP, sampling, clamp
    and log-prob against torch's MLP_Actor_PPO / Normal (float32), 256 rows per agent."""
    d = golden("actor_ppo")
    tags = [f"{kind}0"] if kind == "coupled" else [f"{kind}0", f"{kind}1"]
    n = d[f"{tags[0]}_obs"].shape[0]
    env = _env(kind, n)
    env.reset("train")
    actors = [_actor(d, t) for t in tags]
    obs = [torch.from_numpy(d[f"{t}_obs"]).cuda() for t in tags]
    eps = torch.from_numpy(np.concatenate([d[f"{t}_eps"] for t in tags], 1)[None].copy()).cuda()
    out = env.rollout_actor(actors, 1, obs=obs, noise=eps)
    act, logp = _np(out["action"][0]), _np(out["logprob"][0])
    want_a = np.concatenate([d[f"{t}_action"] for t in tags], 1)
    want_l = np.concatenate([d[f"{t}_logprob"] for t in tags], 1)
    assert np.abs(act - want_a).max() <= 1e-6
    assert np.array_equal(np.abs(act) == 1.0, np.abs(want_a) == 1.0)
    assert np.abs(logp - want_l).max() <= 5e-5      # (a - mean)^2 / (2 std^2) with |z| up to 16 in float32
    det = env.rollout_actor(actors, 1, obs=obs, deterministic=True)
    want_m = np.clip(np.concatenate([d[f"{t}_mean"] for t in tags], 1), -1, 1)
    assert np.abs(_np(det["action"][0]) - want_m).max() <= 1e-6
    ls = np.concatenate([d[f"{t}_log_std"] for t in tags])
    assert np.abs(_np(det["logprob"][0]) - (-ls - ao.LOG_SQRT_2PI)).max() <= 1e-6


@pytest.mark.parametrize("kind", ["coupled", "decoupled"])
def test_closed_loop_vs_reference_env_and_actor(golden, kind):
    """The whole loop against the reference: its wrapper env stepped by its actor modules for 200
    steps (4 envs, injected noise), reproduced by ONE qr_rollout_actor launch."""
    d = golden(f"actorloop_{kind}")
    nag = orc.N_AGENTS[kind]
    T, n = d["eps"].shape[:2]
    env = _env(kind, n, layout="f64", want_raw_reward=False)
    env.set_state(d["init_state"], integ=np.zeros((n, 8)), params=d["params"])
    env.get_norm_error_state()
    actors = [_actor(d, f"actor{k}") for k in range(nag)]
    out = env.rollout_actor(actors, T, noise=torch.from_numpy(d["eps"]).cuda())
    obs = [out["obs0"]] if nag == 1 else [out["obs0"], out["obs1"]]
    for k in range(nag):
        assert np.abs(_np(obs[k]) - d[f"obs{k}"][1:]).max() <= 2e-5   # |ex| reaches 5 (free run): 5e-7 x magnitude
    assert np.abs(_np(out["action"]) - d["actions"]).max() <= 2e-6
    assert np.abs(_np(out["logprob"]) - d["logprobs"]).max() <= 5e-5
    assert np.abs(_np(out["reward"]) - d["rewards"]).max() <= 1e-5
    assert np.array_equal(_np(out["terminated"]), d["dones"])
    assert grouped_rel_err(_np(env.get_current_state()), d["states"][T]) <= 1e-6


@pytest.mark.parametrize("kind", ["coupled", "decoupled"])
def test_closed_loop_vs_oracle_1000_envs(kind):
    """1000 envs x 64 steps, random actors, injected noise: GPU loop vs oracle env + oracle actor."""
    from gym_rotor_amd import random_actors
    n, T = 1000, 64
    rng = np.random.default_rng(31 + len(kind))
    env = _env(kind, n, layout="f64")
    state = _round_trip(env, orc.sample_reset_state(rng, n).astype(np.float32).astype(np.float64))
    params = orc.sample_params(rng, n).astype(np.float32).astype(np.float64)
    env.set_state(state, integ=np.zeros((n, 8)), params=params)
    env.get_norm_error_state()
    gen = torch.Generator("cuda").manual_seed(5)
    actors = random_actors(kind, "cuda", generator=gen, log_std=-1.0)
    for a in actors:
        a.mean_w.mul_(10.0)
    A = env.action_dim
    eps = rng.standard_normal((T, n, A)).astype(np.float32)
    out = env.rollout_actor(actors, T, noise=torch.from_numpy(eps).cuda())
    pw = [{f: _np(getattr(a, f)).astype(np.float64) for f in FIELDS} for a in actors]
    adims = [p["mean_w"].shape[0] for p in pw]
    o = orc.error_obs_batch(kind, state, None, np.zeros((n, 8)))
    s, integ, obs = state, o["integ"], o["obs"]
    worst_a = 0.0
    for t in range(T):
        col, acts = 0, []
        for k, p in enumerate(pw):
            a_, _, _ = ao.choose_action(p, obs[k], eps[t, :, col:col + adims[k]])
            acts.append(a_); col += adims[k]
        act = np.concatenate(acts, 1).astype(np.float32)
        worst_a = max(worst_a, np.abs(act - _np(out["action"][t])).max())
        o = orc.step_batch(kind, s, act.astype(np.float64), params, None, integ)
        s, integ, obs = o["state"], o["integ"], o["obs"]
    assert worst_a <= 2e-5                                           # closed loop: actor rounding feeds back
    assert grouped_rel_err(_np(env.get_current_state()), s) <= 2e-6
    assert np.abs(_np(out["obs0"][T - 1]) - obs[0]).max() <= 2e-5


def test_in_kernel_noise_statistics_and_streams():
    """Philox + Box-Muller action noise: standard normal, independent across envs / steps /
    components, reproducible, independent of sharding and of how a horizon is split into calls."""
    from gym_rotor_amd import random_actors
    n, T = 4096, 32
    actors = random_actors("decoupled", "cuda", generator=torch.Generator("cuda").manual_seed(1), log_std=0.0)
    for a in actors:
        a.mean_w.zero_(); a.mean_b.zero_()       # mean = 0, std = 1: the unclamped action IS the noise

    def run(n_envs, offset=0, splits=(T,), seed=3, **kw):
        env = _env("decoupled", n_envs, seed=seed, env_offset=offset, **kw)
        env.reset("train")
        env.get_norm_error_state()
        acts = [env.rollout_actor(actors, k, max_action=1e6)["action"] for k in splits]
        return _np(torch.cat(acts, 0))

    z = run(n)
    assert z.shape == (T, n, 5)
    assert abs(z.mean()) < 4 / np.sqrt(z.size) and abs(z.var() - 1) < 6 * np.sqrt(2 / z.size)
    assert abs((z ** 3).mean()) < 0.02 and abs((z ** 4).mean() - 3) < 0.06
    assert np.abs(z).max() < 6.5
    c = np.corrcoef(np.stack([z[:-1, :, 0].ravel(), z[1:, :, 0].ravel(), z[:-1, :, 1].ravel(), z[:-1, :, 4].ravel(),
                              np.roll(z[:-1, :, 0], 1, axis=1).ravel()]))
    assert np.abs(c - np.eye(5)).max() < 0.02
    assert np.array_equal(z, run(n))                                   # reproducible
    assert not np.array_equal(z, run(n, seed=4))                       # seeded
    assert np.array_equal(z[:, 1000:1600], run(600, offset=1000))      # a shard draws its global envs' numbers
    assert np.array_equal(z, run(n, splits=(8, 24)))                   # step_base advances across calls
    # with in-launch resets the launch carries a helper wavefront per tile, which samples the noise a step ahead: the
    # same numbers (the mean is zero here, so the action is the noise whatever the env does)
    assert np.array_equal(z, run(n, auto_reset=True)) and np.array_equal(z, run(n, splits=(5, 27), auto_reset=True))


def test_policy_rollout_with_auto_reset_and_storage():
    """PPO-shaped use: auto-reset on, rows written straight into a RolloutStorage horizon; the
    rollout equals the same loop driven from the host (torch actor + env.step) bit-for-bit in the
    env outputs given the same actions."""
    from gym_rotor_amd import random_actors
    n, T = 2048, 96
    actors = random_actors("coupled", "cuda", generator=torch.Generator("cuda").manual_seed(2), log_std=-0.5)
    from gym_rotor_amd import RolloutStorage
    env = _env("coupled", n, seed=11, auto_reset=True)
    env.reset("train")
    first_obs = env.get_norm_error_state()[0].clone()
    st = RolloutStorage(env, T)
    out = st.collect(env, actors)
    assert out["obs0"].data_ptr() == st.obs[0][1].data_ptr() and out["action"].data_ptr() == st.act[0].data_ptr()
    assert torch.equal(st.obs[0][0], first_obs)
    assert torch.isfinite(out["obs0"]).all() and torch.isfinite(out["logprob"]).all()
    assert (out["action"].abs() <= 1).all()
    done = out["terminated"][..., 0]
    assert 2e-4 < done.float().mean() < 0.2          # episodes end and restart inside the launch
    assert (out["reward"][..., 0][done] == -1).all()
    # replay the recorded actions through qr_rollout on a twin env: identical env outputs
    twin = _env("coupled", n, seed=11, auto_reset=True)
    twin.reset("train")
    twin.get_norm_error_state()
    ro = twin.rollout(out["action"])
    for k in ("obs0", "reward", "terminated"):
        assert torch.equal(ro[k], out[k]), k
    assert torch.equal(twin.get_current_state(), env.get_current_state())
    # the actor saw obs[t-1]: recompute the means with torch from the stored observations
    a = actors[0]
    x = st.obs[0][:-1]
    h = torch.relu(x @ a.fc1_w.T + a.fc1_b); h = torch.relu(h @ a.fc2_w.T + a.fc2_b)
    mean = torch.tanh(h @ a.mean_w.T + a.mean_b)
    z = (out["action"] - mean) / a.log_std.exp()
    inside = out["action"].abs() < 1
    lp = -0.5 * z * z - a.log_std - ao.LOG_SQRT_2PI
    assert (lp - out["logprob"])[inside].abs().max() <= 2e-4
    assert abs(float(z[inside].mean())) < 0.01          # (the variance is that of a normal truncated by the clamp)


def test_rollout_actor_api_errors():
    from gym_rotor_amd import QuadVecEnv, random_actors
    env = _env("coupled", 64)
    env.reset("train")
    actors = random_actors("coupled", "cuda")
    with pytest.raises(ValueError):
        env.rollout_actor(actors, 4)                                   # no current observation yet
    env.get_norm_error_state()
    with pytest.raises(ValueError):
        env.rollout_actor(random_actors("decoupled", "cuda"), 4)       # wrong number / sizes of actors
    with pytest.raises(ValueError):
        env.rollout_actor(actors, 0)
    with pytest.raises(ValueError):
        env.rollout_actor(actors, 4, noise=torch.zeros(4, 64, 5, device="cuda"))
    with pytest.raises(ValueError):
        QuadVecEnv("quad", 64, device="cuda", obs_rows=True).rollout_actor(actors, 4)
    out = env.rollout_actor(actors, 4)
    assert out["action"].shape == (4, 64, 4) and out["obs0"].shape == (4, 64, 23)
    # caller-owned outputs are written by raw pointer: anything but the exact contiguous tensor is refused
    for key, bad in (("action", torch.empty(4, 64, 8, device="cuda")[..., :4]), ("reward", out["reward"].double()),
                     ("obs0", out["obs0"][:3]), ("terminated", out["terminated"].cpu())):
        with pytest.raises(ValueError):
            env.rollout_actor(actors, 4, out={**{k: v for k, v in out.items() if k != "obs"}, key: bad})
    with pytest.raises(ValueError):
        env.rollout(out["action"], out={k: v for k, v in out.items() if k not in ("obs", "reward")})
    with pytest.raises(ValueError):
        env.step(out["action"][0], out={"obs0": out["obs0"][0], "reward": out["reward"][0, :, 0], "terminated": out["terminated"][0]})


def test_checkpoint_resume_reproduces_the_rollout():
    """state_dict()/load_state_dict() carry the env buffers, episode counters, the position of the
    action-noise stream and the current observation: a resumed env continues bit-for-bit."""
    from gym_rotor_amd import random_actors
    n = 1500
    actors = random_actors("decoupled", "cuda", generator=torch.Generator("cuda").manual_seed(8), log_std=-0.7)
    env = _env("decoupled", n, seed=21, auto_reset=True, goal_mode=1)
    env.reset("train")
    env.get_desired(store_goal=True)
    env.get_norm_error_state()
    env.rollout_actor(actors, 20)
    sd = env.state_dict()
    want = env.rollout_actor(actors, 30)
    twin = _env("decoupled", n, seed=999, auto_reset=True, goal_mode=1)   # different seed: must come from the checkpoint
    twin.load_state_dict(sd)
    got = twin.rollout_actor(actors, 30)
    for k in ("obs0", "obs1", "action", "logprob", "reward", "terminated"):
        assert torch.equal(got[k], want[k]), k
    assert torch.equal(twin.get_current_state(), env.get_current_state())


@pytest.mark.parametrize("kind", ["coupled", "decoupled"])
def test_td3_actor_in_kernel_vs_reference_module(golden, kind):
    """TD3.choose_action (td3.py:93-96): clip(MLP_Actor_TD3(obs) + sigma eps) is the same kernel path
    with mean_linear = fc3 and log_std = log sigma (ActorParams.from_td3_module)."""
    import types
    from gym_rotor_amd import ActorParams
    d = golden("actor_td3")
    tags = [f"{kind}0"] if kind == "coupled" else [f"{kind}0", f"{kind}1"]
    n = d[f"{tags[0]}_obs"].shape[0]
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    mods = [types.SimpleNamespace(**{f"fc{i}": types.SimpleNamespace(weight=types.SimpleNamespace(data=cu(d[f"{t}_fc{i}_w"]), shape=d[f"{t}_fc{i}_w"].shape, device=torch.device("cuda", 0)),
                                                                   bias=types.SimpleNamespace(data=cu(d[f"{t}_fc{i}_b"]))) for i in (1, 2, 3)}) for t in tags]
    env = _env(kind, n)
    env.reset("train")
    obs = [cu(d[f"{t}_obs"]) for t in tags]
    eps = cu(np.concatenate([d[f"{t}_eps"] for t in tags], 1)[None])
    noisy = env.rollout_actor([ActorParams.from_td3_module(m, float(d["sigma"])) for m in mods], 1, obs=obs, noise=eps)
    want = np.concatenate([d[f"{t}_action"] for t in tags], 1)
    assert np.abs(_np(noisy["action"][0]) - want).max() <= 1e-6
    greedy = env.rollout_actor([ActorParams.from_td3_module(m, 0.0) for m in mods], 1, obs=obs, deterministic=True)
    assert np.abs(_np(greedy["action"][0]) - np.concatenate([d[f"{t}_mean"] for t in tags], 1)).max() <= 1e-6


@pytest.mark.parametrize("kind", ["coupled", "decoupled"])
def test_sac_actor_in_kernel_vs_reference_module(golden, kind):
    """MLP_Actor_SAC.sample (sac_mlp.py:60-82) in the kernel: state-dependent log_std head (second MFMA
    head), clamp to [-20, 2], action = tanh(mean + std eps), squashed log-prob; eval = tanh(mean)."""
    from gym_rotor_amd import ActorParams, _lib
    d = golden("actor_sac")
    tags = [f"{kind}0"] if kind == "coupled" else [f"{kind}0", f"{kind}1"]
    n = d[f"{tags[0]}_obs"].shape[0]
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    actors = [ActorParams(cu(d[f"{t}_fc1_w"]), cu(d[f"{t}_fc1_b"]), cu(d[f"{t}_fc2_w"]), cu(d[f"{t}_fc2_b"]), cu(d[f"{t}_mean_w"]),
                          cu(d[f"{t}_mean_b"]), None, cu(d[f"{t}_log_std_w"]), cu(d[f"{t}_log_std_b"]), _lib.ACTOR_TANH_SAMPLE) for t in tags]
    env = _env(kind, n)
    env.reset("train")
    obs = [cu(d[f"{t}_obs"]) for t in tags]
    eps = cu(np.concatenate([d[f"{t}_eps"] for t in tags], 1)[None])
    out = env.rollout_actor(actors, 1, obs=obs, noise=eps)
    want_a = np.concatenate([d[f"{t}_action"] for t in tags], 1)
    want_l = np.concatenate([d[f"{t}_logprob"] for t in tags], 1)
    ls = np.concatenate([d[f"{t}_log_std"] for t in tags], 1)
    ok = np.abs(ls) < 1.99
    assert np.abs(_np(out["action"][0]) - want_a)[ok].max() <= 3e-5
    inner = ok & (np.abs(want_a) < 0.99)
    assert np.abs(_np(out["logprob"][0]) - want_l)[inner].max() <= 1e-3
    det = env.rollout_actor(actors, 1, obs=obs, deterministic=True)
    want_m = np.tanh(np.concatenate([d[f"{t}_mean"] for t in tags], 1).astype(np.float64))
    # Two float32 evaluations of one network (pre-activations up to 65 here) differ by their summation order: the reference
    # module's own output sits up to 1.9e-6 from the float64 evaluation of the same weights (checked below), and so may the
    # kernel's.  Bounds: 4e-6 against the exact value, 6e-6 against the module's float32 output.
    def exact(t):
        f = lambda k: d[f"{t}_{k}"].astype(np.float64)
        h = np.maximum(f("obs") @ f("fc1_w").T + f("fc1_b"), 0)
        h = np.maximum(h @ f("fc2_w").T + f("fc2_b"), 0)
        return np.tanh(h @ f("mean_w").T + f("mean_b"))
    want_x = np.concatenate([exact(t) for t in tags], 1)
    assert np.abs(want_m - want_x).max() <= 2.5e-6
    got = _np(det["action"][0])
    assert np.abs(got - want_x).max() <= 4e-6
    assert np.abs(got - want_m).max() <= 6e-6


@pytest.mark.parametrize("kind", ["coupled", "decoupled"])
@pytest.mark.parametrize("algo", ["ppo", "sac"])
def test_actor_rollout_helper_launch_equals_the_plain_one(kind, algo):
    """qr_rollout_actor picks a helper-wave instantiation (noise, reset pool and observation rows on a second wavefront per tile) for
    grids up to 1024 tiles — since round 5 also for the general actor form (SAC's log_std head and tanh-of-sample rule).  The same
    horizon under helper_rollout=True / False: every output row (observations, actions, log-probs, rewards, flags, terminal
    observations), the state, the integrators, the parameters and every counter agree to the bit — through in-launch resets,
    in-kernel noise, a time limit and a ragged last tile."""
    from gym_rotor_amd import random_actors
    n, T = 64 * 37 + 21, 48
    actors = random_actors(kind, "cuda", generator=torch.Generator("cuda").manual_seed(4), log_std=-0.3, algo=algo)
    outs, envs = [], []
    for h in (True, False):
        env = _env(kind, n, seed=9, auto_reset=True, max_episode_steps=30, final_obs=True, helper_rollout=h)
        env.reset("train")
        env.get_norm_error_state()
        assert env.kernel_info(T)[2] == (128 if h else 64)          # (the rollout family's own override bit)
        o1 = env.rollout_actor(actors, T)
        o2 = env.rollout_actor(actors, 7)                            # a second launch continues the noise stream and the counters
        outs.append((o1, o2)); envs.append(env)
    for a, b in zip(outs[0], outs[1]):
        for k in ("obs0", "obs1", "action", "logprob", "reward", "terminated", "truncated"):
            if k in a:
                assert torch.equal(a[k], b[k]), k
    assert outs[0][0]["truncated"].any() and int(envs[0]._episode.sum()) >= n     # (the time limit alone ended every episode once)
    for name in ("_pos_vel", "_att_rate", "_integ", "_params", "_episode", "_steps", "_reset_count"):
        assert torch.equal(getattr(envs[0], name), getattr(envs[1], name)), name


@pytest.mark.parametrize("kind", ["coupled", "decoupled"])
@pytest.mark.parametrize("helper", [True, False])
def test_actor_rollout_equals_steps_on_the_actions_it_sampled(kind, helper):
    """With in-launch resets rate adaptivity cannot trigger (every env that leaves the regime is re-sampled at the end of that
    step; w_adapt >= 2.5 W_lim), so qr_rollout_actor runs the plain stage arithmetic — the instantiation family of qr_step — and
    a horizon of it IS T calls of env.step() on the actions it sampled: observations, rewards, flags, state, integrators,
    parameters and counters agree to the bit, through in-launch resets and a time limit, with and without the helper wave."""
    from gym_rotor_amd import random_actors
    n, T = 64 * 9 + 5, 40
    actors = random_actors(kind, "cuda", generator=torch.Generator("cuda").manual_seed(6), log_std=-0.3)
    a = _env(kind, n, seed=11, auto_reset=True, max_episode_steps=25, helper_rollout=helper)
    a.reset("train")
    a.get_norm_error_state()
    po = a.rollout_actor(actors, T)
    b = _env(kind, n, seed=11, auto_reset=True, max_episode_steps=25)
    b.reset("train")
    b.get_norm_error_state()
    for t in range(T):
        o, r, d, tr, _ = b.step(po["action"][t])
        obs = [o] if isinstance(o, torch.Tensor) else list(o)
        for j, ob in enumerate(obs):
            assert torch.equal(ob, po[f"obs{j}"][t]), (t, j)
        assert torch.equal(torch.as_tensor(r).reshape(-1), po["reward"][t].reshape(-1)), t
        assert torch.equal(torch.as_tensor(d).reshape(-1), po["terminated"][t].reshape(-1)), t
        assert torch.equal(tr.reshape(-1), po["truncated"][t].reshape(-1)), t
    assert po["truncated"].any() and int(a._episode.sum()) >= n           # (the time limit alone ended every episode once)
    for name in ("_pos_vel", "_att_rate", "_integ", "_params", "_episode", "_steps", "_reset_count"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name


@pytest.mark.parametrize("kind", ["coupled", "decoupled"])
def test_actor_rollout_in_chunks_of_resident_tiles_equals_the_single_launch(kind):
    """Beyond 1024 tiles qr_rollout_actor runs its helper-wave instantiation over chunks of 1024 tiles, one launch after the other
    (QrEnv flag bit QR_FLAG_NO_HELPER_ROLLOUT: the plain instantiation over the whole grid in one launch).  The split changes no
    result: a ragged grid of 2.6 chunks, in-launch resets, a time limit, in-kernel noise, two launches in a row."""
    from gym_rotor_amd import random_actors
    n, T = 64 * 2700 + 29, 24
    actors = random_actors(kind, "cuda", generator=torch.Generator("cuda").manual_seed(12), log_std=-0.4)
    outs, envs = [], []
    for h in (None, False):
        env = _env(kind, n, seed=21, auto_reset=True, max_episode_steps=15, helper_rollout=h)
        env.reset("train")
        env.get_norm_error_state()
        eps = torch.randn(4, n, env.action_dim, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
        outs.append((env.rollout_actor(actors, T), env.rollout_actor(actors, 5), env.rollout_actor(actors, 4, noise=eps),
                     env.rollout_actor(actors, 3, deterministic=True)))          # in-kernel noise, injected draws, the mean action
        envs.append(env)
    for a, b in zip(outs[0], outs[1]):
        for k in ("obs0", "obs1", "action", "logprob", "reward", "terminated", "truncated"):
            if k in a:
                assert torch.equal(a[k], b[k]), k
    assert outs[0][0]["truncated"].any()
    for name in ("_pos_vel", "_att_rate", "_integ", "_params", "_episode", "_steps", "_reset_count"):
        assert torch.equal(getattr(envs[0], name), getattr(envs[1], name)), name
