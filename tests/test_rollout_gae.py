"""Rollout storage + GAE: oracle pinned on the reference's own GAE lines
(CPU), the HIP reverse-scan kernel and the storage/step plumbing against it (GPU), and the
all-reduced normalisation statistics over two gloo ranks (CPU)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import gae_oracle as go


@pytest.mark.parametrize("case", ["a", "b"])
def test_gae_oracle_matches_reference(case, golden):
    d = golden("gae")
    gamma, lam = d[f"{case}_gamma_lam"]
    val = d[f"{case}_value"]
    adv, tgt = go.gae(d[f"{case}_reward"], d[f"{case}_done"], val[:-1], val[1:], gamma, lam)
    assert np.abs(tgt - d[f"{case}_td_target"]).max() <= 5e-6      # reference runs the scan in float32/python floats
    assert np.abs(adv - d[f"{case}_advantage"]).max() <= 5e-6
    for c in range(adv.shape[1]):                                   # ppo.py:147, one env horizon per call
        assert np.abs(go.normalize(adv[:, c]) - d[f"{case}_normalized_per_column"][:, c]).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("T,M", [(64, 48), (7, 1), (33, 1000), (128, 65536)])
def test_gae_kernel_vs_oracle(T, M, golden):
    from gym_rotor_amd import _lib
    lib = _lib.load()
    if (T, M) == (64, 48):
        d = golden("gae")
        rwd, done, val = d["a_reward"], d["a_done"], d["a_value"]
        gamma, lam = d["a_gamma_lam"]
    else:
        rng = np.random.default_rng(T * M)
        rwd = rng.uniform(-1, 1, (T, M)).astype(np.float32)
        done = rng.uniform(size=(T, M)) < 0.05
        val = rng.normal(0, 2, (T + 1, M)).astype(np.float32)
        gamma, lam = 0.99, 0.9
    dev = "cuda"
    r, dn, v = (torch.from_numpy(x).to(dev).contiguous() for x in (rwd, done, val))
    adv = torch.empty(T, M, device=dev); tgt = torch.empty(T, M, device=dev)
    grid = (M + 63) // 64
    part = torch.zeros(grid, 2, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    # (a) Vnext = value[t+1] (bootstrap row), (b) explicit next_value
    for nv in (None, v[1:].contiguous()):
        rc = lib.qr_gae(r.data_ptr(), dn.data_ptr(), v.data_ptr(), None if nv is None else nv.data_ptr(), T, M,
                        float(gamma), float(lam), adv.data_ptr(), tgt.data_ptr(), part.data_ptr(), s)
        assert rc == 0
        a_ref, t_ref = go.gae(rwd, done, val[:-1], val[1:], gamma, lam)
        scale = max(1.0, np.abs(a_ref).max())
        assert np.abs(adv.cpu().numpy() - a_ref).max() <= 2e-6 * scale
        assert np.abs(tgt.cpu().numpy() - t_ref).max() <= 2e-6 * scale
        tot = part.sum(0).cpu().numpy()
        assert abs(tot[0] - a_ref.sum()) <= 1e-4 * max(1.0, abs(a_ref).sum()) and abs(tot[1] - (a_ref ** 2).sum()) <= 1e-5 * (a_ref ** 2).sum()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["coupled", "decoupled"])
def test_rollout_storage_with_env(kind):
    """env.step(out=storage.slot(t)) fills the [T,N,..] buffers exactly like stepping and copying;
    GAE + normalisation on them match the oracle; sample() has the reference's PPO batch shapes."""
    from gym_rotor_amd import QuadVecEnv, RolloutStorage
    N, T = 1536, 24
    e1 = QuadVecEnv(kind, N, device="cuda", seed=2, auto_reset=True)
    e2 = QuadVecEnv(kind, N, device="cuda", seed=2, auto_reset=True)
    for e in (e1, e2):
        e.reset("train")
    st = RolloutStorage(e1, T)
    st.set_initial_obs(e1.get_norm_error_state())
    e2.get_norm_error_state()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for t in range(T):
        a = torch.rand(N, e1.action_dim, device="cuda", generator=g) * 2 - 1
        e1.step(a, out=st.slot(t))
        obs, r, d, _, _ = e2.step(a)
        obs = [obs] if isinstance(obs, torch.Tensor) else list(obs)
        for k, o in enumerate(obs):
            assert torch.equal(st.obs[k][t + 1], o)
        assert torch.equal(st.reward[t], r) and torch.equal(st.done[t], d)
        val = torch.randn(N, e1.n_agents, device="cuda", generator=g)
        st.insert(t, act=list(a.split(st.action_dims, 1)), logprob=list((a * 0.1).split(st.action_dims, 1)), value=val)
    last = torch.randn(N, e1.n_agents, device="cuda", generator=g)
    adv, tgt, stats = st.compute_gae(0.99, 0.9, last_value=last)
    v = st.value.cpu().numpy()
    M = N * e1.n_agents
    a_ref, t_ref = go.gae(st.reward.cpu().numpy().reshape(T, M), st.done.cpu().numpy().reshape(T, M),
                          v[:-1].reshape(T, M), v[1:].reshape(T, M), 0.99, 0.9)
    assert np.abs(adv.cpu().numpy().reshape(T, M) - a_ref).max() <= 2e-6 * max(1, np.abs(a_ref).max())
    assert np.abs(tgt.cpu().numpy().reshape(T, M) - t_ref).max() <= 2e-6 * max(1, np.abs(a_ref).max())
    nrm = RolloutStorage.normalize(adv, stats).cpu().numpy()
    for k in range(e1.n_agents):
        ref = go.normalize(a_ref.reshape(T, N, e1.n_agents)[..., k])
        assert np.abs(nrm[..., k] - ref).max() <= 2e-5
    obs, act, rwd, obs_next, done, logp = st.sample()
    assert len(obs) == e1.n_agents and obs[0].shape == (T * N, e1.obs_dims[0]) and rwd[0].shape == (T * N, 1)
    assert act[0].shape[1] == st.action_dims[0] and done[0].dtype == torch.float32 and logp[0].shape == act[0].shape
    keep = ~st.reset_mask()[: T - 1].reshape(-1)               # obs_next is obs[t+1] except where an episode ended
    assert torch.equal(obs_next[0][: (T - 1) * N][keep], obs[0][N:][keep])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["coupled", "decoupled", "quad"])
def test_final_observation_and_bootstrap_values(kind):
    """Same-step auto-reset overwrites the observation of a finished episode with the new episode's first one;
    the step kernel keeps the terminal observation in QrStepOut.final_obs* for exactly those envs.  Checked
    against a twin env WITHOUT auto-reset that is loaded with the same pre-step state every step (its
    observation IS the terminal one).  RolloutStorage.next_values then gives the reference's V(obs_next)
    (main.py:163-178, ppo.py:128-138) — also for a time-limit truncation and for the MODUL agent that did not
    terminate — and compute_gae(next_value=...) matches the oracle fed with the true next observations."""
    from gym_rotor_amd import QuadVecEnv, RolloutStorage
    N, T = 2048, 48
    kw = dict(device="cuda", seed=4, max_episode_steps=20, obs_rows=True)
    env = QuadVecEnv(kind, N, auto_reset=True, **kw)
    ref = QuadVecEnv(kind, N, auto_reset=False, w_adapt=0.0, **kw)   # (w_adapt = 0: the plain instantiation, the auto-reset launch's arithmetic)
    env.reset("train"); ref.reset("train")
    st = RolloutStorage(env, T)
    assert st.final_obs is not None
    first = env.get_norm_error_state() if kind != "quad" else env.get_current_state().float()
    st.set_initial_obs(first)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    ws = [torch.randn(d, device="cuda", generator=g) for d in env.obs_dims]
    if kind == "quad":
        ws = [ws[0]]

    def critic(rows):  # one value per agent: a fixed map of that agent's observation
        rows = [rows] if isinstance(rows, torch.Tensor) else list(rows)
        cols = [torch.tanh(r @ w) for r, w in zip(rows, ws)]
        return torch.stack(cols, 1) if len(cols) == env.n_agents else cols[0][:, None].expand(-1, env.n_agents).contiguous()

    true_next = [torch.zeros(T, N, d, device="cuda") for d in env.obs_dims]
    n_reset = n_trunc = n_other_agent = 0
    for t in range(T):
        a = torch.rand(N, env.action_dim, device="cuda", generator=g) * 2 - 1
        ref.load_state_dict({k: v for k, v in env.state_dict().items() if k != "last_obs"})
        env.step(a, out=st.slot(t))
        o_ref, _, d_ref, t_ref, _ = ref.step(a)
        o_ref = [o_ref] if isinstance(o_ref, torch.Tensor) else list(o_ref)
        reset = st.done[t].any(-1) | st.truncated[t]
        assert torch.equal(st.done[t], d_ref) and torch.equal(st.truncated[t], t_ref)
        n_reset += int(reset.sum()); n_trunc += int((st.truncated[t] & ~st.done[t].any(-1)).sum())
        n_other_agent += int((reset[:, None] & ~st.done[t]).sum()) if env.n_agents > 1 else 0
        for k, o in enumerate(o_ref):
            assert torch.equal(st.final_obs[k][t][reset], o[reset])          # terminal observation, bit for bit
            assert torch.equal(st.obs[k][t + 1][~reset], o[~reset])          # everyone else: the plain next observation
            true_next[k][t] = o
    assert n_reset > 200 and n_trunc > 0
    if kind == "decoupled":
        assert n_other_agent > 0                                             # an agent cut off by the other one's termination
    for t in range(T + 1):
        st.value[t] = critic([o[t] for o in st.obs])
    nv = st.next_values(critic)
    nv_true = torch.stack([critic([o[t] for o in true_next]) for t in range(T)])
    assert torch.equal(nv, nv_true)
    adv, tgt = st.compute_gae(0.99, 0.9, next_value=nv, want_stats=False)
    M = N * env.n_agents
    a_ref, t_ref = go.gae(st.reward.cpu().numpy().reshape(T, M), st.done.cpu().numpy().reshape(T, M),
                          st.value[:-1].cpu().numpy().reshape(T, M), nv_true.cpu().numpy().reshape(T, M), 0.99, 0.9)
    assert np.abs(adv.cpu().numpy().reshape(T, M) - a_ref).max() <= 2e-6 * max(1, np.abs(a_ref).max())
    # bootstrapping from value[t+1] instead is visibly different on the rows that ended an episode without `done`
    a_naive, _ = go.gae(st.reward.cpu().numpy().reshape(T, M), st.done.cpu().numpy().reshape(T, M),
                        st.value[:-1].cpu().numpy().reshape(T, M), st.value[1:].cpu().numpy().reshape(T, M), 0.99, 0.9)
    assert np.abs(a_naive - a_ref).max() > 1e-2
    obs, act, rwd, obs_next, done, logp = st.sample()
    for k in range(len(env.obs_dims)):
        assert torch.equal(obs_next[k], true_next[k].reshape(T * N, -1))


_GLOO = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, {root!r})
    from gym_rotor_amd.rollout import RolloutStorage
    from gym_rotor_amd import shard_range
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    T, N, A = 16, 1001, 2
    g = torch.Generator(); g.manual_seed(0)
    adv = torch.randn(T, N, A, generator=g) * torch.tensor([1.0, 3.0]) + torch.tensor([0.5, -2.0])
    s, e = shard_range(N, rank, world)
    loc = adv[:, s:e]
    a64 = loc.double()
    stats = torch.cat([torch.stack([a64.sum((0, 1)), (a64 * a64).sum((0, 1))], 1), torch.full((A, 1), float(T * (e - s)), dtype=torch.float64)], 1)
    nrm = RolloutStorage.normalize(loc, stats)
    ref = (adv - adv.mean((0, 1))) / (adv.reshape(-1, A).std(0) + 1e-4)
    assert torch.allclose(nrm, ref[:, s:e], atol=1e-5), rank
    dist.barrier(); dist.destroy_process_group(); print("ok", rank)
""")


def test_normalisation_stats_all_reduce_gloo(tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    script = tmp_path / "w.py"
    script.write_text(_GLOO.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2"),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err
