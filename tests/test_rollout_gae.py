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
    assert torch.equal(obs_next[0][: (T - 1) * N], obs[0][N:])


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
