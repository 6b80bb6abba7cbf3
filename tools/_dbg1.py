import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gym_rotor_amd import QuadVecEnv
kind="decoupled"; n=4096
rng = np.random.default_rng(0)
env = QuadVecEnv(kind, n, device="cuda", seed=9, auto_reset=True, max_episode_steps=50, obs_rows=True)
ref = QuadVecEnv(kind, n, device="cuda", seed=9, auto_reset=False, max_episode_steps=50, obs_rows=True)
for e in (env, ref): e.reset("train")
a = torch.from_numpy(rng.uniform(-1, 1, (n, env.action_dim)).astype(np.float32)).cuda()
for t in range(60):
    ref.load_state_dict(env.state_dict())
    obs, rwd, done, trunc, _ = env.step(a)
    o2, r2, d2, t2, _ = ref.step(a)
    hit = (done.any(1) | trunc).cpu().numpy()
    st, sr = env.get_current_state().cpu().numpy(), ref.get_current_state().cpu().numpy()
    bad = np.argwhere((st != sr).any(1) & ~hit)[:, 0]
    if len(bad):
        print("t", t, "bad lanes", bad[:20], "tiles", np.unique(bad // 64)[:10], "n", len(bad))
        i = bad[0]
        print("cols", np.flatnonzero(st[i] != sr[i]), (st[i] - sr[i])[st[i] != sr[i]])
        tile = i // 64
        print("hit in tile:", np.flatnonzero(hit[tile*64:(tile+1)*64]))
        ig, ir = env.integ.cpu().numpy(), ref.integ.cpu().numpy()
        print("integ differs:", np.abs(ig[~hit] - ir[~hit]).max())
        break
else:
    print("no mismatch")
