#!/usr/bin/env python3
"""NumPy emulation of candidate arithmetic for the attitude stages of the default (`mixed`) layout in the FREE RUN far beyond
termination, where the Decoupled action map feeds (R, W) back into the torque and amplifies every perturbation ~100x over 900
steps.  Against the float64 DOP853 oracle, 256 envs x 1000 random-action steps, per-env adaptive substep
multiplier ceil(max|W| / 16) like the kernel's (wave-level there).

  f32q    the kernel's arithmetic today: W1, W2 in float64, quaternion stages in float32 from the float32-rounded q
  delta   k1 = q (0, W/2) in float64; the later stages as DIFFERENCES from k1 in float32:
          k_i - k1 = qdot(q, w_i - w_1) + qdot(qt_i - q, w_i)   (bilinear; both factors small or exactly representable)
  f64q    every quaternion stage in float64 (the `f64` layout's arithmetic for q; x, v still float32 words)
  X>thr   arithmetic X only in env-steps that start with max|W_i| >= thr [rad/s], f32q otherwise

    python tools/numerics_delta.py [--kind decoupled] [--n 256] [--T 1000]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import quad_oracle as orc  # noqa: E402
from tests.conftest import grouped_rel_err  # noqa: E402
from tools.numerics_f32stage import quat_to_R, R_to_quat  # noqa: E402

f32 = np.float32


def qdot(q, w, T):
    """q (0, w) with w already HALF rates; q [n,4], w [n,3]"""
    qw, qx, qy, qz = q.T
    a, b, c = w.T
    return np.stack([-(qx * a + qy * b + qz * c), qw * a + qy * c - qz * b, qw * b + qz * a - qx * c, qw * c + qx * b - qy * a], 1).astype(T)


def thrust(q):
    qw, qx, qy, qz = q.T
    return np.stack([qx * qz + qw * qy, qy * qz - qw * qx, qx * qx + qy * qy], 1)


def substep(x, v, q, W, h, c, A1, U, mode):
    """One RK4 substep.  W chain float64 always; q per `mode`; thrust sums float32."""
    def Wdot(Wv):
        return np.stack([A1 * Wv[:, 1] * Wv[:, 2] + U[:, 0], U[:, 1] - A1 * Wv[:, 2] * Wv[:, 0], U[:, 2]], 1)
    k1 = Wdot(W); Wa = W + 0.5 * h * k1
    k2 = Wdot(Wa); Wb = W + 0.5 * h * k2
    k3 = Wdot(Wb); Wc = W + h * k3
    k4 = Wdot(Wc)
    Wn = W + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
    Ws = [W, Wa, Wb, Wc]
    cs = [0.5 * h, 0.5 * h, h]
    if mode == "f64q":
        ks, qt = [], q
        for i in range(4):
            k = qdot(qt, 0.5 * Ws[i], np.float64); ks.append(k)
            if i < 3:
                qt = q + cs[i] * k
        dq = h / 6.0 * (ks[0] + 2 * ks[1] + 2 * ks[2] + ks[3])
        g = [thrust(q.astype(f32))] + [thrust((q + cs[i] * ks[i]).astype(f32)) for i in range(3)]
    elif mode == "f32q":
        q0 = q.astype(f32)
        w = [(0.5 * a).astype(f32) for a in Ws]
        ks, qt, g = [], q0, []
        for i in range(4):
            g.append(thrust(qt))
            k = qdot(qt, w[i], f32); ks.append(k)
            if i < 3:
                qt = q0 + f32(cs[i]) * k
        dq = (f32(h / 6.0) * (ks[0] + f32(2) * ks[1] + f32(2) * ks[2] + ks[3])).astype(np.float64)
    elif mode == "delta":
        q0 = q.astype(f32)
        k1q = qdot(q, 0.5 * Ws[0], np.float64)                # float64
        k1f = k1q.astype(f32)
        dw = [((0.5 * (Ws[i] - Ws[0]))).astype(f32) for i in range(4)]   # stage rate minus start rate (formed in float64, rounded)
        w = [(0.5 * a).astype(f32) for a in Ws]
        d, g = [np.zeros_like(q0)], [thrust(q0)]
        dqt = f32(cs[0]) * k1f                                 # qt_2 - q
        for i in range(1, 4):
            g.append(thrust(q0 + dqt))
            di = qdot(q0, dw[i], f32) + qdot(dqt, w[i], f32)   # k_i - k1
            d.append(di)
            if i < 3:
                dqt = f32(cs[i]) * (k1f + di)
        dq = h * k1q + (f32(h / 6.0) * (f32(2) * d[1] + f32(2) * d[2] + d[3])).astype(np.float64)
    else:
        raise ValueError(mode)
    g1, g23, g4 = g[0], g[1] + g[2], g[3]
    cf, hf = c.astype(f32)[:, None], f32(h)
    sgn = np.array([-1, -1, 1], f32)
    G = g1 + f32(2) * g23 + g4
    gc = (9.81 - c).astype(f32)
    dv = sgn * (hf * cf * f32(1 / 3.0)) * G.astype(f32); dv[:, 2] += hf * gc
    dx = hf * v.astype(f32) + sgn * (hf * hf * cf * f32(1 / 3.0)) * (g1 + g23).astype(f32); dx[:, 2] += f32(0.5) * hf * hf * gc
    return x + dx.astype(np.float64), v + dv.astype(np.float64), q + dq, Wn


def run(kind, n, T, seed, mode, thr, w_adapt=16.0):
    rng = np.random.default_rng(seed)
    A = orc.ACTION_DIM[kind]
    state = orc.sample_reset_state(rng, n).astype(f32).astype(np.float64)
    params = orc.sample_params(rng, n).astype(f32).astype(np.float64)
    acts = rng.uniform(-1, 1, (T, n, A)).astype(f32)
    dvp = orc.derive(params)
    s = state.copy()
    x, v, W = state[:, 0:3].copy(), state[:, 3:6].copy(), state[:, 15:18].copy()
    q = R_to_quat(state[:, 6:15])
    s[:, 6:15] = quat_to_R(q)
    used = 0
    for t in range(T):
        a = acts[t].astype(np.float64)
        f, M = orc.action_map_batch(kind, a, s, dvp)
        s = orc.integrate_batch(s, f, M, dvp.m, dvp.J1, dvp.J1, dvp.J3)
        mine = np.concatenate([x, v, quat_to_R(q), W], 1)
        f, M = orc.action_map_batch(kind, a, mine, dvp)        # the emulation's OWN state feeds its action map
        c = f / dvp.m
        A1 = (dvp.J1 - dvp.J3) / dvp.J1
        U = np.stack([M[:, 0] / dvp.J1, M[:, 1] / dvp.J1, M[:, 2] / dvp.J3], 1)
        wmax = np.abs(W).max(1)
        mul = np.clip(np.ceil(wmax / w_adapt), 1, 16).astype(int)
        special = wmax >= thr
        used += int(special.sum())
        for m_ in np.unique(mul):
            for sp in (False, True):
                sel = (mul == m_) & (special == sp)
                if not sel.any():
                    continue
                h = orc.DT / int(m_)
                xs, vs, qs, Ws_ = x[sel], v[sel], q[sel], W[sel]
                for _ in range(int(m_)):
                    xs, vs, qs, Ws_ = substep(xs, vs, qs, Ws_, h, c[sel], A1[sel], U[sel], mode if sp else "f32q")
                x[sel], v[sel], q[sel], W[sel] = xs, vs, qs, Ws_
        q *= (1.5 - 0.5 * (q * q).sum(1))[:, None]
        x = x.astype(f32).astype(np.float64); v = v.astype(f32).astype(np.float64)
    got = np.concatenate([x, v, quat_to_R(q), W], 1)
    return grouped_rel_err(got, s), np.abs(s[:, 15:18]).max(), used / (n * T)


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--kind", default="decoupled")
    p.add_argument("--n", type=int, default=256)
    p.add_argument("--T", type=int, default=1000)
    p.add_argument("--seeds", default="79,500")
    a = p.parse_args()
    for seed in map(int, a.seeds.split(",")):
        for mode, thr in (("f32q", 0.0), ("delta", 0.0), ("f64q", 0.0), ("delta", 2 * np.pi), ("f64q", 2 * np.pi), ("delta", 12.0), ("f64q", 16.0)):
            err, wmax, frac = run(a.kind, a.n, a.T, seed, mode, thr)
            print(f"{a.kind} seed {seed} {mode:5s} for max|W| >= {thr:5.2f}: grouped error {err:.2e} (special arithmetic in {100 * frac:.0f} % of the env-steps; max|W| {wmax:.1f})", flush=True)
