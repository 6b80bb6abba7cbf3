#!/usr/bin/env python3
"""NumPy emulation of candidate arithmetic for the RK4 substep, against the float64 DOP853 oracle.

Variants (state always ACCUMULATED in float64 for q, W; x, v rounded to float32 at every env-step
boundary = the `mixed` layout):
  f64      every stage in float64 (the round-1 kernel)
  f32k     stage states and derivatives in float32 (from the float32-rounded substep-start state),
           increment summed in float32, added to the float64 state; the constant torque term h*U is
           added in float64
  f32q     as f32k for q (and the thrust direction), but W1, W2 integrated in float64
Metric: conftest.grouped_rel_err after T free-run random-action steps (no reset), per-env adaptive
substep multiplier ceil(max|W| / 16) like the kernel's wave-level one (a wave takes >= this)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import quad_oracle as orc  # noqa: E402
from tests.conftest import grouped_rel_err  # noqa: E402

G = 9.81


def quat_to_R(q):
    w, x, y, z = q.T
    R = np.empty((q.shape[0], 9), q.dtype)
    R[:, 0] = 1 - 2 * (y * y + z * z); R[:, 1] = 2 * (x * y + w * z); R[:, 2] = 2 * (x * z - w * y)
    R[:, 3] = 2 * (x * y - w * z); R[:, 4] = 1 - 2 * (x * x + z * z); R[:, 5] = 2 * (y * z + w * x)
    R[:, 6] = 2 * (x * z + w * y); R[:, 7] = 2 * (y * z - w * x); R[:, 8] = 1 - 2 * (x * x + y * y)
    return R


def R_to_quat(Rv):
    from scipy.spatial.transform import Rotation
    R = np.swapaxes(Rv.reshape(-1, 3, 3), 1, 2)
    q = Rotation.from_matrix(R).as_quat()  # x y z w
    return np.stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]], 1)


def rhs(q, W, c, A1, U, T):
    """(vdot, qdot, Wdot) in dtype T."""
    qw, qx, qy, qz = q.T
    W1, W2, W3 = W.T
    two = T(2)
    vd = np.stack([-two * c * (qx * qz + qw * qy), -two * c * (qy * qz - qw * qx),
                   (T(G) - c) + two * c * (qx * qx + qy * qy)], 1)
    h = T(0.5)
    qd = np.stack([-h * (qx * W1 + qy * W2 + qz * W3), h * (qw * W1 + qy * W3 - qz * W2),
                   h * (qw * W2 + qz * W1 - qx * W3), h * (qw * W3 + qx * W2 - qy * W1)], 1)
    Wd = np.stack([A1 * W2 * W3 + U[:, 0], U[:, 1] - A1 * W3 * W1, U[:, 2]], 1)
    return vd, qd, Wd


def rk4_substep(x, v, q, W, h, c, A1, U, mode):
    f32 = np.float32
    if mode == "f64":
        T = np.float64
        y0 = (v, q, W)
        cc, AA, UU = c, A1, U
    else:
        T = f32
        y0 = (v.astype(f32), q.astype(f32), W.astype(f32))
        cc, AA, UU = c.astype(f32), A1.astype(f32), U.astype(f32)
    hT, h2 = T(h), T(0.5 * h)
    k1 = rhs(y0[1], y0[2], cc, AA, UU, T)
    y2 = [a + h2 * k for a, k in zip(y0, k1)]
    k2 = rhs(y2[1], y2[2], cc, AA, UU, T)
    y3 = [a + h2 * k for a, k in zip(y0, k2)]
    k3 = rhs(y3[1], y3[2], cc, AA, UU, T)
    y4 = [a + hT * k for a, k in zip(y0, k3)]
    k4 = rhs(y4[1], y4[2], cc, AA, UU, T)
    h6 = T(h / 6.0)
    inc = [h6 * (a + T(2) * b + T(2) * cc_ + d) for a, b, cc_, d in zip(k1, k2, k3, k4)]
    # x' = v from the stage velocities: x += h v + h^2/6 (k1v + k2v + k3v)
    dx = hT * y0[0] + T(h * h / 6.0) * (k1[0] + k2[0] + k3[0])
    if mode == "f64":
        return x + dx, v + inc[0], q + inc[1], W + inc[2]
    if mode == "f32k":
        # constant torque term exactly in float64, gyroscopic part from the float32 stages
        gyro = inc[2].astype(np.float64) - (T(h) * UU).astype(np.float64)
        # (the kernel would form the gyro sum separately; emulate: sum of float32 (k - U))
        g = [k[2] - UU for k in (k1, k2, k3, k4)]
        gyro = (h6 * (g[0] + T(2) * g[1] + T(2) * g[2] + g[3])).astype(np.float64)
        return x + dx.astype(np.float64), v + inc[0].astype(np.float64), q + inc[1].astype(np.float64), W + h * U + gyro
    raise ValueError(mode)


def rk4_substep_f32q(x, v, q, W, h, c, A1, U):
    """W in float64 (cheap: 2 components + closed-form W3), q and thrust direction in float32 using the
    float32-rounded stage W."""
    f32 = np.float32
    # float64 W stages
    def Wdot(Wv):
        return np.stack([A1 * Wv[:, 1] * Wv[:, 2] + U[:, 0], U[:, 1] - A1 * Wv[:, 2] * Wv[:, 0], U[:, 2]], 1)
    kW1 = Wdot(W); W2s = W + 0.5 * h * kW1
    kW2 = Wdot(W2s); W3s = W + 0.5 * h * kW2
    kW3 = Wdot(W3s); W4s = W + h * kW3
    kW4 = Wdot(W4s)
    Wn = W + h / 6.0 * (kW1 + 2 * kW2 + 2 * kW3 + kW4)
    q0 = q.astype(f32); cc = c.astype(f32)
    Ws = [a.astype(f32) for a in (W, W2s, W3s, W4s)]
    z3 = np.zeros((q.shape[0], 3), f32); z1 = np.zeros(q.shape[0], f32)
    hT, h2 = f32(h), f32(0.5 * h)
    kv1, kq1, _ = rhs(q0, Ws[0], cc, z1, z3, f32)
    kv2, kq2, _ = rhs(q0 + h2 * kq1, Ws[1], cc, z1, z3, f32)
    kv3, kq3, _ = rhs(q0 + h2 * kq2, Ws[2], cc, z1, z3, f32)
    kv4, kq4, _ = rhs(q0 + hT * kq3, Ws[3], cc, z1, z3, f32)
    h6 = f32(h / 6.0)
    dq = h6 * (kq1 + f32(2) * kq2 + f32(2) * kq3 + kq4)
    dv = h6 * (kv1 + f32(2) * kv2 + f32(2) * kv3 + kv4)
    dx = hT * v.astype(f32) + f32(h * h / 6.0) * (kv1 + kv2 + kv3)
    return x + dx.astype(np.float64), v + dv.astype(np.float64), q + dq.astype(np.float64), Wn


def run(kind_mode, n, T, seed, w_adapt=16.0, substeps=1):
    rng = np.random.default_rng(seed)
    state = orc.sample_reset_state(rng, n).astype(np.float32).astype(np.float64)
    params = orc.sample_params(rng, n).astype(np.float32).astype(np.float64)
    acts = rng.uniform(-1, 1, (T, n, 4)).astype(np.float32)
    dv = orc.derive(params)
    # oracle
    s = state.copy()
    x, v, W = state[:, 0:3].copy(), state[:, 3:6].copy(), state[:, 15:18].copy()
    q = R_to_quat(state[:, 6:15])
    s[:, 6:15] = quat_to_R(q)
    worst = 0.0
    for t in range(T):
        a = acts[t].astype(np.float64)
        f, M = orc.action_map_batch("quad", a, s, dv)
        s = orc.integrate_batch(s, f, M, dv.m, dv.J1, dv.J1, dv.J3)
        c = f / dv.m
        A1 = (dv.J1 - dv.J3) / dv.J1
        U = np.stack([M[:, 0] / dv.J1, M[:, 1] / dv.J1, M[:, 2] / dv.J3], 1)
        mul = np.clip(np.ceil(np.abs(W).max(1) / w_adapt), 1, 16).astype(int) if w_adapt > 0 else np.ones(n, int)
        for m_ in np.unique(mul):
            sel = mul == m_
            ns = substeps * int(m_)
            h = orc.DT / ns
            xs, vs, qs, Ws = x[sel], v[sel], q[sel], W[sel]
            for _ in range(ns):
                if kind_mode == "f32q":
                    xs, vs, qs, Ws = rk4_substep_f32q(xs, vs, qs, Ws, h, c[sel], A1[sel], U[sel])
                else:
                    xs, vs, qs, Ws = rk4_substep(xs, vs, qs, Ws, h, c[sel], A1[sel], U[sel], kind_mode)
            x[sel], v[sel], q[sel], W[sel] = xs, vs, qs, Ws
        q *= (1.5 - 0.5 * (q * q).sum(1))[:, None]
        x = x.astype(np.float32).astype(np.float64); v = v.astype(np.float32).astype(np.float64)
        if t % 100 == 99 or t == T - 1:
            got = np.concatenate([x, v, quat_to_R(q), W], 1)
            worst = max(worst, grouped_rel_err(got, s))
    got = np.concatenate([x, v, quat_to_R(q), W], 1)
    per_env = np.zeros(n)
    from tests.conftest import GROUPS
    for sl in GROUPS:
        per_env = np.maximum(per_env, np.abs(got[:, sl] - s[:, sl]).max(1) / np.maximum(np.abs(s[:, sl]).max(1), 1.0))
    return worst, np.abs(s[:, 15:18]).max(), np.percentile(per_env, [50, 99]), per_env.max()


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--n", type=int, default=1024)
    p.add_argument("--T", type=int, default=1000)
    p.add_argument("--modes", default="f64,f32k,f32q")
    p.add_argument("--substeps", default="1")
    p.add_argument("--seeds", default="500")
    a = p.parse_args()
    for seed in map(int, a.seeds.split(",")):
        for sub in map(int, a.substeps.split(",")):
            for mode in a.modes.split(","):
                worst, wmax, pct, mx = run(mode, a.n, a.T, seed, substeps=sub)
                print(f"seed {seed} substeps {sub} mode {mode:5s}: worst grouped err {worst:.2e} (final: median {pct[0]:.1e}, p99 {pct[1]:.1e}, max {mx:.1e}); max|W| {wmax:.1f}", flush=True)
