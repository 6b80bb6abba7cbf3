#!/usr/bin/env python3
"""NumPy emulation of a 4th-order Lie-group (Magnus) substep for EVEN substep counts, against the float64 DOP853 oracle and
against the RK4 substep the kernel runs (tools/numerics_f32stage.py, mode f32k = the round-6 kernel's arithmetic).

Per env-step (zero-order-hold f, M; J1 = J2):
  W3(t) = W3 + U3 t exactly; w = W1 + i W2 obeys w' = -i a(t) w + u, a(t) = A1 W3(t): w(t) = w0 + u t + C(t), C a polynomial
  whose Taylor coefficients are formed once per env-step (degree `deg`).
Per substep [t, t + h], midpoint tm (all from closed forms at tm, no stages):
  Theta = h Wm + h^3/24 Wm'' + sigma h^3/12 (Wm x Wm')       (Magnus, terms 1 and 2 about the midpoint: local error O(h^5))
  q <- q (x) exp(Theta / 2)                                    (increment q (x) (cos - 1, sin .) so that float32 rounds small terms)
Thrust: the direction u(q) at the substep BOUNDARIES, composite Simpson over the env-step for v and (with weights T - t) for x.
Modes: mag64 (everything float64: the truncation error), mag32 (per-substep arithmetic float32, sums float32, state float64)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import quad_oracle as orc  # noqa: E402
from tests.conftest import grouped_rel_err, GROUPS  # noqa: E402
from tools.numerics_f32stage import quat_to_R, R_to_quat, rk4_substep  # noqa: E402

G = 9.81
SIGMA = 1.0


def uvec(q):
    w, x, y, z = q.T
    return np.stack([x * z + w * y, y * z - w * x, x * x + y * y], 1)


def magnus_step(x, v, q, W, dt, nsub, c, A1, U, T, deg=4):
    """One env-step of nsub (even) Magnus substeps; stage arithmetic in dtype T; returns float64 state."""
    f = T
    h = dt / nsub
    w0 = (W[:, 0] + 1j * W[:, 1])
    u = (U[:, 0] + 1j * U[:, 1])
    a0 = A1 * W[:, 2]
    ad = A1 * U[:, 2]
    ctype = np.complex64 if T == np.float32 else np.complex128
    # Taylor coefficients of w(t) = sum c_k t^k
    cs = [w0.astype(ctype), (-1j * a0 * w0 + u).astype(ctype)]
    a0T, adT = a0.astype(f), ad.astype(f)
    for k in range(1, deg):
        cs.append((-1j * (a0T * cs[k] + adT * cs[k - 1]) / f(k + 1)).astype(ctype))
    coup1 = (-1j * a0T * cs[0]).astype(ctype)   # c1 - u: the coupling part's linear coefficient

    def C(t):   # coupling part of w(t): w(t) - w0 - u t
        acc = cs[deg]
        for k in range(deg - 1, 1, -1):
            acc = (acc * f(t) + cs[k]).astype(ctype)
        return ((acc * f(t) + coup1) * f(t)).astype(ctype)

    w0T, uT = w0.astype(ctype), u.astype(ctype)
    W3_0, U3 = W[:, 2].astype(f), U[:, 2].astype(f)
    U1, U2 = U[:, 0].astype(f), U[:, 1].astype(f)
    qs = q.astype(f)
    dqs = np.zeros_like(qs)
    n = q.shape[0]
    g_sum = np.zeros((n, 3), f)      # sum w_k u_k
    gx_sum = np.zeros((n, 3), f)     # sum w_k (T - t_k) u_k
    uk = uvec(qs)
    g_sum += uk
    gx_sum += f(dt) * uk
    hT = f(h)
    k1c, k2c = f(h ** 3 / 24.0), f(SIGMA * h ** 3 / 12.0)
    for s in range(nsub):
        tm = (s + 0.5) * h
        wm = (w0T + uT * f(tm) + C(tm)).astype(ctype)
        W1m, W2m = wm.real.astype(f), wm.imag.astype(f)
        W3m = W3_0 + U3 * f(tm)
        am = a0T + adT * f(tm)
        d1 = am * W2m + U1
        d2 = -am * W1m + U2
        d3 = U3
        dd1 = adT * W2m + am * d2
        dd2 = -adT * W1m - am * d1
        # Wm x Wm'
        cx = W2m * d3 - W3m * d2
        cy = W3m * d1 - W1m * d3
        cz = W1m * d2 - W2m * d1
        half = f(0.5)
        th1 = half * (hT * W1m + k1c * dd1 + k2c * cx)
        th2 = half * (hT * W2m + k1c * dd2 + k2c * cy)
        th3 = half * (hT * W3m + k2c * cz)
        n2 = th1 * th1 + th2 * th2 + th3 * th3
        sn = f(1) + n2 * (f(-1 / 6.0) + n2 * f(1 / 120.0))
        cm1 = n2 * (f(-0.5) + n2 * (f(1 / 24.0) + n2 * f(-1 / 720.0)))
        A, B, Cc = sn * th1, sn * th2, sn * th3
        q0, q1, q2, q3 = qs.T
        dq = np.stack([q0 * cm1 - (q1 * A + q2 * B + q3 * Cc),
                       q1 * cm1 + (q0 * A + q2 * Cc - q3 * B),
                       q2 * cm1 + (q0 * B + q3 * A - q1 * Cc),
                       q3 * cm1 + (q0 * Cc + q1 * B - q2 * A)], 1).astype(f)
        dqs = (dqs + dq).astype(f)
        qs = (qs + dq).astype(f)
        wk = f(1.0) if s == nsub - 1 else (f(4.0) if s % 2 == 0 else f(2.0))
        # node k = s + 1: Simpson weight 4 for odd k, 2 for even interior k, 1 for the last
        k = s + 1
        wk = f(1.0) if k == nsub else (f(4.0) if k % 2 == 1 else f(2.0))
        uk = uvec(qs)
        g_sum = (g_sum + wk * uk).astype(f)
        gx_sum = (gx_sum + wk * f(dt - k * h) * uk).astype(f)
    cT = c.astype(f)
    gc = (G - c).astype(f)
    sgn = np.array([-1.0, -1.0, 1.0], f)
    two_c_h3 = (f(2.0) * cT * f(h / 3.0))[:, None]
    vf, xf = v.astype(f), x.astype(f)
    dtT = f(dt)
    v_new = vf + two_c_h3 * sgn * g_sum
    v_new[:, 2] += dtT * gc
    x_new = xf + dtT * vf + two_c_h3 * sgn * gx_sum
    x_new[:, 2] += f(0.5) * dtT * dtT * gc
    Cend = C(dt)
    Wn = W.copy()
    Wn[:, 0] = W[:, 0] + dt * U[:, 0] + Cend.real.astype(np.float64)
    Wn[:, 1] = W[:, 1] + dt * U[:, 1] + Cend.imag.astype(np.float64)
    Wn[:, 2] = W[:, 2] + dt * U[:, 2]
    return x_new.astype(np.float64), v_new.astype(np.float64), q + dqs.astype(np.float64), Wn


def bdot_terms(q, W1, W2):
    """d/dt of u-vector's parent: b3 = (2 u0, 2 u1, 1 - 2 u2);  b3' = W2 b1 - W1 b2 (columns of R).  Returns b3'/2 in the u convention
    (so that it combines with u): (b3'_0 / 2, b3'_1 / 2, -b3'_2 / 2)."""
    w, x, y, z = q.T
    b1 = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)], 1)
    b2 = np.stack([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)], 1)
    bd = W2[:, None] * b1 - W1[:, None] * b2
    return np.stack([0.5 * bd[:, 0], 0.5 * bd[:, 1], -0.5 * bd[:, 2]], 1).astype(q.dtype)


def magnus_step_em(x, v, q, W, dt, nsub, c, A1, U, T, deg=4):
    """As magnus_step, any nsub >= 1: the thrust integrals by the trapezoid rule over the substep boundaries + the Euler-Maclaurin
    end correction h^2/12 (g'(0) - g'(T)) (interior derivative terms telescope)."""
    f = T
    h = dt / nsub
    w0 = (W[:, 0] + 1j * W[:, 1]); u = (U[:, 0] + 1j * U[:, 1])
    a0 = A1 * W[:, 2]; ad = A1 * U[:, 2]
    ctype = np.complex64 if T == np.float32 else np.complex128
    cs = [w0.astype(ctype), (-1j * a0 * w0 + u).astype(ctype)]
    a0T, adT = a0.astype(f), ad.astype(f)
    for k in range(1, deg):
        cs.append((-1j * (a0T * cs[k] + adT * cs[k - 1]) / f(k + 1)).astype(ctype))
    coup1 = (-1j * a0T * cs[0]).astype(ctype)

    def wpoly(t):
        acc = cs[deg]
        for k in range(deg - 1, -1, -1):
            acc = (acc * f(t) + cs[k]).astype(ctype)
        return acc

    def C(t):
        acc = cs[deg]
        for k in range(deg - 1, 1, -1):
            acc = (acc * f(t) + cs[k]).astype(ctype)
        return ((acc * f(t) + coup1) * f(t)).astype(ctype)

    W3_0, U3 = W[:, 2].astype(f), U[:, 2].astype(f)
    U1, U2 = U[:, 0].astype(f), U[:, 1].astype(f)
    qs = q.astype(f); dqs = np.zeros_like(qs)
    n = q.shape[0]
    u0 = uvec(qs)
    bd0 = bdot_terms(qs, W[:, 0].astype(f), W[:, 1].astype(f))
    g_sum = f(0.5) * u0               # trapezoid weights 1/2, 1, ..., 1, 1/2
    gx_sum = f(0.5) * f(dt) * u0      # of (T - t) u
    hT = f(h); k1c, k2c = f(h ** 3 / 24.0), f(SIGMA * h ** 3 / 12.0)
    for s in range(nsub):
        tm = (s + 0.5) * h
        wm = wpoly(tm)
        W1m, W2m = wm.real.astype(f), wm.imag.astype(f)
        W3m = W3_0 + U3 * f(tm); am = a0T + adT * f(tm)
        d1 = am * W2m + U1; d2 = -am * W1m + U2; d3 = U3
        dd1 = adT * W2m + am * d2; dd2 = -adT * W1m - am * d1
        cx = W2m * d3 - W3m * d2; cy = W3m * d1 - W1m * d3; cz = W1m * d2 - W2m * d1
        half = f(0.5)
        th1 = half * (hT * W1m + k1c * dd1 + k2c * cx); th2 = half * (hT * W2m + k1c * dd2 + k2c * cy); th3 = half * (hT * W3m + k2c * cz)
        n2 = th1 * th1 + th2 * th2 + th3 * th3
        sn = f(1) + n2 * (f(-1 / 6.0) + n2 * f(1 / 120.0))
        cm1 = n2 * (f(-0.5) + n2 * (f(1 / 24.0) + n2 * f(-1 / 720.0)))
        A, B, Cc = sn * th1, sn * th2, sn * th3
        q0, q1, q2, q3 = qs.T
        dq = np.stack([q0 * cm1 - (q1 * A + q2 * B + q3 * Cc), q1 * cm1 + (q0 * A + q2 * Cc - q3 * B),
                       q2 * cm1 + (q0 * B + q3 * A - q1 * Cc), q3 * cm1 + (q0 * Cc + q1 * B - q2 * A)], 1).astype(f)
        dqs = (dqs + dq).astype(f); qs = (qs + dq).astype(f)
        k = s + 1
        wk = f(0.5) if k == nsub else f(1.0)
        uk = uvec(qs)
        g_sum = (g_sum + wk * uk).astype(f)
        gx_sum = (gx_sum + wk * f(dt - k * h) * uk).astype(f)
    wT = wpoly(dt)
    bdT = bdot_terms(qs, wT.real.astype(f), wT.imag.astype(f))
    # Euler-Maclaurin: + h^2/12 (g'(0) - g'(T));  v: g = u -> (bd0 - bdT);  x: g = (T - t) u -> (-u0 + T bd0) - (-uT) = (uT - u0) + T bd0
    em = f(h * h / 12.0)
    Iv = hT * g_sum + em * (bd0 - bdT)
    Ix = hT * gx_sum + em * ((uk - u0) + f(dt) * bd0)
    cT = c.astype(f); gc = (G - c).astype(f)
    sgn = np.array([-1.0, -1.0, 1.0], f)
    two_c = (f(2.0) * cT)[:, None]
    vf, xf = v.astype(f), x.astype(f); dtT = f(dt)
    v_new = vf + two_c * sgn * Iv; v_new[:, 2] += dtT * gc
    x_new = xf + dtT * vf + two_c * sgn * Ix; x_new[:, 2] += f(0.5) * dtT * dtT * gc
    Cend = C(dt)
    Wn = W.copy()
    Wn[:, 0] = W[:, 0] + dt * U[:, 0] + Cend.real.astype(np.float64)
    Wn[:, 1] = W[:, 1] + dt * U[:, 1] + Cend.imag.astype(np.float64)
    Wn[:, 2] = W[:, 2] + dt * U[:, 2]
    return x_new.astype(np.float64), v_new.astype(np.float64), q + dqs.astype(np.float64), Wn


STORE_F32 = True


def run(mode, n, T, seed, substeps, w_adapt=16.0, deg=4, in_regime=False):
    rng = np.random.default_rng(seed)
    state = orc.sample_reset_state(rng, n).astype(np.float32).astype(np.float64)
    params = orc.sample_params(rng, n).astype(np.float32).astype(np.float64)
    acts = rng.uniform(-1, 1, (T, n, 4)).astype(np.float32)
    dv = orc.derive(params)
    s = state.copy()
    x, v, W = state[:, 0:3].copy(), state[:, 3:6].copy(), state[:, 15:18].copy()
    q = R_to_quat(state[:, 6:15])
    s[:, 6:15] = quat_to_R(q)
    alive = np.ones(n, bool)
    worst = 0.0
    for t in range(T):
        a = acts[t].astype(np.float64)
        f_, M = orc.action_map_batch("quad", a, s, dv)
        s = orc.integrate_batch(s, f_, M, dv.m, dv.J1, dv.J1, dv.J3)
        c = f_ / dv.m
        A1 = (dv.J1 - dv.J3) / dv.J1
        U = np.stack([M[:, 0] / dv.J1, M[:, 1] / dv.J1, M[:, 2] / dv.J3], 1)
        mul = np.clip(np.ceil(np.abs(W).max(1) / w_adapt), 1, 16).astype(int) if (w_adapt > 0 and not in_regime) else np.ones(n, int)
        for m_ in np.unique(mul):
            sel = mul == m_
            ns = substeps * int(m_)
            if mode.startswith("em"):
                Tt = np.float64 if mode == "em64" else np.float32
                x[sel], v[sel], q[sel], W[sel] = magnus_step_em(x[sel], v[sel], q[sel], W[sel], orc.DT, ns, c[sel], A1[sel], U[sel], Tt, deg)
            elif mode.startswith("mag"):
                Tt = np.float64 if mode == "mag64" else np.float32
                x[sel], v[sel], q[sel], W[sel] = magnus_step(x[sel], v[sel], q[sel], W[sel], orc.DT, ns, c[sel], A1[sel], U[sel], Tt, deg)
            else:
                h = orc.DT / ns
                xs, vs, qs, Ws = x[sel], v[sel], q[sel], W[sel]
                for _ in range(ns):
                    xs, vs, qs, Ws = rk4_substep(xs, vs, qs, Ws, h, c[sel], A1[sel], U[sel], mode)
                x[sel], v[sel], q[sel], W[sel] = xs, vs, qs, Ws
        q *= (1.5 - 0.5 * (q * q).sum(1))[:, None]
        if STORE_F32:
            x = x.astype(np.float32).astype(np.float64); v = v.astype(np.float32).astype(np.float64)
        if in_regime:   # follow an env only while the reference has not terminated it (production mode re-samples it there)
            alive &= (np.abs(s[:, 0:3]).max(1) < 1.0) & (np.abs(s[:, 3:6]).max(1) < 4.0) & (np.abs(s[:, 15:18]).max(1) < 2 * np.pi)
        got = np.concatenate([x, v, quat_to_R(q), W], 1)
        if alive.any() and (in_regime or t % 100 == 99 or t == T - 1):
            worst = max(worst, grouped_rel_err(got[alive], s[alive]))
    return worst, int(alive.sum())


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--n", type=int, default=1024)
    p.add_argument("--T", type=int, default=1000)
    p.add_argument("--modes", default="f64,f32k,mag64,mag32")
    p.add_argument("--substeps", default="2,4,10")
    p.add_argument("--seeds", default="500")
    p.add_argument("--deg", type=int, default=4)
    p.add_argument("--in-regime", action="store_true")
    p.add_argument("--no-f32-storage", action="store_true", help="keep x, v float64 between steps: exposes the integrators' own error")
    a = p.parse_args()
    STORE_F32 = not a.no_f32_storage
    for seed in map(int, a.seeds.split(",")):
        for sub in map(int, a.substeps.split(",")):
            for mode in a.modes.split(","):
                worst, alive = run(mode, a.n, a.T, seed, sub, deg=a.deg, in_regime=a.in_regime)
                print(f"seed {seed} substeps {sub:2d} mode {mode:6s}: worst grouped err {worst:.2e} ({alive} envs followed to the end)", flush=True)
