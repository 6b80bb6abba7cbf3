"""CPU oracle for goal generation.  TEST INFRASTRUCTURE ONLY.

NumPy restatement of modes 0 (idle / warm-up) and 1 (hovering) of the reference's
utils/trajectory_generator.py — the caller that feeds `set_goal_state` before every step
(main.py:145-147, 226-229).  Random draws are arguments, so the oracle can be pinned against
the reference with injected draws (tools/gen_golden.py -> tests/golden/trajgoal_*.npz).
Only tests/ and __graft_entry__.smoke() may import this file.
"""
from __future__ import annotations

import numpy as np

from .quad_oracle import DT


def _R(state):
    return np.swapaxes(np.asarray(state, dtype=np.float64)[..., 6:15].reshape(-1, 3, 3), 1, 2)


def traj_start_batch(state, mode, theta_b1d=None, t_traj=None, w_b1d=None):
    """mark_traj_start(state) (trajectory_generator.py:176-204) + the episode-start branch of
    calculate_desired: mode 0 (:141-148) b1d = Rz(theta_b1d) b1_proj; mode 1 (:253-266) x_init,
    smooth_term = -ln(0.001)/t_traj, w_b1d.  Returns the generator state as a dict of arrays."""
    state = np.atleast_2d(np.asarray(state, dtype=np.float64))
    n = state.shape[0]
    R = _R(state)
    theta_init = np.arctan2(R[:, 1, 0], R[:, 0, 0])  # update_initial_state (:199-204)
    tr = {"calls": np.zeros(n), "theta_init": theta_init, "mode": mode}
    if mode == 0:
        th = theta_init + np.asarray(theta_b1d, dtype=np.float64)
        tr["b1d"] = np.stack([np.cos(th), np.sin(th), np.zeros(n)], 1)
    elif mode == 1:
        tr["x_init"] = state[:, 0:3].copy()
        tr["smooth"] = -np.log(0.001) / (np.asarray(t_traj, dtype=np.float64) * np.ones(n))
        tr["w_b1d"] = np.asarray(w_b1d, dtype=np.float64) * np.ones(n)
    elif mode == 6:  # eight_shaped_curve (:427-449): centre = current position, no draws
        tr["center"] = state[:, 0:3].copy()
    else:
        raise ValueError("only TrajectoryGenerator modes 0, 1 and 6 are in scope")
    return tr


EIGHT = dict(T=9.0, A1=1.5, A2=1.0, w_b1d=0.349066, alt_d=-0.6, eps=0.01, count=3)  # :98-110


def get_desired_batch(tr, state, dt=DT, eight=None):
    """get_desired(state, mode) (:113-173): advances t by dt, returns xd, vd, b1d, b1d_dot, Wd [N,3]."""
    state = np.atleast_2d(np.asarray(state, dtype=np.float64))
    n = state.shape[0]
    tr["calls"] = tr["calls"] + 1.0  # update_current_time (:224-229)
    if tr["mode"] == 6:  # eight_shaped_curve (:451-505)
        p = dict(EIGHT, **(eight or {}))
        t = np.minimum(tr["calls"] * dt, p["count"] * p["T"])[:, None]
        w1, w2, k = 2 * np.pi / p["T"], 4 * np.pi / p["T"], -np.log(p["eps"]) / p["T"]
        e, de = 1.0 - np.exp(-k * t), k * np.exp(-k * t)
        c = tr["center"]
        za = (c[:, 2:3] - p["alt_d"]) / 2
        xd = np.concatenate([p["A2"] * np.sin(w2 * t) * e + c[:, 0:1], p["A1"] * (np.cos(w1 * t) - 1.0) * e + c[:, 1:2],
                             za * (1 - np.cos(w1 * t)) + c[:, 2:3]], 1)
        vd = np.concatenate([p["A2"] * (w2 * np.cos(w2 * t) * e + np.sin(w2 * t) * de),
                             p["A1"] * (-w1 * np.sin(w1 * t) * e + (np.cos(w1 * t) - 1.0) * de), za * w1 * np.sin(w1 * t)], 1)
        term = p["w_b1d"] * t * e + tr["theta_init"][:, None]
        dterm = p["w_b1d"] * (e + t * de)
        z = np.zeros_like(term)
        b1d = np.concatenate([np.cos(term), np.sin(term), z], 1)
        b1d_dot = np.concatenate([-np.sin(term) * dterm, np.cos(term) * dterm, z], 1)
    elif tr["mode"] == 0:
        xd, vd = np.zeros((n, 3)), np.zeros((n, 3))
        b1d, b1d_dot = tr["b1d"], np.zeros((n, 3))
    else:  # hovering (:268-277), x_goal = 0
        t = (tr["calls"] * dt)[:, None]
        sm, w = tr["smooth"][:, None], tr["w_b1d"][:, None]
        e = np.exp(-sm * t)
        xd, vd = tr["x_init"] * e, -tr["x_init"] * sm * e
        ang = w * t + tr["theta_init"][:, None]
        z = np.zeros_like(ang)
        b1d = np.concatenate([np.cos(ang), np.sin(ang), z], 1)
        b1d_dot = np.concatenate([-w * np.sin(ang), w * np.cos(ang), z], 1)
    # Wd (:165-172)
    R, W = _R(state), state[:, 15:18]
    b1, b2, b3 = R[:, :, 0], R[:, :, 1], R[:, :, 2]
    b3_dot = W[:, 1:2] * b1 - W[:, 0:1] * b2  # R hat(W) e3
    dot = lambda a, b: (a * b).sum(1, keepdims=True)
    b1c = b1d - dot(b1d, b3) * b3
    b1c_dot = b1d_dot - (dot(b1d_dot, b3) * b3 + dot(b1d, b3_dot) * b3 + dot(b1d, b3) * b3_dot)
    Wd = np.zeros((n, 3))
    Wd[:, 2] = (b3 * np.cross(b1c, b1c_dot)).sum(1)
    return xd, vd, b1d, b1d_dot, Wd
