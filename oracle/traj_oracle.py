"""CPU oracle for goal generation.  TEST INFRASTRUCTURE ONLY.

NumPy restatement of the reference's utils/trajectory_generator.py — modes 0 (idle / warm-up), 1 (hovering), 6 (eight-shaped curve)
and, with the generator's persistent fields and flags carried env by env, 2 (take-off), 3 (landing), 4 (stay), 5 (circle) — the caller that feeds `set_goal_state` before every step
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
    elif mode in (2, 3, 4, 5):  # take-off / landing / stay / circle (:279-416): mark_traj_start (:176-196) of a FRESH generator
        tr["x_init"] = state[:, 0:3].copy()
        tr["t"] = np.zeros(n)                                     # accumulated like the reference (t = t + dt, :224-229), not calls * dt
        for k in ("started", "complete", "manual", "manual_init", "landed"):
            tr[k] = np.zeros(n, bool)
        tr["xd"], tr["vd"], tr["Wd"] = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))   # __init__ (:54-55, 67-68)
        tr["b1d"] = np.tile(np.array([1.0, 0.0, 0.0]), (n, 1))
        tr["b1d_dot"] = np.zeros((n, 3))
    else:
        raise ValueError("TrajectoryGenerator mode must be 0..6")
    return tr


# take-off / landing / circle constants (trajectory_generator.py:81-94)
MODES = dict(takeoff_end_height=-0.5, takeoff_velocity=-0.05, landing_velocity=1.0, landing_motor_cutoff_height=-0.25,
             num_circles=2, circle_radius=0.7, circle_linear_v=0.4, circle_W=0.4)


def _heading(R):
    """get_current_b1 (:215-218): (cos theta, sin theta, 0), theta = atan2(b1[1], b1[0])."""
    th = np.arctan2(R[1, 0], R[0, 0])
    return np.array([np.cos(th), np.sin(th), 0.0]), th


def _stateful_modes(tr, state, dt):
    """calculate_desired (:135-160) for modes 2-5, env by env, with the generator's persistent fields (xd, vd, b1d, b1d_dot, Wd and
    the flags) carried in `tr` exactly as the reference object carries them.  Returns the mask of envs that returned through
    manual() — for those the reference skips the Wd computation (early `return`, :137-139) and Wd keeps its last value."""
    m, mode = MODES, tr["mode"]
    n = state.shape[0]
    R = _R(state)
    skip_wd = np.zeros(n, bool)
    for i in range(n):
        x, v = state[i, 0:3], state[i, 3:6]
        xd, vd = tr["xd"][i], tr["vd"][i]
        if tr["manual"][i]:  # manual() (:232-250)
            if not tr["manual_init"][i]:
                xd[:], vd[:] = x, v                                # set_desired_states_to_current
                tr["b1d"][i], tr["theta_init"][i] = _heading(R[i])
                tr["manual_init"][i] = True
            vd[:] = 0.0
            th = tr["theta_init"][i]
            tr["b1d"][i] = [np.cos(th), np.sin(th), 0.0]
            skip_wd[i] = True
            continue
        if mode == 2:  # takeoff (:279-309)
            if not tr["started"][i]:
                xd[:], vd[:], tr["Wd"][i] = 0.0, 0.0, 0.0          # set_desired_states_to_zero
                xd[0], xd[1] = x[0], x[1]
                tr["x_init"][i] = x
                tr["b1d"][i], _ = _heading(R[i])
                tr["started"][i] = True
            tr["t"][i] += dt
            t_traj = (m["takeoff_end_height"] - tr["x_init"][i, 2]) / m["takeoff_velocity"]
            if tr["t"][i] < t_traj:
                xd[2] = tr["x_init"][i, 2] + m["takeoff_velocity"] * tr["t"][i]
            elif np.linalg.norm(xd - x) < 0.04:                   # waypoint_reached (:312-318)
                xd[2], vd[2] = m["takeoff_end_height"], 0.0
                tr["complete"][i] = tr["manual"][i] = True         # mark_traj_end(True)
        elif mode == 3:  # land (:321-349)
            if not tr["started"][i]:
                xd[:], vd[:] = x, v
                tr["b1d"][i], _ = _heading(R[i])
                tr["started"][i] = True
            tr["t"][i] += dt
            t_traj = (m["landing_motor_cutoff_height"] - tr["x_init"][i, 2]) / m["landing_velocity"]   # (x_init: mark_traj_start's; x at the first call is the same state)
            if tr["t"][i] < t_traj:
                xd[2] = tr["x_init"][i, 2] + m["landing_velocity"] * tr["t"][i]
            elif x[2] > m["landing_motor_cutoff_height"]:
                xd[2], vd[2] = m["landing_motor_cutoff_height"], 0.0
                tr["complete"][i] = tr["landed"][i] = True         # mark_traj_end(False)
            else:
                xd[2], vd[2] = m["landing_motor_cutoff_height"], m["landing_velocity"]
        elif mode == 4:  # stay (:352-357)
            if not tr["started"][i]:
                xd[:], vd[:] = x, v
                tr["b1d"][i], _ = _heading(R[i])
                tr["started"][i] = True
            tr["complete"][i] = tr["manual"][i] = True
        else:  # circle (:360-416); the centre is the state of the first call = x_init
            if not tr["started"][i]:
                xd[:], vd[:] = x, v
                tr["b1d"][i], _ = _heading(R[i])
                tr["started"][i] = True
            tr["t"][i] += dt
            r, lv, w = m["circle_radius"], m["circle_linear_v"], m["circle_W"]
            c = tr["x_init"][i]
            t_traj = r / lv + m["num_circles"] * 2 * np.pi / w
            if tr["t"][i] < r / lv:
                xd[0], vd[0] = c[0] + lv * tr["t"][i], lv
            elif tr["t"][i] < t_traj:
                t = tr["t"][i] - r / lv
                th = w * t
                xd[0], vd[0] = r * np.cos(th) + c[0], -r * w * np.sin(th)
                xd[1], vd[1] = r * np.sin(th) + c[1], r * w * np.cos(th)
                tb = w * t + np.pi
                tr["b1d"][i] = [np.cos(tb), np.sin(tb), 0.0]
                tr["b1d_dot"][i] = [-w * np.sin(tb), w * np.cos(tb), 0.0]
            else:
                tr["complete"][i] = tr["manual"][i] = True
    return skip_wd


EIGHT = dict(T=9.0, A1=1.5, A2=1.0, w_b1d=0.349066, alt_d=-0.6, eps=0.01, count=3)  # :98-110


def get_desired_batch(tr, state, dt=DT, eight=None):
    """get_desired(state, mode) (:113-173): advances t by dt, returns xd, vd, b1d, b1d_dot, Wd [N,3]."""
    state = np.atleast_2d(np.asarray(state, dtype=np.float64))
    n = state.shape[0]
    tr["calls"] = tr["calls"] + 1.0  # update_current_time (:224-229)
    skip_wd = np.zeros(n, bool)
    if tr["mode"] in (2, 3, 4, 5):
        skip_wd = _stateful_modes(tr, state, dt)
        xd, vd, b1d, b1d_dot = tr["xd"].copy(), tr["vd"].copy(), tr["b1d"].copy(), tr["b1d_dot"].copy()
    elif tr["mode"] == 6:  # eight_shaped_curve (:451-505)
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
    if tr["mode"] in (2, 3, 4, 5):  # (manual mode returns before this computation: Wd keeps its last value)
        Wd[skip_wd] = tr["Wd"][skip_wd]
        tr["Wd"] = Wd.copy()
    return xd, vd, b1d, b1d_dot, Wd
