"""Flight-log writer / reader compatible with the reference's `.dat` logs.

Writer = what `main.py:344-352,382-389` does for one eval episode: per step the row
`[action | state(18) eIx(3) eb1 eIb1 | xd(3) vd(3) b1c(3) Wd(3)]` (b1c = b1d - (b1d.b3) b3), saved
with `np.savetxt(..., header="Actions and States\\naction[0], ..., state[0], ..., command[0], ...",
fmt='%.10f')`.  Reader = the column map of `draw_plot.py:24-47,74-77` (MODUL: 5 action columns,
MONO: 4).  Lets `draw_plot.py`-style tooling consume episodes of the batched env unchanged.
"""
from __future__ import annotations

import numpy as np

HEADER = "Actions and States\naction[0], ..., state[0], ..., command[0], ..."
ACTION_COLS = {"MODUL": 5, "MONO": 4}


def save_flight_log(path, actions, states, eIx, eb1, eIb1, xd, vd, b1c, Wd):
    """All arguments are per-step arrays [T, ...]; `states` is the [T,18] state BEFORE the action
    (main.py:346), eIx/eb1/eIb1 the de-normalised integral / heading errors (utils.py:21-40)."""
    T = len(actions)
    cols = [np.reshape(np.asarray(a, dtype=np.float64), (T, -1)) for a in (actions, states, eIx, eb1, eIb1, xd, vd, b1c, Wd)]
    data = np.column_stack(cols)
    np.savetxt(path, data, header=HEADER, fmt="%.10f")
    return data


def load_flight_log(path, framework: str = "MODUL") -> dict:
    """draw_plot.py:14-47: split a `.dat` log into named columns."""
    data = np.loadtxt(path) if isinstance(path, (str, bytes)) or hasattr(path, "read") else np.asarray(path)
    na = ACTION_COLS[framework]
    if data.shape[1] != na + 23 + 12:
        raise ValueError(f"{framework} log must have {na + 35} columns, got {data.shape[1]}")
    obs, cmd = data[:, na:na + 23], data[:, na + 23:]
    return {"action": data[:, :na], "state": obs[:, 0:18], "x": obs[:, 0:3], "v": obs[:, 3:6], "R_vec": obs[:, 6:15],
            "W": obs[:, 15:18], "eIx": obs[:, 18:21], "eb1": obs[:, 21], "eIb1": obs[:, 22],
            "xd": cmd[:, 0:3], "vd": cmd[:, 3:6], "b1c": cmd[:, 6:9], "Wd": cmd[:, 9:12]}


class FlightLogger:
    """Collects one env of a QuadVecEnv (wrapper kinds) step by step, like the `save_log` branch of
    `Learner.eval_policy` (main.py:344-352), then writes the reference's `.dat` format."""

    def __init__(self, env, index: int = 0):
        if env.kind == "quad":
            raise ValueError("flight logs are defined for the coupled / decoupled wrappers")
        self.env, self.i = env, int(index)
        self.rows = {k: [] for k in ("act", "state", "eIx", "eb1", "eIb1", "xd", "vd", "b1c", "Wd")}

    def record(self, action, obs, goal=None):
        """Call BEFORE env.step(action): `obs` is the current observation (tuple for decoupled),
        `goal` = (xd, vd, b1d, Wd) rows or None for the hover default."""
        e, i = self.env, self.i
        st = e.get_current_state()[i].cpu().numpy()
        obs = [obs] if not isinstance(obs, (tuple, list)) else list(obs)
        o0 = obs[0][i].cpu().numpy().astype(np.float64)
        if e.kind == "decoupled":
            o1 = obs[1][i].cpu().numpy().astype(np.float64)
            eIx, eb1, eIb1 = o0[3:6] * e.eIx_lim, o1[0] * np.pi, o1[1] * e.eIb1_lim
        else:
            eIx, eb1, eIb1 = o0[3:6] * e.eIx_lim, o0[18] * np.pi, o0[19] * e.eIb1_lim
        if goal is None:
            xd, vd, b1d, Wd = np.zeros(3), np.zeros(3), np.array([1.0, 0.0, 0.0]), np.zeros(3)
        else:
            xd, vd, b1d, Wd = (np.asarray(g[i].cpu() if hasattr(g, "cpu") else g[i], dtype=np.float64) for g in goal)
        b3 = st[12:15]
        b1c = b1d - np.dot(b1d, b3) * b3  # main.py:349-351
        a = action[i].cpu().numpy() if hasattr(action, "cpu") else np.asarray(action[i])
        for k, v in zip(self.rows, (a, st, eIx, eb1, eIb1, xd, vd, b1c, Wd)):
            self.rows[k].append(np.asarray(v, dtype=np.float64))

    def save(self, path):
        r = self.rows
        return save_flight_log(path, r["act"], r["state"], r["eIx"], r["eb1"], r["eIb1"], r["xd"], r["vd"], r["b1c"], r["Wd"])
