"""`torch.ops.gym_rotor_amd.*` — every entry point of the C-ABI (include/quadrotor_hip.h) registered as a
PyTorch custom op, for callers that want the env step inside a torch program:
`torch.compile(fullgraph=True)`, export, CUDA-graph capture.  Tensors carry the device buffers, the op
body is one ctypes call into libquadrotor_hip.so on the current stream of the buffers' device; there is
no CPU kernel behind these ops (a CPU tensor raises).

Argument convention (the same for every env op):
    state   pos_vel [6, N] (x, v), att_rate [6, N] (the three smaller quaternion components k0 k1 k2, W)
                                                       SoA views (row stride = QrEnv.field_stride)
    per-env integ [8, N]?, params [6, N]?, goal [12, N]?, traj [8, N]?, episode [N] i32?, steps [N] i32?,
            reset_count [ceil(N/64)] i32?              None where the env has no such buffer
    cfg     List[int] = [kind, layout, flags, seed, env_offset, goal_mode, max_episode_steps]   (QrEnv scalars)
    coeffs  List[float] = the QrCoeffs fields in header order (`env_args(env)[2]` / `default_coeffs_list()`)
`env_args(env)` gives (state + per-env tensors, cfg, coeffs) of a QuadVecEnv; `step(env, action, out)` etc.
are thin functional wrappers that call the ops with the env's own buffers — what QuadVecEnv.step does, but
visible to torch's tracer as ONE opaque, mutating op.

Reference interface mirrored: QuadEnv.step / reset / set_goal_state(-> goal buffer) / get_norm_error_state /
get_current_state (gym_rotor/envs/quad.py:142, 171, 409, 413, 421).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib
from .constants import OBS_DIMS

_NS = "gym_rotor_amd"
_COEFF_NAMES = [n for n, _ in _lib.QrCoeffs._fields_]
N_CFG = 7


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _gpu(t: torch.Tensor):
    if t.device.type != "cuda":
        raise RuntimeError("gym_rotor_amd ops run on the GPU only (no CPU kernel exists)")


def default_coeffs_list() -> List[float]:
    c = _lib.default_coeffs()
    return [float(getattr(c, n)) for n in _COEFF_NAMES]


def env_args(env):
    """(tensors, cfg, coeffs) of a QuadVecEnv for the ops below: tensors = (pos_vel, att_rate, integ, params, goal, traj,
    episode, steps, reset_count); cfg / coeffs are the plain-Python copies the env keeps of its QrEnv scalars and
    QrCoeffs (so that torch.compile can trace through this function).  Seeds are passed as int64 (QuadVecEnv accepts
    0 <= seed < 2^63 only, so the op path and the env path draw the same streams)."""
    tensors = (env._pos_vel, env._att_rate, env._integ, env._params, env._goal, env._traj, env._episode, env._steps, env._reset_count)
    return tensors, env._op_cfg, env._op_coeffs


def _env_struct(pos_vel, att_rate, integ, params, goal, traj, episode, steps, reset_count, cfg: Sequence[int], coeffs: Sequence[float]):
    _gpu(pos_vel)
    if len(cfg) != N_CFG:
        raise ValueError(f"cfg must hold {N_CFG} ints: kind, layout, flags, seed, env_offset, goal_mode, max_episode_steps")
    if len(coeffs) != len(_COEFF_NAMES):
        raise ValueError(f"coeffs must hold the {len(_COEFF_NAMES)} QrCoeffs fields")
    e = _lib.QrEnv()
    e.kind, e.layout, e.flags = int(cfg[0]), int(cfg[1]), int(cfg[2])
    e.seed, e.env_offset, e.goal_mode, e.max_episode_steps = int(cfg[3]) & (2 ** 64 - 1), int(cfg[4]), int(cfg[5]), int(cfg[6])
    if pos_vel.shape[0] != 6 or att_rate.shape[0] != 6 or att_rate.shape[1] != pos_vel.shape[1] or att_rate.stride(0) != pos_vel.stride(0):
        raise ValueError("pos_vel and att_rate must be [6, N] SoA views with the same row stride (quadrotor_hip.h: field_stride)")
    e.num_envs, e.field_stride = pos_vel.shape[1], pos_vel.stride(0)
    e.pos_vel, e.att_rate, e.integ, e.params, e.goal, e.traj = _p(pos_vel), _p(att_rate), _p(integ), _p(params), _p(goal), _p(traj)
    e.episode, e.steps, e.reset_count = _p(episode), _p(steps), _p(reset_count)
    for n, v in zip(_COEFF_NAMES, coeffs):
        setattr(e.coeffs, n, float(v))
    return e


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _out_struct(obs0, obs1, reward, reward_raw, done, truncated, final_obs0, final_obs1):
    o = _lib.QrStepOut()
    o.obs0, o.obs1, o.reward, o.reward_raw = _p(obs0), _p(obs1), _p(reward), _p(reward_raw)
    o.done, o.truncated, o.final_obs0, o.final_obs1 = _p(done), _p(truncated), _p(final_obs0), _p(final_obs1)
    return o


# (goal: written by the fused goal generator's stateful modes 2-5, whose xd / vd / b1d / Wd persist there)
_ENV_MUT = ("pos_vel", "att_rate", "integ", "params", "goal", "traj", "episode", "steps", "reset_count")
_OUT_MUT = ("obs0", "obs1", "reward", "reward_raw", "done", "truncated", "final_obs0", "final_obs1")


@torch.library.custom_op(f"{_NS}::qr_step", mutates_args=_ENV_MUT + _OUT_MUT)
def qr_step(pos_vel: torch.Tensor, att_rate: torch.Tensor, integ: Optional[torch.Tensor], params: Optional[torch.Tensor],
            goal: Optional[torch.Tensor], traj: Optional[torch.Tensor], episode: Optional[torch.Tensor], steps: Optional[torch.Tensor],
            reset_count: Optional[torch.Tensor], action: torch.Tensor,
            obs0: Optional[torch.Tensor], obs1: Optional[torch.Tensor], reward: torch.Tensor, reward_raw: Optional[torch.Tensor],
            done: torch.Tensor, truncated: Optional[torch.Tensor], final_obs0: Optional[torch.Tensor], final_obs1: Optional[torch.Tensor],
            substeps: int, cfg: List[int], coeffs: List[float]) -> None:
    """QuadEnv.step for all envs (qr_step).  action [N, A] float32 contiguous; outputs as in QrStepOut."""
    e = _env_struct(pos_vel, att_rate, integ, params, goal, traj, episode, steps, reset_count, cfg, coeffs)
    o = _out_struct(obs0, obs1, reward, reward_raw, done, truncated, final_obs0, final_obs1)
    with torch.cuda.device(pos_vel.device):
        _lib.check(_lib.load().qr_step(C.byref(e), action.data_ptr(), substeps, C.byref(o), _stream(pos_vel)), "qr_step")


@torch.library.custom_op(f"{_NS}::qr_rollout", mutates_args=_ENV_MUT + _OUT_MUT)
def qr_rollout(pos_vel: torch.Tensor, att_rate: torch.Tensor, integ: Optional[torch.Tensor], params: Optional[torch.Tensor],
               goal: Optional[torch.Tensor], traj: Optional[torch.Tensor], episode: Optional[torch.Tensor], steps: Optional[torch.Tensor],
               reset_count: Optional[torch.Tensor], action: torch.Tensor,
               obs0: Optional[torch.Tensor], obs1: Optional[torch.Tensor], reward: torch.Tensor, reward_raw: Optional[torch.Tensor],
               done: torch.Tensor, truncated: Optional[torch.Tensor], final_obs0: Optional[torch.Tensor], final_obs1: Optional[torch.Tensor],
               substeps: int, cfg: List[int], coeffs: List[float]) -> None:
    """T env-steps in one launch (qr_rollout).  action [T, N, A]; outputs [T, N, ...]."""
    e = _env_struct(pos_vel, att_rate, integ, params, goal, traj, episode, steps, reset_count, cfg, coeffs)
    o = _out_struct(obs0, obs1, reward, reward_raw, done, truncated, final_obs0, final_obs1)
    with torch.cuda.device(pos_vel.device):
        _lib.check(_lib.load().qr_rollout(C.byref(e), action.data_ptr(), action.shape[0], substeps, C.byref(o), _stream(pos_vel)), "qr_rollout")


@torch.library.custom_op(f"{_NS}::qr_rollout_actor", mutates_args=_ENV_MUT + _OUT_MUT + ("action_out", "logprob_out"))
def qr_rollout_actor(pos_vel: torch.Tensor, att_rate: torch.Tensor, integ: Optional[torch.Tensor], params: Optional[torch.Tensor],
                     goal: Optional[torch.Tensor], traj: Optional[torch.Tensor], episode: Optional[torch.Tensor], steps: Optional[torch.Tensor],
                     reset_count: Optional[torch.Tensor],
                     actor0: List[torch.Tensor], actor1: List[torch.Tensor], squash: List[int],
                     obs0_in: torch.Tensor, obs1_in: Optional[torch.Tensor], noise: Optional[torch.Tensor],
                     action_out: torch.Tensor, logprob_out: Optional[torch.Tensor],
                     obs0: Optional[torch.Tensor], obs1: Optional[torch.Tensor], reward: torch.Tensor, reward_raw: Optional[torch.Tensor],
                     done: torch.Tensor, truncated: Optional[torch.Tensor], final_obs0: Optional[torch.Tensor], final_obs1: Optional[torch.Tensor],
                     n_steps: int, substeps: int, noise_seed: int, step_base: int, max_action: float, deterministic: bool,
                     cfg: List[int], coeffs: List[float]) -> None:
    """The collection loop with the reference's MLP actor(s) inside the step kernel (qr_rollout_actor).  actorK = the agent's
    tensors in the order fc1_w, fc1_b, fc2_w, fc2_b, mean_w, mean_b, then EITHER log_std OR log_std_w, log_std_b (7 or 8
    tensors; actor1 = [] for COUPLED); squash = QR_ACTOR_* per agent."""
    from .policy import ActorParams, c_actor_array
    e = _env_struct(pos_vel, att_rate, integ, params, goal, traj, episode, steps, reset_count, cfg, coeffs)
    actors = []
    for k, ts in enumerate((actor0, actor1)):
        if not ts:
            continue
        if len(ts) == 7:
            actors.append(ActorParams(*ts[:6], ts[6], None, None, int(squash[k])))
        elif len(ts) == 8:
            actors.append(ActorParams(*ts[:6], None, ts[6], ts[7], int(squash[k])))
        else:
            raise ValueError("an actor is 7 tensors (.., log_std) or 8 (.., log_std_w, log_std_b)")
    arr = c_actor_array(actors)
    pol = _lib.QrPolicyRollout()
    pol.actors = arr
    pol.obs0_in, pol.obs1_in, pol.noise = obs0_in.data_ptr(), _p(obs1_in), _p(noise)
    pol.noise_seed, pol.step_base = int(noise_seed) & (2 ** 64 - 1), int(step_base)
    pol.max_action, pol.deterministic = float(max_action), int(bool(deterministic))
    pol.action_out, pol.logprob_out = action_out.data_ptr(), _p(logprob_out)
    o = _out_struct(obs0, obs1, reward, reward_raw, done, truncated, final_obs0, final_obs1)
    with torch.cuda.device(pos_vel.device):
        _lib.check(_lib.load().qr_rollout_actor(C.byref(e), C.byref(pol), n_steps, substeps, C.byref(o), _stream(pos_vel)), "qr_rollout_actor")


@torch.library.custom_op(f"{_NS}::qr_error_obs", mutates_args=("integ", "obs0", "obs1"))
def qr_error_obs(pos_vel: torch.Tensor, att_rate: torch.Tensor, integ: torch.Tensor, goal: Optional[torch.Tensor],
                 obs0: torch.Tensor, obs1: Optional[torch.Tensor], cfg: List[int], coeffs: List[float], fmt: int = -1) -> None:
    """QuadEnv.get_norm_error_state(framework) for all envs (qr_error_obs / qr_error_obs_format): advances the integral terms like
    the reference.  fmt = -1: the env's own format; 1 (= QR_KIND_COUPLED): MONO rows [N,23]; 2 (= QR_KIND_DECOUPLED): MODUL rows
    [N,15] + [N,3] — on either wrapper kind (quad.py:452-466)."""
    e = _env_struct(pos_vel, att_rate, integ, None, goal, None, None, None, None, cfg, coeffs)
    e.goal_mode = 0
    with torch.cuda.device(pos_vel.device):
        if fmt < 0:
            _lib.check(_lib.load().qr_error_obs(C.byref(e), obs0.data_ptr(), _p(obs1), _stream(pos_vel)), "qr_error_obs")
        else:
            _lib.check(_lib.load().qr_error_obs_format(C.byref(e), int(fmt), obs0.data_ptr(), _p(obs1), _stream(pos_vel)), "qr_error_obs_format")


@torch.library.custom_op(f"{_NS}::qr_reset", mutates_args=("pos_vel", "att_rate", "integ", "params", "episode", "steps"))
def qr_reset(pos_vel: torch.Tensor, att_rate: torch.Tensor, integ: Optional[torch.Tensor], params: Optional[torch.Tensor],
             episode: torch.Tensor, steps: Optional[torch.Tensor], mask: Optional[torch.Tensor], cfg: List[int], coeffs: List[float]) -> None:
    """QuadEnv.reset for masked envs (qr_reset); cfg's flags select train / eval (QR_FLAG_EVAL_RESET) and UDM."""
    e = _env_struct(pos_vel, att_rate, integ, params, None, None, episode, steps, None, cfg, coeffs)
    e.goal_mode = 0
    with torch.cuda.device(pos_vel.device):
        _lib.check(_lib.load().qr_reset(C.byref(e), _p(mask), _stream(pos_vel)), "qr_reset")


@torch.library.custom_op(f"{_NS}::qr_traj_start", mutates_args=("traj", "goal"))
def qr_traj_start(pos_vel: torch.Tensor, att_rate: torch.Tensor, traj: torch.Tensor, goal: Optional[torch.Tensor], episode: Optional[torch.Tensor],
                  mask: Optional[torch.Tensor], draws: Optional[torch.Tensor], cfg: List[int], coeffs: List[float]) -> None:
    """TrajectoryGenerator.mark_traj_start for masked envs from the current state (qr_traj_start); draws [3, N] injects
    theta_b1d, t_traj, w_b1d, default: the env's stream (seed, global env id, episode).  goal: the env's goal buffer — required
    (and reset to a fresh generator's xd = vd = Wd = 0, b1d = e1) in the stateful modes 2-5, untouched otherwise."""
    e = _env_struct(pos_vel, att_rate, None, None, goal, traj, episode, None, None, cfg, coeffs)
    with torch.cuda.device(pos_vel.device):
        _lib.check(_lib.load().qr_traj_start(C.byref(e), _p(mask), _p(draws), _stream(pos_vel)), "qr_traj_start")


@torch.library.custom_op(f"{_NS}::qr_get_state", mutates_args=("rows",))
def qr_get_state(pos_vel: torch.Tensor, att_rate: torch.Tensor, rows: torch.Tensor, cfg: List[int], coeffs: List[float]) -> None:
    """QuadEnv.get_current_state: float64 rows [N, 18] = (x, v, vec_F(R(q)), W) (qr_get_state)."""
    e = _env_struct(pos_vel, att_rate, None, None, None, None, None, None, None, cfg, coeffs)
    e.goal_mode = 0
    with torch.cuda.device(pos_vel.device):
        _lib.check(_lib.load().qr_get_state(C.byref(e), rows.data_ptr(), _stream(pos_vel)), "qr_get_state")


@torch.library.custom_op(f"{_NS}::qr_set_state", mutates_args=("pos_vel", "att_rate", "rejected"))
def qr_set_state(pos_vel: torch.Tensor, att_rate: torch.Tensor, rows: torch.Tensor, mask: Optional[torch.Tensor],
                 rejected: Optional[torch.Tensor], cfg: List[int], coeffs: List[float]) -> None:
    """Assignment to QuadEnv.state from float64 rows [N, 18] (qr_set_state); `rejected` (int32 [1], zeroed by the caller)
    counts rows whose attitude block has no nearest rotation."""
    e = _env_struct(pos_vel, att_rate, None, None, None, None, None, None, None, cfg, coeffs)
    e.goal_mode = 0
    with torch.cuda.device(pos_vel.device):
        _lib.check(_lib.load().qr_set_state(C.byref(e), rows.data_ptr(), _p(mask), _p(rejected), _stream(pos_vel)), "qr_set_state")


@torch.library.custom_op(f"{_NS}::qr_gae", mutates_args=("advantage", "td_target"))
def qr_gae(reward: torch.Tensor, done: torch.Tensor, value: torch.Tensor, gamma: float, lam: float,
           advantage: torch.Tensor, td_target: torch.Tensor, next_value: Optional[torch.Tensor] = None) -> None:
    """GAE reverse scan (qr_gae): reward/done [T, M]; value [T+1, M] (row T = bootstrap) — or, with next_value [T, M]
    (the reference's V(obs_next)), value [T, M]."""
    _gpu(reward)
    T, M = reward.shape[0], reward[0].numel()
    with torch.cuda.device(reward.device):
        rc = _lib.load().qr_gae(reward.data_ptr(), done.data_ptr(), value.data_ptr(), _p(next_value), T, M, gamma, lam,
                                advantage.data_ptr(), td_target.data_ptr(), None, _stream(reward))
    _lib.check(rc, "qr_gae")


# ----------------------------------------------------------------------------------------------------------------
# functional wrappers over a QuadVecEnv's own buffers
# ----------------------------------------------------------------------------------------------------------------
def _outs(env, out):
    if out is None:
        out = {"obs0": env._obs0, "obs1": env._obs1, "reward": env._reward, "reward_raw": env._reward_raw, "terminated": env._done,
               "truncated": env._trunc if env._steps is not None else None, "final_obs0": env._final0, "final_obs1": env._final1}
    trunc = out.get("truncated") if env._steps is not None else None
    return (out.get("obs0"), out.get("obs1"), out["reward"], out.get("reward_raw"), out["terminated"], trunc,
            out.get("final_obs0"), out.get("final_obs1"))


def step(env, action: torch.Tensor, out: Optional[dict] = None) -> None:
    """`env.step(action)` as ONE torch op (same buffers, same bits): results land in the env's output tensors (or `out`)."""
    t, cfg, co = env_args(env)
    torch.ops.gym_rotor_amd.qr_step(*t, action, *_outs(env, out), env.substeps, cfg, co)


def rollout(env, actions: torch.Tensor, out: dict) -> None:
    t, cfg, co = env_args(env)
    torch.ops.gym_rotor_amd.qr_rollout(*t, actions, *_outs(env, out), env.substeps, cfg, co)


def rollout_actor(env, actors, n_steps: int, obs, out: dict, noise: Optional[torch.Tensor] = None, noise_seed: Optional[int] = None,
                  step_base: int = 0, max_action: float = 1.0, deterministic: bool = False) -> None:
    t, cfg, co = env_args(env)
    obs = [obs] if isinstance(obs, torch.Tensor) else list(obs)
    lists, squash = [], []
    for a in actors:
        ts = [a.fc1_w, a.fc1_b, a.fc2_w, a.fc2_b, a.mean_w, a.mean_b]
        ts += [a.log_std] if a.log_std_w is None else [a.log_std_w, a.log_std_b]
        lists.append(ts); squash.append(int(a.squash))
    while len(lists) < 2:
        lists.append([]); squash.append(0)
    torch.ops.gym_rotor_amd.qr_rollout_actor(*t, lists[0], lists[1], squash, obs[0], obs[1] if len(obs) > 1 else None, noise,
                                             out["action"], out.get("logprob"), *_outs(env, out), n_steps, env.substeps,
                                             (env.seed if noise_seed is None else noise_seed) & (2 ** 63 - 1), step_base, max_action,
                                             deterministic, cfg, co)


def error_obs(env, framework: Optional[str] = None, out=None) -> None:
    """get_norm_error_state through the op: the env's own format into its own buffers, or (framework = "MONO" | "MODUL", out = the
    row tensor(s) to fill) the format the argument names."""
    t, cfg, co = env_args(env)
    if framework is None or framework == env.framework:
        torch.ops.gym_rotor_amd.qr_error_obs(t[0], t[1], env._integ, env._goal, env._obs0, env._obs1, cfg, co)
        return
    if framework not in ("MONO", "MODUL"):
        raise ValueError(f"framework must be 'MONO' or 'MODUL', got {framework!r}")
    dims = OBS_DIMS["coupled" if framework == "MONO" else "decoupled"]
    if out is None:
        raise ValueError(f"error_obs(framework={framework!r}) on a {env.framework} env writes the OTHER format: pass out= the row tensor(s) "
                         f"to fill, float32 {[(env.num_envs, d) for d in dims]}")
    rows = [out] if isinstance(out, torch.Tensor) else list(out)
    if len(rows) != len(dims):
        raise ValueError(f"framework {framework!r} has {len(dims)} row tensor(s), got {len(rows)}")
    for r, d in zip(rows, dims):
        if tuple(r.shape) != (env.num_envs, d) or r.dtype != torch.float32 or r.device != env.device or not r.is_contiguous():
            raise ValueError(f"out rows must be contiguous float32 [{env.num_envs}, {d}] on {env.device}")
    fmt = _lib.KIND_ID["coupled" if framework == "MONO" else "decoupled"]
    torch.ops.gym_rotor_amd.qr_error_obs(t[0], t[1], env._integ, env._goal, rows[0], rows[1] if len(rows) > 1 else None, cfg, co, fmt)


def reset(env, env_type: str = "train", mask: Optional[torch.Tensor] = None) -> None:
    """What QuadVecEnv.reset launches: qr_reset, then (fused goal generator) qr_traj_start for the same envs — main.py:226-227;
    the previous episode's observation rows stop being the env's current observation."""
    t, cfg, co = env_args(env)
    rcfg = list(cfg)
    rcfg[2] = (rcfg[2] & ~3) | (2 if env_type == "eval" else 0)  # clear QR_FLAG_AUTO_RESET, select QR_FLAG_EVAL_RESET
    torch.ops.gym_rotor_amd.qr_reset(t[0], t[1], env._integ, env._params, env._episode, env._steps, mask, rcfg, co)
    if env.goal_mode is not None:
        torch.ops.gym_rotor_amd.qr_traj_start(t[0], t[1], env._traj, env._goal, env._episode, mask, None, cfg, co)
    env._last_obs = None


def get_state(env, rows: torch.Tensor) -> None:
    t, cfg, co = env_args(env)
    torch.ops.gym_rotor_amd.qr_get_state(t[0], t[1], rows, cfg, co)


def set_state(env, rows: torch.Tensor, mask: Optional[torch.Tensor] = None, rejected: Optional[torch.Tensor] = None) -> None:
    t, cfg, co = env_args(env)
    torch.ops.gym_rotor_amd.qr_set_state(t[0], t[1], rows, mask, rejected, cfg, co)
