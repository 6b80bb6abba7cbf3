"""QuadVecEnv — N independent quadrotors stepped by one fused HIP launch on an MI355X.

Host-side mirror of the reference's env interface for the env.step() hot path
(QuadEnv / CoupledWrapper / DecoupledWrapper: gym_rotor/envs/quad.py:142-466,
gym_rotor/wrappers/{coupled,decoupled}_yaw_wrapper.py).  Same method names and argument
meaning, batched over a leading env axis:

    step(actions[N,A]) -> (obs, reward[N,n_agents], terminated[N,n_agents], truncated[N], info)
    reset(env_type='train'|'eval', seed=None, options=None, mask=None) -> float32 state [N,18]
    set_goal_state(xd, vd, b1d, b1d_dot, Wd); get_norm_error_state(framework=None)
    get_current_state(); close()

All tensors live on the GPU; `step` performs no host synchronisation.  The arithmetic is
done ONLY by libquadrotor_hip.so (include/quadrotor_hip.h) — there is no torch/NumPy
fallback, and constructing the env without the library or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from . import _lib, launch_cache
from .constants import ACTION_DIM, FRAMEWORK, KINDS, N_AGENTS, OBS_DIMS, QuadConstants
from .spaces import Box


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# Raw handle of torch's current stream / current device with as little Python as possible: step()
# is called once per env-step and the launch itself is only a few microseconds.
try:
    _raw_stream = torch._C._cuda_getCurrentRawStream
    _get_device = torch._C._cuda_getDevice
    if not callable(_raw_stream) or not callable(_get_device):
        raise AttributeError
except AttributeError:  # pragma: no cover - older/newer torch without the private accessors
    def _raw_stream(idx):
        return torch.cuda.current_stream(idx).cuda_stream

    def _get_device():
        return torch.cuda.current_device()


def _pack_att_rate(old: torch.Tensor) -> torch.Tensor:
    """[7, N] rows (q = w,x,y,z; W) -> the [6, N] storage form (quadrotor_hip.h: the three components of q of smaller
    magnitude, the dropped one made positive, its index in the two lowest mantissa bits of the first kept one; W)."""
    q, W = old[:4], old[4:]
    idx = q.abs().argmax(0)                                             # ties: the lowest index, like pack_quat
    sgn = torch.where(q.gather(0, idx[None])[0] < 0, -1.0, 1.0).to(q.dtype)
    rows = torch.arange(3, device=old.device)[:, None]
    keep = rows + (rows >= idx[None]).to(rows.dtype)                    # the three indices != idx, in order
    k = q.gather(0, keep) * sgn[None]
    ibits = torch.int64 if k.dtype == torch.float64 else torch.int32
    k0 = (k[0].contiguous().view(ibits) & ~3) | idx.to(ibits)
    return torch.cat([k0.view(k.dtype)[None], k[1:], W], 0)


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


def _field_stride(n: int) -> int:
    """Elements between consecutive fields of the SoA buffers: N rounded up to a multiple of 4.
    (Padding the stride off powers of two was measured on MI355X and makes no difference: the
    memory system hashes channels; tools/microbench.py sweep)"""
    return (n + 3) // 4 * 4


class QuadVecEnv:
    """Batched Quad-v0 / CoupledWrapper / DecoupledWrapper.

    kind            'quad' | 'coupled' | 'decoupled'  (MONO: coupled, MODUL: decoupled)
    num_envs        N envs owned by this object (this GPU's shard)
    substeps        fixed 4th-order substeps per env-step replacing solve_ivp(DOP853) (quad.py:265): 1 = one RK4 step; with two or
                    more the default layout takes Magnus substeps (half the instructions), the uniform layouts RK4
    w_adapt         [rad/s] rate-adaptive substepping, the stand-in for DOP853's error control: a
                    wavefront holding an env with max|W_i| > w_adapt takes ceil(max|W_i| / w_adapt)
                    times the substeps.  Never active in regime (|W| < 2 pi), and with auto_reset it
                    cannot trigger (the plain kernel is launched); it keeps envs that are stepped on
                    far beyond termination inside the 1e-5 trajectory bar.  0 disables it.
    layout          internal state precision (the state is 12 words: x, v, the unit quaternion q stored as its three
                    smaller components, W):
                    'mixed' (default) x,v float32 + q,W float64; W integrated and q accumulated in
                    float64, RK4 stage quaternions / thrust direction in float32: 1000-step
                    trajectories within ~4e-6 of the reference (the float32 ulp of x, v);
                    'f64' all float64 storage and arithmetic: within ~3e-7, 25 % more bytes per env-step;
                    'f32' all float32 (fast, outside the 1e-5 parity bar)
    use_UDM         per-env domain randomisation at reset (quad.py:359-404)
    auto_reset      re-sample terminated/truncated envs inside the step launch; the returned
                    observation is then the first observation of the new episode
    reset_on_done   (without auto_reset) a promise: the caller resets every env that step() reports terminated / truncated
                    before stepping it again — the reference's own training loop (main.py:183-186, 212-230).  No env then
                    starts a step outside the termination bounds, the rate-adaptive path cannot trigger, and step() runs the
                    kernel compiled without it: the same arithmetic, bit for bit, as with auto_reset, 8 % faster than the
                    free-run default.  Leave False for free runs / evaluation flights that fly on after termination.  The promise
                    is about what happens BETWEEN step() calls: a multi-step rollout() keeps the rate-adaptive kernel whatever
                    this flag says (nobody can reset an env between two steps of one launch)
    max_episode_steps  >0 sets truncated when an episode reaches that many steps
    env_offset      global index of local env 0 (multi-GPU sharding; part of the RNG key)
    goal_mode       None: goals come from set_goal_state() (hover default).  0..6: the reference's
                    TrajectoryGenerator mode 0 (idle/warm-up: xd = vd = 0, b1d drawn per episode),
                    1 (hovering: exponential approach of the origin + yaw rate), 2 (take-off), 3 (landing),
                    4 (stay), 5 (circle) or 6 (eight-shaped curve; parameters eight_* of QuadConstants) is evaluated
                    INSIDE the step launch from the pre-step state, as main.py:145-147 does on the
                    host every step; see mark_traj_start() / get_desired()
    final_obs       with auto_reset: also keep the TERMINAL observation rows of the envs that were
                    re-sampled in the last step (`final_observation()`; the reference's obs_next of
                    that transition, main.py:163-178) — what a learner bootstraps V(s') from
    helper          launch-rule override of step() (speed only, no result bit changes): None = the library's rule (a helper
                    wavefront per 64-env tile on small grids, thresholds measured on MI355X; the environment variables
                    QR_HELPER_GRID / QR_HELPER_GRID_WRAP / QR_HELPER_GRID_ROLLOUT override them per process); True / False force /
                    forbid it.  helper_rollout: the same for rollout() / rollout_actor(), a separate choice (their crossover
                    is a different one; a choice timed on step() says nothing about them)
    autotune        the launch rule's compiled-in thresholds are crossovers measured on other boxes; boxes differ.  None (default):
                    if `helper` is None and this env's grid lies within +-25 % of the rule's threshold, use the choice RECORDED for
                    (device, library, kind, size, substeps, ...) in ~/.cache/gym_rotor_amd/launch.json (launch_cache.py;
                    QR_LAUNCH_CACHE moves / disables it) — a file read, no launch; without a record the compiled rule stays in
                    force (QR_AUTOTUNE=time in the environment: time it then and record).  Away from the threshold the rule is
                    unambiguous and nothing is read.  True: time both instantiations now (autotune_launch(), ~0.2 s, a hipGraph
                    capture and a synchronise: not something a constructor should do unasked) and record the result.
                    False (or QR_AUTOTUNE=0 in the environment): the rule as compiled, never the cache
    obs_rows        write float32 observation rows [N,D].  Always on for the wrappers.  For
                    kind='quad' the observation is the next state (quad.py:269-271): True writes
                    it as float32 [N,18] rows each step; False (default) writes nothing and
                    step() returns obs=None (use get_current_state() when it is needed)
    """

    metadata = {"render_modes": []}

    def __init__(self, kind: str = "decoupled", num_envs: int = 1, device="cuda", seed: int = 0,
                 substeps: int = 1, layout: str = "mixed", use_UDM: bool = True,
                 UDM_percentage: float = 10.0, auto_reset: bool = False, max_episode_steps: int = 0,
                 env_offset: int = 0, want_raw_reward: bool = False, obs_rows: Optional[bool] = None,
                 field_stride: Optional[int] = None, goal_mode: Optional[int] = None, w_adapt: float = 16.0,
                 constants: Optional[QuadConstants] = None, final_obs: bool = False,
                 helper: Optional[bool] = None, autotune: Optional[bool] = None, reset_on_done: bool = False,
                 helper_rollout: Optional[bool] = None):
        if kind not in KINDS:
            raise ValueError(f"kind must be one of {KINDS}, got {kind!r}")
        if num_envs < 1:
            raise ValueError("num_envs must be >= 1")
        if substeps < 1:
            raise ValueError("substeps must be >= 1")
        if layout not in _lib.LAYOUT_ID:
            raise ValueError(f"layout must be one of {tuple(_lib.LAYOUT_ID)}, got {layout!r}")
        self._lib = _lib.load()  # raises if the HIP library is missing
        if not torch.cuda.is_available():
            raise RuntimeError("QuadVecEnv needs an AMD GPU (torch.cuda.is_available() is False); "
                               "gym_rotor_amd has no CPU execution path")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("device must be a cuda (ROCm) device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.kind, self.num_envs, self.substeps = kind, int(num_envs), int(substeps)
        self.framework = FRAMEWORK[kind]
        self.n_agents, self.action_dim, self.obs_dims = N_AGENTS[kind], ACTION_DIM[kind], OBS_DIMS[kind]
        self.layout = layout
        self.use_UDM, self.UDM_percentage = bool(use_UDM), float(UDM_percentage)
        self.auto_reset, self.max_episode_steps = bool(auto_reset), int(max_episode_steps)
        self.env_offset, self.seed = int(env_offset), self._check_seed(seed)
        self._helper, self._helper_rollout = helper, helper_rollout
        self._act_align = 16 if self.action_dim == 4 else 4   # (quadrotor_hip.h: A = 4 one 16-byte load per lane, A = 5 dword loads)
        self.reset_on_done = bool(reset_on_done)
        if auto_reset and self.env_offset % 64:
            import warnings
            warnings.warn(f"QuadVecEnv(auto_reset=True, env_offset={self.env_offset}): the in-launch reset stream is keyed by the global id of each "
                          "64-env tile's first env, so results are independent of the sharding only when every shard starts at a multiple of 64 "
                          "envs (shard_range() cuts there); this shard is still deterministic, but not bit-equal to the same envs inside another "
                          "partition", stacklevel=2)
        if self.reset_on_done and auto_reset:
            raise ValueError("reset_on_done is the caller's promise for auto_reset=False; with auto_reset=True the launch resets itself")
        c = self.constants = constants or QuadConstants(UDM_percentage=UDM_percentage)

        # ---- attributes the reference's callers read (trajectory_generator.py:44-46,
        #      policy_regularization.py:31-41, draw_plot.py:37,51-71, main.py:68-73) ----
        self.dt, self.freq, self.g = c.dt, c.freq, c.g
        self.x_lim, self.v_lim, self.W_lim, self.euler_lim = c.x_lim, c.v_lim, c.W_lim, c.euler_lim
        self.eIx_lim, self.eIb1_lim, self.sat_sigma = c.eIx_lim, c.eIb1_lim, c.sat_sigma
        self.alpha, self.beta = c.alpha, c.beta
        self.m_nominal, self.d_nominal, self.J_nominal = c.m_nominal, c.d_nominal, c.J_nominal
        self.c_tf_nominal, self.c_tw_nominal = c.c_tf_nominal, c.c_tw_nominal
        self.hover_force, self.min_force, self.max_force = c.hover_force, c.min_force, c.max_force
        self.avrg_act, self.scale_act = c.avrg_act, c.scale_act
        self.forces_to_fM = c.forces_to_fM
        self.fM_to_forces = np.linalg.inv(self.forces_to_fM)
        self.reward_min, self.reward_min_1, self.reward_min_2 = c.reward_min, c.reward_min_1, c.reward_min_2
        self.reward_crash = c.reward_crash
        self.e1, self.e2, self.e3 = np.eye(3)
        low = np.concatenate([-c.x_lim * np.ones(3), -c.v_lim * np.ones(3), -np.ones(9), -c.W_lim * np.ones(3)])
        self.single_observation_space = Box(low.astype(np.float32), (-low).astype(np.float32), dtype=np.float32)
        self.single_action_space = Box(-1.0, 1.0, shape=(self.action_dim,), dtype=np.float32)
        self.observation_space, self.action_space = self.single_observation_space, self.single_action_space

        # ---- device buffers (SoA [field][ld], ld >= N: see field_stride in quadrotor_hip.h) ----
        N, dev = self.num_envs, self.device
        self._ld = _field_stride(N) if field_stride is None else int(field_stride)
        if self._ld < N or self._ld % 4:
            raise ValueError("field_stride must be a multiple of 4 and >= num_envs")
        xv_dt = torch.float64 if layout == "f64" else torch.float32
        qw_dt = torch.float32 if layout == "f32" else torch.float64
        self._pos_vel = self._soa(6, xv_dt)     # x(3), v(3)
        self._att_rate = self._soa(6, qw_dt)    # the 3 smaller components of q (index of the dropped one in k0's low bits), W(3):
        #                                         all zeros = identity attitude at rest
        self._integ = None if kind == "quad" else self._soa(8, torch.float32)
        self._params = None
        if self.use_UDM:
            self._params = self._soa(6, torch.float32)
            self._params.copy_(torch.tensor(c.nominal_params, dtype=torch.float32, device=dev)[:, None].expand(6, N))
        self._goal = None  # default hover goal until set_goal_state is called (quad.py:98-101)
        if goal_mode not in _lib.GOAL_ID:
            raise ValueError("goal_mode must be None or 0..6 (TrajectoryGenerator modes fused into the step)")
        self.goal_mode = goal_mode
        self._traj = None if goal_mode is None else self._soa(8, torch.float32)
        if goal_mode in (2, 3, 4, 5):   # the stateful modes keep the generator's xd, vd, b1d, Wd in the goal buffer
            self._goal = self._soa(12, torch.float32)
            self._goal[6] = 1.0
        self._episode = torch.zeros(N, dtype=torch.int32, device=dev)
        # stream position of the in-launch reset, one counter per 64-env tile (quadrotor_hip.h: reset_count)
        self._reset_count = torch.zeros((N + 63) // 64, dtype=torch.int32, device=dev)
        self._steps = torch.zeros(N, dtype=torch.int32, device=dev) if self.max_episode_steps > 0 else None
        # caller-facing rows
        self.obs_rows = (kind != "quad") if obs_rows is None else bool(obs_rows)
        if kind != "quad" and not self.obs_rows:
            raise ValueError("obs_rows=False is only meaningful for kind='quad'")
        self._obs0 = torch.empty(N, self.obs_dims[0], dtype=torch.float32, device=dev) if self.obs_rows else None
        self._obs1 = torch.empty(N, self.obs_dims[1], dtype=torch.float32, device=dev) if len(self.obs_dims) > 1 else None
        self._reward = torch.empty(N, self.n_agents, dtype=torch.float32, device=dev)
        self._reward_raw = torch.empty(N, self.n_agents, dtype=torch.float32, device=dev) if want_raw_reward else None
        self._done = torch.zeros(N, self.n_agents, dtype=torch.bool, device=dev)
        self._final0 = self._final1 = None
        if final_obs:
            if not self.auto_reset:
                raise ValueError("final_obs=True needs auto_reset=True (without it step() already returns the terminal observation)")
            self._final0 = torch.zeros(N, self.obs_dims[0], dtype=torch.float32, device=dev)
            self._final1 = torch.zeros(N, self.obs_dims[1], dtype=torch.float32, device=dev) if len(self.obs_dims) > 1 else None
        self._rejected = torch.zeros(1, dtype=torch.int32, device=dev)  # qr_set_state: rows without a nearest rotation
        self._last_obs = None    # observation rows the next policy action is computed from (rollout_actor)
        self._policy_steps = 0   # global step index of the in-kernel action-noise stream
        self._trunc = torch.zeros(N, dtype=torch.bool, device=dev)

        # ---- C structs (pointers refreshed lazily) ----
        self._cenv = _lib.QrEnv()
        self._cout = _lib.QrStepOut()
        co = _lib.default_coeffs()
        for name in ("Cx", "CIx", "Cv", "Cb1", "CIb1", "Cw12", "CW3", "alpha", "beta", "x_lim", "v_lim", "W_lim",
                     "eIx_lim", "eIb1_lim"):
            setattr(co, name, float(getattr(c, name)))
        co.CW, co.dt, co.euler_lim_deg, co.udm_fraction = c.CW, c.dt, c.euler_lim, self.UDM_percentage / 100.0
        co.w_adapt = float(w_adapt)
        for name in ("eight_T", "eight_A1", "eight_A2", "eight_w_b1d", "eight_alt_d", "eight_eps", "eight_count",
                     "m_nominal", "d_nominal", "J1_nominal", "J3_nominal", "c_tf_nominal", "c_tw_nominal", "g", "min_force"):
            setattr(co, name, float(getattr(c, name)))
        self._cenv.coeffs = co
        self._sync_structs()
        self._closed = False
        self.autotune_report = None
        if autotune is None and os.environ.get("QR_AUTOTUNE", "1").lower() in ("0", "off", "false"):
            autotune = False
        if autotune:
            self.autotune_launch()
        elif autotune is None and helper is None:
            self._autotune_from_cache(time_if_absent=os.environ.get("QR_AUTOTUNE", "").lower() in ("time", "timed"))

    # ------------------------------------------------------------------------------
    @staticmethod
    def _check_seed(seed) -> int:
        """Seeds are Philox keys of 64 bits; the torch custom ops carry them as int64, so both paths accept 0 <= seed < 2^63."""
        seed = int(seed)
        if not 0 <= seed < 2 ** 63:
            raise ValueError("seed must satisfy 0 <= seed < 2**63")
        return seed

    def _soa(self, fields: int, dtype) -> torch.Tensor:
        """[fields, N] view of a zeroed [fields, ld] buffer (the view starts at the buffer's base)."""
        return torch.zeros(fields, self._ld, dtype=dtype, device=self.device)[:, :self.num_envs]

    def _sync_structs(self):
        self._epoch = getattr(self, "_epoch", 0) + 1   # launches captured before this point hold stale pointers / flags (CapturedStep)
        e, o = self._cenv, self._cout
        e.kind, e.layout = _lib.KIND_ID[self.kind], _lib.LAYOUT_ID[self.layout]
        e.num_envs, e.field_stride = self.num_envs, self._ld
        e.env_offset, e.seed = self.env_offset, self.seed
        e.pos_vel, e.att_rate = _ptr(self._pos_vel), _ptr(self._att_rate)
        e.integ, e.params, e.goal = _ptr(self._integ), _ptr(self._params), _ptr(self._goal)
        e.traj = _ptr(self._traj)
        e.goal_mode = _lib.GOAL_ID[self.goal_mode]
        e.episode, e.steps, e.reset_count = _ptr(self._episode), _ptr(self._steps), _ptr(self._reset_count)
        e.max_episode_steps = self.max_episode_steps
        e.flags = (_lib.FLAG_AUTO_RESET if self.auto_reset else 0) | (0 if self.use_UDM else _lib.FLAG_NO_UDM)
        e.flags |= {None: 0, True: _lib.FLAG_FORCE_HELPER, False: _lib.FLAG_NO_HELPER}[self._helper]
        e.flags |= {None: 0, True: _lib.FLAG_FORCE_HELPER_ROLLOUT, False: _lib.FLAG_NO_HELPER_ROLLOUT}[self._helper_rollout]
        e.flags |= _lib.FLAG_CALLER_RESETS if self.reset_on_done else 0
        o.obs0, o.obs1, o.reward, o.reward_raw = _ptr(self._obs0), _ptr(self._obs1), _ptr(self._reward), _ptr(self._reward_raw)
        o.done, o.truncated = _ptr(self._done), (_ptr(self._trunc) if self._steps is not None else None)
        o.final_obs0, o.final_obs1 = _ptr(self._final0), _ptr(self._final1)
        # the same QrEnv scalars / QrCoeffs as plain Python values, for the torch custom ops (torch_ops.env_args):
        # [kind, layout, flags, seed, env_offset, goal_mode, max_episode_steps]
        self._op_cfg = [int(e.kind), int(e.layout), int(e.flags), int(self.seed), int(self.env_offset),
                        int(e.goal_mode), int(self.max_episode_steps)]
        self._op_coeffs = [float(getattr(e.coeffs, n)) for n, _ in _lib.QrCoeffs._fields_]

    def _stream(self):
        """Raw handle of torch's current stream on the env's device (the launch stream)."""
        return _raw_stream(self.device.index)

    def _on_device(self):
        """Context for a launch: kernels run on the env's device; if another device is current it is made current
        for the call only and restored afterwards (the process-wide current device is not left changed)."""
        return _NULL_CTX if _get_device() == self.device.index else torch.cuda.device(self.device)

    def _check_actions(self, actions: torch.Tensor, lead=()):
        if not isinstance(actions, torch.Tensor):
            raise TypeError("actions must be a torch.Tensor on the env's device")
        if actions.dtype != torch.float32:
            raise TypeError(f"actions must be float32, got {actions.dtype}")
        if actions.device != self.device:
            raise ValueError(f"actions on {actions.device}, env on {self.device}")
        want = tuple(lead) + (self.num_envs, self.action_dim)
        if actions.shape != want:
            raise ValueError(f"actions shape {tuple(actions.shape)} != {want}")
        if not actions.is_contiguous():
            actions = actions.contiguous()
        if actions.data_ptr() % self._act_align:  # e.g. a row slice of a bigger tensor: A = 4 rows are read with 16-byte loads
            actions = actions.clone()
        return actions

    def _check_out(self, out: dict, lead: tuple, policy: bool = False):
        """Caller-owned output tensors are written by raw pointer: validate shape, dtype, device and
        contiguity (a mismatch would silently corrupt memory)."""
        N, dev = self.num_envs, self.device
        want = {"reward": (lead + (N, self.n_agents), torch.float32), "terminated": (lead + (N, self.n_agents), torch.bool)}
        if self.obs_rows or self.kind != "quad":
            want["obs0"] = (lead + (N, self.obs_dims[0]), torch.float32)
        if len(self.obs_dims) > 1:
            want["obs1"] = (lead + (N, self.obs_dims[1]), torch.float32)
        if policy:
            want["action"] = (lead + (N, self.action_dim), torch.float32)
        optional = {"reward_raw": (lead + (N, self.n_agents), torch.float32), "truncated": (lead + (N,), torch.bool),
                    "logprob": (lead + (N, self.action_dim), torch.float32),
                    "final_obs0": (lead + (N, self.obs_dims[0]), torch.float32)}
        if len(self.obs_dims) > 1:
            optional["final_obs1"] = (lead + (N, self.obs_dims[1]), torch.float32)
        if out.get("final_obs0") is not None:
            if not self.auto_reset:
                raise ValueError("out['final_obs0'] needs auto_reset=True")
            if len(self.obs_dims) > 1 and out.get("final_obs1") is None:
                raise ValueError("out['final_obs1'] is required with out['final_obs0'] for kind 'decoupled'")
        for k, (shape, dtype) in list(want.items()) + [(k, v) for k, v in optional.items() if out.get(k) is not None]:
            t = out.get(k)
            if t is None:
                if k == "obs0" and self.kind == "quad":
                    continue
                raise ValueError(f"out[{k!r}] is required")
            if tuple(t.shape) != shape or t.dtype != dtype or t.device != dev or not t.is_contiguous():
                raise ValueError(f"out[{k!r}] must be a contiguous {dtype} tensor of shape {shape} on {dev}, got "
                                 f"{tuple(t.shape)} {t.dtype} on {t.device}")
        if self._steps is not None and out.get("truncated") is None:
            raise ValueError("out['truncated'] is required when max_episode_steps is set")

    def _obs(self):
        if self._obs0 is None:  # kind='quad' without observation rows: fetch with get_current_state()
            return None
        return self._obs0 if self._obs1 is None else (self._obs0, self._obs1)

    # ------------------------------------------------------------------------------
    def step(self, actions: torch.Tensor, out: Optional[dict] = None):
        """QuadEnv.step (quad.py:142-168) for all envs.  Returned tensors are the env's
        output buffers: valid until the next step()/rollout() call.  `out` (see
        RolloutStorage.slot) redirects the observation / reward / done rows of this step into
        caller-owned contiguous tensors, e.g. slices of a [T, N, ...] rollout buffer."""
        a = self._check_actions(actions)
        if out is None:
            with self._on_device():
                rc = self._lib.qr_step(C.byref(self._cenv), a.data_ptr(), self.substeps, C.byref(self._cout), self._stream())
            _lib.check(rc, "qr_step")
            self._last_obs = self._obs()
            return self._last_obs, self._reward, self._done, self._trunc, {}
        self._check_out(out, ())
        o = _lib.QrStepOut()
        o.obs0, o.obs1, o.reward = _ptr(out.get("obs0")), _ptr(out.get("obs1")), _ptr(out["reward"])
        o.reward_raw, o.done = _ptr(out.get("reward_raw")), _ptr(out["terminated"])
        o.truncated = _ptr(out.get("truncated")) if self._steps is not None else None
        o.final_obs0, o.final_obs1 = _ptr(out.get("final_obs0")), _ptr(out.get("final_obs1"))
        with self._on_device():
            rc = self._lib.qr_step(C.byref(self._cenv), a.data_ptr(), self.substeps, C.byref(o), self._stream())
        _lib.check(rc, "qr_step")
        obs = out.get("obs0") if "obs1" not in out else (out["obs0"], out["obs1"])
        self._last_obs = obs
        return obs, out["reward"], out["terminated"], out.get("truncated", self._trunc), {}

    def rollout(self, actions: torch.Tensor, out: Optional[dict] = None):
        """T env-steps in one launch (state stays in registers).  actions [T,N,A].
        Returns dict(obs, reward[T,N,n_agents], terminated, truncated[T,N])."""
        if actions.dim() != 3:
            raise ValueError("rollout actions must be [T, N, A]")
        T = actions.shape[0]
        a = self._check_actions(actions, lead=(T,))
        N, dev = self.num_envs, self.device
        if out is None:
            out = {"obs0": torch.empty(T, N, self.obs_dims[0], dtype=torch.float32, device=dev) if self.obs_rows else None,
                   "reward": torch.empty(T, N, self.n_agents, dtype=torch.float32, device=dev),
                   "terminated": torch.zeros(T, N, self.n_agents, dtype=torch.bool, device=dev),
                   "truncated": torch.zeros(T, N, dtype=torch.bool, device=dev)}
            if self._obs1 is not None:
                out["obs1"] = torch.empty(T, N, self.obs_dims[1], dtype=torch.float32, device=dev)
        else:
            self._check_out(out, (T,))
        o = _lib.QrStepOut()
        o.obs0, o.obs1, o.reward = _ptr(out.get("obs0")), _ptr(out.get("obs1")), _ptr(out["reward"])
        o.reward_raw, o.done = _ptr(out.get("reward_raw")), _ptr(out["terminated"])
        o.truncated = _ptr(out["truncated"]) if self._steps is not None else None
        o.final_obs0, o.final_obs1 = _ptr(out.get("final_obs0")), _ptr(out.get("final_obs1"))
        with self._on_device():
            rc = self._lib.qr_rollout(C.byref(self._cenv), a.data_ptr(), T, self.substeps, C.byref(o), self._stream())
        _lib.check(rc, "qr_rollout")
        out["obs"] = out["obs0"] if self._obs1 is None else (out["obs0"], out["obs1"])
        if out["obs0"] is not None:
            self._last_obs = out["obs0"][T - 1] if self._obs1 is None else (out["obs0"][T - 1], out["obs1"][T - 1])
        return out

    def rollout_actor(self, actors, n_steps: int, obs=None, noise: Optional[torch.Tensor] = None,
                      deterministic: bool = False, max_action: float = 1.0, noise_seed: Optional[int] = None,
                      out: Optional[dict] = None):
        """`n_steps` env-steps in ONE launch with the PPO actor(s) inside the loop — the collection
        loop of main.py:141-166 with PPO.choose_action (ppo.py:82-101): per step every env's
        actor is evaluated on its current observation, the action sampled (mean + std eps),
        clamped and stepped.  actors: one `policy.ActorParams` per agent.  obs: the
        observation(s) the first action is computed from (default: what the last step /
        get_norm_error_state / rollout_actor returned).  noise: optional [T,N,A] standard normals
        (default: drawn in the kernel, stream (noise_seed, global env id, global step)).
        Returns dict(obs0[, obs1], action[T,N,A], logprob[T,N,A], reward, terminated, truncated);
        obs0[t] is the observation after step t.  out: the same dict with preallocated tensors
        (e.g. views of a RolloutStorage)."""
        from . import policy as _policy
        if self.kind == "quad":
            raise ValueError("rollout_actor needs kind 'coupled' or 'decoupled' (the reference trains on the wrappers)")
        if not self.obs_rows:
            raise ValueError("rollout_actor needs obs_rows=True")
        actors = list(actors)
        dims = _policy.ACTOR_DIMS[self.kind]
        if len(actors) != len(dims):
            raise ValueError(f"kind {self.kind!r} needs {len(dims)} actor(s)")
        for a_, d in zip(actors, dims):
            a_.check(d, self.device)
        T, N, A, dev = int(n_steps), self.num_envs, self.action_dim, self.device
        if T < 1:
            raise ValueError("n_steps must be >= 1")
        if obs is None:
            obs = self._last_obs
            if obs is None:
                raise ValueError("no current observation: call get_norm_error_state() / step() first or pass obs=")
        obs = [obs] if isinstance(obs, torch.Tensor) else list(obs)
        for o_, d in zip(obs, self.obs_dims):
            if tuple(o_.shape) != (N, d) or o_.dtype != torch.float32 or o_.device != dev or not o_.is_contiguous():
                raise ValueError(f"obs must be contiguous float32 [{N}, {d}] on {dev}")
        if noise is not None:
            if tuple(noise.shape) != (T, N, A) or noise.dtype != torch.float32 or noise.device != dev or not noise.is_contiguous():
                raise ValueError(f"noise must be contiguous float32 [{T}, {N}, {A}] on {dev}")
        if out is None:
            out = {"obs0": torch.empty(T, N, self.obs_dims[0], dtype=torch.float32, device=dev),
                   "action": torch.empty(T, N, A, dtype=torch.float32, device=dev),
                   "logprob": torch.empty(T, N, A, dtype=torch.float32, device=dev),
                   "reward": torch.empty(T, N, self.n_agents, dtype=torch.float32, device=dev),
                   "terminated": torch.zeros(T, N, self.n_agents, dtype=torch.bool, device=dev),
                   "truncated": torch.zeros(T, N, dtype=torch.bool, device=dev)}
            if len(self.obs_dims) > 1:
                out["obs1"] = torch.empty(T, N, self.obs_dims[1], dtype=torch.float32, device=dev)
        else:
            self._check_out(out, (T,), policy=True)
        arr = _policy.c_actor_array(actors)
        pol = _lib.QrPolicyRollout()
        pol.actors = arr
        pol.obs0_in, pol.obs1_in = obs[0].data_ptr(), (obs[1].data_ptr() if len(obs) > 1 else None)
        pol.noise = _ptr(noise)
        pol.noise_seed = (self.seed if noise_seed is None else int(noise_seed)) & (2 ** 64 - 1)
        pol.step_base = self._policy_steps
        pol.max_action, pol.deterministic = float(max_action), int(bool(deterministic))
        pol.action_out, pol.logprob_out = out["action"].data_ptr(), _ptr(out.get("logprob"))
        o = _lib.QrStepOut()
        o.obs0, o.obs1, o.reward = _ptr(out.get("obs0")), _ptr(out.get("obs1")), _ptr(out["reward"])
        o.reward_raw, o.done = _ptr(out.get("reward_raw")), _ptr(out["terminated"])
        o.truncated = _ptr(out["truncated"]) if self._steps is not None else None
        o.final_obs0, o.final_obs1 = _ptr(out.get("final_obs0")), _ptr(out.get("final_obs1"))
        with self._on_device():
            rc = self._lib.qr_rollout_actor(C.byref(self._cenv), C.byref(pol), T, self.substeps, C.byref(o), self._stream())
        _lib.check(rc, "qr_rollout_actor")
        self._policy_steps += T
        self._last_obs = out["obs0"][T - 1] if len(self.obs_dims) == 1 else (out["obs0"][T - 1], out["obs1"][T - 1])
        out["obs"] = out["obs0"] if len(self.obs_dims) == 1 else (out["obs0"], out["obs1"])
        return out

    def capture(self, actions: Optional[torch.Tensor] = None, n_steps: int = 1, body=None, warmup: int = 1) -> "CapturedStep":
        """step() inside ONE replayable hipGraph, captured once on a side stream and replayed with a single host call.

        Why: an eager step() costs ~5.6-6.0 us of host time per call (4.65 of it the HIP launch path), more than the kernel takes
        at 65 536 envs (4.2 us); a graph replay costs ~10 us of host time whatever it holds (measured on MI355X,
        profiles/r05/eager_cost.json).  So a replay pays as soon as it carries more than one launch:

            # (a) the caller's per-step work and the env step in one replay — e.g. a torch policy (~10 launches) + step():
            step = env.capture(body=lambda: env.step(policy(env_obs)))      # body runs once, under capture; its return value is kept
            ... = step()                                                     # ONE host call per env-step
            # (b) n_steps env-steps per replay, one action slab each (4.2 us per step at 65 536 envs from n_steps ~ 8 on):
            steps = env.capture(n_steps=64); steps.actions.copy_(slabs); obs, reward, terminated, truncated, info = steps()
            # (c) a single step per replay (n_steps=1, no body): correct, but SLOWER than eager step() — use it only to splice the
            #     step into a caller-side graph-driven loop

        `warmup` (with `body`): libraries the body calls (hipBLASLt behind torch.nn.Linear, ...) initialise on first use and must not
        do that under capture, so body() is first run `warmup` times eagerly on the side stream — with the ENV's state, counters and
        RNG position saved before and restored after (like autotune_launch; what the body does to the caller's own tensors is the
        caller's business).  warmup=0 skips it (the caller has already run the same shapes).
        `actions`: the static buffer every replay reads — a contiguous float32 [N, A] tensor on the env's device (A = 4: 16-byte
        aligned), or [n_steps, N, A] (each launch its own slab); default: allocated here (`.actions` of the result).  With `body`
        no action buffer is involved: body() must call env.step(...) itself, on tensors that stay alive and in place.
        The returned tensors are the env's own output buffers (as with step()): with n_steps > 1 they hold the LAST step's rows.
        Replays are bit-identical to eager calls: in-launch resets draw from per-tile counters in device memory, which advance
        under replay exactly as under eager launches (quadrotor_hip.h: reset_count).  capture() itself executes nothing: the
        env's state is untouched until the first replay.  A later call that changes what the launches baked in — the first
        set_goal_state() / set_state(params=) (they allocate a buffer), reset(seed=), set_launch(), load_state_dict() — makes the
        capture stale: replaying it raises, capture again."""
        N, A, dev = self.num_envs, self.action_dim, self.device
        n_steps = int(n_steps)
        if n_steps < 1:
            raise ValueError("n_steps must be >= 1")
        if body is not None:
            if actions is not None or n_steps != 1:
                raise ValueError("capture(body=...) takes neither actions nor n_steps: the body calls env.step() itself")
        else:
            if actions is None:
                actions = torch.zeros((N, A) if n_steps == 1 else (n_steps, N, A), dtype=torch.float32, device=dev)
            if actions.dim() == 2:
                checked = self._check_actions(actions)
            elif actions.dim() == 3 and actions.shape[0] == n_steps:
                checked = self._check_actions(actions, lead=(n_steps,))
                if (N * A * 4) % self._act_align:
                    raise ValueError(f"per-step action slabs need N * A * 4 bytes to be a multiple of {self._act_align}")
            else:
                raise ValueError(f"actions must be [{N}, {A}] or [{n_steps}, {N}, {A}]")
            if checked.data_ptr() != actions.data_ptr():
                raise ValueError(f"the static action buffer must be contiguous and {self._act_align}-byte aligned (it is read in place by every replay)")
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        saved = last = None
        if body is not None and warmup > 0:
            # the snapshot is taken on the CURRENT stream, BEFORE the side stream is made to wait for it: the warm-up body below then
            # runs strictly after these clones (taken after the wait, the clones and the body's writes could overlap)
            saved, last = self.state_dict(), self._last_obs
        policy_steps = self._policy_steps
        side.wait_stream(cur)
        if saved is not None:
            outs = [t for t in (self._obs0, self._obs1, self._reward, self._reward_raw, self._done, self._trunc, self._final0, self._final1)
                    if t is not None]                       # the env's output buffers: a body typically reads its action from them
            with torch.cuda.stream(side):
                kept = [t.clone() for t in outs]
                for _ in range(int(warmup)):
                    body()
                for t, k in zip(outs, kept):
                    t.copy_(k)
            side.synchronize()
            self.load_state_dict(saved)
            self._last_obs = last
        graph = torch.cuda.CUDAGraph()
        epoch = self._epoch
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                if body is not None:
                    ret = body()
                else:
                    for k in range(n_steps):
                        ret = self.step(actions if actions.dim() == 2 else actions[k])
        cur.wait_stream(side)
        if self._epoch != epoch:
            raise RuntimeError("capture(body=...): the body changed the env's buffers / launch configuration (set_goal_state, reset(seed=), ...): "
                               "do that before capturing")
        if self._policy_steps != policy_steps:
            # rollout_actor() bakes the position of its exploration-noise stream (step_base) into the launch: every replay would draw
            # the SAME noise again
            self._policy_steps = policy_steps
            raise RuntimeError("capture(body=...): the body called rollout_actor(), whose in-kernel action-noise stream position is a launch "
                               "argument — a replay would repeat the same noise.  Capture step() (with the policy in torch), or call "
                               "rollout_actor() eagerly / pass noise= from a tensor the caller refreshes")
        return CapturedStep(self, graph, actions, ret, n_steps)

    def reset(self, env_type: str = "train", seed: Optional[int] = None, options: Optional[dict] = None,
              mask: Optional[torch.Tensor] = None):
        """QuadEnv.reset (quad.py:171-222): new parameters (train + use_UDM), initial error
        state, zero integrators, for envs where mask is True (default all).  Returns the
        float32 state [N,18] like the reference; callers then set the goal and call
        get_norm_error_state() for the first observation (main.py:126-129)."""
        if env_type not in ("train", "eval"):
            raise ValueError("env_type must be 'train' or 'eval'")
        if seed is not None:
            self.seed = self._check_seed(seed)
            self._cenv.seed = self.seed
            self._op_cfg[3] = self.seed
            self._epoch += 1
        m = None
        if mask is not None:
            if mask.shape != (self.num_envs,) or mask.device != self.device:
                raise ValueError("mask must be a [num_envs] tensor on the env's device")
            m = mask.to(torch.uint8) if mask.dtype != torch.uint8 else mask
            m = m.contiguous()
        flags = self._cenv.flags
        self._cenv.flags = (flags & ~(_lib.FLAG_EVAL_RESET | _lib.FLAG_AUTO_RESET)) | (_lib.FLAG_EVAL_RESET if env_type == "eval" else 0)
        try:
            with self._on_device():
                rc = self._lib.qr_reset(C.byref(self._cenv), _ptr(m), self._stream())
        finally:
            self._cenv.flags = flags
        _lib.check(rc, "qr_reset")
        self._last_obs = None  # the rows of the previous episode are not this episode's observation
        if self.goal_mode is not None:  # main.py:226-227: reset, then mark_traj_start(state)
            with self._on_device():
                _lib.check(self._lib.qr_traj_start(C.byref(self._cenv), _ptr(m), None, self._stream()), "qr_traj_start")
        return self.get_current_state().to(torch.float32)

    def get_norm_error_state(self, framework: Optional[str] = None):
        """quad.py:421-466.  Advances the integral terms (same side effect as the reference).  `framework` selects the FORMAT as
        in the reference — "MONO": [obs[N,23]], "MODUL": [obs1[N,15], obs2[N,3]] — on either wrapper kind (the method is
        QuadEnv's, both wrappers inherit it unchanged); default: the env's own.  The env's own format is written to its output
        buffers (what step() returns next is unaffected either way); the other one to fresh tensors.  kind='quad' raises
        AttributeError like a bare QuadEnv does (it has no alpha / beta / eIx_lim: quad.py:448)."""
        if self.kind == "quad":
            raise AttributeError("'QuadEnv' object has no attribute 'alpha' — get_norm_error_state needs the integral terms of a "
                                 "wrapper env (kind 'coupled' / 'decoupled'), exactly as in the reference (quad.py:448)")
        fw = self.framework if framework is None else framework
        if fw not in ("MONO", "MODUL"):
            raise ValueError(f"framework must be 'MONO' or 'MODUL', got {framework!r}")
        if fw == self.framework:
            with self._on_device():
                rc = self._lib.qr_error_obs(C.byref(self._cenv), _ptr(self._obs0), _ptr(self._obs1), self._stream())
            _lib.check(rc, "qr_error_obs")
            self._last_obs = self._obs()
            return [self._obs0] if self._obs1 is None else [self._obs0, self._obs1]
        dims = OBS_DIMS["coupled" if fw == "MONO" else "decoupled"]
        rows = [torch.empty(self.num_envs, d, dtype=torch.float32, device=self.device) for d in dims]
        with self._on_device():
            rc = self._lib.qr_error_obs_format(C.byref(self._cenv), _lib.KIND_ID["coupled" if fw == "MONO" else "decoupled"],
                                               rows[0].data_ptr(), rows[1].data_ptr() if len(rows) > 1 else None, self._stream())
        _lib.check(rc, "qr_error_obs_format")
        self._last_obs = None   # (rows of the other format are not what this env's actor consumes)
        return rows

    # ------------------------------------------------------------------------------
    def _rows3(self, v, name):
        t = torch.as_tensor(v, dtype=torch.float32, device=self.device)
        if t.shape == (3,):
            t = t.expand(self.num_envs, 3)
        if tuple(t.shape) != (self.num_envs, 3):
            raise ValueError(f"{name} must have shape (3,) or ({self.num_envs}, 3)")
        return t.t()

    def set_goal_state(self, xd, vd, b1d, b1d_dot=None, Wd=None):
        """quad.py:413-418.  b1d_dot is accepted and ignored (unused by the step path)."""
        if self.goal_mode is not None:
            raise RuntimeError("this env generates its goals on the device (goal_mode); set_goal_state() would be ignored by step()")
        if self._goal is None:
            self._goal = self._soa(12, torch.float32)
            self._cenv.goal = self._goal.data_ptr()
            self._epoch += 1
        self._goal[0:3] = self._rows3(xd, "xd")
        self._goal[3:6] = self._rows3(vd, "vd")
        self._goal[6:9] = self._rows3(b1d, "b1d")
        self._goal[9:12] = self._rows3(np.zeros(3) if Wd is None else Wd, "Wd")
        self._last_obs = None  # observations are errors w.r.t. the goal: ask get_norm_error_state() again

    # ---- goal generation (utils/trajectory_generator.py modes 0 / 1) ----------------
    def mark_traj_start(self, mask: Optional[torch.Tensor] = None, theta_b1d=None, t_traj=None, w_b1d=None):
        """TrajectoryGenerator.mark_traj_start(state) + the episode-start draws of mode 0
        (theta_b1d ~ U(+-25 deg)) / mode 1 (t_traj ~ U(2,5) s, w_b1d ~ U(+-0.15 pi) rad/s), from the
        current state.  Pass the draws ([N] tensors) to inject them; default: the env's RNG."""
        if self.goal_mode is None:
            raise RuntimeError("mark_traj_start needs a goal_mode (0..6)")
        m = None if mask is None else torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        draws = None
        if theta_b1d is not None or t_traj is not None or w_b1d is not None:
            z = torch.zeros(self.num_envs, dtype=torch.float32, device=self.device)
            one = torch.ones(self.num_envs, dtype=torch.float32, device=self.device)
            cols = [z if theta_b1d is None else torch.as_tensor(theta_b1d, dtype=torch.float32, device=self.device).expand(self.num_envs),
                    3.0 * one if t_traj is None else torch.as_tensor(t_traj, dtype=torch.float32, device=self.device).expand(self.num_envs),
                    z if w_b1d is None else torch.as_tensor(w_b1d, dtype=torch.float32, device=self.device).expand(self.num_envs)]
            draws = torch.stack(cols).contiguous()
        with self._on_device():
            _lib.check(self._lib.qr_traj_start(C.byref(self._cenv), _ptr(m), _ptr(draws), self._stream()), "qr_traj_start")

    def get_desired(self, store_goal: bool = False, mask: Optional[torch.Tensor] = None):
        """TrajectoryGenerator.get_desired(state, mode): (xd, vd, b1d, b1d_dot, Wd) as [N,3] views
        for the current state (rows of envs outside `mask` are left zero); advances the generator's
        clock by dt like every reference call.  store_goal=True also does set_goal_state()."""
        if self.goal_mode is None:
            raise RuntimeError("get_desired needs a goal_mode (0..6)")
        rows = torch.zeros(self.num_envs, 15, dtype=torch.float32, device=self.device)
        m = None if mask is None else torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        if store_goal and self._goal is None:
            self._goal = self._soa(12, torch.float32)
            self._cenv.goal = self._goal.data_ptr()
            self._epoch += 1
        with self._on_device():
            _lib.check(self._lib.qr_get_desired(C.byref(self._cenv), _ptr(m), rows.data_ptr(), int(store_goal), self._stream()), "qr_get_desired")
        return rows[:, 0:3], rows[:, 3:6], rows[:, 6:9], rows[:, 9:12], rows[:, 12:15]

    def get_current_state(self) -> torch.Tensor:
        """quad.py:409-410: float64 [N,18] = (x, v, vec_F(R), W), rebuilt from the 12-word
        internal state (x, v, smallest-three quaternion, W; R = R(q)) by a small kernel; a fresh tensor each call."""
        rows = torch.empty(self.num_envs, 18, dtype=torch.float64, device=self.device)
        with self._on_device():
            _lib.check(self._lib.qr_get_state(C.byref(self._cenv), rows.data_ptr(), self._stream()), "qr_get_state")
        return rows

    def set_state(self, state, integ=None, params=None, mask=None):
        """Inject states (and optionally integrator terms / parameters): [N,18] / [N,8] / [N,6]; with
        `mask` only the masked envs change (state, integ and params alike).  R goes through the
        reference's ensure_SO3 rule and is stored as a unit quaternion.  A row whose attitude block
        has no nearest rotation (det R <= 0, NaN/Inf) makes the WHOLE call fail: the rows are validated first
        (qr_check_state: the kernel's own test, nothing written) and ValueError is raised before any state,
        integrator term or parameter of any env has changed."""
        s = torch.as_tensor(state, device=self.device).to(torch.float64).contiguous()
        if tuple(s.shape) != (self.num_envs, 18):
            raise ValueError(f"state must be [{self.num_envs}, 18]")
        m = None if mask is None else torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        if m is not None and tuple(m.shape) != (self.num_envs,):
            raise ValueError("mask must be a [num_envs] tensor")
        sel = None if m is None else m.bool()[None, :]

        def rows_for(dst_fields, rows, what):   # shape checks BEFORE anything is written
            src = torch.as_tensor(rows, device=self.device).to(torch.float32)
            if tuple(src.shape) != (self.num_envs, dst_fields):
                raise ValueError(f"{what} must be [{self.num_envs}, {dst_fields}]")
            return src.t()

        integ_src = rows_for(8, integ, "integ") if (integ is not None and self._integ is not None) else None
        params_src = rows_for(6, params, "params") if params is not None else None
        self._rejected.zero_()
        with self._on_device():
            _lib.check(self._lib.qr_check_state(C.byref(self._cenv), s.data_ptr(), _ptr(m), self._rejected.data_ptr(), self._stream()), "qr_check_state")
        bad = int(self._rejected.item())  # off the hot path: one host sync per injection
        if bad:
            raise ValueError(f"set_state: {bad} row(s) rejected — the attitude block has det R <= 0 or non-finite entries "
                             "(no nearest rotation); nothing was applied: every env keeps its state, integrator terms and parameters")
        with self._on_device():
            _lib.check(self._lib.qr_set_state(C.byref(self._cenv), s.data_ptr(), _ptr(m), self._rejected.data_ptr(), self._stream()), "qr_set_state")
        self._last_obs = None

        def put(dst, src):
            dst.copy_(src if sel is None else torch.where(sel, src, dst))

        if integ_src is not None:
            put(self._integ, integ_src)
        if params_src is not None:
            if self._params is None:
                self._params = self._soa(6, torch.float32)
                self._params.copy_(torch.tensor(self.constants.nominal_params, dtype=torch.float32, device=self.device)[:, None].expand(6, self.num_envs))
                self._cenv.params = self._params.data_ptr()
                self._epoch += 1
            put(self._params, params_src)

    def final_observation(self):
        """Terminal observation rows of the envs re-sampled by the LAST step() (final_obs=True): rows of
        envs that did not reset hold whatever an earlier reset left there — select with the step's
        terminated/truncated flags."""
        if self._final0 is None:
            raise RuntimeError("construct the env with final_obs=True")
        return self._final0 if self._final1 is None else (self._final0, self._final1)

    def state_dict(self) -> dict:
        """Checkpoint of everything the env owns (18 + 8 words per env + params/goal/counters)."""
        keys = ("_pos_vel", "_att_rate", "_integ", "_params", "_goal", "_traj", "_episode", "_steps", "_reset_count")
        sd = {k[1:]: (None if getattr(self, k) is None else getattr(self, k).clone()) for k in keys}
        sd["policy_steps"] = int(self._policy_steps)   # position of the in-kernel action-noise stream (rollout_actor)
        sd["seed"] = int(self.seed)
        if self._last_obs is not None:  # the observation the next policy action is computed from
            last = (self._last_obs,) if isinstance(self._last_obs, torch.Tensor) else tuple(self._last_obs)
            sd["last_obs"] = tuple(o.clone() for o in last)
        return sd

    def load_state_dict(self, sd: dict):
        sd = dict(sd)
        self._policy_steps = int(sd.pop("policy_steps", 0))
        if "seed" in sd:
            # (checkpoints written before seeds were range-checked may hold a negative / >= 2^63 seed, which the env then masked
            # into the Philox key's 63 usable bits itself: do the same, so that such a checkpoint still loads and resumes its stream)
            self.seed = self._check_seed(int(sd.pop("seed")) & (2 ** 63 - 1))
            self._cenv.seed = self.seed
        last = sd.pop("last_obs", None)
        self._last_obs = None
        if last is not None:
            last = tuple(o.to(self.device).clone() for o in last)
            self._last_obs = last[0] if len(last) == 1 else last
        for k, v in sd.items():
            cur = getattr(self, "_" + k)
            if v is None:
                continue
            if k == "att_rate" and v.shape[0] == 7:  # a checkpoint from before the smallest-three attitude storage: q(4), W(3)
                v = _pack_att_rate(v.to(self.device))
            if cur is None:
                cur = self._soa(v.shape[0], v.dtype) if v.dim() == 2 else torch.zeros_like(v, device=self.device)
                setattr(self, "_" + k, cur)
            cur.copy_(v)
        self._sync_structs()

    @property
    def params(self):
        """Per-env (m, d, J1, J3, c_tf, c_tw) as [N,6] view, or None when nominal."""
        return None if self._params is None else self._params.t()

    @property
    def integ(self):
        return None if self._integ is None else self._integ.t()

    @property
    def episode_steps(self):
        return self._steps

    def set_launch(self, helper: Optional[bool] = None, helper_rollout: Optional[bool] = None):
        """Pin (True / False) or release (None) the launch rule's choice — a helper wavefront per tile — for this env's step()
        (`helper`) and, separately, for its rollout() / rollout_actor() (`helper_rollout`)."""
        self._helper, self._helper_rollout = helper, helper_rollout
        self._sync_structs()

    # ---- launch rule: cached / timed choice -------------------------------------------------------------------------
    def _launch_cache_key(self, action_source: str = "default") -> str:
        lib_id = f"abi{_lib.ABI_VERSION}:{os.path.getsize(_lib.LIB_PATH) if os.path.exists(_lib.LIB_PATH) else 0}"
        goal = "external" if self.goal_mode is None else f"mode{self.goal_mode}"
        variant = f"s{self.substeps}" + ("+rows" if (self.kind == "quad" and self.obs_rows) else "") + ("+final" if self._final0 is not None else "")
        return launch_cache.key(torch.cuda.get_device_name(self.device), lib_id, self.kind, (self.num_envs + 63) // 64, self.layout,
                                goal, action_source, variant)

    def _has_both_step_launches(self) -> bool:
        """Does qr_step have a helper-wave AND a plain instantiation for this env?  (quadrotor_kernels.hip: wants_helper /
        wants_helper_traj — in-launch resets, the default layout, external goals or the stateless generator modes.)"""
        return self.auto_reset and self.layout == "mixed" and self.goal_mode in (None, 0, 1, 6)

    def _effective_step_threshold(self) -> int:
        """The tile count up to which step() of THIS env takes the helper-wave launch under the compiled rule: the process'
        thresholds (qr_launch_thresholds) with the launcher's own reductions for >= 2 substeps and the fused goal generator
        (quadrotor_kernels.hip: wants_helper / wants_helper_traj)."""
        thr = _lib.launch_thresholds()
        if self.kind == "quad":
            return thr["step_quad"] if (self.substeps <= 1 and self.goal_mode is None) else min(thr["step_quad"], 2560)
        if self.substeps > 1:
            return min(thr["step_wrappers"], 1664)
        return thr["step_wrappers"] if self.goal_mode is None else min(thr["step_wrappers"], 2048)

    def _autotune_from_cache(self, time_if_absent: bool = False):
        """The default path of the constructor: nothing unless the grid is near the rule's threshold; then the choice RECORDED for
        this (device, library, kind, size, substeps, ...) by an earlier autotune_launch() — reading a small JSON file, no launch,
        no synchronisation.  Timing is opt-in (autotune=True, or QR_AUTOTUNE=time in the environment).  Never raises: a failure
        leaves the compiled rule in force."""
        if not self._has_both_step_launches():
            return
        if not launch_cache.near_threshold((self.num_envs + 63) // 64, self._effective_step_threshold()):
            return
        try:
            key = self._launch_cache_key()
            hit = launch_cache.lookup(key)
            if hit is not None:
                self.set_launch({"default": None, "helper": True, "no_helper": False}[hit["picked"]], self._helper_rollout)
                self.autotune_report = dict(hit.get("us", {}), picked=hit["picked"], source="cache")
                return
            if not time_if_absent:
                return
            report = self.autotune_launch()
            report["source"] = "timed"
        except Exception as ex:   # pragma: no cover - a tuning failure must not take the env down
            import warnings
            warnings.warn(f"gym_rotor_amd: launch autotuning skipped ({type(ex).__name__}: {ex}); the compiled launch rule stays in force")

    def autotune_launch(self, actions: Optional[torch.Tensor] = None, launches: int = 200, repeats: int = 3):
        """Time step() under the launch-rule choices that exist for this env — the library's default, helper wavefront forced,
        helper wavefront forbidden — and keep the fastest (set_launch).  The env's state, counters and RNG position are restored
        afterwards: tuning changes no later result (the env's own OUTPUT buffers — what the last step() returned — are overwritten
        by the timed launches: copy what you still need before tuning in mid-flight).  `actions`: the [N, A] rows (or a list of them, cycled through) the caller will
        step with; where they come from — cache or HBM — moves the wrappers' crossover, so pass the real source
        when there is one; default: 8 random slabs.  `launches` timed launches per candidate (one hipGraph), best of `repeats`.
        Returns {candidate: us per launch, "picked": name} (also kept as `autotune_report`)."""
        dev, N = self.device, self.num_envs
        saved, saved_choice = self.state_dict(), self._helper
        if actions is None:
            gen = torch.Generator(device=dev); gen.manual_seed(99)
            acts = [torch.rand(N, self.action_dim, device=dev, generator=gen) * 2 - 1 for _ in range(8)]
        else:
            acts = [self._check_actions(a_) for a_ in (actions if isinstance(actions, (list, tuple)) else [actions])]
        cands, seen, report = {"default": None, "helper": True, "no_helper": False}, {}, {}
        for name, h in cands.items():
            self.set_launch(h, self._helper_rollout)
            geom = self.launch_plan()["key"]   # the instantiation the launcher would run with this env's substeps
            if geom in seen:        # the same instantiation as an earlier candidate
                report[name] = report[seen[geom]]
                continue
            seen[geom] = name
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for k in range(10):
                    self.step(acts[k % len(acts)])
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for k in range(launches):
                        self.step(acts[k % len(acts)])
                best = float("inf")
                for _ in range(repeats):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    g.replay(); e0.record(); g.replay(); e1.record()
                    torch.cuda.synchronize(dev)
                    best = min(best, e0.elapsed_time(e1) * 1e3 / launches)
            torch.cuda.current_stream(dev).wait_stream(side)
            report[name] = best
            del g
        self.load_state_dict(saved)
        pick = min(("helper", "no_helper"), key=lambda k: report[k])
        if report["default"] <= report[pick] * 1.01:   # within noise of the rule's own choice: keep the rule
            self.set_launch(saved_choice, self._helper_rollout)
            report["picked"] = "default"
        else:
            self.set_launch(cands[pick], self._helper_rollout)
            report["picked"] = pick
        self.autotune_report = report
        launch_cache.store(self._launch_cache_key("default" if actions is None else "caller"), report)
        return report

    def launch_plan(self, n_steps: int = 1, actor: Optional[str] = None) -> dict:
        """What the launcher runs for step() (n_steps=1) / rollout(T) / rollout_actor(actors, T) with `actor` in ("ppo", "td3",
        "sac"): dict(name = the kernel instantiation as it appears in a rocprofv3 trace, grid, block, launches, key, and the
        template arguments traj, adapt, policy, single, help, hrew, mag) — the launcher's own decision function with this env's
        substeps (qr_launch_plan; host-side, nothing is launched)."""
        plan = _lib.QrLaunchPlan()
        form = {None: 0, "ppo": 1, "td3": 1, "sac": 2}[actor]
        _lib.check(self._lib.qr_launch_plan(C.byref(self._cenv), int(n_steps), self.substeps, form, C.byref(plan)), "qr_launch_plan")
        return {"name": plan.name.decode(), "grid": plan.grid, "block": plan.block, "launches": plan.launches, "key": int(plan.key),
                **{k: int(getattr(plan, k)) for k in ("traj", "adapt", "policy", "single", "help", "hrew", "mag")}}

    def kernel_info(self, n_steps=1, actor: Optional[str] = None):
        """(kernel family, workgroups, threads per workgroup) of the launch `step` (n_steps=1) / `rollout` / `rollout_actor` uses."""
        plan = self.launch_plan(n_steps, actor)
        return f"qr::step_kernel<{_lib.KIND_ID[self.kind]},...>", plan["grid"], plan["block"]

    def touch(self, actions: torch.Tensor):
        """The step's memory traffic and nothing else (qr_touch): one launch that reads and writes per env exactly what step()
        does, without arithmetic — the yardstick bench.py prices a step against.  The state is left as it was (bit for bit); the
        env's output rows (what the last step() returned) hold zeros afterwards."""
        a = self._check_actions(actions)
        with self._on_device():
            rc = self._lib.qr_touch(C.byref(self._cenv), a.data_ptr(), C.byref(self._cout), self._stream())
        _lib.check(rc, "qr_touch")

    def render(self, *a, **k):
        raise NotImplementedError("render (VPython GUI, quad.py:469-754) is out of scope")

    def close(self):
        self._closed = True


class CapturedStep:
    """A captured env.step(): see QuadVecEnv.capture().  `actions` is the static buffer every replay reads; calling the object
    replays the graph on torch's current stream and returns step()'s tuple (the env's own output buffers)."""

    def __init__(self, env: QuadVecEnv, graph, actions: torch.Tensor, ret, n_steps: int):
        self.env, self.graph, self.actions, self.n_steps = env, graph, actions, n_steps
        self._ret, self._epoch = ret, env._epoch

    def __call__(self, actions: Optional[torch.Tensor] = None):
        env = self.env
        if env._epoch != self._epoch:
            raise RuntimeError("the env changed after capture() (a buffer was allocated, the seed or the launch rule changed, or a state_dict "
                               "was loaded): the captured launch holds stale arguments — call env.capture() again")
        if actions is not None and actions is not self.actions:
            if self.actions is None:
                raise ValueError("this capture runs a caller-supplied body: it has no action buffer")
            self.actions.copy_(actions)      # (a copy kernel per step: write into .actions in place to avoid it)
        self.graph.replay()
        env._last_obs = env._obs()
        return self._ret

    replay = __call__


def as_gymnasium_vector_env(env: "QuadVecEnv"):
    """`env` as a `gymnasium.vector.VectorEnv` (the reference env is a `gym.Env`, quad.py:19, with the spaces of
    quad.py:108-132) when gymnasium is importable; raises ImportError otherwise — the engine itself never needs gymnasium.

    Everything stays a torch tensor on the GPU; nothing here synchronises with the host.
      reset(seed=, options=)  -> (obs[N, D], info).  options: {"env_type": "train" | "eval"} (the reference's extra reset argument,
                                 quad.py:171), {"reset_mask": bool[N]} (Gymnasium's partial reset).  The first observation is what the
                                 reference's loop forms after a reset: [goal] -> get_norm_error_state() (main.py:126-129) — the
                                 integral terms advance by that call exactly as they do there.
      step(actions[N, A])     -> (obs[N, D], reward, terminated, truncated, info).  obs: the agents' rows side by side (Decoupled:
                                 15 + 3 = 18 columns; info["obs_per_agent"] holds the tuple).  Single-agent kinds: reward[N],
                                 terminated[N].  Decoupled: reward[N, 2] (one column per agent, as the reference returns one reward
                                 per agent) and terminated[N] = either agent's flag — the episode ends and the env is re-sampled on
                                 either (main.py:183-186); info["terminated_per_agent"] is the [N, 2] tensor.
      autoreset               metadata["autoreset_mode"] = SAME_STEP when the env was built with auto_reset=True (terminated envs are
                                 re-sampled inside the step launch: the returned obs row is the new episode's first observation), else
                                 DISABLED (call reset(options={"reset_mask": ...})).  With final_obs=True the TERMINAL observation of
                                 every env that ended is reachable as info["final_obs"][N, D] under the mask info["_final_obs"][N]
                                 (Gymnasium >= 1.0 names; the same tensors under "final_observation" / "_final_observation", the names
                                 of the Gymnasium 0.28 the reference pins).
      spaces                  single_observation_space: Box(-inf, inf, (D,)) for the wrappers (normalised errors are not bounded by
                                 the state box), the reference's state box for kind='quad' (quad.py:108-125); single_action_space:
                                 Box(-1, 1, (A,)) (quad.py:127-132); observation_space / action_space: their batch_space; all seedable
                                 (utils.py:17-18 calls .seed()).
    """
    import gymnasium as gym

    D = sum(env.obs_dims)
    if env.kind == "quad":
        so = env.single_observation_space
        single_obs = gym.spaces.Box(so.low, so.high, dtype=np.float32)
    else:
        single_obs = gym.spaces.Box(-np.inf, np.inf, shape=(D,), dtype=np.float32)
    single_act = gym.spaces.Box(-1.0, 1.0, shape=(env.action_dim,), dtype=np.float32)
    modes = getattr(gym.vector, "AutoresetMode", None)
    mode = (modes.SAME_STEP if env.auto_reset else modes.DISABLED) if modes is not None else ("SameStep" if env.auto_reset else "Disabled")

    class GymnasiumQuadVecEnv(gym.vector.VectorEnv):
        metadata = {"render_modes": [], "autoreset_mode": mode}

        def __init__(self):
            batch = getattr(getattr(gym.vector, "utils", None), "batch_space", None)
            obs_space = batch(single_obs, env.num_envs) if batch else single_obs
            act_space = batch(single_act, env.num_envs) if batch else single_act
            try:      # Gymnasium 0.28 (the reference's pin): VectorEnv.__init__(num_envs, observation_space, action_space)
                super().__init__(env.num_envs, single_obs, single_act)
            except TypeError:   # Gymnasium >= 1.0: no constructor arguments, attributes are set by the subclass
                super().__init__()
            self.env = env
            self.num_envs = env.num_envs
            self.single_observation_space, self.single_action_space = single_obs, single_act
            self.observation_space, self.action_space = obs_space, act_space
            self.render_mode = None

        @staticmethod
        def _cat(obs):
            return obs if isinstance(obs, torch.Tensor) else torch.cat(list(obs), 1)

        def _first_obs(self):
            if env.kind == "quad":
                return env.get_current_state().to(torch.float32)
            if env.goal_mode is not None:
                env.get_desired(store_goal=True)    # main.py:226-229: mark_traj_start (done by reset), get_desired, set_goal_state, then the observation
            return self._cat(env.get_norm_error_state())

        def reset(self, *, seed=None, options=None):
            options = dict(options or {})
            mask = options.get("reset_mask")
            if mask is not None:
                mask = torch.as_tensor(mask, device=env.device).to(torch.bool)
            if mask is None or env.kind == "quad":
                env.reset(options.get("env_type", "train"), seed=seed, mask=mask)
                return self._first_obs(), {}
            # partial reset: the envs outside the mask keep their integral terms and the observation rows of their last step
            # (get_norm_error_state() forms rows for the whole batch and advances every env's integrators)
            keep_integ = env._integ.clone()
            keep_rows = [t.clone() for t in (env._obs0, env._obs1) if t is not None]
            env.reset(options.get("env_type", "train"), seed=seed, mask=mask)
            self._first_obs()
            env._integ.copy_(torch.where(mask[None, :], env._integ, keep_integ))
            for t, k in zip([t for t in (env._obs0, env._obs1) if t is not None], keep_rows):
                t.copy_(torch.where(mask[:, None], t, k))
            env._last_obs = env._obs()
            return self._cat(env._last_obs), {}

        def step(self, actions):
            obs, rwd, term, trunc, _ = env.step(torch.as_tensor(actions, dtype=torch.float32, device=env.device))
            if obs is None:
                obs = env.get_current_state().to(torch.float32)
            info = {}
            if env.n_agents == 1:
                rwd, ended = rwd[:, 0], term[:, 0]
            else:
                info["terminated_per_agent"], info["obs_per_agent"] = term, obs
                ended = term.any(dim=1)
            if env._final0 is not None:
                fin, m = self._cat(env.final_observation()), ended | trunc
                info["final_obs"], info["_final_obs"] = fin, m
                info["final_observation"], info["_final_observation"] = fin, m
            return self._cat(obs), rwd, ended, trunc, info

        def close_extras(self, **kwargs):
            env.close()

        def close(self, **kwargs):
            parent = getattr(super(), "close", None)
            if parent is not None:
                parent(**kwargs)
            env.close()
            self.closed = True

    return GymnasiumQuadVecEnv()
