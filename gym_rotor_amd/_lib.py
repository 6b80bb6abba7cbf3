"""ctypes binding of the C-ABI in include/quadrotor_hip.h (libquadrotor_hip.so).

The library is built in-tree by `__graft_entry__.build()` / `make -C gym_rotor_amd/csrc`.
There is deliberately NO fallback: if the shared library is missing or a symbol is absent
the import of the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QR_LIB overrides the path for measurement builds (tools/microbench.py ablations) only.
LIB_PATH = os.environ.get("QR_LIB", os.path.join(_HERE, "libquadrotor_hip.so"))

KIND_QUAD, KIND_COUPLED, KIND_DECOUPLED = 0, 1, 2
KIND_ID = {"quad": KIND_QUAD, "coupled": KIND_COUPLED, "decoupled": KIND_DECOUPLED}
FLAG_AUTO_RESET, FLAG_EVAL_RESET, FLAG_NO_UDM = 1, 2, 4
FLAG_FORCE_HELPER, FLAG_NO_HELPER = 8, 16   # launch-rule overrides of the one-step launch (speed only)
FLAG_CALLER_RESETS = 32                     # the caller resets every done env before stepping it again (one-step launches)
FLAG_FORCE_HELPER_ROLLOUT, FLAG_NO_HELPER_ROLLOUT = 64, 128   # the same overrides for qr_rollout / qr_rollout_actor
ABI_VERSION = 15
GOAL_EXTERNAL, GOAL_MODE0, GOAL_MODE1, GOAL_MODE6, GOAL_MODE2, GOAL_MODE3, GOAL_MODE4, GOAL_MODE5 = 0, 1, 2, 3, 4, 5, 6, 7
GOAL_ID = {None: 0, 0: 1, 1: 2, 6: 3, 2: 4, 3: 5, 4: 6, 5: 7}  # TrajectoryGenerator mode -> QR_GOAL_*
LAYOUT_ID = {"mixed": 0, "f64": 1, "f32": 2}

ERRORS = {-1: "QR_E_NULL: a required pointer is NULL", -2: "QR_E_KIND: bad env kind",
          -3: "QR_E_SIZE: bad num_envs / substeps / n_steps / coefficients", -4: "QR_E_ALIGN: buffer not 16-byte aligned"}

# every symbol include/quadrotor_hip.h declares
SYMBOLS = ("qr_step", "qr_rollout", "qr_rollout_actor", "qr_error_obs", "qr_error_obs_format", "qr_reset", "qr_get_state", "qr_set_state", "qr_check_state",
           "qr_traj_start", "qr_get_desired", "qr_gae",
           "qr_default_coeffs", "qr_abi_version", "qr_step_kernel_info", "qr_launch_thresholds",
           "qr_launch_plan", "qr_launch_stats", "qr_instance_table", "qr_touch")


class QrCoeffs(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "Cx", "CIx", "Cv", "Cb1", "CIb1", "CW", "Cw12", "CW3", "alpha", "beta", "dt",
        "x_lim", "v_lim", "W_lim", "eIx_lim", "eIb1_lim", "euler_lim_deg", "udm_fraction",
        "eight_T", "eight_A1", "eight_A2", "eight_w_b1d", "eight_alt_d", "eight_eps", "eight_count", "w_adapt",
        "m_nominal", "d_nominal", "J1_nominal", "J3_nominal", "c_tf_nominal", "c_tw_nominal", "g", "min_force")]


class QrEnv(C.Structure):
    _fields_ = [("kind", C.c_int32), ("layout", C.c_int32), ("num_envs", C.c_int64), ("field_stride", C.c_int64),
                ("env_offset", C.c_int64), ("seed", C.c_uint64),
                ("pos_vel", C.c_void_p), ("att_rate", C.c_void_p),
                ("integ", C.c_void_p), ("params", C.c_void_p), ("goal", C.c_void_p),
                ("traj", C.c_void_p), ("goal_mode", C.c_int32), ("reserved0", C.c_int32),
                ("episode", C.c_void_p), ("steps", C.c_void_p), ("reset_count", C.c_void_p),
                ("max_episode_steps", C.c_int32), ("flags", C.c_uint32), ("coeffs", QrCoeffs)]


class QrStepOut(C.Structure):
    _fields_ = [("obs0", C.c_void_p), ("obs1", C.c_void_p), ("reward", C.c_void_p),
                ("reward_raw", C.c_void_p), ("done", C.c_void_p), ("truncated", C.c_void_p),
                ("final_obs0", C.c_void_p), ("final_obs1", C.c_void_p)]


class QrActor(C.Structure):
    _fields_ = [("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p), ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p),
                ("mean_w", C.c_void_p), ("mean_b", C.c_void_p), ("log_std", C.c_void_p),
                ("log_std_w", C.c_void_p), ("log_std_b", C.c_void_p),
                ("obs_dim", C.c_int32), ("hidden_dim", C.c_int32), ("action_dim", C.c_int32), ("squash", C.c_int32)]


ACTOR_TANH_MEAN, ACTOR_TANH_SAMPLE = 0, 1


class QrPolicyRollout(C.Structure):
    _fields_ = [("actors", C.POINTER(QrActor)), ("obs0_in", C.c_void_p), ("obs1_in", C.c_void_p), ("noise", C.c_void_p),
                ("noise_seed", C.c_uint64), ("step_base", C.c_uint64), ("max_action", C.c_float), ("deterministic", C.c_int32),
                ("action_out", C.c_void_p), ("logprob_out", C.c_void_p)]


class QrLaunchPlan(C.Structure):
    _fields_ = [("grid", C.c_int32), ("block", C.c_int32), ("launches", C.c_int32),
                ("traj", C.c_int32), ("adapt", C.c_int32), ("policy", C.c_int32), ("single", C.c_int32), ("help", C.c_int32), ("hrew", C.c_int32),
                ("mag", C.c_int32), ("key", C.c_uint32), ("name", C.c_char * 96)]


class QuadrotorLibError(RuntimeError):
    pass


_lib = None


def load():
    """Load libquadrotor_hip.so (once) and type its entry points.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QuadrotorLibError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gym_rotor_amd/csrc`.  gym_rotor_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise QuadrotorLibError(f"{LIB_PATH} does not export {s}")
    P = C.POINTER
    lib.qr_abi_version.restype = C.c_int
    lib.qr_abi_version.argtypes = []
    lib.qr_default_coeffs.restype = None
    lib.qr_default_coeffs.argtypes = [P(QrCoeffs)]
    lib.qr_step.restype = C.c_int
    lib.qr_step.argtypes = [P(QrEnv), C.c_void_p, C.c_int32, P(QrStepOut), C.c_void_p]
    lib.qr_rollout.restype = C.c_int
    lib.qr_rollout.argtypes = [P(QrEnv), C.c_void_p, C.c_int32, C.c_int32, P(QrStepOut), C.c_void_p]
    lib.qr_rollout_actor.restype = C.c_int
    lib.qr_rollout_actor.argtypes = [P(QrEnv), P(QrPolicyRollout), C.c_int32, C.c_int32, P(QrStepOut), C.c_void_p]
    lib.qr_error_obs.restype = C.c_int
    lib.qr_error_obs.argtypes = [P(QrEnv), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.qr_error_obs_format.restype = C.c_int
    lib.qr_error_obs_format.argtypes = [P(QrEnv), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.qr_reset.restype = C.c_int
    lib.qr_reset.argtypes = [P(QrEnv), C.c_void_p, C.c_void_p]
    lib.qr_get_state.restype = C.c_int
    lib.qr_get_state.argtypes = [P(QrEnv), C.c_void_p, C.c_void_p]
    lib.qr_set_state.restype = C.c_int
    lib.qr_set_state.argtypes = [P(QrEnv), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.qr_check_state.restype = C.c_int
    lib.qr_check_state.argtypes = [P(QrEnv), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.qr_traj_start.restype = C.c_int
    lib.qr_traj_start.argtypes = [P(QrEnv), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.qr_get_desired.restype = C.c_int
    lib.qr_get_desired.argtypes = [P(QrEnv), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.qr_gae.restype = C.c_int
    lib.qr_gae.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_float,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.qr_step_kernel_info.restype = C.c_char_p
    lib.qr_step_kernel_info.argtypes = [P(QrEnv), C.c_int32, P(C.c_int32), P(C.c_int32)]
    lib.qr_launch_plan.restype = C.c_int
    lib.qr_launch_plan.argtypes = [P(QrEnv), C.c_int32, C.c_int32, C.c_int32, P(QrLaunchPlan)]
    lib.qr_launch_stats.restype = C.c_int32
    lib.qr_launch_stats.argtypes = [P(C.c_uint32), P(C.c_uint32), C.c_int32, C.c_int32]
    lib.qr_instance_table.restype = C.c_int32
    lib.qr_instance_table.argtypes = [P(C.c_uint32), C.c_int32]
    lib.qr_touch.restype = C.c_int
    lib.qr_touch.argtypes = [P(QrEnv), C.c_void_p, P(QrStepOut), C.c_void_p]
    lib.qr_launch_thresholds.restype = None
    lib.qr_launch_thresholds.argtypes = [P(C.c_int32), P(C.c_int32), P(C.c_int32)]
    if lib.qr_abi_version() != ABI_VERSION:
        raise QuadrotorLibError(f"ABI mismatch: library {lib.qr_abi_version()} vs binding {ABI_VERSION}")
    _lib = lib
    return lib


def default_coeffs() -> QrCoeffs:
    c = QrCoeffs()
    load().qr_default_coeffs(C.byref(c))
    return c


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc < 0:
        raise ValueError(f"{what}: {ERRORS.get(rc, rc)}")
    raise QuadrotorLibError(f"{what}: hipError_t {rc}")


def launch_thresholds() -> dict:
    """The launch rule's helper-wavefront thresholds of this process, in 64-env tiles (qr_launch_thresholds)."""
    q, w, r = C.c_int32(), C.c_int32(), C.c_int32()
    load().qr_launch_thresholds(C.byref(q), C.byref(w), C.byref(r))
    return {"step_quad": q.value, "step_wrappers": w.value, "rollout": r.value}


LAYOUT_NAME = {v: k for k, v in LAYOUT_ID.items()}
KIND_NAME = {v: k for k, v in KIND_ID.items()}


def describe_key(key: int) -> str:
    """'layout/kind TRAJ=.. ADAPT=.. POLICY=.. SINGLE=.. HELP=.. HREW=.. MAG=..' of a qr_launch_stats / qr_instance_table key."""
    b = key & 0xFF
    return (f"{LAYOUT_NAME[key >> 16]}/{KIND_NAME[(key >> 8) & 0xF]} TRAJ={b & 3} ADAPT={(b >> 2) & 1} POLICY={(b >> 3) & 3} "
            f"SINGLE={(b >> 5) & 1} HELP={(b >> 6) & 1} HREW={(b >> 7) & 1} MAG={(key >> 12) & 1}")


def launch_stats(reset: bool = False) -> dict:
    """{key: launches} of every step-kernel instantiation this process has launched (qr_launch_stats)."""
    lib = load()
    cap = 512
    keys, counts = (C.c_uint32 * cap)(), (C.c_uint32 * cap)()
    n = lib.qr_launch_stats(keys, counts, cap, int(reset))
    return {int(keys[i]): int(counts[i]) for i in range(min(n, cap))}


def instance_table() -> list:
    """Keys of every step-kernel instantiation the library holds (qr_instance_table)."""
    lib = load()
    cap = 512
    keys = (C.c_uint32 * cap)()
    n = lib.qr_instance_table(keys, cap)
    return [int(keys[i]) for i in range(min(n, cap))]
