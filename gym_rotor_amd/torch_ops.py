"""`torch.ops.gym_rotor_amd.*` — the C-ABI entry points registered as PyTorch custom ops
(SURVEY.md §8b), for callers that want the step inside a torch program (torch.compile /
export / CUDA-graph capture) rather than through QuadVecEnv.  Tensors carry the device buffers,
the op body is a ctypes call into libquadrotor_hip.so on the current stream; there is no CPU
kernel behind these ops (a CPU tensor raises).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

_NS = "gym_rotor_amd"


_RESET_COUNTS: dict = {}


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _env_struct(kind, layout, pos_vel, att_rate, integ, params, goal, episode, seed, env_offset, flags):
    if pos_vel.device.type != "cuda":
        raise RuntimeError("gym_rotor_amd ops run on the GPU only (no CPU kernel exists)")
    e = _lib.QrEnv()
    e.kind, e.layout = kind, layout
    e.num_envs, e.field_stride = pos_vel.shape[1], pos_vel.stride(0)
    e.env_offset, e.seed, e.flags = env_offset, seed & (2 ** 64 - 1), flags
    e.pos_vel, e.att_rate, e.integ, e.params, e.goal, e.episode = _p(pos_vel), _p(att_rate), _p(integ), _p(params), _p(goal), _p(episode)
    e.coeffs = _lib.default_coeffs()
    if flags & _lib.FLAG_AUTO_RESET:  # per-tile stream position of the in-launch reset (quadrotor_hip.h: reset_count)
        key = (pos_vel.data_ptr(), pos_vel.shape[1])
        if key not in _RESET_COUNTS:
            _RESET_COUNTS[key] = torch.zeros((pos_vel.shape[1] + 63) // 64, dtype=torch.int32, device=pos_vel.device)
        e.reset_count = _RESET_COUNTS[key].data_ptr()
    return e


@torch.library.custom_op(f"{_NS}::qr_step", mutates_args=("pos_vel", "att_rate", "integ", "params", "episode", "obs0", "obs1", "reward", "done"))
def qr_step(pos_vel: torch.Tensor, att_rate: torch.Tensor, integ: Optional[torch.Tensor], params: Optional[torch.Tensor],
            goal: Optional[torch.Tensor], episode: Optional[torch.Tensor], action: torch.Tensor,
            obs0: Optional[torch.Tensor], obs1: Optional[torch.Tensor], reward: torch.Tensor, done: torch.Tensor,
            kind: int, layout: int, substeps: int, flags: int, seed: int, env_offset: int) -> None:
    """QuadEnv.step for all envs (include/quadrotor_hip.h: qr_step).  SoA buffers [F, N] (row stride
    = field stride), action [N, A] float32 contiguous, outputs as in QrStepOut."""
    e = _env_struct(kind, layout, pos_vel, att_rate, integ, params, goal, episode, seed, env_offset, flags)
    o = _lib.QrStepOut()
    o.obs0, o.obs1, o.reward, o.done = _p(obs0), _p(obs1), _p(reward), _p(done)
    rc = _lib.load().qr_step(C.byref(e), action.data_ptr(), substeps, C.byref(o), torch.cuda.current_stream(pos_vel.device).cuda_stream)
    _lib.check(rc, "qr_step")


@torch.library.custom_op(f"{_NS}::qr_reset", mutates_args=("pos_vel", "att_rate", "integ", "params", "episode"))
def qr_reset(pos_vel: torch.Tensor, att_rate: torch.Tensor, integ: Optional[torch.Tensor], params: Optional[torch.Tensor],
             episode: torch.Tensor, mask: Optional[torch.Tensor], kind: int, layout: int, flags: int, seed: int, env_offset: int) -> None:
    """QuadEnv.reset for masked envs (qr_reset)."""
    e = _env_struct(kind, layout, pos_vel, att_rate, integ, params, None, episode, seed, env_offset, flags)
    rc = _lib.load().qr_reset(C.byref(e), _p(mask), torch.cuda.current_stream(pos_vel.device).cuda_stream)
    _lib.check(rc, "qr_reset")


@torch.library.custom_op(f"{_NS}::qr_gae", mutates_args=("advantage", "td_target"))
def qr_gae(reward: torch.Tensor, done: torch.Tensor, value: torch.Tensor, gamma: float, lam: float,
           advantage: torch.Tensor, td_target: torch.Tensor) -> None:
    """GAE reverse scan (qr_gae): reward/done [T, M], value [T+1, M] (row T = bootstrap)."""
    if reward.device.type != "cuda":
        raise RuntimeError("gym_rotor_amd ops run on the GPU only (no CPU kernel exists)")
    T, M = reward.shape[0], reward[0].numel()
    rc = _lib.load().qr_gae(reward.data_ptr(), done.data_ptr(), value.data_ptr(), None, T, M, gamma, lam,
                            advantage.data_ptr(), td_target.data_ptr(), None, torch.cuda.current_stream(reward.device).cuda_stream)
    _lib.check(rc, "qr_gae")
