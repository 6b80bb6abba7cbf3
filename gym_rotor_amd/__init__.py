"""gym_rotor_amd — MI355X-native batched quadrotor dynamics engine.

Drop-in for the env.step() hot path of fdcl-gwu/gym-rotor (Quad-v0, CoupledWrapper,
DecoupledWrapper), executed by hand-written HIP kernels for gfx950 behind a C-ABI
(include/quadrotor_hip.h).  See DESIGN.md / INTEGRATION.md.
"""
from .constants import ACTION_DIM, ALGO_BYTES, FRAMEWORK, KINDS, N_AGENTS, OBS_DIMS, QuadConstants  # noqa: F401
from .spaces import Box  # noqa: F401
from .sharding import shard_range, make_sharded_env, all_gather_rows  # noqa: F401
from .vec_env import CapturedStep, QuadVecEnv, as_gymnasium_vector_env  # noqa: F401
from .compat import QuadEnv, CoupledWrapper, DecoupledWrapper  # noqa: F401
from .rollout import RolloutStorage  # noqa: F401
from .policy import ActorParams, random_actors  # noqa: F401
from . import torch_ops  # noqa: F401  (registers torch.ops.gym_rotor_amd.*)

__all__ = ["QuadVecEnv", "QuadEnv", "CoupledWrapper", "DecoupledWrapper", "QuadConstants", "Box",
           "shard_range", "make_sharded_env", "all_gather_rows", "RolloutStorage"]
