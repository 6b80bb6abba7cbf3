"""A stand-in for the `gymnasium` package (not installed in this image, no network) with just the surface
gym_rotor_amd.as_gymnasium_vector_env touches: Env, spaces.Box, vector.VectorEnv, vector.AutoresetMode,
vector.utils.batch_space.  Original, arithmetic-free; used by tests/test_gymnasium_adapter.py only (put on sys.path there)."""
from . import spaces, vector  # noqa: F401

__version__ = "0.0-standin"


class Env:
    metadata = {"render_modes": []}
    observation_space = action_space = None

    def reset(self, *, seed=None, options=None):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def close(self):
        pass
