"""Minimal stand-in for `gymnasium`, used ONLY by tools/gen_golden.py in the build
container so that /root/reference imports (gymnasium is not installed, no network).
Not part of the product: gym_rotor_amd never imports this."""
from . import spaces, utils, envs  # noqa: F401


class Env:
    metadata = {}

    def reset(self, *, seed=None, options=None):
        return None

    def close(self):
        pass
