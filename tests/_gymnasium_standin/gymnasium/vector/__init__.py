import enum

from . import utils  # noqa: F401


class AutoresetMode(enum.Enum):
    NEXT_STEP = "NextStep"
    SAME_STEP = "SameStep"
    DISABLED = "Disabled"


class VectorEnv:
    metadata = {}
    num_envs = 1
    closed = False
    render_mode = None

    def reset(self, *, seed=None, options=None):
        raise NotImplementedError

    def step(self, actions):
        raise NotImplementedError

    def close_extras(self, **kwargs):
        pass

    def close(self, **kwargs):
        if not self.closed:
            self.close_extras(**kwargs)
            self.closed = True

    @property
    def unwrapped(self):
        return self
