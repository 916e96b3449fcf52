"""pettingzoo.utils stand-in: identity wrappers (the AEC wrappers are never on the step() path)."""
from . import wrappers  # noqa: F401


def parallel_to_aec(env):
    return env
