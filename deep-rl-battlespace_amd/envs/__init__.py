"""Mirror of the reference's `envs` package: `envs.battle_env.parallel_env` (reference envs/battle_env.py:61)."""
from . import battle_env  # noqa: F401
