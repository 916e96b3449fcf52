"""deep-rl-battlespace_amd -- the Battlespace step() path, batched, on MI355X.

Only what the hot path needs lives here:
  csrc/          hand-written HIP kernels for gfx950 + the C ABI of include/battlespace_hip.h
  build.py       hipcc recipe for csrc/ (in-tree libbattlespace_hip.so)
  _lib.py        ctypes binding of the C ABI (fails loudly when the library is missing)
  envs/battle_env.py   `parallel_env`: the reference's PettingZoo ParallelEnv surface, batched over n_envs
  spaces.py      Box / Discrete metadata containers (gym is not a dependency)
  sharding.py    one contiguous env range per rank (no collective on the step path)
  instinct/      the reference's scripted opponent, evaluated on device (next row f-2)
  rollout.py     on-device actor + step loop in one HIP graph (next row f-1); csrc/bsx_actor.hip is its MFMA actor
  replay.py      device-resident transition ring, mirror of maddpg/buffer.py (next row f-3)
  render.py      one game's exported state as an RGB image on the host, diagnostic (next row f-4)
There is no CPU fallback: without the HIP library the env cannot be constructed.
"""
from . import envs, instinct  # noqa: F401
from .envs.battle_env import parallel_env  # noqa: F401

__all__ = ["envs", "instinct", "parallel_env"]
