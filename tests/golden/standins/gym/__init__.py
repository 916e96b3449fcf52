"""gym 0.25.0 stand-in -- FIXTURE GENERATION ONLY (build-authored; gym is absent from this image).
Only what /root/reference/envs/battle_env.py touches: spaces.Box / spaces.Discrete containers and utils.EzPickle."""
from . import spaces, utils  # noqa: F401
