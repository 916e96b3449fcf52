"""Observation / action space metadata (reference envs/battle_env.py:132-135,144-162).  gym is not a dependency: these
are plain containers with the attributes callers read (low, high, shape, dtype, n) plus sample()."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def sample(self):
        lo, hi = np.minimum(self.low, self.high), np.maximum(self.low, self.high)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return f"Box({self.low.flat[0]}, {self.high.flat[0]}, {self.shape}, {self.dtype})"


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def sample(self):
        return int(np.random.randint(self.n))

    def __repr__(self):
        return f"Discrete({self.n})"
