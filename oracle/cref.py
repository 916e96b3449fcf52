"""ctypes wrapper of oracle/battlespace_ref.c -- the C restatement of the reference step() path.  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load(build=True):
    global _LIB
    if _LIB is None:
        path = os.environ.get("BSR_LIB") or os.path.join(_HERE, "libbattlespace_ref.so")   # BSR_LIB: e.g. the ASan build
        src = os.path.join(_HERE, "battlespace_ref.c")
        if build and not os.environ.get("BSR_LIB") and (not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src)):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        lib = ctypes.CDLL(path)
        vp, i64, u64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int
        lib.bsr_create.restype = vp
        lib.bsr_create.argtypes = [i64, ci, ci, ctypes.POINTER(ctypes.c_double)]
        lib.bsr_destroy.argtypes = [vp]
        lib.bsr_reset.argtypes = [vp, vp, vp, u64, u64, i64, vp]
        lib.bsr_step.argtypes = [vp, vp, ci, vp, ci, ci, u64, i64, vp, vp, vp, vp, vp]
        lib.bsr_observe.argtypes = [vp, vp]
        lib.bsr_export.argtypes = [vp] * 16
        for f in (lib.bsr_destroy, lib.bsr_reset, lib.bsr_step, lib.bsr_observe, lib.bsr_export):
            f.restype = None
        _LIB = lib
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


EXPORT = (("px", np.int32, "A"), ("py", np.int32, "A"), ("pdir", np.float64, "A"), ("php", np.int32, "A"),
          ("palive", np.uint8, "A"), ("base_xy", np.int32, 4), ("bhp", np.int32, 2), ("tick", np.int32, 0),
          ("env_done", np.uint8, 0), ("winner", np.uint8, 0), ("bl_live", np.uint8, "AK"), ("bl_x", np.int32, "AK"),
          ("bl_y", np.int32, "AK"), ("bl_dir", np.float64, "AK"), ("counters", np.int32, 4))


class CRefBatch:
    """E independent games advanced by the C oracle (OpenMP over games)."""

    def __init__(self, n_envs, n_agents=1, hit_base_reward=100, hit_plane_reward=10, miss_punishment=-1,
                 die_punishment=-5, lose_punishment=-20, continuous_actions=False, seed=0, env_offset=0,
                 auto_reset=False, **_ignored):
        self.lib = load()
        self.E, self.n, self.A, self.D = int(n_envs), int(n_agents), 2 * int(n_agents), 3 * int(n_agents) + 2
        self.continuous = bool(continuous_actions)
        cfg = (ctypes.c_double * 5)(hit_base_reward, hit_plane_reward, miss_punishment, die_punishment, lose_punishment)
        self.h = self.lib.bsr_create(self.E, self.n, int(self.continuous), cfg)
        if not self.h:
            raise ValueError("bsr_create failed")
        self.seed, self.env_offset, self.auto_reset = int(seed), int(env_offset), bool(auto_reset)
        self.nonce = 0
        self.obs = np.zeros((self.E, self.A, self.D), np.float32)
        self.rew = np.zeros((self.E, self.A), np.float64)
        self.done = np.ones((self.E, self.A), np.uint8)
        self.env_done = np.ones(self.E, np.uint8)
        self.winner = np.zeros(self.E, np.uint8)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.bsr_destroy(self.h)
            self.h = None

    def reset(self, spawn=None, mask=None):
        self.nonce += 1
        sp = None if spawn is None else np.ascontiguousarray(spawn, np.int32)
        mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.lib.bsr_reset(self.h, _p(mk), _p(sp), self.seed, self.nonce, self.env_offset, _p(self.obs))
        sel = slice(None) if mk is None else mk.astype(bool)     # the flags step() reports, as they stand after the reset
        self.env_done[sel] = 0; self.winner[sel] = 0; self.done[sel] = 0
        return self.obs

    def step(self, actions, u=None, empty=False):
        if self.continuous:
            a = np.ascontiguousarray(actions)
            kind = 1 if a.dtype == np.float64 else 0
            if kind == 0:
                a = np.ascontiguousarray(a, np.float32)
        else:
            a = np.asarray(actions)
            if a.ndim == 3:
                a, kind = np.ascontiguousarray(a, np.float32), 1
            else:
                a, kind = np.ascontiguousarray(a, np.int32), 0
        uu = None if u is None else np.ascontiguousarray(u, np.float64)
        self.lib.bsr_step(self.h, _p(a), kind, _p(uu), int(empty), int(self.auto_reset), self.seed, self.env_offset,
                          _p(self.obs), _p(self.rew), _p(self.done), _p(self.env_done), _p(self.winner))
        return self.obs, self.rew, self.done.astype(bool)

    def export_state(self):
        E, A = self.E, self.A
        out = {}
        for name, dt, shp in EXPORT:
            s = {"A": (E, A), "AK": (E, A, 12), 0: (E,)}.get(shp, (E, shp) if isinstance(shp, int) and shp else None)
            out[name] = np.zeros(s, dt)
        self.lib.bsr_export(self.h, *[_p(out[name]) for name, _, _ in EXPORT])
        return out
