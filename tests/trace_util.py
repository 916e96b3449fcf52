"""Helpers shared by the parity tests: load golden traces, replay them, compare field by field.

A *trace* is the flat-step record documented in tests/golden/make_golden.py."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

STATE_FIELDS = ("px", "py", "pdir", "php", "palive", "bhp", "tick", "bl_live", "bl_x", "bl_y", "bl_dir",
                "total_games", "ties", "wins_red", "wins_blue")
OUT_FIELDS = ("obs", "rew", "done", "env_done", "winner")
WINNER_CODE = {"none": 0, "red": 1, "blue": 2, "tie": 3}


def trace_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "g[1-5]_*.npz")))


def load_trace(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    t = {k: z[k] for k in z.files if k != "meta"}
    t["meta"] = json.loads(str(z["meta"]))
    t["name"] = name
    return t


def episodes(t):
    p = t["ep_ptr"]
    for e in range(len(p) - 1):
        yield e, int(p[e]), int(p[e + 1])


def assert_step_equal(name, s, got, t, obs_rtol=0.0, rew_rtol=1e-12):
    """got: dict with OUT_FIELDS + STATE_FIELDS for one step; t: the golden trace; s: flat step index."""
    def fail(f, a, b):
        raise AssertionError(f"{name}: step {s}: field {f} differs\n got {a!r}\n exp {b!r}")

    for f in ("done", "env_done", "winner", "px", "py", "php", "palive", "bhp", "tick",
              "total_games", "ties", "wins_red", "wins_blue", "bl_live"):
        if f in got and not np.array_equal(np.asarray(got[f]), t[f][s]):
            fail(f, np.asarray(got[f]), t[f][s])
    if "pdir" in got and not np.array_equal(np.asarray(got["pdir"], np.float64), t["pdir"][s]):
        fail("pdir", got["pdir"], t["pdir"][s])
    if "bl_live" in got:
        m = t["bl_live"][s]
        for f in ("bl_x", "bl_y", "bl_dir"):
            a, b = np.asarray(got[f])[m], t[f][s][m]
            if not np.array_equal(a, b):
                fail(f, a, b)
    if obs_rtol == 0.0:
        if not np.array_equal(np.asarray(got["obs"], np.float32), t["obs"][s]):
            fail("obs", got["obs"], t["obs"][s])
    else:
        np.testing.assert_allclose(np.asarray(got["obs"], np.float64), t["obs"][s].astype(np.float64), rtol=obs_rtol,
                                   atol=0, err_msg=f"{name}: step {s}: obs")
    np.testing.assert_allclose(np.asarray(got["rew"], np.float64), t["rew"][s], rtol=rew_rtol, atol=0,
                               err_msg=f"{name}: step {s}: rew")


def step_actions(t, s, ids):
    """The actions dict the reference was called with at flat step s."""
    if t["empty_call"][s]:
        return {}
    if "logits" in t:
        return {a: t["logits"][s, i].copy() for i, a in enumerate(ids)}
    if t["meta"]["continuous"]:
        return {a: t["actions"][s, i].copy() for i, a in enumerate(ids)}
    return {a: int(t["actions"][s, i]) for i, a in enumerate(ids)}
