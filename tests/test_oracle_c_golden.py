"""The C oracle (oracle/battlespace_ref.c) against outputs of the reference itself and against the Python oracle.

Every golden trace is replayed with all its episodes side by side as one batch; the C restatement must reproduce
every output and every piece of game state bit for bit (observations: identical float32)."""
import random

import numpy as np
import pytest

from oracle import battlespace_ref as pyref
from oracle import cref
from trace_util import episode_groups, load_trace, replay_batched, trace_names


class CAdapter:
    def __init__(self, E, cfg):
        self.b = cref.CRefBatch(E, **cfg)

    def reset(self, spawn):
        return self.b.reset(spawn=spawn).copy()

    def step(self, act, u, empty):
        return self.b.step(act, u=u, empty=empty)

    def export(self):
        return self.b.export_state()

    def env_done(self):
        return self.b.env_done

    def winner(self):
        return self.b.winner


@pytest.mark.parametrize("name", trace_names())
def test_c_oracle_reproduces_reference_trace(name):
    t = load_trace(name)
    for group in episode_groups(t):
        n_exact, n_vals = replay_batched(CAdapter, t, group, exact_obs=True)
        assert n_exact == n_vals


@pytest.mark.parametrize("n,cont", [(1, False), (2, False), (4, False), (1, True), (3, True)])
def test_c_oracle_equals_python_oracle_on_random_play(n, cont):
    """Same spawns, actions and random() values through both restatements: identical state and outputs."""
    E, T, A = 24, 150, 2 * n
    rng = np.random.default_rng(1000 + n + 10 * cont)
    pys = [pyref.RefEnv(n_agents=n, continuous_actions=cont, rng=random.Random(7 * e + n)) for e in range(E)]
    spawn = np.zeros((E, 4 + 3 * A), np.int32)
    for e, p in enumerate(pys):
        p.reset()
        spawn[e] = [p.base_x[0], p.base_y[0], p.base_x[1], p.base_y[1]] + [v for i in range(A) for v in (p.px[i], p.py[i], p.pdir[i])]
    cb = cref.CRefBatch(E, n_agents=n, continuous_actions=cont)
    cb.reset(spawn=spawn)
    for t in range(T):
        if cont:
            act = rng.uniform(-1.2, 1.2, (E, A, 3))
        else:
            act = np.where(rng.random((E, A)) < 0.5, 1, rng.integers(-1, 5, (E, A))).astype(np.int32)
        u = rng.random((E, A))
        obs, rew, done = cb.step(act, u=u)
        st = cb.export_state()
        for e, p in enumerate(pys):
            ids = p.possible_agents
            a = {ids[i]: (act[e, i].copy() if cont else int(act[e, i])) for i in range(A)}
            ob, rw, dn, _ = p.step(a, u=list(u[e]))
            snap = p.snapshot()
            for f in ("px", "py", "php", "bhp", "pdir"):
                assert np.array_equal(st[f][e], snap[f]), (t, e, f)
            assert np.array_equal(st["bl_live"][e].astype(bool), snap["bl_live"]), (t, e)
            m = snap["bl_live"]
            assert np.array_equal(st["bl_dir"][e][m], snap["bl_dir"][m]) and np.array_equal(st["bl_x"][e][m], snap["bl_x"][m])
            assert np.array_equal(obs[e], np.stack([ob[i] for i in ids])), (t, e)
            assert np.array_equal(rew[e], np.asarray([float(rw[i]) for i in ids])), (t, e)
            assert [bool(dn[i]) for i in ids] == done[e].tolist()
            assert bool(cb.env_done[e]) == p.env_done and int(cb.winner[e]) == pyref.WINNER_CODE[p.winner]


def test_c_oracle_production_draws_have_the_reference_distribution():
    """The C oracle's Philox spawns (reset and auto-reset streams) and shot jitter against what the REFERENCE draws (fixture g7,
    sprites.py:314) -- the same checks tests/test_hip_rng_pin.py runs on the GPU's production path, so that "GPU == C oracle under
    Philox" (the full-size and soak tests) also says "both draw from the reference's ranges and support"."""
    from rng_pin_util import check_jitter, check_spawn_table, spawn_table
    E = 200_000
    c = cref.CRefBatch(E, n_agents=2, seed=20261004, auto_reset=True)
    c.reset()
    st = c.export_state()
    d = spawn_table(st)
    check_spawn_table(d, "C oracle, stream RESET")
    d0 = st["pdir"].copy()
    c.step(np.ones((E, 4), np.int32))                         # every plane fires: one fresh bullet each
    st = c.export_state()
    live = st["bl_live"].astype(bool)
    assert (live.sum(-1) == 1).all()
    check_jitter(st["bl_dir"][live], d0.reshape(-1), "C oracle, shot jitter")
    c.step(np.ones((E, 4), np.int32), empty=True)            # tie ...
    assert c.env_done.all()
    c.step(np.ones((E, 4), np.int32))                         # ... and the in-step re-spawn (stream AUTORESET)
    d2 = spawn_table(c.export_state())
    check_spawn_table(d2, "C oracle, stream AUTORESET")
    assert (d2 != d).any(1).mean() > 0.999
