"""The CPU oracle (oracle/battlespace_ref.py) against outputs of the reference itself.

Every golden trace was produced by the unmodified reference (tests/golden/make_golden.py).  The
oracle is replayed on the same spawn, action and random() inputs and must reproduce every output
and every piece of game state bit for bit (observations: identical float32; rewards: identical sums
up to float re-association, none here)."""
import math
import random

import numpy as np
import pytest

from oracle import battlespace_ref as ref
from trace_util import assert_step_equal, episodes, load_trace, step_actions, trace_names, GOLDEN, WINNER_CODE


def replay(t):
    cfg = dict(t["meta"]["cfg"])
    env = ref.RefEnv(**cfg)
    ids = env.possible_agents
    name = t["name"]
    for e, a, b in episodes(t):
        obs = env.reset(spawn=t["spawn"][e].tolist())
        got0 = np.stack([obs[i] for i in ids])
        assert np.array_equal(got0, t["obs0"][e]), f"{name}: ep {e}: reset obs"
        for s in range(a, b):
            u = [None if math.isnan(v) else float(v) for v in t["u"][s]]
            obs, rew, done, info = env.step(step_actions(t, s, ids), u=u)
            assert done is env.dones
            got = env.snapshot()
            got.update(obs=np.stack([obs[i] for i in ids]), rew=[float(rew[i]) for i in ids],
                       done=[done[i] for i in ids], env_done=env.env_done, winner=WINNER_CODE[env.winner])
            assert_step_equal(name, s, got, t, obs_rtol=0.0, rew_rtol=0.0)
            assert got["total_time"] == t["total_time"][s]


@pytest.mark.parametrize("name", trace_names())
def test_oracle_reproduces_reference_trace(name):
    replay(load_trace(name))


def test_oracle_same_seed_same_game_as_reference():
    """With the stdlib generator seeded as the fixture generator seeded it, the oracle draws spawns and
    jitters in the reference's order and so replays the whole C1 rollout from the seed alone."""
    t = load_trace("g2_1v1_random")
    random.seed(t["meta"]["seed"])
    env = ref.RefEnv(**t["meta"]["cfg"])      # the constructor consumes 4 + 3A draws, as the reference's does
    ids = env.possible_agents
    for e, a, b in episodes(t):
        obs = env.reset()
        assert np.array_equal(np.stack([obs[i] for i in ids]), t["obs0"][e])
        for s in range(a, b):
            obs, rew, done, _ = env.step(step_actions(t, s, ids))
            got = env.snapshot()
            got.update(obs=np.stack([obs[i] for i in ids]), rew=[float(rew[i]) for i in ids],
                       done=[done[i] for i in ids], env_done=env.env_done, winner=WINNER_CODE[env.winner])
            assert_step_equal(t["name"], s, got, t, rew_rtol=0.0)


def test_rel_angle_and_dist_table():
    z = np.load(f"{GOLDEN}/g6_rel_angle_table.npz")
    for x0, y0, x1, y1, a0, ra, di in z["table"]:
        a = int(a0) if float(a0).is_integer() else float(a0)
        assert ref.rel_angle((int(x0), int(y0)), a, (int(x1), int(y1))) == ra
        assert ref.dist((int(x0), int(y0)), (int(x1), int(y1))) == di


def test_spawn_ranges_match_reference_draws():
    z = np.load(f"{GOLDEN}/g7_spawn_stats.npz")
    rng = random.Random(11)
    env = ref.RefEnv(n_agents=2, rng=rng)
    rows = []
    for _ in range(20000):
        env.reset()
        rows.append([env.base_x[0], env.base_y[0], env.base_x[1], env.base_y[1]] +
                    [v for i in range(4) for v in (env.px[i], env.py[i], env.pdir[i])])
    d = np.asarray(rows)
    assert np.array_equal(d.min(0), z["lo"]) and np.array_equal(d.max(0), z["hi"])
    np.testing.assert_allclose(d.mean(0), z["mean"], rtol=0.02)
    # red directions: uniform over {270..359} U {0..90}; blue: uniform over 90..270 -- same support as the reference's
    assert np.array_equal(np.bincount(d[:, 6], minlength=361) > 0, z["red_dir_hist"] > 0)
    assert np.array_equal(np.bincount(d[:, 12], minlength=361) > 0, z["blue_dir_hist"] > 0)


def test_tie_tick_follows_float_accumulation():
    assert [ref.tie_tick(n) for n in (1, 2, 3, 4, 5, 8)] == [121, 141, 161, 181, 200, 260]


def test_spaces_and_ids():
    env = ref.RefEnv(n_agents=2)
    assert env.possible_agents == ["plane0", "plane1", "plane2", "plane3"]
    assert env.possible_red == ["plane0", "plane1"] and env.possible_blue == ["plane2", "plane3"]
    sp = env.observation_space("plane0")
    assert sp.shape == (8,) and sp.dtype == np.float32 and (sp.low == 1).all() and (sp.high == -1).all()
    assert env.action_space("plane3").n == 4 and env.n_actions == 4
    c = ref.RefEnv(n_agents=1, continuous_actions=True)
    assert c.n_actions == 3 and c.action_space("plane0").shape == (3,) and c.max_turn == 35


def test_instinct_oracle_reproduces_reference_agent():
    from oracle import instinct_ref as ir
    z = np.load(f"{GOLDEN}/g8_instinct_pairs.npz")
    for n in (1, 2, 4):
        obs, act = z[f"disc_{n}v{n}/obs"], z[f"disc_{n}v{n}/action"]
        assert [ir.discrete_action(o, n) for o in obs] == act.tolist()
        obs, act = z[f"cont_{n}v{n}/obs"], z[f"cont_{n}v{n}/action"]
        rnd, noise = z[f"cont_{n}v{n}/rand"], z[f"cont_{n}v{n}/noise"]
        got = np.stack([ir.continuous_action(o, n, r, nz) for o, r, nz in zip(obs, rnd, noise)])
        assert np.array_equal(got, act)
