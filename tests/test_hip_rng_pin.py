"""The PRODUCTION random path of reset() / auto-reset / the shot on the MI355X (in-kernel Philox4x32, no injected draws) against the
reference's own draws: fixture g7_spawn_stats.npz (the unmodified reference's `Plane.reset` / `Base.reset`, sprites.py:74-91,238-252,
20 000 resets at 2v2) and sprites.py:314 for the bullet jitter.  The parity traces inject the reference's draws; these tests say that
what the kernels draw when nothing is injected has the reference's ranges, support and uniformity (tests/rng_pin_util.py)."""
import numpy as np
import pytest
import torch

import deep_rl_battlespace_amd as bsx
from rng_pin_util import check_jitter, check_spawn_table, spawn_table

pytestmark = pytest.mark.gpu
E = 1 << 20                                                 # >= 1 M games, n = 2 as g7 was drawn


def _state(env, fields=("px", "py", "pdir", "base_xy")):
    return {k: v.cpu().numpy() for k, v in env.export_state(fields).items()}


def test_reset_spawns_have_the_reference_distribution():
    env = bsx.parallel_env(n_agents=2, n_envs=E, seed=20261004, device="cuda:0")
    env.reset()
    d = spawn_table(_state(env))
    check_spawn_table(d, "bsx_reset, stream RESET")
    # a second reset() of the same env draws NEW spawns (the nonce advances), with the same distribution
    env.reset()
    d2 = spawn_table(_state(env))
    assert (d2 != d).any(1).mean() > 0.999
    check_spawn_table(d2, "bsx_reset, second nonce")
    # another seed, another shard offset: other games; the same seed and offset: the same games
    other = bsx.parallel_env(n_agents=2, n_envs=E, seed=20261005, device="cuda:0")
    other.reset()
    assert (spawn_table(_state(other)) != d).any(1).mean() > 0.999
    del other
    shard = bsx.parallel_env(n_agents=2, n_envs=E // 4, seed=20261004, env_offset=E // 2, device="cuda:0")
    shard.reset()
    assert np.array_equal(spawn_table(_state(shard)), d[E // 2:E // 2 + E // 4])     # keyed by the GLOBAL game index


def test_auto_reset_spawns_have_the_reference_distribution():
    env = bsx.parallel_env(n_agents=2, n_envs=E, seed=77, auto_reset=True, device="cuda:0")
    env.reset()
    first = spawn_table(_state(env))
    tables = [first]
    for game in range(2):
        env.step({})                                        # step({}) ends every game as a tie (battle_env.py:309-311) ...
        assert bool(env.export_state(("env_done",))["env_done"].all())
        env.step({})                                        # ... and the next call re-spawns it in-kernel (stream AUTORESET, episode = games so far)
        st = _state(env, ("px", "py", "pdir", "base_xy", "env_done", "tick", "php", "bhp"))
        assert not st["env_done"].any() and (st["tick"] == 0).all() and (st["php"] == 4).all() and (st["bhp"] == 10).all()
        d = spawn_table(st)
        check_spawn_table(d, f"in-kernel auto-reset, game {game + 1}")
        assert all((d != t).any(1).mean() > 0.999 for t in tables)      # every episode of a game draws afresh
        tables.append(d)


@pytest.mark.parametrize("n,continuous", [(1, False), (2, False), (4, False), (1, True)])
def test_shot_jitter_has_the_reference_distribution(n, continuous):
    """Every plane fires on the first call after reset(): the new bullet's heading minus the shooter's pre-move heading is the
    reference's `random.random() * 8 - 4` -- every kernel family draws it the same way (1v1 table shot, sincos shot, continuous)."""
    En = (1 << 19) // n
    A = 2 * n
    env = bsx.parallel_env(n_agents=n, n_envs=En, seed=5, continuous_actions=continuous, device="cuda:0")
    env.reset()
    d0 = env.export_state(("pdir",))["pdir"].cpu().numpy().copy()
    if continuous:
        act = torch.zeros((En, A, 3), dtype=torch.float32, device="cuda:0"); act[..., 2] = 1.0     # no turn, shoot
    else:
        act = torch.ones((En, A), dtype=torch.int32, device="cuda:0")
    env.step_batch(act)
    st = _state(env, ("bl_live", "bl_dir"))
    live = st["bl_live"].astype(bool)
    assert (live.sum(-1) == 1).all()                        # one bullet per plane, none can have ended on its first update ... unless off the field
    bl = st["bl_dir"][live].reshape(En, A)
    check_jitter(bl.reshape(-1), d0.reshape(-1), f"{n}v{n} {'continuous' if continuous else 'discrete'}")
    # the same seed draws the same jitters again; another seed draws others
    for seed, same in ((5, True), (6, False)):
        e2 = bsx.parallel_env(n_agents=n, n_envs=En, seed=seed, continuous_actions=continuous, device="cuda:0")
        e2.reset(spawn=None)
        if seed == 5:
            assert np.array_equal(e2.export_state(("pdir",))["pdir"].cpu().numpy(), d0)
        e2.step_batch(act)
        s2 = _state(e2, ("bl_live", "bl_dir"))
        b2 = s2["bl_dir"][s2["bl_live"].astype(bool)].reshape(En, A)
        if same:
            assert np.array_equal(b2, bl)
        del e2
