"""Property tests of the CPU oracle on random play (hypothesis): invariants that hold for every reachable state.
They document what the HIP path is also checked for at full size (tests/test_hip_fullsize.py)."""
import random

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import battlespace_ref as ref


@settings(max_examples=25, deadline=None)
@given(n=st.integers(1, 3), seed=st.integers(0, 10_000), p_shoot=st.floats(0.0, 1.0), cont=st.booleans())
def test_oracle_state_invariants(n, seed, p_shoot, cont):
    rng = random.Random(seed)
    env = ref.RefEnv(n_agents=n, continuous_actions=cont, rng=rng)
    env.reset()
    ids = env.possible_agents
    total = {a: 0.0 for a in ids}
    for t in range(260):
        if env.env_done:
            assert env.winner in ("red", "blue", "tie") and all(env.dones.values())
            before = env.snapshot()
            obs, rew, done, _ = env.step({a: 0 for a in ids} if not cont else {a: np.zeros(3) for a in ids})
            after = env.snapshot()
            assert all(np.array_equal(before[k], after[k]) for k in ("px", "py", "php", "bhp", "bl_live")) and not any(rew.values())
            env.reset()
            assert env.tick == 0 and not env.bullets and all(h == 4 for h in env.php)
            continue
        if cont:
            acts = {a: np.asarray([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), 1.0 if rng.random() < p_shoot else -1.0]) for a in ids}
        else:
            acts = {a: (1 if rng.random() < p_shoot else rng.randint(-1, 4)) for a in ids}
        obs, rew, done, _ = env.step(acts)
        for i, a in enumerate(ids):
            assert 25 <= env.px[i] <= 1175 and 24 <= env.py[i] <= 776 and 0 <= env.pdir[i] <= 360
            assert 0 <= env.php[i] <= 4 and env.palive[i] == (env.php[i] > 0)
            o = obs[a]
            assert o.dtype == np.float32 and o.shape == (3 * n + 2,)
            if env.palive[i]:
                assert -1 <= o[0] <= 1 and -0.5 <= o[1] <= 0.5
            else:
                assert (o == -1).all() and (done[a] or env.env_done)
            total[a] += rew[a]
        assert len(env.bullets) <= 11 * 2 * n and all(1 <= b[4] <= 11 for b in env.bullets)
        assert all(0 <= b[1] <= 1200 and 0 <= b[2] <= 800 for b in env.bullets)
        assert env.tick <= ref.tie_tick(n)
        assert env.total_games == env.ties + env.team["red"]["wins"] + env.team["blue"]["wins"]
