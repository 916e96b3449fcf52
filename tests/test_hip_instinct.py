"""The on-device scripted opponent (bsx_instinct_*, instinct.Team) against the reference agent's own outputs (fixture
g8) and, closed-loop, against the CPU oracles.  Needs the MI355X."""
import numpy as np
import pytest
import torch

from oracle import cref
from oracle import instinct_ref as ir
from trace_util import GOLDEN

pytestmark = pytest.mark.gpu


def _env(**kw):
    import deep_rl_battlespace_amd as bsx
    return bsx.parallel_env(**kw)


@pytest.mark.parametrize("n", [1, 2, 4])
def test_discrete_instinct_reproduces_reference_agent(n):
    from deep_rl_battlespace_amd import instinct
    z = np.load(f"{GOLDEN}/g8_instinct_pairs.npz")
    obs, agent, act = z[f"disc_{n}v{n}/obs"], z[f"disc_{n}v{n}/agent"], z[f"disc_{n}v{n}/action"]
    M, A, D = len(obs), 2 * n, 3 * n + 2
    env = _env(n_agents=n, n_envs=M)
    full = torch.zeros((M, A, D), dtype=torch.float32, device="cuda")
    full[torch.arange(M), torch.as_tensor(agent, dtype=torch.long)] = torch.as_tensor(obs).cuda()
    red = instinct.Team(env.possible_red, env.possible_blue, env)
    blue = instinct.Team(env.possible_blue, env.possible_red, env)
    out = torch.full((M, A), -7, dtype=torch.int32, device="cuda")
    red.write_actions(out=out, obs=full)
    assert bool((out[:, n:] == -7).all())                       # only the red rows were written
    blue.write_actions(out=out, obs=full)
    got = out.cpu().numpy()[np.arange(M), agent]
    assert np.array_equal(got, act)
    onehot = torch.zeros((M, A, 4), dtype=torch.float32, device="cuda")
    red.write_actions(out=onehot, obs=full); blue.write_actions(out=onehot, obs=full)
    assert np.array_equal(onehot.argmax(-1).cpu().numpy()[np.arange(M), agent], act)
    assert bool(((onehot == 1).sum(-1) == 1).all()) and bool(((onehot == -1).sum(-1) == 3).all())


@pytest.mark.parametrize("n", [1, 2, 4])
def test_continuous_instinct_reproduces_reference_agent(n):
    from deep_rl_battlespace_amd import instinct
    z = np.load(f"{GOLDEN}/g8_instinct_pairs.npz")
    t = f"cont_{n}v{n}"
    obs, agent, act, rnd, noise = z[t + "/obs"], z[t + "/agent"], z[t + "/action"], z[t + "/rand"], z[t + "/noise"]
    M, A, D = len(obs), 2 * n, 3 * n + 2
    env = _env(n_agents=n, n_envs=M, continuous_actions=True)
    full = torch.zeros((M, A, D), dtype=torch.float32, device="cuda")
    full[torch.arange(M), torch.as_tensor(agent, dtype=torch.long)] = torch.as_tensor(obs).cuda()
    r = np.zeros((M, A, 4))
    r[np.arange(M), agent, 0] = np.nan_to_num(rnd, nan=0.99)
    r[np.arange(M), agent, 1:] = noise
    both = [instinct.Team(env.possible_red, env.possible_blue, env), instinct.Team(env.possible_blue, env.possible_red, env)]
    out = torch.zeros((M, A, 3), dtype=torch.float64, device="cuda")
    for tm in both:
        tm.write_actions(out=out, obs=full, rnd=r)
    got = out.cpu().numpy()[np.arange(M), agent]
    assert np.array_equal(got, act)                              # binary64, same operation order: bit-exact
    # production draws: in range, fresh per call
    a1 = both[0].write_actions(obs=full).clone(); a2 = both[0].write_actions(obs=full).clone()
    assert float(a1.abs().max()) <= 1.0 and not torch.equal(a1, a2)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 6])
def test_closed_loop_instinct_vs_instinct_matches_cpu_oracles(n):
    """Both teams scripted, env and opponent on device, against C-oracle env + Python-oracle opponent: same games."""
    from deep_rl_battlespace_amd import instinct
    E, A, T = (96 if n <= 2 else 40), 2 * n, 10 * (10 + 2 * n) + 30
    env = _env(n_agents=n, n_envs=E, seed=17, auto_reset=True)
    c = cref.CRefBatch(E, n_agents=n, seed=17, auto_reset=True)
    obs_h = env.reset(); obs_c = c.reset().copy()
    red = instinct.Team(env.possible_red, env.possible_blue, env)
    blue = instinct.Team(env.possible_blue, env.possible_red, env)
    acts = torch.zeros((E, A), dtype=torch.int32, device="cuda")
    for t in range(T):
        red.write_actions(out=acts); blue.write_actions(out=acts)
        want = np.asarray([[ir.discrete_action(obs_c[e, a], n) for a in range(A)] for e in range(E)], np.int32)
        assert np.array_equal(acts.cpu().numpy(), want), f"tick {t}"
        o, r, d = env.step_batch(acts)
        co, cr, cd = c.step(want)
        np.testing.assert_allclose(o.cpu().numpy(), co, rtol=1e-5, atol=1e-7)
        assert np.array_equal(d.cpu().numpy(), cd) and np.array_equal(r.cpu().numpy().astype(np.float64), cr)
        obs_c = co.copy()
    cnt = env.counters().sum(0)
    assert cnt[2] + cnt[3] > 0.5 * cnt[0] > 0                    # scripted play is decisive: most games end in a win


def test_dropin_team_surface():
    """Reference call pattern (main.py:119-122,179): Team(agent_list, enemy_list, env).choose_actions(obs_dict) -> dict."""
    import random
    from deep_rl_battlespace_amd import instinct
    random.seed(4)
    env = _env(n_agents=2)
    obs = env.reset()
    blue = instinct.Team(env.possible_blue, env.possible_red, env)
    red = instinct.Team(env.possible_red, env.possible_blue, env)
    steps = 0
    while not env.env_done and steps < 400:
        a = dict(blue.choose_actions({k: obs[k] for k in env.possible_blue}))
        a.update(red.choose_actions({k: obs[k] for k in env.possible_red}))
        assert set(a) == set(env.possible_agents) and all(isinstance(v, int) and v in (1, 2, 3) for v in a.values())
        assert [a[k] for k in env.possible_agents] == [ir.discrete_action(obs[k], 2) for k in env.possible_agents]
        obs, rew, done, _ = env.step(a)
        steps += 1
    assert env.env_done and env.winner in ("red", "blue", "tie")


def test_rollout_with_learned_red_and_scripted_blue():
    """The reference's training setup on device (main.py:119-122,179): red = actor, blue = instinct.Team; one HIP graph."""
    from deep_rl_battlespace_amd import instinct
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    E, n, T = 1024, 2, 20
    env = _env(n_agents=n, n_envs=E, seed=9, auto_reset=True); env.reset()
    torch.manual_seed(0)
    actor = StackedActor(2 * n, 3 * n + 2, 4, device="cuda")
    blue = instinct.Team(env.possible_blue, env.possible_red, env)
    ro = PolicyRollout(env, actor, T, noise_std=0.3, opponent=blue)
    ro.start(); ro.capture()
    for rep in range(6):
        ro.run()
    torch.cuda.synchronize()
    blue_scores = ro.scores[:, :, n:]
    assert bool(((blue_scores == 1).sum(-1) == 1).all()) and bool(((blue_scores == -1).sum(-1) == 3).all())   # one-hot rows
    want = np.asarray([[ir.discrete_action(o, n) for o in row] for row in ro.obs[3, :64, n:].cpu().numpy()])
    assert np.array_equal(blue_scores[3, :64].argmax(-1).cpu().numpy(), want)
    assert not bool(((ro.scores[:, :, :n].abs() == 1).all(-1)).all())                                             # red rows are the actor's
    c = env.counters().sum(0)
    assert c[3] > c[2] and c[0] > 0                               # the scripted team beats a random-weight actor


@pytest.mark.parametrize("n", [1, 2, 3, 4])
@pytest.mark.parametrize("scripted", ["blue", "red"])
def test_one_launch_rollout_plays_the_scripted_opponent_in_kernel(scripted, n):
    """bsx_rollout_discrete with scripted_team: the reference's training setup (main.py:119-122: learned team vs
    instinct.Team) as ONE launch -- the scripted side's rows are decided in-kernel from the observation rows in LDS and
    its actor is skipped.  Same transitions, bit for bit, as the two-kernel rollout with `opponent.write_actions`."""
    from deep_rl_battlespace_amd import instinct
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    E, T = (4100 if n == 1 else 1030), 40
    torch.manual_seed(1)
    actor = StackedActor(2 * n, 3 * n + 2, 4, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(60.0)
    ros = []
    for one in (False, True):
        env = _env(n_agents=n, n_envs=E, seed=17, auto_reset=True); env.reset()
        opp = (instinct.Team(env.possible_blue, env.possible_red, env) if scripted == "blue"
               else instinct.Team(env.possible_red, env.possible_blue, env))
        ro = PolicyRollout(env, actor, T, noise_std=0.2, seed=5, opponent=opp, one_launch=one); ro.start(); ro.capture()
        ros.append(ro)
    a, b = ros
    for rep in range(5):
        a.run(); b.run()
        torch.cuda.synchronize()
        assert torch.equal(a.obs, b.obs) and torch.equal(a.scores, b.scores), rep
        assert torch.equal(a.rew, b.rew) and torch.equal(a.done, b.done), rep
    cols = slice(n, 2 * n) if scripted == "blue" else slice(0, n)
    assert bool(((b.scores[:, :, cols] == 1).sum(-1) == 1).all()) and bool(((b.scores[:, :, cols].abs() == 1).all()))
    sa, sb = a.env.export_state(), b.env.export_state()
    for k in ("px", "py", "pdir", "php", "bhp", "tick", "env_done", "winner", "bl_live", "counters"):
        assert torch.equal(sa[k], sb[k]), k
    c = b.env.counters().sum(0)
    assert c[0] > 0 and (c[3] > c[2] if scripted == "blue" else c[2] > c[3])    # the scripted side wins more often
