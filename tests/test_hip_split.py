"""The two-wave 1v1 step kernels (csrc/bsx_step_split.h) against the one-wave kernels (BSX_F_ONE_WAVE / `one_wave=True`): the same step() of
envs/battle_env.py:281-381, so every output of every call and the complete game state must be IDENTICAL -- the kernels include the same
phase files, each wave with the side effects of its role.  The product runs the MULTI-TICK form (a game wave + an outputs wave per 64
agents) for bsx_step_many_discrete up to 65 536 games and the PER-CALL form (a wave for everything but the observation geometry + a
geometry wave fed the post-move poses) for bsx_step_discrete / _range up to 114 688 games.  These are EQUIVALENCE tests between two
kernels of this build; the parity tests of each kernel against the C oracle, at every size the launcher gives it, are in
tests/test_hip_fullsize.py (test_every_*_kernel_the_launcher_selects_vs_c_oracle, test_one_wave_1v1_kernels_kept_by_flag_vs_c_oracle)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _env(**kw):
    import deep_rl_battlespace_amd as bsx
    return bsx.parallel_env(**kw)


def _same_state(a, b):
    sa, sb = a.export_state(), b.export_state()
    for k in sa:
        if k in ("bl_x", "bl_y", "bl_dir"):                     # slots without a live bullet hold leftovers
            m = sa["bl_live"].bool()
            assert torch.equal(sa["bl_live"], sb["bl_live"]) and torch.equal(sa[k][m], sb[k][m]), k
        else:
            assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("E,enc,wide,auto", [(65536, "int", False, True), (1000, "scores", False, True), (31, "int", True, False),
                                             (4097, "scores", True, True), (114688, "int", False, True), (131072, "int", False, True)])
def test_per_call_two_wave_kernel_equals_the_one_wave_kernel(E, enc, wide, auto):
    """Random play with many shots (so that pools fill, planes die, bases fall, games end and -- auto -- re-spawn in place), masked resets by
    hand otherwise, an empty call in between: outputs equal on every call, state equal at the end and at a few calls in between.
    (131 072 games take the one-wave kernel either way: the case checks that the size switch changes nothing.)"""
    kw = dict(n_agents=1, n_envs=E, seed=99, auto_reset=auto, wide_offsets=wide)
    a, b = _env(**kw), _env(one_wave=True, **kw)
    oa, ob = a.reset(), b.reset()
    assert all(torch.equal(oa[k], ob[k]) for k in oa)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    T = 260 if E <= 65536 else 150
    for t in range(T):
        act = torch.randint(-1, 5, (E, 2), generator=g, device="cuda", dtype=torch.int32)       # incl. the out-of-range "do not move" values
        act = torch.where(torch.rand((E, 2), generator=g, device="cuda") < 0.55, torch.ones_like(act), act)
        if enc == "scores":
            sc = torch.rand((E, 2, 4), generator=g, device="cuda") * 0.5
            sc.scatter_(2, act.clamp(0, 3).long().unsqueeze(-1), 1.0)
            ra, rb = a.step_batch(sc), b.step_batch(sc)
        elif t == 97:
            ra = a.step({}); rb = b.step({})                                                     # battle_env.py:309: every running game ties
            ra = (torch.stack([ra[0][k] for k in a.possible_agents], 1), torch.stack([ra[1][k] for k in a.possible_agents], 1), torch.stack([ra[2][k] for k in a.possible_agents], 1))
            rb = (torch.stack([rb[0][k] for k in b.possible_agents], 1), torch.stack([rb[1][k] for k in b.possible_agents], 1), torch.stack([rb[2][k] for k in b.possible_agents], 1))
        else:
            ra, rb = a.step_batch(act), b.step_batch(act)
        for u, v, name in zip(ra, rb, ("obs", "rew", "done")):
            assert torch.equal(u, v), (t, name)
        assert torch.equal(a.env_done, b.env_done) and torch.equal(a.winner, b.winner), t
        if not auto and t % 40 == 39:
            m = a.env_done.clone()
            a.reset(mask=m); b.reset(mask=m)
        if t in (60, 130, T - 1):
            _same_state(a, b)
    c = a.counters()
    assert (c == b.counters()).all() and int(c[:, 0].sum()) > 0 and int(c[:, 2:].sum()) > 0        # games ended, some by a base kill


@pytest.mark.parametrize("E,dtype,wide,auto", [(65536, torch.float32, False, True), (81920, torch.float32, False, True), (1000, torch.float64, False, True), (31, torch.float32, True, False),
                                              (4097, torch.float64, True, True), (131072, torch.float32, False, True)])
def test_per_call_two_wave_kernel_with_continuous_actions_equals_the_one_wave_kernel(E, dtype, wide, auto):
    """bsx_step_continuous takes the two-wave form too (up to 81 920 games): only the first wave's loads differ -- the action triple in its
    encodings (float32 / float64 [.,3] here; float32 rows of four come from the policy rollout's graph form, whose tests run against the C
    oracle) and the float64 heading beside the plane record; the geometry wave is the discrete kernel's.  Fractional headings, sincos moves and shots, clipped actions, masked resets: outputs and state identical."""
    kw = dict(n_agents=1, n_envs=E, seed=17, auto_reset=auto, wide_offsets=wide, continuous_actions=True)
    a, b = _env(**kw), _env(one_wave=True, **kw)
    oa, ob = a.reset(), b.reset()
    assert all(torch.equal(oa[k], ob[k]) for k in oa)
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    T = 200 if E <= 65536 else 100
    for t in range(T):
        act = (torch.rand((E, 2, 3), generator=g, device="cuda", dtype=torch.float32) * 2.6 - 1.3).to(dtype)   # beyond [-1, 1]: clipped in-kernel
        act[..., 2] = torch.where(torch.rand((E, 2), generator=g, device="cuda") < 0.5, torch.ones((), device="cuda", dtype=dtype), act[..., 2])
        ra, rb = a.step_batch(act), b.step_batch(act)
        for u, v, name in zip(ra, rb, ("obs", "rew", "done")):
            assert torch.equal(u, v), (t, name)
        assert torch.equal(a.env_done, b.env_done) and torch.equal(a.winner, b.winner), t
        if not auto and t % 40 == 39:
            m = a.env_done.clone()
            a.reset(mask=m); b.reset(mask=m)
        if t in (50, T - 1):
            _same_state(a, b)
    assert int(a.counters()[:, 0].sum()) > 0


def test_per_call_two_wave_kernel_with_injected_jitter_and_as_range_launches():
    """Host-drawn random() values for the shots (the parity traces' form) and the batch as two chains of range launches in one graph
    (bsx_step_discrete_range): the launcher takes the split kernel for each range; same games as the one-wave kernel per call."""
    E, T = 8192, 64
    kw = dict(n_agents=1, n_envs=E, seed=3, auto_reset=True)
    a, b = _env(**kw), _env(one_wave=True, **kw)
    a.reset(); b.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(8)
    acts = torch.where(torch.rand((T, E, 2), generator=g, device="cuda") < 0.5, 1, torch.randint(0, 4, (T, E, 2), generator=g, device="cuda")).to(torch.int32)
    u = torch.rand((T, E, 2), generator=g, device="cuda", dtype=torch.float64)
    for t in range(T // 2):
        ra, rb = a.step_batch(acts[t], u=u[t]), b.step_batch(acts[t], u=u[t])
        assert all(torch.equal(x, y) for x, y in zip(ra, rb)), t
    _same_state(a, b)
    ga, _ = a.capture_steps(acts[T // 2:], chains=2)
    ga.replay()
    for t in range(T // 2, T):
        b.step_batch(acts[t])
    torch.cuda.synchronize()
    _same_state(a, b)


def test_per_call_two_wave_kernel_runs_the_drop_in_game_too():
    """One game behind the reference's surface (n_envs=None): the launch is one workgroup of two waves; same game as the one-wave kernel."""
    import random
    outs = []
    for one in (False, True):
        random.seed(77)
        env = _env(n_agents=1, one_wave=one)
        rng = np.random.default_rng(4)
        obs = env.reset()
        log = [np.concatenate([obs[k] for k in env.possible_agents])]
        for t in range(400):
            if env.env_done:
                obs = env.reset()
            o, r, d, _ = env.step({k: int(rng.integers(0, 4)) if rng.random() < 0.6 else 1 for k in env.possible_agents})
            log.append(np.concatenate([o[k] for k in env.possible_agents] + [[float(r[k]) for k in env.possible_agents], [float(d[k]) for k in env.possible_agents]]))
        outs.append((np.concatenate(log), env.total_games, env.winner))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:]


@pytest.mark.parametrize("E,enc,auto,inject", [(65536, "int", True, False), (3000, "scores", True, False), (33, "int", False, True), (131072, "int", True, False),
                                               (32768, "int", True, False), (40000, "scores", False, True), (49152, "int", True, True)])
def test_two_wave_multi_tick_kernel_equals_the_one_wave_multi_tick_kernel(E, enc, auto, inject):
    """The PRODUCT's multi-tick 1v1 launches of up to 65 536 games (bsx_step_many_discrete; csrc/bsx_step_split.h, MANY): a GAME wave (the
    whole state machine, up to a tick ahead) and an OUTPUTS wave (geometry, rewards, rows, flags) per 64 agents, against the one-wave
    multi-tick kernel (BSX_F_ONE_WAVE): every tick's outputs, env_done per tick and the complete state identical.  Up to 32 768 games the
    outputs wave carries the state too and repeats classify, move and outcome (form 1); above, it takes 16 bytes per agent and tick from the
    game wave and repeats nothing (form 2): both are here.  (131 072 games take the one-wave kernel either way: the case checks that the
    size switch changes nothing.)"""
    kw = dict(n_agents=1, n_envs=E, seed=41, auto_reset=auto)
    a, b = _env(**kw), _env(one_wave=True, **kw)
    a.reset(); b.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(6)
    T = 50
    for rep in range(6 if E <= 65536 else 3):
        act = torch.where(torch.rand((T, E, 2), generator=g, device="cuda") < 0.5, 1, torch.randint(-1, 5, (T, E, 2), generator=g, device="cuda")).to(torch.int32)
        u = torch.rand((T, E, 2), generator=g, device="cuda", dtype=torch.float64) if inject else None
        if enc == "scores":
            sc = torch.rand((T, E, 2, 4), generator=g, device="cuda") * 0.5
            sc.scatter_(3, act.clamp(0, 3).long().unsqueeze(-1), 1.0)
            act = sc
        eda, edb = torch.zeros((T, E), dtype=torch.uint8, device="cuda"), torch.zeros((T, E), dtype=torch.uint8, device="cuda")
        ra = a.step_many(act, store=True, u=u, env_done_out=eda)
        rb = b.step_many(act, store=True, u=u, env_done_out=edb)
        for x, y, name in zip(ra, rb, ("obs", "rew", "done")):
            assert torch.equal(x, y), (rep, name)
        assert torch.equal(eda, edb), rep
        _same_state(a, b)
        if not auto:
            m = a.env_done.clone()
            a.reset(mask=m); b.reset(mask=m)
    # and a launch whose results are the env-owned last-tick tensors (store=False)
    act = torch.randint(0, 4, (7, E, 2), generator=g, device="cuda", dtype=torch.int32)
    ra, rb = a.step_many(act), b.step_many(act)
    assert all(torch.equal(x, y) for x, y in zip(ra, rb))
    _same_state(a, b)
    assert int(a.counters()[:, 0].sum()) > 0
