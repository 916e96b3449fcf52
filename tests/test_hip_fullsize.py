"""Full-size checks of the HIP step() path on the MI355X (BASELINE.json configs[1] and [2] sizes): bit-exact against
the C oracle in the PRODUCTION configuration (auto-reset, in-kernel Philox jitter and spawns), plus size-independent
properties: determinism, shard invariance, state invariants, counter conservation."""
import numpy as np
import pytest
import torch

from oracle import cref
from trace_util import OBS_ATOL, OBS_RTOL

pytestmark = pytest.mark.gpu


def _env(**kw):
    import deep_rl_battlespace_amd as bsx
    return bsx.parallel_env(**kw)


def _actions(T, E, A, seed, p_shoot=0.25, device="cuda"):
    g = torch.Generator(device=device); g.manual_seed(seed)
    a = torch.randint(0, 4, (T, E, A), generator=g, device=device, dtype=torch.int32)
    if p_shoot != 0.25:
        m = torch.rand((T, E, A), generator=g, device=device) < p_shoot
        a = torch.where(m, torch.ones_like(a), a)
    return a


def _compare_with_c_oracle(E, n, T, seed, p_shoot, check_every, env_offset=0):
    A = 2 * n
    env = _env(n_agents=n, n_envs=E, seed=seed, auto_reset=True, env_offset=env_offset)
    c = cref.CRefBatch(E, n_agents=n, seed=seed, auto_reset=True, env_offset=env_offset)
    o_h = env.reset()
    o_c = c.reset()
    np.testing.assert_allclose(torch.stack([o_h[a] for a in env.possible_agents], 1).cpu().numpy(), o_c, rtol=OBS_RTOL, atol=OBS_ATOL)
    acts = _actions(T, E, A, seed + 1, p_shoot)
    acts_h = acts.cpu().numpy()
    n_exact = n_vals = 0
    for t in range(T):
        obs, rew, done = env.step_batch(acts[t])
        co, cr, cd = c.step(acts_h[t])
        o = obs.cpu().numpy()
        assert np.array_equal(done.cpu().numpy(), cd), f"step {t}: done"
        assert np.array_equal(rew.cpu().numpy().astype(np.float64), cr), f"step {t}: rew"
        assert np.array_equal(env.env_done.cpu().numpy(), c.env_done.astype(bool)) and np.array_equal(env.winner.cpu().numpy(), c.winner)
        np.testing.assert_allclose(o, co, rtol=OBS_RTOL, atol=OBS_ATOL, err_msg=f"step {t}: obs")
        n_exact += int((o == co).sum()); n_vals += o.size
        if t % check_every == check_every - 1 or t == T - 1:
            sh = {k: v.cpu().numpy() for k, v in env.export_state().items()}
            sc = c.export_state()
            for f in ("px", "py", "pdir", "php", "palive", "base_xy", "bhp", "tick", "env_done", "winner", "bl_live", "counters"):
                assert np.array_equal(sh[f], sc[f]), f"step {t}: {f}"
            m = sc["bl_live"].astype(bool)
            for f in ("bl_x", "bl_y", "bl_dir"):
                assert np.array_equal(sh[f][m], sc[f][m]), f"step {t}: {f}"
    assert n_exact >= n_vals * (1 - 1e-3), f"only {n_exact}/{n_vals} observation values bit-identical"
    return env


def test_C2_65536x1v1_bit_exact_vs_c_oracle():
    """configs[1]: 65 536 games of 1v1, uniform random actions, 260 calls (two full games + auto-resets)."""
    env = _compare_with_c_oracle(65536, 1, 260, seed=1234, p_shoot=0.25, check_every=65)
    c = env.counters().sum(0)
    assert c[0] == c[1] + c[2] + c[3] and c[0] >= 2 * 65536      # every finished game is a tie or a win


def test_C3_65536x4v4_bit_exact_vs_c_oracle():
    """configs[2]: 65 536 games of 4v4 (8 agents, all-pairs staged in LDS), 200 calls across the 181-call tie."""
    _compare_with_c_oracle(65536, 4, 200, seed=99, p_shoot=0.25, check_every=100)


def _compare_many_with_c_oracle(E, chunks, seed, p_shoot, inject, one_wave=False, scores=False):
    """bsx_step_many_discrete against the C oracle: each chunk of T ticks is ONE launch (store=True: every tick's rows, rewards, flags and
    env_done kept), the oracle walks the same action table a call at a time; every tick's outputs and the state after every launch."""
    env = _env(n_agents=1, n_envs=E, seed=seed, auto_reset=True, one_wave=one_wave)
    c = cref.CRefBatch(E, n_agents=1, seed=seed, auto_reset=True)
    env.reset(); c.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(seed + 5)
    n_exact = n_vals = 0
    for k, T in enumerate(chunks):
        acts = _actions(T, E, 2, seed + 10 + k, p_shoot)
        if k == 0:
            acts[T // 2, : E // 3] = 7                          # out-of-range indices: the plane does not move (battle_env.py:399-417)
        u = torch.rand((T, E, 2), generator=g, device="cuda", dtype=torch.float64) if inject else None
        arg = acts
        if scores:                                              # [T, E, A, 4] score rows arg-maxed in-kernel (battle_env.py:327-328)
            arg = torch.rand((T, E, 2, 4), generator=g, device="cuda") * 0.5
            arg.scatter_(3, acts.clamp(0, 3).long().unsqueeze(-1), 1.0)
            acts = acts.clamp(0, 3)
        ed = torch.zeros((T, E), dtype=torch.uint8, device="cuda")
        mo, mr, md = env.step_many(arg, store=True, u=u, env_done_out=ed)
        mo, mr, md, ed = mo.cpu().numpy(), mr.cpu().numpy(), md.cpu().numpy(), ed.cpu().numpy()
        acts_h = acts.cpu().numpy()
        u_h = u.cpu().numpy() if inject else None
        for t in range(T):
            co, cr, cd = c.step(acts_h[t], u=u_h[t] if inject else None)
            assert np.array_equal(md[t], cd), f"launch {k} tick {t}: done"
            assert np.array_equal(mr[t].astype(np.float64), cr), f"launch {k} tick {t}: rew"
            assert np.array_equal(ed[t], c.env_done), f"launch {k} tick {t}: env_done"
            np.testing.assert_allclose(mo[t], co, rtol=OBS_RTOL, atol=OBS_ATOL, err_msg=f"launch {k} tick {t}: obs")
            n_exact += int((mo[t] == co).sum()); n_vals += co.size
        _same_state_as_c_oracle(env, c, f"launch {k}")
    assert n_exact >= n_vals * (1 - 1e-3), f"only {n_exact}/{n_vals} observation values bit-identical"
    assert int(env.counters()[:, 0].sum()) >= E                 # the launches crossed the 121-call tie and the re-spawns
    return env


def _same_state_as_c_oracle(env, c, where):
    sh = {k: v.cpu().numpy() for k, v in env.export_state().items()}
    sc = c.export_state()
    for f in ("px", "py", "pdir", "php", "palive", "base_xy", "bhp", "tick", "env_done", "winner", "bl_live", "counters"):
        assert np.array_equal(sh[f], sc[f]), f"{where}: {f}"
    m = sc["bl_live"].astype(bool)
    for f in ("bl_x", "bl_y", "bl_dir"):
        assert np.array_equal(sh[f][m], sc[f][m]), f"{where}: {f}"


# The launcher picks a 1v1 kernel BY SIZE (csrc/bsx_kernels.hip, split_applies / launch_for_n): each of them restates
# envs/battle_env.py:281-381, so each is run against the C oracle here, at the sizes where the launcher takes it.
@pytest.mark.parametrize("E,inject,scores,kernel", [
    (65536, False, False, "two-wave multi-tick kernel whose outputs wave takes what the game wave publishes (32 768 < games <= 65 536)"),
    (65536, True, False, "the same with the shots' random() values injected"),
    (49152, True, True, "the same, score rows, a size that is not a power of two"),
    (32768, False, False, "two-wave multi-tick kernel whose outputs wave carries the state too (games <= 32 768), at its largest size"),
    (81920, False, False, "one-wave multi-tick kernel (games > 65 536)"),
    (81920, True, True, "the same, score rows and injected random() values"),
], ids=["65536-form2", "65536-form2-injected", "49152-form2-scores-injected", "32768-form1", "81920-one-wave", "81920-one-wave-scores-injected"])
def test_every_multi_tick_1v1_kernel_the_launcher_selects_vs_c_oracle(E, inject, scores, kernel):
    """step_many at 1v1: two launches of 130 ticks (across the 121-call tie, the re-spawns and the second game's first shots), shots every
    other call so that pools fill and planes die; per tick obs / rew / done / env_done, after each launch the whole state."""
    _compare_many_with_c_oracle(E, (130, 130), seed=700 + E % 1000 + int(inject), p_shoot=0.5, inject=inject, scores=scores)


@pytest.mark.parametrize("E,kernel", [
    (98304, "two-wave per-call kernel whose geometry wave also computes the call's Philox block, at the last size it takes"),
    (114688, "two-wave per-call kernel without that (98 304 < games <= 114 688), at the last size it takes"),
    (131072, "one-wave per-call kernel (games > 114 688): the kernel of the 262 144- and 1 M-game bench rows"),
], ids=["98304-two-wave-draw", "114688-two-wave", "131072-one-wave"])
def test_every_per_call_1v1_kernel_the_launcher_selects_vs_c_oracle(E, kernel):
    """step() per launch at 1v1 on either side of the size switch: 130 calls, across the tie and the re-spawns."""
    _compare_with_c_oracle(E, 1, 130, seed=811 + E % 1000, p_shoot=0.4, check_every=65)


def test_continuous_two_wave_kernel_at_its_last_size_and_the_one_wave_kernel_above_it_vs_c_oracle():
    """Continuous actions take the two-wave per-call form up to 81 920 games, the one-wave kernel above: both sizes against the C oracle."""
    _compare_generic(81920, 1, 125, seed=902, cont=True, f32=True)
    _compare_generic(98304, 1, 125, seed=903, cont=True, f32=False)


@pytest.mark.parametrize("cont", [False, True])
def test_one_wave_1v1_kernels_kept_by_flag_vs_c_oracle(cont):
    """BSX_F_ONE_WAVE (`one_wave=True`): the one-wave 1v1 kernels at a size where the launcher would take the two-wave forms -- per call
    (discrete / continuous) and as a multi-tick launch -- against the C oracle, not only against the two-wave kernels."""
    if cont:
        _compare_generic(8192, 1, 130, seed=31, cont=True, f32=True, one_wave=True)
    else:
        _compare_generic(8192, 1, 130, seed=32, one_wave=True)
        _compare_many_with_c_oracle(8192, (130, 40), seed=33, p_shoot=0.5, inject=False, one_wave=True)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 6])
def test_shoot_heavy_play_bit_exact_vs_c_oracle(n):
    """Every ring slot in use, same-step multi-hits and kill chains, wins by base kill; incl. the generic-n kernel (6)."""
    _compare_with_c_oracle(4096, n, 190 + 20 * n, seed=5 + n, p_shoot=0.8, check_every=50)


@pytest.mark.parametrize("n,G", [(1, 2), (3, 8), (4, 8), (8, 16)])
def test_bullet_pools_run_over_several_rounds_and_stay_exact(n, G):
    """Layout v2 keeps a wavefront's bullets as ONE pool (64 / G games): under keep-shooting play a pool holds several times the 64
    entries of a round, entries move at every compaction, games end and re-spawn next to running ones (their entries are dropped,
    the neighbours' stay).  The state must equal the C oracle's all the same -- and the test says that it really got there: the
    fullest pool of the final state spans at least three rounds (five at 1v1)."""
    E = 2048
    _compare_with_c_oracle(E, n, 150, seed=40 + n, p_shoot=0.9, check_every=25)              # through deaths, game ends and re-spawns
    env = _compare_with_c_oracle(E, n, 25, seed=60 + n, p_shoot=0.9, check_every=5)          # ... and the moment the pools are fullest
    live = env.export_state(("bl_live",))["bl_live"].cpu().numpy().astype(np.int64)          # [E, A, 12]
    epb = 64 // G
    per_pool = live.reshape(E // epb, -1).sum(1)
    assert per_pool.max() > (256 if n == 1 else 128), per_pool.max()
    assert per_pool.max() <= 64 * 12


def test_deterministic_and_shard_invariant():
    """Same seed -> same games; and a job split into shards (env_offset) plays the same games as one batch:
    the in-kernel generator is keyed by the GLOBAL env index (what makes the 8-GPU layout a pure partition)."""
    E, n, T = 8192, 2, 170
    A = 2 * n
    acts = _actions(T, E, A, 77, p_shoot=0.6)
    outs = []
    for parts in (1, 1, 4):
        q = E // parts
        envs = [_env(n_agents=n, n_envs=q, seed=42, auto_reset=True, env_offset=i * q) for i in range(parts)]
        for v in envs:
            v.reset()
        trace = []
        for t in range(T):
            o = [v.step_batch(acts[t, i * q:(i + 1) * q].contiguous()) for i, v in enumerate(envs)]
            trace.append(tuple(torch.cat([x[k] for x in o]).clone() for k in range(3)))
        outs.append((trace, torch.cat([v.export_state(("px", "py", "bl_live", "counters"))["px"] for v in envs])))
    for other in outs[1:]:
        for (o0, r0, d0), (o1, r1, d1) in zip(outs[0][0], other[0]):
            assert torch.equal(o0, o1) and torch.equal(r0, r1) and torch.equal(d0, d1)
        assert torch.equal(outs[0][1], other[1])


def test_C4_524288_games_as_eight_shards_equal_one_batch():
    """BASELINE.json configs[3] at its real size, on one card: 524 288 games of 1v1 as ONE batch and as the eight contiguous shards
    of 65 536 the 8-GPU layout gives its ranks (sharding.make_shard: env_offset = first global index), stepped with the same global
    action table, end in the same state bit for bit -- spawns, bullet jitter and auto-resets are keyed by the global game index."""
    from deep_rl_battlespace_amd import sharding
    E, n, T, W = 524288, 1, 130, 8
    acts = _actions(T, E, 2, 4321)
    whole = _env(n_agents=n, n_envs=E, seed=77, auto_reset=True); whole.reset()
    shards = [sharding.make_shard(E, r, W, n_agents=n, seed=77, auto_reset=True) for r in range(W)]
    assert [s.env_offset for s in shards] == [r * 65536 for r in range(W)] and all(s.n_envs == 65536 for s in shards)
    for s in shards:
        s.reset()
    for t in range(T):
        ow, rw, dw = whole.step_batch(acts[t])
        for r, s in enumerate(shards):
            o, rr, d = s.step_batch(acts[t, r * 65536:(r + 1) * 65536].contiguous())
            if t % 43 == 0 or t == T - 1:
                sl = slice(r * 65536, (r + 1) * 65536)
                assert torch.equal(o, ow[sl]) and torch.equal(rr, rw[sl]) and torch.equal(d, dw[sl]), (t, r)
    sw = whole.export_state()
    for r, s in enumerate(shards):
        ss = s.export_state(); sl = slice(r * 65536, (r + 1) * 65536)
        for f in ("px", "py", "pdir", "php", "bhp", "tick", "env_done", "winner", "bl_live", "bl_x", "bl_y", "bl_dir", "counters"):
            assert torch.equal(ss[f], sw[f][sl]), (r, f)
    assert int(sw["counters"][:, 0].sum()) >= E                  # every game crossed its first game end


@pytest.mark.parametrize("n,cont", [(1, False), (4, False), (2, True), (6, False)])
def test_wide_and_narrow_offset_kernels_play_the_same_games(n, cont):
    """A job whose arrays stay below 4 GB runs kernels with 32-bit row / byte offsets (SGPR base + 32-bit VGPR offset), larger jobs
    the 64-bit ones; BSX_F_WIDE_OFFSETS (parallel_env(wide_offsets=True)) takes the 64-bit kernels at any size.  Both must leave the
    same state and the same outputs, per step, as one launch per step and as multi-tick launches."""
    E, T = 3000, 150
    A = 2 * n
    kw = dict(n_agents=n, n_envs=E, seed=123, auto_reset=True, continuous_actions=cont)
    a, b = _env(**kw), _env(wide_offsets=True, **kw)
    a.reset(); b.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    if cont:
        acts = torch.rand((T, E, A, 3), generator=g, device="cuda") * 2 - 1
    else:
        acts = torch.randint(0, 4, (T, E, A), generator=g, device="cuda", dtype=torch.int32)
        acts = torch.where(torch.rand((T, E, A), generator=g, device="cuda") < 0.4, torch.ones_like(acts), acts)
    for t in range(T // 2):
        oa, ra, da = a.step_batch(acts[t]); ob, rb, db = b.step_batch(acts[t])
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), t
    oa, ra, da = a.step_many(acts[T // 2:].contiguous(), store=True); ob, rb, db = b.step_many(acts[T // 2:].contiguous(), store=True)
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    sa, sb = a.export_state(), b.export_state()
    for f in sa:
        assert torch.equal(sa[f], sb[f]), f


def test_largest_batch_offsets_past_4GB_play_the_same_games():
    """Maximum sizes: 16 M + 5 games of 1v1 in ONE batch (12 GB of state; bullet-step rows start beyond 4 GB, the grid is
    524 289 wavefronts, the last one ragged).  The in-kernel generator is keyed by the global env index, so the last 4 096
    games of the big batch must be, bit for bit, the games a 4 096-game shard with that env_offset plays."""
    E, n, T, q = (1 << 24) + 5, 1, 70, 4096
    big = _env(n_agents=n, n_envs=E, seed=21, auto_reset=True)
    small = _env(n_agents=n, n_envs=q, seed=21, auto_reset=True, env_offset=E - q)
    ob = big.reset(); os_ = small.reset()
    for a in big.possible_agents:
        assert torch.equal(ob[a][E - q:], os_[a])
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    act = torch.empty((E, 2), dtype=torch.int32, device="cuda")
    for t in range(T):
        tail = torch.randint(0, 4, (q, 2), generator=g, device="cuda", dtype=torch.int32)
        tail = torch.where(torch.rand((q, 2), generator=g, device="cuda") < 0.6, torch.ones_like(tail), tail)
        act.fill_(2 if t % 4 == 0 else 1); act[E - q:] = tail   # everyone else: three shots, one turn (planes circle instead of parking on a wall)
        o1, r1, d1 = big.step_batch(act)
        o2, r2, d2 = small.step_batch(tail)
        assert torch.equal(o1[E - q:], o2) and torch.equal(r1[E - q:], r2) and torch.equal(d1[E - q:], d2), f"step {t}"
    s1 = big.export_state(); s2 = small.export_state()
    for f in ("px", "py", "pdir", "php", "bhp", "tick", "bl_live", "bl_x", "bl_y", "bl_dir", "counters"):
        assert torch.equal(s1[f][E - q:], s2[f]), f
    # bullets in flight all over the batch, not only in the tail: this schedule holds 3.8 live bullets per plane after 70 calls
    # (measured with the C oracle on 8 192 games, same seed); ask for 2.5
    assert int(s1["bl_live"].sum()) > 2.5 * 2 * E


def test_state_invariants_at_full_size():
    """Properties that hold for any number of games: poses inside the clamp box, headings in [0, 360], hit points in
    range, live bullets inside the field and younger than 12 updates, observation ranges, done/alive consistency."""
    E, n = 65536, 1
    env = _env(n_agents=n, n_envs=E, seed=3, auto_reset=True)
    env.reset()
    acts = _actions(50, E, 2, 11, p_shoot=0.7)
    for rep in range(5):
        for t in range(50):
            obs, rew, done = env.step_batch(acts[t])
        st = env.export_state()
        assert int(st["px"].min()) >= 25 and int(st["px"].max()) <= 1175
        assert int(st["py"].min()) >= 24 and int(st["py"].max()) <= 776
        assert float(st["pdir"].min()) >= 0 and float(st["pdir"].max()) <= 360
        assert int(st["php"].min()) >= 0 and int(st["php"].max()) <= 4
        assert torch.equal(st["palive"].bool(), st["php"] > 0)
        live = st["bl_live"].bool()
        assert int(st["bl_x"][live].min()) >= 0 and int(st["bl_x"][live].max()) <= 1200
        assert int(st["bl_y"][live].min()) >= 0 and int(st["bl_y"][live].max()) <= 800
        assert int(live.sum(-1).max()) <= 11
        assert int(st["tick"].max()) <= env.tie_tick
        running = ~st["env_done"].bool()
        assert torch.equal(done[running], ~st["palive"].bool()[running]) and bool(done[~running].all())
        o = obs[st["palive"].bool() & running[:, None]]
        assert float(o[:, 0].min()) >= -1 and float(o[:, 0].max()) <= 1 and float(o[:, 1].abs().max()) <= 0.5
        c = st["counters"].sum(0)
        assert int(c[0]) == int(c[1] + c[2] + c[3])


def test_graph_replay_equals_eager_steps():
    """capture_steps(): T launches in one HIP graph play the same games as T step_batch() calls."""
    E, n, T = 4096, 1, 40
    acts = _actions(T, E, 2, 5, p_shoot=0.5)
    a = _env(n_agents=n, n_envs=E, seed=8, auto_reset=True); a.reset()
    b = _env(n_agents=n, n_envs=E, seed=8, auto_reset=True); b.reset()
    graph, (go, gr, gd) = b.capture_steps(acts, store=True)
    for rep in range(4):
        graph.replay()
        torch.cuda.synchronize()
        for t in range(T):
            o, r, d = a.step_batch(acts[t])
            assert torch.equal(o, go[t]) and torch.equal(r, gr[t]) and torch.equal(d, gd[t]), (rep, t)
    sa, sb = a.export_state(), b.export_state()
    assert all(torch.equal(sa[k], sb[k]) for k in ("px", "py", "php", "tick", "counters", "bl_live"))


@pytest.mark.parametrize("E,n,chains,mode", [(4096, 1, 2, "int"), (1000, 1, 3, "int"), (1300, 2, 2, "int"), (2049, 4, 3, "int"), (600, 6, 2, "int"),
                                             (777, 3, 4, "cont64"), (1024, 1, 4, "scores"), (1025, 2, 8, "cont32")])
def test_chained_graph_plays_the_same_games(E, n, chains, mode):
    """capture_steps(chains=P): the batch as P game ranges, each its own chain of launches on a branch of ONE graph
    (bsx_step_*_range) -- the games of chains=1, bit for bit: every tick's outputs, the final state, the counters; ragged last
    range, ranges of unequal length, more chains asked for than there are 256-game blocks."""
    A = 2 * n
    cont = mode.startswith("cont")
    a = _env(n_agents=n, n_envs=E, seed=21, auto_reset=True, continuous_actions=cont); a.reset()
    b = _env(n_agents=n, n_envs=E, seed=21, auto_reset=True, continuous_actions=cont); b.reset()
    ranges = b.chain_ranges(chains)
    assert len(b.chain_ranges("auto")) == 1                    # a batch this small gains nothing from chains
    assert ranges[0][0] == 0 and sum(c for _, c in ranges) == E and all(f % 256 == 0 for f, _ in ranges)
    assert all(ranges[i][0] + ranges[i][1] == ranges[i + 1][0] for i in range(len(ranges) - 1)) and len(ranges) == min(chains, -(-E // 256))
    T = 60
    g = torch.Generator(device="cuda"); g.manual_seed(E + n)
    if mode == "int":
        acts = _actions(T, E, A, 300 + E, p_shoot=0.7)
    elif mode == "scores":
        acts = torch.randn((T, E, A, 4), generator=g, device="cuda"); acts[..., 1] += 0.9
    else:
        acts = (torch.rand((T, E, A, 3), generator=g, device="cuda", dtype=torch.float64) * 2.6 - 1.3)
        acts = acts.to(torch.float32).contiguous() if mode == "cont32" else acts
    ga, outs_a = a.capture_steps(acts, store=True)
    gb, outs_b = b.capture_steps(acts, store=True, chains=chains)
    for rep in range(5):                                      # 300 calls: across the time-limit tie and the re-spawns
        ga.replay(); gb.replay()
        torch.cuda.synchronize()
        for x, y, name in zip(outs_a, outs_b, ("obs", "rew", "done")):
            assert torch.equal(x, y), (rep, name)
    sa, sb = a.export_state(), b.export_state()
    assert all(torch.equal(sa[k], sb[k]) for k in sa), [k for k in sa if not torch.equal(sa[k], sb[k])]
    assert torch.equal(a._env_done, b._env_done) and torch.equal(a._winner, b._winner)


def test_range_calls_in_any_order_make_one_step():
    """bsx_step_discrete_range: three launches over disjoint game ranges, issued last range first, are one step() of the batch;
    rows outside a launch's range are left alone."""
    E, n = 1500, 2
    a = _env(n_agents=n, n_envs=E, seed=5, auto_reset=True); a.reset()
    b = _env(n_agents=n, n_envs=E, seed=5, auto_reset=True); b.reset()
    acts = _actions(30, E, 2 * n, 77, p_shoot=0.6)
    for t in range(30):
        oa, ra, da = a.step_batch(acts[t], copy=True)
        b._obs.fill_(7.0)
        for i, games in enumerate(reversed(b.chain_ranges(3))):
            b._launch(acts[t].data_ptr(), 0, False, None, b._obs.data_ptr(), b._rew.data_ptr(), b._done.data_ptr(), games=games)
            if i == 0:                                        # only the last range has been written so far
                assert bool((b._obs[:games[0]] == 7.0).all()) and torch.equal(b._obs[games[0]:], oa[games[0]:])
        assert torch.equal(b._obs, oa) and torch.equal(b._rew, ra) and torch.equal(b._done.view(torch.bool), da.view(torch.bool)), t
    sa, sb = a.export_state(), b.export_state()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)


@pytest.mark.parametrize("E,n,mode", [(4096, 1, "int"), (1000, 2, "int"), (515, 4, "int"), (300, 6, "int"), (33, 16, "int"), (70, 3, "cont64"),
                                      (2048, 1, "scores"), (1024, 2, "cont32"), (777, 1, "cont64")])
def test_step_many_equals_consecutive_step_calls(E, n, mode):
    """bsx_step_many_*: T ticks in ONE launch (a wavefront walks its games through all of them, state through the L2)
    play the same games as T step_batch() launches, bit for bit -- every tick's outputs and the final state, across
    auto-resets and the time-limit tie, with long bullet lists (shoot-heavy), in chunks of different lengths."""
    A = 2 * n
    cont = mode.startswith("cont")
    a = _env(n_agents=n, n_envs=E, seed=13, auto_reset=True, continuous_actions=cont); a.reset()
    b = _env(n_agents=n, n_envs=E, seed=13, auto_reset=True, continuous_actions=cont); b.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(E + n)
    for T in (1, 7, 150, 64):
        if mode == "int":
            acts = _actions(T, E, A, 100 + T, p_shoot=0.7)
            acts[T // 2, : E // 3] = 9                          # out-of-range indices: the plane does not move
        elif mode == "scores":
            acts = torch.randn((T, E, A, 4), generator=g, device="cuda"); acts[..., 1] += 0.9
        else:
            acts = (torch.rand((T, E, A, 3), generator=g, device="cuda", dtype=torch.float64) * 2.6 - 1.3)
            acts = acts.to(torch.float32).contiguous() if mode == "cont32" else acts
        mo, mr, md = b.step_many(acts, store=True)
        for t in range(T):
            o, r, d = a.step_batch(acts[t])
            assert torch.equal(o, mo[t]) and torch.equal(r, mr[t]) and torch.equal(d, md[t]), (T, t)
        assert torch.equal(a.env_done, b.env_done) and torch.equal(a.winner, b.winner)
        sa, sb = a.export_state(), b.export_state()
        for k in sa:
            if k.startswith("bl_") and k != "bl_live":
                m = sa["bl_live"].bool()
                assert torch.equal(sa[k][m], sb[k][m]), (T, k)
            else:
                assert torch.equal(sa[k], sb[k]), (T, k)
    # store=False: the env-owned tensors hold the last tick
    acts = (_actions(5, E, A, 1) if mode == "int" else acts[:5].contiguous())
    o2, r2, d2 = b.step_many(acts)
    for t in range(5):
        o, r, d = a.step_batch(acts[t])
    assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(d, d2)
    assert b.tie_tick > 227 or int(b.counters()[:, 0].sum()) > 0     # the 227 ticks crossed game ends (n = 16 ties on call 421 only)


def _compare_generic(E, n, T, seed, cont=False, logits=False, f32=False, one_wave=False):
    """HIP vs C oracle in the production configuration for any action encoding; ragged sizes (E not a multiple of the
    games-per-wave count) exercise the clamped-index lanes and the partial last wavefront."""
    A = 2 * n
    env = _env(n_agents=n, n_envs=E, seed=seed, auto_reset=True, continuous_actions=cont, one_wave=one_wave)
    c = cref.CRefBatch(E, n_agents=n, seed=seed, auto_reset=True, continuous_actions=cont)
    env.reset(); c.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    n_exact = n_vals = 0
    for t in range(T):
        if cont:
            a = (torch.rand((E, A, 3), generator=g, device="cuda", dtype=torch.float64) * 2.6 - 1.3)
            a = a.to(torch.float32) if f32 else a
            a[..., 2] = torch.where(torch.rand((E, A), generator=g, device="cuda") < 0.5, a[..., 2].abs(), a[..., 2])
        elif logits:
            a = torch.randn((E, A, 4), generator=g, device="cuda", dtype=torch.float32)
            a[..., 1] += 0.8
        else:
            a = torch.randint(-1, 5, (E, A), generator=g, device="cuda", dtype=torch.int32)
            a = torch.where(torch.rand((E, A), generator=g, device="cuda") < 0.5, torch.ones_like(a), a)
        obs, rew, done = env.step_batch(a)
        co, cr, cd = c.step(a.cpu().numpy())
        o = obs.cpu().numpy()
        assert np.array_equal(done.cpu().numpy(), cd), f"step {t}: done"
        np.testing.assert_allclose(rew.cpu().numpy(), cr, rtol=1e-6, atol=1e-6, err_msg=f"step {t}: rew")
        np.testing.assert_allclose(o, co, rtol=OBS_RTOL, atol=OBS_ATOL, err_msg=f"step {t}: obs")
        n_exact += int((o == co).sum()); n_vals += o.size
    sh = {k: v.cpu().numpy() for k, v in env.export_state().items()}
    sc = c.export_state()
    for f in ("px", "py", "pdir", "php", "bhp", "tick", "env_done", "winner", "bl_live", "counters"):
        assert np.array_equal(sh[f], sc[f]), f
    m = sc["bl_live"].astype(bool)
    assert np.array_equal(sh["bl_x"][m], sc["bl_x"][m]) and np.array_equal(sh["bl_dir"][m], sc["bl_dir"][m])
    assert n_exact >= n_vals * (1 - 2e-3)


@pytest.mark.parametrize("E,n", [(1, 1), (31, 1), (33, 1), (1000, 2), (7, 3), (129, 4), (50, 5), (40, 8), (9, 16)])
def test_ragged_sizes_and_every_team_size_vs_c_oracle(E, n):
    _compare_generic(E, n, 10 * (10 + 2 * n) + 25, seed=100 + E + n)


@pytest.mark.parametrize("n,f32", [(1, False), (1, True), (2, False), (4, True)])
def test_continuous_actions_vs_c_oracle(n, f32):
    """battle_env.py:418-424 at scale: float64 and float32 action tensors (clipped in-kernel), in-kernel jitter."""
    _compare_generic(8192, n, 150, seed=300 + n, cont=True, f32=f32)


def test_C2_size_continuous_actions_vs_c_oracle():
    """The continuous mode (the reference's own driver test_env.py:22-43 steps the env with [speed, turn, shoot] boxes) at the
    headline batch size: 65 536 games of 1v1, float32 actions, 135 calls across the 121-call tie and the re-spawns."""
    _compare_generic(65536, 1, 135, seed=901, cont=True, f32=True)


def test_score_vector_actions_vs_c_oracle():
    """[E, A, 4] score vectors arg-maxed in-kernel (battle_env.py:327-328)."""
    _compare_generic(8192, 2, 150, seed=77, logits=True)


@pytest.mark.parametrize("fused", [False, True])
def test_policy_rollout_graph_equals_eager_loop(fused):
    """configs[4] plumbing: actor -> score vectors -> step() captured in one HIP graph plays the same games as the
    tick-by-tick loop, and the transition buffers are consistent (obs[t+1] of tick t is obs[t] of tick t+1's input)."""
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    E, n, T = 2048, 2, 24
    torch.manual_seed(3)
    actor = StackedActor(2 * n, 3 * n + 2, 4, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(200.0)                                    # decisive scores (default init is +-0.003)
    a = _env(n_agents=n, n_envs=E, seed=21, auto_reset=True); a.reset()
    b = _env(n_agents=n, n_envs=E, seed=21, auto_reset=True); b.reset()
    ro = PolicyRollout(b, actor, T, fused=fused)
    ro.start(); ro.capture()
    obs = a._obs.clone()
    from deep_rl_battlespace_amd.rollout import FusedActor
    fa = FusedActor(actor, n)
    for rep in range(3):
        ro.run()
        torch.cuda.synchronize()
        assert torch.equal(ro.obs[0], obs)
        for t in range(T):
            with torch.no_grad():
                s = fa(obs) if fused else actor(obs)
            o, r, d = a.step_batch(s.contiguous())
            assert torch.equal(ro.scores[t], s) and torch.equal(ro.obs[t + 1], o) and torch.equal(ro.rew[t], r) and torch.equal(ro.done[t], d), (rep, t)
            obs = o.clone()
    acts = ro.scores.argmax(-1)
    assert len(torch.unique(acts)) >= 3                        # the policy actually uses several actions


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 8, 12, 16])
def test_fused_actor_matches_torch_fp32_reference(n):
    """bsx_actor_forward (hand-written HIP) against the torch fp32 composition of the same op (StackedActor), which is
    itself pinned on the reference ActorNetwork's forward (tests/test_rollout_cpu.py).  fp32, different summation
    order: 2e-5 absolute on tanh outputs."""
    from deep_rl_battlespace_amd.rollout import FusedActor, StackedActor
    torch.manual_seed(10 + n)
    E, A, D = 5000 - n, 2 * n, 3 * n + 2                     # (ragged: the last wave's tile is partly empty)
    actor = StackedActor(A, D, 4 if n % 3 else 3, device="cuda")   # three outputs = the continuous-action head
    with torch.no_grad():
        actor.w3.mul_(50.0); actor.g1.uniform_(0.5, 1.5); actor.h1.uniform_(-0.3, 0.3); actor.g2.uniform_(0.5, 1.5); actor.h2.uniform_(-0.3, 0.3)
    fused = FusedActor(actor, n)
    obs = torch.rand((E, A, D), device="cuda") * 2 - 1
    obs[::7] = -1.0                                             # dead observers see all -1
    with torch.no_grad():
        want = actor(obs)
    na = want.shape[-1]                                        # (the kernel's rows are 4 wide: three outputs + one unused column)
    got = fused(obs)[..., :na]
    torch.testing.assert_close(got, want, rtol=0, atol=2e-5)
    assert float((got.argmax(-1) == want.argmax(-1)).float().mean()) > 0.999
    # weights refresh after an update
    with torch.no_grad():
        actor.b3.add_(0.1)
    fused.refresh()
    torch.testing.assert_close(fused(obs)[..., :na], actor(obs).detach(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("n", [1, 4])
def test_fused_actor_bf16x3_is_close_to_fp32(n):
    """precision="bf16x3": the 64 x 64 layer as three bf16 matrix products of two-term splits.  Against the torch fp32
    reference: 1e-4 absolute on the tanh outputs (measured ~1e-5), far from plain-bf16 accuracy (~4e-3); the arg-max
    agrees on all but a sliver of rows; and it is NOT the exact-f32 path (some bits differ)."""
    from deep_rl_battlespace_amd.rollout import FusedActor, StackedActor
    torch.manual_seed(20 + n)
    E, A, D = 20000, 2 * n, 3 * n + 2
    actor = StackedActor(A, D, 4, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(50.0); actor.g1.uniform_(0.5, 1.5); actor.h1.uniform_(-0.3, 0.3); actor.g2.uniform_(0.5, 1.5); actor.h2.uniform_(-0.3, 0.3)
    obs = torch.rand((E, A, D), device="cuda") * 2 - 1
    with torch.no_grad():
        want = actor(obs)
    exact = FusedActor(actor, n)(obs)
    got = FusedActor(actor, n, precision="bf16x3")(obs)
    err = float((got - want).abs().max())
    assert err < 1e-4, err
    assert float((got - want).abs().mean()) < 1e-5
    assert float((got.argmax(-1) == want.argmax(-1)).float().mean()) > 0.999
    assert not torch.equal(got, exact)
    # "bf16x6": three terms per operand, six products -- float32-class accuracy (the same 2e-5 the exact-f32 kernel is held to
    # against torch, measured ~1e-6 from the exact kernel), still not the fmaf chain's bits
    six = FusedActor(actor, n, precision="bf16x6")(obs)
    torch.testing.assert_close(six, want, rtol=0, atol=2e-5)
    assert float((six - exact).abs().max()) < 5e-6 and float((six - exact).abs().mean()) < 2e-7
    assert float((six.argmax(-1) == exact.argmax(-1)).float().mean()) > 0.9999
    with pytest.raises(ValueError):
        FusedActor(actor, n, precision="fp8")


def test_fused_actor_noise_is_gaussian_clamped_and_rekeyed():
    from deep_rl_battlespace_amd.rollout import FusedActor, StackedActor
    torch.manual_seed(1)
    n, E = 1, 200000
    actor = StackedActor(2, 5, 4, device="cuda")                 # default init: scores ~ 0 (+-0.003 head)
    fused = FusedActor(actor, n, seed=5)
    obs = torch.rand((E, 2, 5), device="cuda") * 2 - 1
    base = fused(obs, noise_std=0.0)
    a = fused(obs, noise_std=0.25, seq=7)
    b = fused(obs, noise_std=0.25, seq=7)
    c = fused(obs, noise_std=0.25, seq=8)
    assert torch.equal(a, b) and not torch.equal(a, c)           # deterministic per key, fresh per seq
    z = (a - base) / 0.25
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    assert abs(float((z ** 3).mean())) < 0.03 and abs(float((z ** 4).mean()) - 3.0) < 0.1
    big = fused(obs, noise_std=5.0, seq=9)
    assert float(big.max()) <= 1.0 and float(big.min()) >= -1.0 and float((big.abs() == 1.0).float().mean()) > 0.5
    sb = torch.tensor([1], dtype=torch.int64, device="cuda")
    d = torch.empty_like(a); fused.forward_into(obs, d, 0.25, seq=6, seq_base=sb)
    assert torch.equal(d, a)                                     # seq + *seq_base


def test_fused_actor_ou_noise_follows_the_reference_process():
    """Ornstein-Uhlenbeck option (utils/noise.py:17-21): x += theta*(mu - x) + sigma*N(0,1); action += scale*x; clamp.
    The normals are Philox draws, so the recursion is checked by replaying the SAME key: from x = 0 the first update gives
    x1 = sigma*z, a second one with the same z gives x2 = (1 - theta)*x1 + sigma*z = (2 - theta)*x1; rows whose game has
    finished restart from mu (main.py:155)."""
    from deep_rl_battlespace_amd.rollout import FusedActor, StackedActor
    torch.manual_seed(2)
    n, E = 2, 50000
    A, D = 2 * n, 3 * n + 2
    actor = StackedActor(A, D, 4, device="cuda")
    fused = FusedActor(actor, n, seed=9)
    obs = torch.rand((E, A, D), device="cuda") * 2 - 1
    base = fused(obs)
    st = torch.zeros((E, A, 4), device="cuda")
    out = torch.empty_like(base)
    theta, sigma, scale = 0.15, 0.2, 0.5
    fused.forward_into(obs, out, 0.0, seq=3, ou=dict(scale=scale, state=st))
    x1 = st.clone()
    z = x1 / sigma
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    torch.testing.assert_close(out, (base + scale * x1).clamp(-1, 1), rtol=0, atol=1e-6)
    fused.forward_into(obs, out, 0.0, seq=3, ou=dict(scale=scale, state=st))
    torch.testing.assert_close(st, (2 - theta) * x1, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(out, (base + scale * st).clamp(-1, 1), rtol=0, atol=1e-6)
    # finished games restart from mu; the others continue; non-default theta / sigma / mu are honoured
    done = torch.zeros(E, dtype=torch.uint8, device="cuda"); done[::3] = 1
    prev = st.clone()
    fused.forward_into(obs, out, 0.0, seq=3, ou=dict(scale=scale, state=st, env_done=done, theta=0.5, sigma=0.1, mu=0.25))
    want = torch.where(done.bool()[:, None, None], torch.full_like(prev, 0.25), prev)
    want = want + 0.5 * (0.25 - want) + 0.1 * z
    torch.testing.assert_close(st, want, rtol=1e-5, atol=1e-6)
    # Gaussian and OU may be combined: the OU increment keeps its draw, the white term takes a second, independent one
    both = torch.empty_like(base); st2 = torch.zeros_like(st)
    fused.forward_into(obs, both, 0.25, seq=3, ou=dict(scale=scale, state=st2))
    assert torch.equal(st2, x1)
    zg = (both - (base + scale * x1)) / 0.25
    inside = (both.abs() < 0.999)
    assert abs(float(zg[inside].mean())) < 0.02 and abs(float(zg[inside].std()) - 1.0) < 0.05
    corr = float(torch.corrcoef(torch.stack([zg[inside], z[inside]]))[0, 1])
    assert abs(corr) < 0.01, corr                                # not the OU process's normals again
    # a missing state tensor is refused
    with pytest.raises(ValueError):
        fused.forward_into(obs, out, 0.0, seq=3, ou=dict(scale=scale, state=st[:10]))


def test_rollout_with_ou_noise_restarts_per_game():
    """PolicyRollout(ou_scale=...): the OU state lives in HBM across graph replays and is re-zeroed for a game on the tick
    its env_done flag is up (that tick's actions are the ignored ones of the auto-reset call)."""
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    E, n, T = 2048, 1, 24
    env = _env(n_agents=n, n_envs=E, seed=8, auto_reset=True); env.reset()
    actor = StackedActor(2, 5, 4, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(100.0)
    ro = PolicyRollout(env, actor, T, ou_scale=0.3, seed=4); ro.start(); ro.capture()
    for _ in range(12):
        ro.run()
    torch.cuda.synchronize()
    x = ro.ou["state"]
    # stationary std of the discrete process: sigma / sqrt(1 - (1 - theta)^2) = 0.2 / sqrt(0.2775) = 0.38
    assert 0.3 < float(x.std()) < 0.45 and abs(float(x.mean())) < 0.02
    assert float(ro.scores.max()) <= 1.0 and float(ro.scores.min()) >= -1.0
    assert int(env.counters()[:, 0].sum()) > 0                    # games did finish (and restart) along the way


@pytest.mark.parametrize("n", [1, 2, 3, 4])
@pytest.mark.parametrize("noise", ["none", "gaussian", "ou", "gaussian-bf16x3", "ou-bf16x6"])
def test_one_launch_rollout_equals_two_kernel_rollout(noise, n):
    """bsx_rollout_discrete (T ticks of actor -> step in ONE launch, observation rows handed over in LDS) against the
    two-kernel form (bsx_actor_forward + bsx_step_discrete per tick): the same transitions bit for bit -- observations,
    the actors' score vectors (incl. exploration noise and OU state), rewards, dones, final game state -- over several
    runs, across auto-resets, with a ragged last wavefront."""
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    E, T = (4100 if n == 1 else 1030), 50
    torch.manual_seed(3)
    actor = StackedActor(2 * n, 3 * n + 2, 4, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(60.0); actor.g1.uniform_(0.5, 1.5); actor.h1.uniform_(-0.3, 0.3)
    kw = dict(noise_std=0.3) if noise.startswith("gaussian") else (dict(ou_scale=0.4) if noise.startswith("ou") else {})
    if noise.endswith("bf16x3"):
        kw["precision"] = "bf16x3"
    if noise.endswith("bf16x6"):
        kw["precision"] = "bf16x6"                               # (every team size since round 3: the three-term weights as a rolling window)
    ros = []
    for one in (False, True):
        env = _env(n_agents=n, n_envs=E, seed=31, auto_reset=True); env.reset()
        ro = PolicyRollout(env, actor, T, seed=7, one_launch=one, **kw); ro.start()
        if one:
            ro.capture()
        ros.append(ro)
    a, b = ros
    for rep in range(4):
        a.run(); b.run()
        torch.cuda.synchronize()
        assert torch.equal(a.obs, b.obs), rep
        assert torch.equal(a.scores, b.scores), rep
        assert torch.equal(a.rew, b.rew) and torch.equal(a.done, b.done), rep
        if noise.startswith("ou"):
            assert torch.equal(a.ou["state"], b.ou["state"]), rep
    sa, sb = a.env.export_state(), b.env.export_state()
    for k in ("px", "py", "pdir", "php", "bhp", "tick", "env_done", "winner", "bl_live", "counters"):
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(a.env.env_done, b.env.env_done)
    assert int(a.env.counters()[:, 0].sum()) >= E                # the runs crossed game ends
    assert len(torch.unique(a.scores.argmax(-1))) >= 3


@pytest.mark.parametrize("fused", [False, True])
def test_continuous_policy_rollout(fused):
    """The env's other action mode on the rollout path: an actor with 3 outputs [speed, turn, shoot] (maddpg/networks.py:75
    with env.n_actions = 3, battle_env.py:151-153), written as 4-wide rows and read by bsx_step_continuous as BSX_ACT_F32X4.
    The graph plays what an eager loop plays, and the 4-wide path steps exactly like plain float32 [E, A, 3] actions."""
    from deep_rl_battlespace_amd.replay import ReplayBuffer
    from deep_rl_battlespace_amd.rollout import FusedActor, PolicyRollout, StackedActor
    E, n, T = 3000, 2, 20
    A, D = 2 * n, 3 * n + 2
    torch.manual_seed(5)
    actor = StackedActor(A, D, 3, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(80.0)
    env = _env(n_agents=n, n_envs=E, seed=9, auto_reset=True, continuous_actions=True); env.reset()
    twin = _env(n_agents=n, n_envs=E, seed=9, auto_reset=True, continuous_actions=True); twin.reset()
    ro = PolicyRollout(env, actor, T, noise_std=0.2 if fused else 0.0, fused=fused, seed=3); ro.start(); ro.capture()
    for rep in range(3):
        ro.run()
        torch.cuda.synchronize()
        assert float(ro.scores[..., 3].abs().max()) <= (1.0 if fused else 0.0)       # the padding column: tanh(0) (+ noise), never read
        for t in range(T):
            o, r, d = twin.step_batch(ro.scores[t][..., :3].contiguous())
            assert torch.equal(o, ro.obs[t + 1]) and torch.equal(r, ro.rew[t]) and torch.equal(d, ro.done[t]), (rep, t)
    if fused:
        with torch.no_grad():
            want = actor(ro.obs[3])
        got = FusedActor(actor, n)(ro.obs[3])
        torch.testing.assert_close(got[..., :3], want, rtol=0, atol=2e-5)
        assert float(got[..., 3].abs().max()) == 0.0
    buf = ReplayBuffer(100000, 64, env.possible_red, env.obs_size, env.obs_size * n, 3, device="cuda")
    stored = buf.store_rollout(ro, range(n))
    keep = ro.valid.reshape(-1).nonzero().squeeze(1)
    assert stored == buf.mem_cntr == int(ro.valid.sum()) and 0 < stored <= T * E
    k5 = int(keep[5]); assert torch.equal(buf.action_mem[5], ro.scores[k5 // E, k5 % E, :n, :3])
    with pytest.raises(ValueError):
        PolicyRollout(env, StackedActor(A, D, 4, device="cuda"), T)                 # discrete head on a continuous env


@pytest.mark.parametrize("E,n", [(1, 1), (31, 1), (5, 4), (33, 3), (2, 2)])
def test_one_launch_rollout_tiny_and_ragged_batches(E, n):
    """Workgroups of the fused rollout cover 32 games: batches smaller than that, and waves that lie entirely beyond the
    batch, must neither read nor write out of range -- and still play what the per-tick kernels play."""
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    T = 12
    torch.manual_seed(E + n)
    actor = StackedActor(2 * n, 3 * n + 2, 4, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(60.0)
    ros = []
    for one in (False, True):
        env = _env(n_agents=n, n_envs=E, seed=4, auto_reset=True); env.reset()
        guard = torch.full((64,), 7.0, device="cuda")          # allocated right behind the rollout buffers
        ro = PolicyRollout(env, actor, T, noise_std=0.3, seed=2, one_launch=one); ro.start()
        ros.append((ro, guard))
    (a, _), (b, guard) = ros
    for rep in range(3):
        a.run(); b.run(); torch.cuda.synchronize()
        assert torch.equal(a.obs, b.obs) and torch.equal(a.scores, b.scores) and torch.equal(a.rew, b.rew) and torch.equal(a.done, b.done)
    assert bool((guard == 7.0).all())
    sa, sb = a.env.export_state(), b.env.export_state()
    assert all(torch.equal(sa[k], sb[k]) for k in ("px", "py", "php", "bhp", "tick", "bl_live", "counters"))


def test_rollout_into_replay_buffer_on_device():
    """f-1 -> f-3: a rollout's transitions go into the device replay ring without a host copy of the data, and ONLY transitions
    do: a tick that found its game finished -- the auto-reset call, whose row would pair the old game's last observation with
    the new game's first under done=False -- is not stored (the reference's loop never produces one: `while not env.env_done`,
    main.py:177-181).  Checked row by row against the rollout's own record of env_done before every tick."""
    from deep_rl_battlespace_amd.replay import ReplayBuffer
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    from deep_rl_battlespace_amd import instinct
    E, n, T = 512, 2, 16
    for one_launch in (False, True):
        env = _env(n_agents=n, n_envs=E, seed=5, auto_reset=True); env.reset()
        actor = StackedActor(2 * n, 3 * n + 2, 4, device="cuda")
        with torch.no_grad():
            actor.w3.mul_(100.0)
        # the scripted opponent makes games decisive early, so game ends (and re-spawns) fall inside these few ticks
        ro = PolicyRollout(env, actor, T, noise_std=0.2, one_launch=one_launch, opponent=instinct.Team(env.possible_blue, env.possible_red, env))
        ro.start(); ro.capture()
        buf = ReplayBuffer(40000, 256, env.possible_red, env.obs_size, env.obs_size * n, 4, device="cuda")
        total, dropped, ends = 0, 0, 0
        for rep in range(12):
            ro.run()
            edone = ro.env_done.clone()
            # the record is env_done BEFORE each tick: row t+1 follows from tick t's dones, and a finished game is re-spawned by
            # the very next tick (auto_reset), so two consecutive set rows never occur
            assert bool(ro.done[edone[1:] != 0].all())           # a finished game reports every plane done
            assert not bool(((edone[:-1] != 0) & (edone[1:] != 0)).any())
            reset_rows = edone[:T] != 0
            assert float(ro.rew[reset_rows].abs().max() if reset_rows.any() else 0.0) == 0.0 and not bool(ro.done[reset_rows].any())
            before = buf.mem_cntr
            stored = buf.store_rollout(ro, range(n))
            keep = (~reset_rows).reshape(-1).nonzero().squeeze(1)
            assert stored == keep.numel() == buf.mem_cntr - before
            # every stored row, in order, is the kept row of the rollout
            rows = (before + torch.arange(stored, device="cuda")) % buf.mem_size
            t_i, e_i = keep // E, keep % E
            assert torch.equal(buf.actor_states[rows], ro.obs[t_i, e_i, :n]) and torch.equal(buf.actor_new_states[rows], ro.obs[t_i + 1, e_i, :n])
            assert torch.equal(buf.action_mem[rows], ro.scores[t_i, e_i, :n]) and torch.equal(buf.rew_mem[rows], ro.rew[t_i, e_i, :n])
            assert torch.equal(buf.done_mem[rows], ro.done[t_i, e_i, :n])
            total += stored; dropped += int(reset_rows.sum()); ends += int((edone[1:] != 0).sum())
        torch.cuda.synchronize()
        assert dropped > 0 and total + dropped == 12 * T * E and buf.is_ready()
        waiting = int((env.env_done != 0).sum())
        assert dropped == ends - waiting                          # one re-spawn call per game end (those still waiting excepted)
        games = int(env.counters()[:, 0].sum())
        assert ends <= games <= 2 * ends                          # both bases in one call count as two games (battle_env.py:363-372)
        a_s, s2_, a, r, a_s2, s2, d = buf.sample()
        assert s2_.shape == (256, n * env.obs_size) and a.shape == (n, 256, 4) and d.dtype == torch.bool
        assert float(a.abs().max()) <= 1.0
        assert torch.equal(env.env_done, ro.env_done[T] != 0)   # the env's own flag is the last row of the record


@pytest.mark.parametrize("case", range(12))
def test_randomised_configurations_vs_c_oracle(case):
    """A seeded sweep over the configuration space -- team size, batch size (ragged), action encoding, float rewards,
    auto-reset on / off with masked host resets, env_offset, per-step vs multi-tick launches -- HIP against the C oracle:
    dones equal, rewards within 1e-6, observations within 1e-5, complete integer state identical at the end."""
    rs = np.random.RandomState(1000 + case)
    n = int(rs.choice([1, 1, 2, 3, 4, 5, 7]))
    E = int(rs.randint(1, 400))
    cont = bool(rs.rand() < 0.3)
    auto = bool(rs.rand() < 0.6)
    many = (not cont or rs.rand() < 0.5) and bool(rs.rand() < 0.5)
    rewards = dict(hit_base_reward=float(rs.choice([100, 50.5])), hit_plane_reward=float(rs.choice([10, 2.25])),
                   miss_punishment=float(rs.choice([-1, -0.125])), die_punishment=-5.0, lose_punishment=float(rs.choice([-20, -7.5])))
    off = int(rs.randint(0, 1 << 20))
    A, T = 2 * n, 130
    kw = dict(n_agents=n, n_envs=E, seed=case, auto_reset=auto, continuous_actions=cont, env_offset=off, **rewards)
    env = _env(**kw)
    c = cref.CRefBatch(E, **{k: v for k, v in kw.items() if k != "n_envs"})
    env.reset(); c.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(case)
    t = 0
    while t < T:
        k = int(rs.randint(1, 40)) if many else 1
        k = min(k, T - t)
        if cont:
            acts = (torch.rand((k, E, A, 3), generator=g, device="cuda", dtype=torch.float64) * 2.4 - 1.2).to(torch.float32)
        elif rs.rand() < 0.5:
            acts = torch.randint(-1, 5, (k, E, A), generator=g, device="cuda", dtype=torch.int32)
            acts = torch.where(torch.rand((k, E, A), generator=g, device="cuda") < 0.5, torch.ones_like(acts), acts)
        else:
            acts = torch.randn((k, E, A, 4), generator=g, device="cuda"); acts[..., 1] += 0.7
        if many:
            outs = env.step_many(acts.contiguous(), store=True)
        for i in range(k):
            o, r, d = (x[i] for x in outs) if many else env.step_batch(acts[i])
            co, cr, cd = c.step(acts[i].cpu().numpy())
            assert np.array_equal(d.cpu().numpy(), cd), (case, t + i)
            np.testing.assert_allclose(r.cpu().numpy(), cr, rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(o.cpu().numpy(), co, rtol=OBS_RTOL, atol=OBS_ATOL)
        t += k
        if not auto and rs.rand() < 0.3:                     # host-side masked reset of the finished games, same spawns on both sides
            m = env.env_done.cpu().numpy().astype(bool)
            if m.any():
                spawn = np.zeros((E, 4 + 3 * A), np.int32)
                spawn[:, 0:4] = [200, 300, 900, 400]
                for a in range(A):
                    spawn[:, 4 + 3 * a: 7 + 3 * a] = [100 + 30 * a if a < n else 1100 - 30 * a, 100 + 40 * a, 0 if a < n else 180]
                env.reset(spawn=spawn, mask=m); c.reset(spawn=spawn, mask=m)
    sh = {k2: v.cpu().numpy() for k2, v in env.export_state().items()}
    sc = c.export_state()
    for f in ("px", "py", "pdir", "php", "bhp", "tick", "env_done", "winner", "bl_live", "counters"):
        assert np.array_equal(sh[f], sc[f]), (case, f)
    mlive = sc["bl_live"].astype(bool)
    assert np.array_equal(sh["bl_x"][mlive], sc["bl_x"][mlive]) and np.array_equal(sh["bl_dir"][mlive], sc["bl_dir"][mlive])
