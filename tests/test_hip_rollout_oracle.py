"""The rollout path (BASELINE.json configs[4]) against the oracle and against fixtures recorded from the
unmodified reference: the closed loop  reference ActorNetwork weights -> on-device actor -> arg-max -> step  beside the C
oracle stepping on the same score vectors; the Ornstein-Uhlenbeck exploration noise against utils/noise.py's own
trajectory (g11); the device replay ring against maddpg/buffer.py's memory layout (g10); the continuous-action one-launch
rollout against its two-kernel form; exploration noise under sharding; one game's exported state as an image."""
import os

import numpy as np
import pytest
import torch

from oracle import cref
from trace_util import OBS_ATOL, OBS_RTOL

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _env(**kw):
    import deep_rl_battlespace_amd as bsx
    return bsx.parallel_env(**kw)


def _reference_actor(tag, n_slots, device="cuda"):
    """StackedActor whose every slot holds the reference ActorNetwork's weights recorded in g9 (maddpg/networks.py:54-85)."""
    from deep_rl_battlespace_amd.rollout import StackedActor
    g9 = np.load(os.path.join(GOLD, "g9_actor_forward.npz"))
    sd = {k.split("/", 1)[1]: torch.from_numpy(g9[k]) for k in g9.files if k.startswith(tag + "/") and k.split("/")[1] not in ("x", "y")}
    actor = StackedActor(n_slots, sd["fc1.weight"].shape[1], 4, device=device)
    for i in range(n_slots):
        actor.load_reference_actor(i, {k: v.to(device) for k, v in sd.items()})
    with torch.no_grad():                                        # the stacked module IS the reference forward on the fixture's rows
        x, y = torch.from_numpy(g9[tag + "/x"]).to(device), torch.from_numpy(g9[tag + "/y"]).to(device)
        got = actor(x[:, None, :].expand(-1, n_slots, -1).contiguous())
    torch.testing.assert_close(got[:, 0], y, rtol=0, atol=2e-6)
    return actor


@pytest.mark.parametrize("one_launch", [True, False])
def test_closed_loop_reference_actor_rollout_vs_c_oracle(one_launch):
    """configs[4] at full size, loop closed against the oracle.  Device: 65 536 games of 1v1, both planes run the reference
    ActorNetwork (g9 weights) through the MFMA actor, exploration noise, clamp, in-kernel arg-max, fused step, for 256 ticks
    (two full games and the auto-resets between them) -- as ONE launch per 32 ticks (bsx_rollout_discrete) and as the
    two-kernel graph.  Beside it the C oracle plays the SAME games from the same score vectors' arg-max
    (maddpg/agent.py:25-33 -> battle_env.py:327-328): rewards, dones, game flags and the full state must be identical,
    observations within 1e-5; and on the observations the games actually visited the device actor equals the torch fp32
    restatement of the reference forward."""
    from deep_rl_battlespace_amd.rollout import FusedActor, PolicyRollout
    E, n, T, RUNS = 65536, 1, 32, 8
    actor = _reference_actor("1v1", 2)
    env = _env(n_agents=n, n_envs=E, seed=77, auto_reset=True)
    c = cref.CRefBatch(E, n_agents=n, seed=77, auto_reset=True)
    o_h = env.reset(); o_c = c.reset()
    np.testing.assert_allclose(torch.stack([o_h[a] for a in env.possible_agents], 1).cpu().numpy(), o_c, rtol=OBS_RTOL, atol=OBS_ATOL)
    ro = PolicyRollout(env, actor, T, noise_std=0.6, seed=5, one_launch=one_launch)
    ro.start(); ro.capture()
    n_exact = n_vals = 0
    acts_seen = torch.zeros(4, dtype=torch.long)
    for run in range(RUNS):
        ro.run()
        torch.cuda.synchronize()
        obs, sc, rew, done, edone = (x.cpu().numpy() for x in (ro.obs, ro.scores, ro.rew, ro._done, ro.env_done))
        if run == 0:
            np.testing.assert_allclose(obs[0], o_c, rtol=OBS_RTOL, atol=OBS_ATOL)
        for t in range(T):
            assert np.array_equal(edone[t], c.env_done), (run, t)         # env_done BEFORE the tick
            co, cr, cd = c.step(sc[t])                                     # float32 [E, A, 4] score vectors: arg-maxed (battle_env.py:327-328)
            assert np.array_equal(done[t].astype(bool), cd), (run, t)
            assert np.array_equal(rew[t].astype(np.float64), cr), (run, t)
            np.testing.assert_allclose(obs[t + 1], co, rtol=OBS_RTOL, atol=OBS_ATOL, err_msg=f"run {run} tick {t}")
            n_exact += int((obs[t + 1] == co).sum()); n_vals += co.size
        assert np.array_equal(edone[T], c.env_done) and np.array_equal(env.winner.cpu().numpy(), c.winner)
        sh = {k: v.cpu().numpy() for k, v in env.export_state().items()}
        scx = c.export_state()
        for f in ("px", "py", "pdir", "php", "palive", "base_xy", "bhp", "tick", "env_done", "winner", "bl_live", "counters"):
            assert np.array_equal(sh[f], scx[f]), (run, f)
        m = scx["bl_live"].astype(bool)
        for f in ("bl_x", "bl_y", "bl_dir"):
            assert np.array_equal(sh[f][m], scx[f][m]), (run, f)
        acts_seen += torch.bincount(ro.scores.argmax(-1).flatten().cpu(), minlength=4)
    assert n_exact >= n_vals * (1 - 1e-3)
    cn = scx["counters"].sum(0)
    assert cn[0] >= 2 * E and cn[0] == cn[1] + cn[2] + cn[3]               # two full games per env and more
    assert int((acts_seen > 0.05 * acts_seen.sum()).sum()) >= 3 and int(acts_seen.min()) > 0   # varied play: every action occurs, three of them often
    # the actor itself, on observations these games visited: MFMA kernel == torch fp32 restatement of the reference forward
    with torch.no_grad():
        want = actor(ro.obs[T // 2])
    got = FusedActor(actor, n)(ro.obs[T // 2])
    torch.testing.assert_close(got, want, rtol=0, atol=2e-5)
    clear = (want.topk(2, -1).values[..., 0] - want.topk(2, -1).values[..., 1]) > 1e-4
    assert torch.equal(got.argmax(-1)[clear], want.argmax(-1)[clear])


@pytest.mark.parametrize("one_launch", [True, False])
def test_reference_evaluation_workload_vs_c_oracle(one_launch):
    """The reference's OWN evaluation workload (evaluate.py:32-76; README.md:30: the trained team wins "~80 %") on this path,
    closed against the oracle: 2v2, the shipped reward config cf.json (float rewards), red = the shipped checkpoints
    actor_plane0 / actor_plane1 (fixture g12 holds their weights) through the MFMA actor with Ornstein-Uhlenbeck noise of scale 0.1
    that is never restarted, blue = the scripted instinct opponent played in-kernel (one launch per 32 ticks) or by its own
    kernel (graph form).  Beside it the C oracle plays the SAME games from the same score vectors' arg-max: rewards, dones, game
    flags and the full state identical after every run, observations within 1e-5.  Then the tally: the device's red win rate
    against the one the unmodified evaluate.main() produced here (g12) and against the published figure."""
    from deep_rl_battlespace_amd.rollout import FusedActor, play_reference_evaluation, reference_checkpoint_actor
    g12 = np.load(os.path.join(GOLD, "g12_evaluation.npz"))
    n, E, T = int(g12["n_agents"]), 16384, 32
    cf = dict(zip(("hit_base_reward", "hit_plane_reward", "miss_punishment", "die_punishment", "lose_punishment"), (float(v) for v in g12["cf"])))
    actor = reference_checkpoint_actor(g12, n)
    # the stacked module and the MFMA kernel ARE the reference ActorNetwork.forward of the checkpoints, on rows met in the reference's own play
    x = torch.stack([torch.from_numpy(g12[f"plane{i % n}/x"]) for i in range(2 * n)], 1).cuda().contiguous()     # [rows, A, 8]
    want = torch.stack([torch.from_numpy(g12[f"plane{i % n}/y"]) for i in range(2 * n)], 1).cuda()
    # (float32, different summation orders; the trained weights are larger than g9's fresh ones: 5e-5 on a tanh output)
    with torch.no_grad():
        torch.testing.assert_close(actor(x), want, rtol=0, atol=5e-5)
    torch.testing.assert_close(FusedActor(actor, n)(x), want, rtol=0, atol=5e-5)
    clear = (want.topk(2, -1).values[..., 0] - want.topk(2, -1).values[..., 1]) > 2e-4
    assert torch.equal(FusedActor(actor, n)(x).argmax(-1)[clear], want.argmax(-1)[clear]) and float(clear.float().mean()) > 0.8   # (the trained policy saturates: some rows tie at tanh = 1)
    env = _env(n_agents=n, n_envs=E, seed=212, auto_reset=True, **cf)
    c = cref.CRefBatch(E, n_agents=n, seed=212, auto_reset=True, **cf)
    res = play_reference_evaluation(env, actor, games=1, T=T, one_launch=one_launch, seed=12)        # one rollout: sets everything up
    ro = res["rollout"]
    o_c = c.reset()
    np.testing.assert_allclose(ro.obs[0].cpu().numpy(), o_c, rtol=OBS_RTOL, atol=OBS_ATOL)
    blue = slice(n, 2 * n)

    def check_run(run):
        torch.cuda.synchronize()
        obs, sc, rew, done, edone = (v.cpu().numpy() for v in (ro.obs, ro.scores, ro.rew, ro._done, ro.env_done))
        assert set(np.unique(sc[:, :, blue])) <= {-1.0, 1.0}                # the scripted side's rows are one-hot (instinct/agent.py:56-62)
        for t in range(T):
            assert np.array_equal(edone[t], c.env_done), (run, t)
            co, cr, cd = c.step(sc[t])                                     # arg-max of the score rows (battle_env.py:327-328)
            assert np.array_equal(done[t].astype(bool), cd), (run, t)
            np.testing.assert_allclose(rew[t], cr, rtol=1e-6, atol=1e-6, err_msg=f"run {run} tick {t}")
            np.testing.assert_allclose(obs[t + 1], co, rtol=OBS_RTOL, atol=OBS_ATOL, err_msg=f"run {run} tick {t}")
        sh = {k: v.cpu().numpy() for k, v in env.export_state().items()}
        scx = c.export_state()
        for f in ("px", "py", "pdir", "php", "palive", "base_xy", "bhp", "tick", "env_done", "winner", "bl_live", "counters"):
            assert np.array_equal(sh[f], scx[f]), (run, f)
        return scx["counters"].sum(0)

    cn = check_run(0)
    for run in range(1, 7):                                                 # 224 ticks: every slot finishes a game (time limit: 141) and starts the next
        ro.run()
        cn = check_run(run)
    assert cn[0] >= E and cn[0] == cn[1] + cn[2] + cn[3]
    rate = cn[2] / cn[0]
    ref_rate = float(g12["red_wins"]) / float(g12["games"])
    print(f"evaluation workload: device {cn[0]} games, red wins {rate:.4f}; reference evaluate.main() {int(g12['games'])} games, {ref_rate:.4f}; README ~0.80")
    # binomial spread of the reference's own sample (a few thousand games) dominates; 4 sigma of it, plus the device's
    sigma = (ref_rate * (1 - ref_rate) * (1.0 / float(g12["games"]) + 1.0 / float(cn[0]))) ** 0.5
    assert abs(rate - ref_rate) < 4 * sigma + 0.01, (rate, ref_rate, sigma)
    assert 0.7 < rate < 0.9                                                 # README.md:30 "~80%"


def test_reference_evaluation_tally_with_the_scripts_quirk_is_within_two_sigma_of_evaluate_py():
    """VERDICT r4 item 3.  The unmodified evaluate.main() here: 3 018 games, red 82.57 % (fixture g12).  Round 4 quoted 84.6 % for the
    device and left the 2.9-sigma gap unexplained.  Two causes, both in how the games were SAMPLED, none in the step path: (1) all
    slots were stopped at one moment and the finished games counted -- each slot's unfinished game is dropped, the longer the likelier,
    and ties are the longest games; (2) the script feeds a game's first tick the observations of a discarded reset (evaluate.py:53-66).
    With the first k whole games of every slot tallied (_FirstGamesTally) and the quirk reproduced (first_tick_stale_obs) the device
    plays the reference's evaluation: within two standard deviations of its tally; without the quirk a little above it."""
    from deep_rl_battlespace_amd.rollout import play_reference_evaluation, reference_checkpoint_actor
    g12 = np.load(os.path.join(GOLD, "g12_evaluation.npz"))
    n, E, k = int(g12["n_agents"]), 32768, 2
    cf = dict(zip(("hit_base_reward", "hit_plane_reward", "miss_punishment", "die_punishment", "lose_punishment"), (float(v) for v in g12["cf"])))
    actor = reference_checkpoint_actor(g12, n)
    p_ref = float(g12["red_wins"]) / float(g12["games"])
    tie_ref = float(g12["ties"]) / float(g12["games"])
    got = {}
    for stale in (True, False):
        env = _env(n_agents=n, n_envs=E, seed=77, auto_reset=True, **cf)
        r = play_reference_evaluation(env, actor, games=0, T=4, one_launch=not stale, seed=5, first_tick_stale_obs=stale, games_per_slot=k)
        r.pop("rollout")
        assert k * E <= r["games"] <= k * E + E // 50 and r["games"] == r["ties"] + r["red_wins"] + r["blue_wins"]   # exactly k per slot (+ the rare two-at-once call)
        sigma = (p_ref * (1 - p_ref) * (1.0 / float(g12["games"]) + 1.0 / r["games"])) ** 0.5
        got[stale] = (r["win_rate_red"], (r["win_rate_red"] - p_ref) / sigma, r["ties"] / r["games"])
    print(f"evaluation tally: with the script's stale first observation {got[True][0]:.4f} ({got[True][1]:+.2f} sigma), every tick on its own "
          f"game {got[False][0]:.4f} ({got[False][1]:+.2f} sigma); evaluate.main() {p_ref:.4f}; ties {got[True][2]:.4f} vs {tie_ref:.4f}")
    assert abs(got[True][1]) < 2.0, got
    assert abs(got[False][1]) < 3.0 and got[False][0] > got[True][0] - 0.004, got     # the quirk costs red a little: its first action is blind
    assert abs(got[True][2] - tie_ref) < 0.012, got


@pytest.mark.parametrize("n", [1, 2])
def test_categorical_policy_head_and_value_head_vs_torch(n):
    """The policy-gradient heads next to the reference's deterministic-plus-noise one (BASELINE.json configs[4] words C5 as a PPO
    rollout): with injected uniforms the kernel's Gumbel-max row, the drawn action and its log-probability equal a torch fp32
    restatement (softmax(scores / T), g = -log(-log u)); the value head equals the torch forward of a second MLP of the same shape;
    and with the kernel's own Philox draws the action frequencies on one observation row follow softmax(scores / T) (chi-square)."""
    from deep_rl_battlespace_amd.rollout import FusedActor, StackedActor
    torch.manual_seed(31)
    E, A, D, tau = 4096, 2 * n, 3 * n + 2, 0.5
    actor, critic = StackedActor(A, D, 4, device="cuda"), StackedActor(A, D, 1, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(60.0); critic.w3.mul_(100.0)                              # spread the fresh heads (0.003-uniform) so that rows differ
    obs = (torch.rand((E, A, D), device="cuda") * 2 - 1).contiguous()
    u = (torch.rand((E, A, 4), device="cuda") * 0.998 + 0.001).contiguous()
    fa = FusedActor(actor, n, seed=9)
    scores = torch.empty((E, A, 4), device="cuda"); lp = torch.zeros((E, A), device="cuda"); val = torch.zeros((E, A), device="cuda")
    fa.forward_into(obs, scores, sample=dict(temperature=tau, logp=lp, u=u), value=dict(weights=critic.pack(), out=val))
    with torch.no_grad():
        z = actor(obs) / tau
        want = z - torch.log(-torch.log(u))
        a = want.argmax(-1)
        want_lp = torch.log_softmax(z, -1).gather(-1, a[..., None])[..., 0]
        want_v = critic(obs, squash=False)[..., 0]
    torch.testing.assert_close(scores, want, rtol=0, atol=2e-4)
    top = want.topk(2, -1).values
    clear = (top[..., 0] - top[..., 1]) > 2e-3
    assert torch.equal(scores.argmax(-1)[clear], a[clear]) and float(clear.float().mean()) > 0.99
    torch.testing.assert_close(lp[clear], want_lp[clear], rtol=0, atol=2e-4)
    torch.testing.assert_close(val, want_v, rtol=0, atol=5e-5)
    # the kernel's own draws: one observation row for every game, 4096 x A draws -> frequencies against softmax(z)
    obs1 = obs[:1].expand(E, -1, -1).contiguous()
    fa.forward_into(obs1, scores, sample=dict(temperature=tau, logp=lp))
    with torch.no_grad():
        pr = torch.softmax(actor(obs1[:1]) / tau, -1)[0]                        # [A, 4]
    for i in range(A):
        cnt = torch.bincount(scores[:, i].argmax(-1), minlength=4).double()
        exp = pr[i].double() * E
        keep = exp > 5
        chi2 = float((((cnt - exp) ** 2) / exp)[keep].sum())
        assert chi2 < 25.0, (i, cnt.tolist(), exp.tolist(), chi2)               # 3 degrees of freedom: p(chi2 > 25) ~ 1.5e-5
        got_lp = lp[:, i]
        want_row = torch.log(pr[i])[scores[:, i].argmax(-1)]
        torch.testing.assert_close(got_lp, want_row.float(), rtol=0, atol=2e-4)
    # a second call with another sequence number draws differently
    s2 = torch.empty_like(scores)
    fa.forward_into(obs1, s2, sample=dict(temperature=tau))
    assert float((s2.argmax(-1) != scores.argmax(-1)).float().mean()) > 0.05


@pytest.mark.parametrize("n,one_launch,precision", [(1, True, "f32"), (1, False, "f32"), (2, False, "f32"), (1, True, "bf16x3")])
def test_categorical_rollout_records_logp_and_value_and_one_launch_equals_the_graph(n, one_launch, precision):
    """PolicyRollout(sample='categorical', value_actor=...): T ticks of (draw an action from softmax(scores / T), V(obs), step) with
    logp / value records [T, E, A]; the step takes exactly the drawn action (its arg-max of the perturbed row), so the C oracle
    stepping on the recorded score rows plays the same games; at 1v1 the one-launch form equals the per-tick graph bit for bit."""
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    E, T, A, D = 8192, 16, 2 * n, 3 * n + 2
    torch.manual_seed(5)
    actor, critic = StackedActor(A, D, 4, device="cuda"), StackedActor(A, D, 1, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(60.0); critic.w3.mul_(100.0)

    def play(ol):
        env = _env(n_agents=n, n_envs=E, seed=41, auto_reset=True)
        env.reset()
        ro = PolicyRollout(env, actor, T, seed=3, one_launch=ol, sample="categorical", temperature=0.7, value_actor=critic, precision=precision)
        ro.start(); ro.capture()
        ro.run(); ro.run()
        torch.cuda.synchronize()
        return env, ro
    env, ro = play(one_launch)
    c = cref.CRefBatch(E, n_agents=n, seed=41, auto_reset=True)
    c.reset()
    env_b, ro_b = play(False) if one_launch else (env, ro)
    if one_launch:
        for f in ("obs", "scores", "rew", "_done", "logp", "value", "env_done"):
            assert torch.equal(getattr(ro, f), getattr(ro_b, f)), f
    # the log-probabilities are those of the arg-max of the recorded rows under softmax(actor(obs) / T); values are the critic's
    with torch.no_grad():
        z = actor(ro.obs[3]) / 0.7
        a = ro.scores[3].argmax(-1)
        want_lp = torch.log_softmax(z, -1).gather(-1, a[..., None])[..., 0]
        want_v = critic(ro.obs[3], squash=False)[..., 0]
    tol = 1.0 if precision == "f32" else 40.0                                   # bf16x3: ~1e-5 on a tanh score, ~1e-3 on the (unsquashed, x100) value head
    torch.testing.assert_close(ro.logp[3], want_lp, rtol=0, atol=3e-4 * tol)
    torch.testing.assert_close(ro.value[3], want_v, rtol=0, atol=5e-5 * tol)
    assert float(ro.logp.max()) <= 0.0 and float(ro.logp.min()) > -20.0
    acts = torch.bincount(ro.scores.argmax(-1).flatten(), minlength=4)
    assert int(acts.min()) > 0.02 * int(acts.sum())                              # a stochastic policy: every action is drawn


def test_ou_noise_reproduces_the_reference_trajectory():
    """g11: utils/noise.py OUNoise(4) run unmodified -- 40 noise() calls with the np.random.randn values it drew, a reset()
    (main.py:155) and a re-scale (main.py:154) on the way.  The in-kernel process (bsx_actor_forward) is fed the same normals
    (BsxActorNoise.z_inject) with an all-zero actor, so its scores ARE scale * state: state and noise must follow the
    reference's binary64 trajectory to float32 accuracy, for every row."""
    from deep_rl_battlespace_amd.rollout import FusedActor, StackedActor
    g = np.load(os.path.join(GOLD, "g11_ou_noise.npz"))
    scale, mu, theta, sigma = (float(v) for v in g["params"])
    events = {int(e[0]): (int(e[1]), float(e[2])) for e in g["events"]}
    n, E = 2, 37
    A, D = 2 * n, 3 * n + 2
    actor = StackedActor(A, D, 4, device="cuda")
    with torch.no_grad():
        for p in (actor.w1, actor.b1, actor.w2, actor.b2, actor.w3, actor.b3):
            p.zero_()                                            # tanh(0) = 0: the score rows are the noise alone
    fused = FusedActor(actor, n, seed=1)
    obs = torch.rand((E, A, D), device="cuda")
    st = torch.full((E, A, 4), mu, device="cuda")
    out = torch.empty((E, A, 4), device="cuda")
    done = torch.zeros(E, dtype=torch.uint8, device="cuda")
    for t in range(g["z"].shape[0]):
        done.zero_()
        if t in events:
            kind, val = events[t]
            if kind == 1:
                done.fill_(1)                                    # reset_noise(): rows of finished games restart from mu
            else:
                scale = val
        z = torch.from_numpy(g["z"][t]).float().cuda().expand(E, A, 4).contiguous()
        fused.forward_into(obs, out, 0.0, seq=t, ou=dict(scale=scale, state=st, env_done=done, theta=theta, sigma=sigma, mu=mu), z=z)
        want_state = torch.from_numpy(g["state"][t]).float().cuda().expand(E, A, 4)
        want_noise = torch.from_numpy(g["noise"][t]).float().cuda().expand(E, A, 4)
        torch.testing.assert_close(st, want_state, rtol=2e-6, atol=2e-7, msg=f"state, call {t}")
        torch.testing.assert_close(out, want_noise.clamp(-1, 1), rtol=2e-6, atol=2e-7, msg=f"noise, call {t}")
    # Gaussian term with injected normals: scores = std * z exactly as maddpg/agent.py:30-31 adds and clamps
    z = torch.randn((E, A, 4), device="cuda")
    fused.forward_into(obs, out, 0.7, seq=0, z=z)
    torch.testing.assert_close(out, (0.7 * z).clamp(-1, 1), rtol=1e-6, atol=1e-7)


def test_replay_ring_reproduces_the_reference_buffer():
    """g10 (maddpg/buffer.py run unmodified; see trace_util.check_replay_against_reference) with the ring in HBM."""
    from trace_util import check_replay_against_reference
    check_replay_against_reference("cuda")


@pytest.mark.parametrize("n", [1, 2, 3, 4])
@pytest.mark.parametrize("noise", ["none", "gaussian", "ou-bf16x3"])
def test_one_launch_continuous_rollout_equals_two_kernel_rollout(noise, n):
    """bsx_rollout_continuous (T ticks of actor -> continuous step in ONE launch) against bsx_actor_forward +
    bsx_step_continuous per tick (BSX_ACT_F32X4 rows): the same transitions bit for bit across runs and auto-resets, and both
    equal a plain env stepped with the float32 [E, A, 3] actions the actors produced (battle_env.py:295-297,418-424)."""
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    E, T = (4100 if n == 1 else 1030), 40
    torch.manual_seed(13)
    actor = StackedActor(2 * n, 3 * n + 2, 3, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(60.0); actor.g1.uniform_(0.5, 1.5); actor.h1.uniform_(-0.3, 0.3)
    kw = dict(noise_std=0.3) if noise == "gaussian" else (dict(ou_scale=0.4, precision="bf16x3") if noise.startswith("ou") else {})
    ros = []
    for one in (False, True):
        env = _env(n_agents=n, n_envs=E, seed=31, auto_reset=True, continuous_actions=True); env.reset()
        ro = PolicyRollout(env, actor, T, seed=7, one_launch=one, **kw); ro.start()
        if one:
            ro.capture()
        ros.append(ro)
    twin = _env(n_agents=n, n_envs=E, seed=31, auto_reset=True, continuous_actions=True); twin.reset()
    a, b = ros
    for rep in range(5):
        a.run(); b.run()
        torch.cuda.synchronize()
        assert torch.equal(a.obs, b.obs), rep
        assert torch.equal(a.scores, b.scores), rep
        assert torch.equal(a.rew, b.rew) and torch.equal(a.done, b.done) and torch.equal(a.env_done, b.env_done), rep
        if noise.startswith("ou"):
            assert torch.equal(a.ou["state"], b.ou["state"]), rep
        for t in range(T):
            o, r, d = twin.step_batch(b.scores[t][..., :3].contiguous())
            assert torch.equal(o, b.obs[t + 1]) and torch.equal(r, b.rew[t]) and torch.equal(d, b.done[t]), (rep, t)
    sa, sb = a.env.export_state(), b.env.export_state()
    for k in ("px", "py", "pdir", "php", "bhp", "tick", "env_done", "winner", "bl_live", "bl_x", "bl_y", "counters"):
        assert torch.equal(sa[k], sb[k]), k
    assert int(a.env.counters()[:, 0].sum()) >= E                # the runs crossed game ends
    assert float(b.scores[..., 2].max()) > 0 and float(b.scores[..., 2].min()) < 0      # some planes fire, some do not


@pytest.mark.parametrize("n,scripted", [(1, "blue"), (2, "blue"), (4, "red")])
def test_one_launch_continuous_rollout_plays_the_scripted_opponent_in_kernel(n, scripted):
    """The reference's only continuous driver is instinct-vs-instinct (test_env.py:22-43); a learned team against the continuous
    scripted opponent (instinct/agent.py:41-54) hands the env float32 rows from the actors next to float64 rows from the script.
    Per-tick form: actor kernel -> the actors' rows widened into one float64 [E, A, 3] array -> bsx_instinct_continuous fills its
    team's rows (Philox: its seed, sequence number tick + replay counter) -> bsx_step_continuous(BSX_ACT_F64).  One-launch form:
    bsx_rollout_continuous with scripted_team: that team's actor tiles are skipped, its binary64 actions are computed from the LDS
    observation rows with the same draws and reach the step unrounded.  Same transitions bit for bit, over runs and auto-resets."""
    from deep_rl_battlespace_amd import instinct
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    E, T = (4100 if n == 1 else 1030), 40
    torch.manual_seed(17)
    actor = StackedActor(2 * n, 3 * n + 2, 3, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(60.0)
    ros = []
    for one in (False, True):
        env = _env(n_agents=n, n_envs=E, seed=33, auto_reset=True, continuous_actions=True); env.reset()
        opp = instinct.Team(env.possible_blue, env.possible_red, env, seed=91) if scripted == "blue" else instinct.Team(env.possible_red, env.possible_blue, env, seed=91)
        ro = PolicyRollout(env, actor, T, seed=7, noise_std=0.2, one_launch=one, opponent=opp); ro.start(); ro.capture()
        ros.append(ro)
    a, b = ros
    cols = a.opponent._cols
    for rep in range(5):
        a.run(); b.run()
        torch.cuda.synchronize()
        assert torch.equal(a.obs, b.obs), rep
        assert torch.equal(a.scores, b.scores), rep
        assert torch.equal(a.rew, b.rew) and torch.equal(a.done, b.done) and torch.equal(a.env_done, b.env_done), rep
    sa, sb = a.env.export_state(), b.env.export_state()
    for k in ("px", "py", "pdir", "php", "bhp", "tick", "env_done", "winner", "bl_live", "bl_x", "bl_y", "bl_dir", "counters"):
        assert torch.equal(sa[k], sb[k]), k
    c = a.env.counters().sum(0)
    assert c[0] >= E and (c[2] + c[3]) > 0.2 * c[0]               # the runs crossed game ends, and a share of them were decided by play
    sc = b.scores[:, :, cols]
    assert float(sc[..., 3].abs().max()) == 0.0 and float(sc[..., :3].abs().max()) <= 1.0
    assert 0.02 < float((sc[..., 2] > 0).float().mean()) < 0.9    # the script fires inside its cone with probability 0.6, not otherwise
    # draws differ from run to run (the replay counter is part of the key): the same tick of two runs is not the same row
    x1 = b.scores[5][:, cols].clone(); b.run(); torch.cuda.synchronize()
    assert not torch.equal(x1, b.scores[5][:, cols])


@pytest.mark.parametrize("n,cont", [(1, False), (3, False), (2, True)])
def test_one_launch_rollout_is_the_same_through_narrow_and_wide_offset_kernels(n, cont):
    """The fused rollout kernels exist in both offset widths as well (32-bit offsets while the job's arrays stay below 4 GB, 64-bit
    above or with wide_offsets=True): same observations, scores, rewards, dones and final state."""
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    E, T = 2100, 24
    torch.manual_seed(3)
    actor = StackedActor(2 * n, 3 * n + 2, 3 if cont else 4, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(40.0)
    ros = []
    for wide in (False, True):
        env = _env(n_agents=n, n_envs=E, seed=11, auto_reset=True, continuous_actions=cont, wide_offsets=wide); env.reset()
        ro = PolicyRollout(env, actor, T, seed=5, noise_std=0.4, one_launch=True); ro.start(); ro.capture()
        ros.append(ro)
    a, b = ros
    for rep in range(8):                                        # 192 ticks: past the time limit of every team size here
        a.run(); b.run()
        torch.cuda.synchronize()
        assert torch.equal(a.obs, b.obs) and torch.equal(a.scores, b.scores) and torch.equal(a.rew, b.rew), rep
        assert torch.equal(a.done, b.done) and torch.equal(a.env_done, b.env_done), rep
    sa, sb = a.env.export_state(), b.env.export_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert int(a.env.counters()[:, 0].sum()) >= E


@pytest.mark.parametrize("one_launch", [False, True])
def test_exploration_noise_does_not_depend_on_the_sharding(one_launch):
    """A job split over ranks (sharding.make_shard: env_offset) explores exactly as the unsplit job: the actor's Philox draws are
    keyed by the GLOBAL row, like the env's own jitter and spawn draws -- Gaussian noise and the OU process alike."""
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    E, n, T, parts = 2048, 2, 24, 4
    torch.manual_seed(2)
    actor = StackedActor(2 * n, 3 * n + 2, 4, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(60.0)

    def play(E_, off):
        env = _env(n_agents=n, n_envs=E_, seed=11, auto_reset=True, env_offset=off); env.reset()
        ro = PolicyRollout(env, actor, T, noise_std=0.3, ou_scale=0.2, seed=9, one_launch=one_launch); ro.start()
        for _ in range(8):
            ro.run()
        torch.cuda.synchronize()
        return ro
    whole = play(E, 0)
    q = E // parts
    pieces = [play(q, i * q) for i in range(parts)]
    for name in ("obs", "scores", "rew", "done", "env_done"):
        assert torch.equal(getattr(whole, name), torch.cat([getattr(p, name) for p in pieces], dim=1)), name
    assert torch.equal(whole.ou["state"], torch.cat([p.ou["state"] for p in pieces]))
    assert int(whole.env.counters()[:, 0].sum()) > 0


@pytest.mark.parametrize("n,cont,chains,heads", [(1, False, 2, "ppo"), (2, False, 3, "noise"), (4, False, 2, "noise"), (2, True, 2, "noise"), (3, False, 4, "ppo")])
def test_chained_rollout_records_the_same_transitions(n, cont, chains, heads):
    """PolicyRollout(chains=P): the games as P ranges, each a chain of (actor -> step) launch pairs on its own branch of the graph --
    the transitions of chains=1 bit for bit (observations, scores, rewards, flags, OU state, log-probabilities, values), eagerly and
    as a captured graph."""
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    E, T = 1100, 20
    torch.manual_seed(n)
    actor = StackedActor(2 * n, 3 * n + 2, 3 if cont else 4, device="cuda")
    critic = StackedActor(2 * n, 3 * n + 2, 1, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(60.0)
    kw = dict(sample="categorical", temperature=0.7, value_actor=critic) if heads == "ppo" else dict(noise_std=0.3, ou_scale=0.2)

    def play(P, graph):
        env = _env(n_agents=n, n_envs=E, seed=17, auto_reset=True, continuous_actions=cont); env.reset()
        ro = PolicyRollout(env, actor, T, seed=9, chains=P, **kw); ro.start()
        if graph:
            ro.capture()
        for _ in range(9):
            ro.run()
        torch.cuda.synchronize()
        return ro
    one = play(1, False)
    for graph in (False, True):
        ch = play(chains, graph)
        for name in ("obs", "scores", "rew", "done", "env_done") + (("logp", "value") if heads == "ppo" else ()):
            assert torch.equal(getattr(one, name), getattr(ch, name)), (name, graph)
        if heads != "ppo":
            assert torch.equal(one.ou["state"], ch.ou["state"])
        sa, sb = one.env.export_state(), ch.env.export_state()
        assert all(torch.equal(sa[k], sb[k]) for k in sa) and torch.equal(one.env._env_done, ch.env._env_done)
    assert int(one.env.counters()[:, 0].sum()) > 0
    with pytest.raises(ValueError):
        PolicyRollout(one.env, actor, T, chains=2, one_launch=True)
    assert PolicyRollout(one.env, actor, T, chains="auto").chains == 1 and PolicyRollout(one.env, actor, T, chains="auto", one_launch=n <= 4).chains == 1   # a batch this small: one chain


@pytest.mark.parametrize("n", [6, 8])
def test_one_launch_beyond_4v4_falls_through_to_the_graph(n):
    """f-1 beyond 4v4 (main.py:177-181 with larger teams): the fused one-launch kernels cover 1v1 ... 4v4; asked for more,
    PolicyRollout(one_launch=True) runs the two-kernel graph instead of refusing, says so in `form_note`, and plays exactly the
    transitions of the graph form asked for directly."""
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    E, T = 1100, 24
    torch.manual_seed(21)
    actor = StackedActor(2 * n, 3 * n + 2, 4, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(80.0)
    ros = []
    for one in (True, False):
        env = _env(n_agents=n, n_envs=E, seed=9, auto_reset=True); env.reset()
        if one:                                             # the form asked for is not the one that runs: the constructor warns
            with pytest.warns(RuntimeWarning, match="two-kernel graph"):
                ro = PolicyRollout(env, actor, T, seed=4, noise_std=0.3, one_launch=True)
        else:
            ro = PolicyRollout(env, actor, T, seed=4, noise_std=0.3, one_launch=False)
        ro.start(); ro.capture()
        ros.append(ro)
    a, b = ros
    assert a.one_launch is False and "two-kernel graph" in a.form_note and b.form_note is None
    for rep in range(13):                                   # past the time-limit ties (221 / 261 ticks at 6v6 / 8v8) and the re-spawns behind them
        a.run(); b.run()
        torch.cuda.synchronize()
        for name in ("obs", "scores", "rew", "done", "env_done"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (name, rep)
    sa, sb = a.env.export_state(), b.env.export_state()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    assert int(a.env.counters()[:, 0].sum()) >= E


def test_exported_game_state_renders_what_the_oracle_holds(tmp_path):
    """f-4 (battle_env.py:498-560 draws planes, bases and bullets of ONE game): the state block of a running batch is exported,
    one game of it equals the oracle's game field by field, and its host-side image shows every live sprite where the state
    says it is (team colour at the centre of each live plane and base, bullet pixels at live bullets)."""
    from deep_rl_battlespace_amd import render
    E, n = 300, 2
    env = _env(n_agents=n, n_envs=E, seed=4, auto_reset=True); env.reset()
    c = cref.CRefBatch(E, n_agents=n, seed=4, auto_reset=True); c.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for t in range(45):
        a = torch.where(torch.rand((E, 2 * n), generator=g, device="cuda") < 0.5, 1, torch.randint(0, 4, (E, 2 * n), generator=g, device="cuda")).to(torch.int32)
        env.step_batch(a); c.step(a.cpu().numpy())
    st = {k: v.cpu().numpy() for k, v in env.export_state().items()}
    sc = c.export_state()
    live = sc["bl_live"].astype(bool)
    e = int(np.argmax(live.reshape(E, -1).sum(1)))               # the game with most bullets in flight
    one = {k: v[e] for k, v in st.items()}
    for f in ("px", "py", "pdir", "php", "base_xy", "bhp", "bl_live"):
        assert np.array_equal(one[f], sc[f][e]), f
    img = render.frame(env, e)                                   # device -> host copy of game e, rasterised
    want = render.frame_from_state({k: sc[k][e] for k in ("px", "py", "pdir", "php", "base_xy", "bhp", "bl_live", "bl_x", "bl_y")}, n)
    assert img.shape == (800, 1200, 3) and np.array_equal(img, want)          # the same picture as the oracle's game
    inked = lambda y, x: tuple(img[min(max(int(y), 0), 799), min(max(int(x), 0), 1199)]) != render.WHITE    # noqa: E731
    for i in range(2 * n):
        assert inked(one["py"][i] - 23, one["px"][i] - 24)      # a corner of the 50 x 48 hit box: filled when alive, outlined when dead
        for k in range(12):
            if one["bl_live"][i, k]:
                assert inked(one["bl_y"][i, k], one["bl_x"][i, k])
    bx = one["base_xy"]
    assert inked(bx[1] - 30, bx[0] - 30) and inked(bx[3] - 30, bx[2] - 30)
    render.save_ppm(tmp_path / "game.ppm", img)
    assert int(live[e].sum()) >= 3 and (tmp_path / "game.ppm").stat().st_size > 800 * 1200 * 3


def test_checkpoint_restores_host_mirrors_and_checks_its_shape():
    """state_dict / load_state_dict in drop-in mode: after a restore the reference-typed surface (`agents`, `dones`, `env_done`,
    `winner`) is what it was at the snapshot -- the next step() draws its random() values for the right planes -- and a
    snapshot of another job shape is refused."""
    import random
    random.seed(5)
    env = _env(n_agents=2)
    env.reset()
    ids = env.possible_agents
    # play until a plane has died but the game is still on, snapshot there
    snap = None
    for t in range(400):
        if env.env_done:
            env.reset()
        obs, rew, dones, _ = env.step({a: 1 if t % 3 else 2 for a in env.agents})
        if not env.env_done and len(env.agents) < len(ids) and snap is None:
            snap = (env.state_dict(), list(env.agents), dict(env.dones), env.winner, random.getstate())
            break
    assert snap is not None, "no plane died in 400 calls of shoot-heavy play"
    sd, agents, dones, winner, rstate = snap
    trace = []
    for t in range(30):
        if env.env_done:
            break
        o, r, d, _ = env.step({a: 1 for a in env.agents})
        trace.append(({k: v.copy() for k, v in o.items()}, dict(r), dict(d)))
    env.load_state_dict(sd)
    random.setstate(rstate)
    assert env.agents == agents and env.dones == dones and env.winner == winner and env.env_done is False
    for t, (o0, r0, d0) in enumerate(trace):
        o, r, d, _ = env.step({a: 1 for a in env.agents})
        assert r == r0 and dict(d) == d0 and all(np.array_equal(o[k], o0[k]) for k in o), t
    other = _env(n_agents=1)
    with pytest.raises(ValueError):
        other.load_state_dict(sd)
