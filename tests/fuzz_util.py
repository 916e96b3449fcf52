"""The randomised differential case generator and checker behind tools/fuzz_parity.py and tests/test_hip_fuzz.py: one random
configuration of the HIP step() path (team size, ragged batch, action encoding, rewards, env_offset, resets, jitter source, offset
width, launch form) played against the C oracle -- test infrastructure, like everything that touches oracle/."""
import numpy as np
import torch

import deep_rl_battlespace_amd as bsx
from oracle import cref

STATE = ("px", "py", "pdir", "php", "palive", "base_xy", "bhp", "tick", "env_done", "winner", "bl_live", "counters")


def draw_case(rng, max_envs=6000):
    n = int(rng.choice([1, 1, 1, 2, 2, 3, 4, 4, 5, 6, 8, 12, 16]))
    E = int(rng.choice([rng.integers(1, 40), rng.integers(40, max(41, max_envs // max(1, n // 2))),
                        256 * rng.integers(1, 9) + rng.integers(-1, 2)]))
    cont = bool(rng.random() < 0.35)
    enc = str(rng.choice(["f32", "f64", "f32x4"])) if cont else str(rng.choice(["int", "int", "scores"]))
    form = str(rng.choice(["step", "step", "many", "graph", "chains"]))
    auto = bool(rng.random() < 0.75)
    host_u = bool(form == "step" and rng.random() < 0.3)
    case = dict(n=n, E=max(1, E), cont=cont, enc=enc, form=form, auto_reset=auto, host_u=host_u,
                wide=bool(rng.random() < 0.3), env_offset=int(rng.choice([0, 0, 777, 2 ** 33 + 5])),
                seed=int(rng.integers(0, 2 ** 31)), T=int(rng.integers(12, 26)) * 10, K=int(rng.choice([5, 10, 30])),
                chains=int(rng.choice([2, 3, 4, 7])), p_shoot=float(rng.choice([0.25, 0.5, 0.8])),
                resume=bool(form in ("step", "many") and rng.random() < 0.3),    # checkpoint mid-run, throw the env away, continue in a new one
                rewards=[int(v) for v in (rng.integers(50, 150), rng.integers(1, 20), -rng.integers(0, 4), -rng.integers(0, 9), -rng.integers(0, 30))]
                if rng.random() < 0.5 else [100, 10, -1, -5, -20])
    return case


def actions_for(case, T, gen):
    E, A = case["E"], 2 * case["n"]
    if case["cont"]:
        a = torch.rand((T, E, A, 3), generator=gen, device="cuda", dtype=torch.float64) * 2.6 - 1.3      # beyond [-1, 1]: clipped in-kernel
        a[..., 2] += case["p_shoot"] - 0.5
        if case["enc"] == "f64":
            return a.contiguous()
        a = a.to(torch.float32)
        if case["enc"] == "f32x4":
            a = torch.cat([a, torch.full((T, E, A, 1), 9.0, device="cuda")], -1)
        return a.contiguous()
    if case["enc"] == "scores":
        a = torch.randn((T, E, A, 4), generator=gen, device="cuda")
        a[..., 1] += 3.0 * (case["p_shoot"] - 0.25)
        return a.contiguous()
    a = torch.randint(0, 4, (T, E, A), generator=gen, device="cuda", dtype=torch.int32)
    a = torch.where(torch.rand((T, E, A), generator=gen, device="cuda") < case["p_shoot"], torch.ones_like(a), a)
    a = torch.where(torch.rand((T, E, A), generator=gen, device="cuda") < 0.01, torch.full_like(a, 7), a)   # out of range: the plane stays
    return a.contiguous()


def oracle_actions(case, a_t):
    a = a_t.cpu().numpy()
    if case["enc"] == "f32x4":
        a = np.ascontiguousarray(a[..., :3])
    return a


def run_case(case):
    n, E, T = case["n"], case["E"], case["T"]
    kw = dict(n_agents=n, seed=case["seed"], auto_reset=case["auto_reset"], env_offset=case["env_offset"], continuous_actions=case["cont"],
              hit_base_reward=case["rewards"][0], hit_plane_reward=case["rewards"][1], miss_punishment=case["rewards"][2],
              die_punishment=case["rewards"][3], lose_punishment=case["rewards"][4])
    env = bsx.parallel_env(n_envs=E, wide_offsets=case["wide"], **kw)
    c = cref.CRefBatch(E, **kw)
    env.reset(); c.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(case["seed"] ^ 0x5EED)
    form, K = case["form"], case["K"]
    T -= T % K
    acts = actions_for(case, T, gen)
    u_all = torch.rand((T, E, 2 * n), generator=gen, device="cuda", dtype=torch.float64) if case["host_u"] else None
    stats = dict(vals=0, exact=0)
    graph = None
    if form in ("graph", "chains"):
        g_act = torch.zeros_like(acts[:K])
        graph, g_out = env.capture_steps(g_act, store=True, chains=case["chains"] if form == "chains" else 1)

    def check_state(t):
        sh = {k: v.cpu().numpy() for k, v in env.export_state().items()}
        sc = c.export_state()
        for f in STATE:
            if not np.array_equal(sh[f], sc[f]):
                return f"state {f} after call {t}"
        m = sc["bl_live"].astype(bool)
        for f in ("bl_x", "bl_y", "bl_dir"):
            if not np.array_equal(sh[f][m], sc[f][m]):
                return f"state {f} after call {t}"
        return None

    for t0 in range(0, T, K):
        if case.get("resume") and t0 // K == 2:
            sd = env.state_dict()
            env = bsx.parallel_env(n_envs=E, wide_offsets=case["wide"], **kw)      # (never reset: everything comes from the snapshot)
            env.load_state_dict(sd)
        if form == "step":
            outs = None
        elif form == "many":
            outs = env.step_many(acts[t0:t0 + K], store=True)
        else:
            g_act.copy_(acts[t0:t0 + K]); graph.replay()
            outs = g_out
        for k in range(K):
            t = t0 + k
            if form == "step":
                a_t = acts[t][..., :3].contiguous() if case["enc"] == "f32x4" else acts[t]     # (4-wide rows are a T-call encoding: an actor's output buffer)
                obs, rew, done = env.step_batch(a_t, u=u_all[t] if u_all is not None else None)
            else:
                obs, rew, done = outs[0][k], outs[1][k], outs[2][k]
            co, cr, cd = c.step(oracle_actions(case, acts[t]), u=u_all[t].cpu().numpy() if u_all is not None else None)
            o = obs.cpu().numpy()
            if not np.array_equal(done.cpu().numpy().astype(bool), cd):
                return f"done at call {t}", stats
            if not np.array_equal(rew.cpu().numpy().astype(np.float64), cr):
                return f"rew at call {t}", stats
            diff = np.abs(o.astype(np.float64) - co)
            if ((diff > 1e-7) & (diff / np.maximum(np.abs(co), 1e-30) > 1e-5)).any():
                return f"obs at call {t}", stats
            stats["vals"] += o.size; stats["exact"] += int((o == co).sum())
        if form == "step" and not np.array_equal(env.env_done.cpu().numpy(), c.env_done.astype(bool)):
            return f"env_done after call {t0 + K - 1}", stats
        if (t0 // K) % 4 == 3 or t0 + K >= T:
            bad = check_state(t0 + K - 1)
            if bad:
                return bad, stats
        if not case["auto_reset"] and (t0 // K) % 3 == 2:          # re-spawn the finished games by hand (battle_env.py:246-279), both sides
            mask = c.env_done.astype(bool).copy()
            if mask.any():
                env.reset(mask=torch.from_numpy(mask).cuda()); c.reset(mask=mask)
    return None, stats


# ---------------------------------------------------------------------------------------------- the drop-in surface (one game)
def draw_dropin_case(rng):
    """One game behind the reference's own surface (n_envs=None: dict actions, numpy rows, Python numbers, stdlib `random` draws in the
    reference's order) against the Python oracle on the same random stream."""
    n = int(rng.choice([1, 1, 2, 2, 3, 4, 6]))
    cont = bool(rng.random() < 0.35)
    return dict(n=n, cont=cont, seed=int(rng.integers(0, 2 ** 31)), T=int(rng.integers(150, 420)), p_shoot=float(rng.choice([0.25, 0.5, 0.8])),
                encoding=str(rng.choice(["int", "int", "vector"])) if not cont else str(rng.choice(["f64", "f32"])),
                all_agents=bool(rng.random() < 0.5), p_empty=float(rng.choice([0.0, 0.0, 0.01])),
                rewards=[[100, 10, -1, -5, -20], [1.0, 0.9, -0.02, -0.03, -0.05],      # the reference's defaults / its training config (main.py:33-37)
                         [float(v) for v in (rng.integers(50, 150) + 0.5, rng.integers(1, 20) + 0.25, -0.5 * rng.integers(1, 4), -1.5 * rng.integers(1, 6),
                                             -float(rng.integers(1, 30)))]][int(rng.integers(0, 3))])


def _dropin_actions(case, T):
    """The action dicts of T calls, as (list of {agent index: value}, list of bool 'empty call'); drawn once, fed to both sides."""
    g = np.random.default_rng(case["seed"] ^ 0xAC7)
    A = 2 * case["n"]
    out = []
    for _ in range(T):
        if g.random() < case["p_empty"]:
            out.append(None)
            continue
        row = {}
        for i in range(A):
            if case["cont"]:
                v = g.random(3) * 2.6 - 1.3
                v[2] += case["p_shoot"] - 0.5
                row[i] = v.astype(np.float32) if case["encoding"] == "f32" else v
            elif case["encoding"] == "vector":
                v = g.standard_normal(4).astype(np.float32)
                v[1] += 3.0 * (case["p_shoot"] - 0.25)
                row[i] = v
            else:
                a = int(g.integers(0, 4))
                if g.random() < case["p_shoot"]:
                    a = 1
                if g.random() < 0.01:
                    a = int(g.choice([-1, 4, 7]))                  # out of range: the plane stays where it is
                row[i] = a
        out.append(row)
    return out


def _play(env, case, acts):
    """Reset, then T calls; a finished game is reset by hand as the reference's loops do (main.py:166-181).  Every return value."""
    ids = env.possible_agents
    rec = []
    obs = env.reset()
    rec.append(("reset", np.stack([np.asarray(obs[i]) for i in ids])))
    for a in acts:
        if env.env_done:
            obs = env.reset()
            rec.append(("reset", np.stack([np.asarray(obs[i]) for i in ids])))
        if a is None:
            d = {}
        else:
            live = ids if case["all_agents"] else list(env.agents)
            d = {i: (np.array(a[ids.index(i)], copy=True) if isinstance(a[ids.index(i)], np.ndarray) else a[ids.index(i)]) for i in live}
        obs, rew, done, _ = env.step(d)
        rec.append(("step", np.stack([np.asarray(obs[i]) for i in ids]), [rew[i] for i in ids], [bool(done[i]) for i in ids],
                    bool(env.env_done), list(env.agents)))
    tally = (int(env.total_games), int(env.ties), int(env.team["red"]["wins"]), int(env.team["blue"]["wins"]))
    return rec, tally


def run_dropin_case(case):
    """-> (None | what differed, {vals, exact}).  Both sides see the same stdlib random stream: the HIP env through the global generator
    (rng='python', as the reference uses it), the oracle through its own random.Random(seed)."""
    import random
    from oracle import battlespace_ref as ref
    kw = dict(n_agents=case["n"], continuous_actions=case["cont"], hit_base_reward=case["rewards"][0], hit_plane_reward=case["rewards"][1],
              miss_punishment=case["rewards"][2], die_punishment=case["rewards"][3], lose_punishment=case["rewards"][4])
    acts = _dropin_actions(case, case["T"])
    random.seed(case["seed"])
    got, tally_h = _play(bsx.parallel_env(**kw), case, acts)
    want, tally_o = _play(ref.RefEnv(rng=random.Random(case["seed"]), **kw), case, acts)
    stats = dict(vals=0, exact=0)
    if len(got) != len(want):
        return f"{len(got)} records against {len(want)}", stats
    for t, (g, w) in enumerate(zip(got, want)):
        if g[0] != w[0]:
            return f"record {t}: {g[0]} against {w[0]}", stats
        o, co = np.asarray(g[1], np.float64), np.asarray(w[1], np.float64)
        diff = np.abs(o - co)
        if ((diff > 1e-7) & (diff / np.maximum(np.abs(co), 1e-30) > 1e-5)).any():
            return f"obs at record {t}", stats
        stats["vals"] += o.size; stats["exact"] += int((np.asarray(g[1]) == np.asarray(w[1])).sum())
        if g[0] == "step":
            # values: equal for integer constants; float constants reach the caller through the kernel's float32 output (1e-6)
            ints = all(isinstance(v, int) for v in case["rewards"])
            # types: the reference's (int 0 without an event, the constants' type otherwise) -- except that float events cancelling to
            # exactly 0.0 come back as int 0 here (the kernel hands over the sum, not whether anything was added)
            if (g[2] != w[2] if ints else not np.allclose(g[2], w[2], rtol=1e-6, atol=1e-9)) or \
                    any(type(x) != type(y) and not (x == 0 and y == 0) for x, y in zip(g[2], w[2])):
                return f"rewards at record {t}: {g[2]} against {w[2]}", stats
            if g[3:] != w[3:]:
                return f"flags / live agents at record {t}: {g[3:]} against {w[3:]}", stats
    if tally_h != tally_o:
        return f"tally {tally_h} against {tally_o}", stats
    return None, stats


# ---------------------------------------------------------------------------------------------- the policy in the loop (rows f-1, f-2)
def draw_rollout_case(rng):
    """A PolicyRollout in a random form -- one launch for all ticks / the two-kernel graph / the graph as chains over game ranges;
    1v1 ... 4v4 (the graph forms also 5v5 ... 8v8); discrete or continuous; no noise / Gaussian / Ornstein-Uhlenbeck / a categorical
    draw (with or without a value head); a scripted opponent on either side; f32 / bf16x3 / bf16x6 -- whose games the C oracle then
    replays from the recorded score rows."""
    form = str(rng.choice(["one_launch", "one_launch", "graph", "chains"]))
    n = int(rng.choice([1, 1, 2, 2, 3, 4])) if form == "one_launch" else int(rng.choice([1, 2, 3, 4, 5, 6, 8]))
    cont = bool(rng.random() < 0.35)
    noise = str(rng.choice(["none", "gauss", "ou", "gauss+ou"] + ([] if cont else ["categorical"])))
    opp = "none" if form == "chains" else str(rng.choice(["none", "none", "red", "blue"]))
    if cont and opp != "none":
        opp = "none"          # (the scripted rows are played in float64 and RECORDED in float32: the record alone does not replay them)
    return dict(form=form, n=n, cont=cont, noise=noise, opponent=opp, precision=str(rng.choice(["f32", "f32", "bf16x3", "bf16x6"])),
                E=int(rng.choice([rng.integers(33, 300), rng.integers(300, 2500), 256 * rng.integers(1, 6) + rng.integers(-1, 2)])),
                T=int(rng.choice([8, 16, 24])), runs=int(rng.integers(7, 12)), chains=int(rng.choice([2, 3])),
                value=bool(noise == "categorical" and (n == 1 or form != "one_launch") and rng.random() < 0.5),
                ou_restart=bool(rng.random() < 0.7), seed=int(rng.integers(0, 2 ** 31)))


def run_rollout_case(case):
    from deep_rl_battlespace_amd import instinct
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    n, E, T, cont = case["n"], case["E"], case["T"], case["cont"]
    kw = dict(n_agents=n, seed=case["seed"], auto_reset=True, continuous_actions=cont)
    env = bsx.parallel_env(n_envs=E, **kw)
    c = cref.CRefBatch(E, **kw)
    env.reset(); o_c = c.reset().copy()
    torch.manual_seed(case["seed"] & 0xFFFF)
    actor = StackedActor(2 * n, 3 * n + 2, 3 if cont else 4, device="cuda")
    with torch.no_grad():
        actor.w3.mul_(60.0)
    rk = dict(one_launch=case["form"] == "one_launch", precision=case["precision"], seed=case["seed"] % 1000, ou_restart=case["ou_restart"],
              chains=case["chains"] if case["form"] == "chains" else 1)
    if "gauss" in case["noise"]:
        rk["noise_std"] = 0.3
    if "ou" in case["noise"]:
        rk["ou_scale"] = 0.2
    if case["noise"] == "categorical":
        rk.update(sample="categorical", temperature=0.8)
        if case["value"]:
            rk["value_actor"] = StackedActor(2 * n, 3 * n + 2, 1, device="cuda")
    if case["opponent"] != "none":
        mine, theirs = (env.possible_red, env.possible_blue) if case["opponent"] == "red" else (env.possible_blue, env.possible_red)
        rk["opponent"] = instinct.Team(mine, theirs, env)
    ro = PolicyRollout(env, actor, T, **rk)
    ro.start(); ro.capture()
    stats = dict(vals=0, exact=0)
    for run in range(case["runs"]):
        ro.run()
        torch.cuda.synchronize()
        obs, sc, rew, done, edone = (x.cpu().numpy() for x in (ro.obs, ro.scores, ro.rew, ro._done, ro.env_done))
        if run == 0 and not np.allclose(obs[0], o_c, rtol=1e-5, atol=1e-7):
            return "the observations the rollout starts from", stats
        for t in range(T):
            if not np.array_equal(edone[t], c.env_done):
                return f"env_done before tick {t} of run {run}", stats
            co, cr, cd = c.step(np.ascontiguousarray(sc[t][..., :3]) if cont else sc[t])
            if not np.array_equal(done[t].astype(bool), cd):
                return f"done at tick {t} of run {run}", stats
            if not np.array_equal(rew[t].astype(np.float64), cr):
                return f"rew at tick {t} of run {run}", stats
            diff = np.abs(obs[t + 1].astype(np.float64) - co)
            if ((diff > 1e-7) & (diff / np.maximum(np.abs(co), 1e-30) > 1e-5)).any():
                return f"obs after tick {t} of run {run}", stats
            stats["vals"] += co.size; stats["exact"] += int((obs[t + 1] == co).sum())
        if not np.array_equal(edone[T], c.env_done) or not np.array_equal(env.winner.cpu().numpy(), c.winner):
            return f"flags after run {run}", stats
    sh = {k: v.cpu().numpy() for k, v in env.export_state().items()}
    scx = c.export_state()
    for f in STATE:
        if not np.array_equal(sh[f], scx[f]):
            return f"state {f} at the end", stats
    m = scx["bl_live"].astype(bool)
    for f in ("bl_x", "bl_y", "bl_dir"):
        if not np.array_equal(sh[f][m], scx[f][m]):
            return f"state {f} at the end", stats
    if case["noise"] == "categorical":                       # the recorded log-probability belongs to the recorded row's arg-max, and is one
        lp = ro.logp.cpu().numpy()
        if not (np.isfinite(lp).all() and (lp <= 1e-6).all()):
            return "log-probabilities", stats
    return None, stats
