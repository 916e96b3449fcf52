#!/usr/bin/env python3
"""Generate the golden step()-parity fixtures by running the UNMODIFIED reference.

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  Output: tests/golden/*.npz -- pure data (inputs + expected
outputs), no reference text.  Re-run with:

    python tests/golden/make_golden.py

How the reference is driven:
  * /root/reference/envs/battle_env.py, envs/sprites.py and instinct/* are imported
    as they lie, with CWD=/root/reference (sprites load assets/*.png relatively).
  * Five third-party packages the reference imports are not installed here and
    cannot be (no network): pygame 2.1.2, gym 0.25.0, PettingZoo 1.19.0,
    vidmaker, cv2.  tests/golden/standins/ holds build-authored stand-ins for
    them; only pygame.Rect carries arithmetic (restated from pygame 2.1.2, see
    standins/pygame/__init__.py).
  * The stdlib `random` module functions the reference calls
    (`random.randint`, `random.random`; sprites.py:82-91,246-252,314) are
    wrapped by a tap that records every draw, and can replay a queue of forced
    values for the hand-scripted edge cases.  That is stdlib patching, not a
    change to reference code.

Trace schema (flat step axis S, episodes delimited by ep_ptr):
  spawn[ep, 4+3A]  base_red x,y, base_blue x,y, then plane x,y,dir in id order
  actions[S,A] int32 (discrete) | actions[S,A,3] float64 (continuous, pre-clip)
  logits[S,A,4] float64 (only the ndarray-action fixture)
  empty_call[S] bool   step({}) calls
  u[S,A] float64       random.random() consumed by agent a's shot (NaN: none)
  obs0[ep,A,D] f32     reset() observations
  obs[S,A,D] f32, rew[S,A] f64, done[S,A] bool, env_done[S] bool,
  winner[S] i8 (0 none, 1 red, 2 blue, 3 tie)
  px,py[S,A] i32, pdir[S,A] f64, php[S,A] i32, palive[S,A] bool
  bhp[S,2] i32, tick[S] i32 (count of time increments), total_time[S] f64
  bl_live/bl_x/bl_y/bl_dir[S,A,12]: live bullets of shooter a; slot = birth tick % 12 (birth tick = the
                       1-based count of physics ticks at which the shot was fired)
  total_games/ties/wins_red/wins_blue[S] i32 (env counters, persist across resets)
"""
import json
import math
import os
import random
import sys
from collections import deque

import numpy as np

sys.dont_write_bytecode = True          # nothing is ever written under /root/reference (read-only study material)

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
K = 12  # bullet slots per shooter (a bullet lives at most 12 updates: 12*45 >= 500)
WINNER = {"none": 0, "red": 1, "blue": 2, "tie": 3}


# ----------------------------------------------------------------------------- RNG tap
class RngTap:
    """Wraps stdlib random.randint / random.random: records draws, optionally replays forced ones."""

    def __init__(self):
        self._randint = random.randint
        self._random = random.random
        self.ints = []
        self.floats = []
        self.force_ints = deque()
        self.force_floats = deque()

    def install(self):
        random.randint = self.randint
        random.random = self.random

    def uninstall(self):
        random.randint = self._randint
        random.random = self._random

    def randint(self, a, b):
        if self.force_ints:
            v = self.force_ints.popleft()
            if v is not None:
                if not (a <= v <= b):
                    raise ValueError(f"forced randint {v} outside [{a},{b}]")
                self.ints.append(v)
                return v
        v = self._randint(a, b)
        self.ints.append(v)
        return v

    def random(self):
        if self.force_floats:
            v = self.force_floats.popleft()
            if v is not None:
                self.floats.append(v)
                return v
        v = self._random()
        self.floats.append(v)
        return v


# ----------------------------------------------------------------------------- recorder
class Recorder:
    def __init__(self, be, tap, cfg):
        self.be, self.tap, self.cfg = be, tap, dict(cfg)
        self.env = be.parallel_env(**cfg)
        env = self.env
        self.n = env.n_agents
        self.A = 2 * self.n
        self.D = env.obs_size
        self.cont = bool(env.continuous_actions)
        self.ids = list(env.possible_agents)
        self.rows = {k: [] for k in (
            "actions", "logits", "empty_call", "u", "obs", "rew", "done", "env_done", "winner",
            "px", "py", "pdir", "php", "palive", "bhp", "tick", "total_time",
            "bl_live", "bl_x", "bl_y", "bl_dir", "total_games", "ties", "wins_red", "wins_blue")}
        self.spawn, self.obs0, self.ep_ptr = [], [], [0]
        self.has_logits = False

    # -- episodes
    def reset(self, forced_spawn=None):
        """forced_spawn: list of 4+3A ints (None entries = draw normally).  Red plane dirs are
        given post-fold (0..90 | 270..359); they are un-folded to the randint(270,450) domain here."""
        env, tap = self.env, self.tap
        if forced_spawn is not None:
            f = list(forced_spawn)
            assert len(f) == 4 + 3 * self.A
            for i in range(self.n):  # red planes: randint(270, 450) then -360 if >= 360
                d = f[4 + 3 * i + 2]
                if d is not None and d < 270:
                    f[4 + 3 * i + 2] = d + 360
            tap.force_ints.extend(f)
        mark = len(tap.ints)
        obs = env.reset()
        assert not tap.force_ints
        return self.adopt_reset(obs, tap.ints[mark:])

    def adopt_reset(self, obs, draws):
        """Book an episode start from a reset() that has just happened (draws: the randint values it consumed)."""
        env = self.env
        assert len(draws) == 4 + 3 * self.A
        # what the env actually holds (dir after the red fold)
        sp = [*env.team["red"]["base"].rect.center, *env.team["blue"]["base"].rect.center]
        for aid in self.ids:
            p = env.team[env.team_map[aid]]["planes"][aid]
            sp += [p.rect.centerx, p.rect.centery, p.direction]
        for i, d in enumerate(draws):  # cross-check against the recorded draws
            exp = sp[i]
            if i >= 4 and (i - 4) % 3 == 2 and (i - 4) // 3 < self.n and d >= 360:
                d -= 360
            assert d == exp, (i, d, exp)
        self.spawn.append(sp)
        self.obs0.append(np.stack([obs[a] for a in self.ids]))
        self._objs = {a: env.team[env.team_map[a]]["planes"][a] for a in self.ids}
        if len(self.rows["obs"]) != self.ep_ptr[-1]:
            self.ep_ptr.append(len(self.rows["obs"]))
        return obs

    def end_episode(self):
        if self.ep_ptr[-1] != len(self.rows["obs"]):
            self.ep_ptr.append(len(self.rows["obs"]))

    # -- one step() call
    def step(self, actions, logits=None, empty=False, call=None, stepper=None):
        """actions: list per agent (ints, or 3-vectors when continuous); logits: optional [A,4].
        call / stepper: the dict a foreign driver handed to step() and the bound method to run it with (the evaluation fixture
        records calls that evaluate.main() makes; `actions` then holds the integers the env reduces that dict to)."""
        env, tap, ids = self.env, self.tap, self.ids
        alive_before = list(env.agents)
        if call is not None:
            pass
        elif empty:
            call = {}
        elif logits is not None:
            call = {a: np.asarray(logits[i], dtype=np.float64) for i, a in enumerate(ids)}
        elif self.cont:
            call = {a: np.asarray(actions[i], dtype=np.float64) for i, a in enumerate(ids)}
        else:
            call = {a: actions[i] for i, a in enumerate(ids)}
        mark = len(tap.floats)
        bullets_before = len(env.bullets)
        was_done = env.env_done
        obs, rew, done, info = (stepper or env.step)(call)
        draws = tap.floats[mark:]
        # attribute each random() draw to the agent whose shot consumed it: alive-agent order
        u = np.full(self.A, np.nan)
        if not was_done and not empty and len(draws):
            shooters = []
            for a in alive_before:
                i = ids.index(a)
                if self.cont:
                    clipped = np.clip(np.asarray(actions[i], dtype=np.float64), -1, 1)
                    fired = clipped[2] > 0
                elif logits is not None:
                    fired = int(np.argmax(logits[i])) == 1
                else:
                    fired = actions[i] == 1
                if fired:
                    shooters.append(i)
            assert len(shooters) == len(draws), (shooters, draws)
            for i, d in zip(shooters, draws):
                u[i] = d
        else:
            assert len(draws) == 0
        r = self.rows
        if self.cont:
            r["actions"].append(np.asarray([np.asarray(a, dtype=np.float64) for a in actions]).reshape(self.A, 3))
        else:
            r["actions"].append(np.asarray(actions, dtype=np.int32).reshape(self.A))
        if logits is not None:
            self.has_logits = True
            r["logits"].append(np.asarray(logits, dtype=np.float64).reshape(self.A, 4))
        else:
            r["logits"].append(np.zeros((self.A, 4)))
        r["empty_call"].append(bool(empty))
        r["u"].append(u)
        r["obs"].append(np.stack([obs[a] for a in ids]).astype(np.float32))
        assert all(obs[a].dtype == np.float32 for a in ids)
        r["rew"].append(np.asarray([float(rew[a]) for a in ids]))
        r["done"].append(np.asarray([bool(done[a]) for a in ids]))
        assert done is env.dones
        r["env_done"].append(bool(env.env_done))
        r["winner"].append(WINNER[env.winner])
        self._snapshot()
        return obs, rew, done

    def _snapshot(self):
        env, ids, r = self.env, self.ids, self.rows
        px, py, pdir, php, palive = [], [], [], [], []
        for a in ids:
            planes = env.team[env.team_map[a]]["planes"]
            alive = a in env.agents
            assert alive == (a in planes)
            p = self._objs[a]  # a popped (dead) Plane object keeps its final pose and hp
            assert alive == bool(p.alive) and (alive or p.hp <= 0)
            palive.append(alive)
            px.append(p.rect.centerx); py.append(p.rect.centery); pdir.append(float(p.direction)); php.append(p.hp)
        r["px"].append(px); r["py"].append(py); r["pdir"].append(pdir); r["php"].append(php); r["palive"].append(palive)
        r["bhp"].append([env.team["red"]["base"].hp, env.team["blue"]["base"].hp])
        tick = int(round(env.total_time / env.time_step))
        r["tick"].append(tick); r["total_time"].append(float(env.total_time))
        # the time-limit tie call advances the clock but not the bullets (battle_env.py:316-323): label slots by
        # the number of ticks on which bullets were actually updated
        ptick = tick - 1 if (env.env_done and env.winner == "tie" and env.total_time >= env.max_time) else tick
        live = np.zeros((self.A, K), bool); bx = np.zeros((self.A, K), np.int32)
        by = np.zeros((self.A, K), np.int32); bd = np.zeros((self.A, K))
        for b in env.bullets:
            i = ids.index(b.agent_id)
            age = int(round(b.dist_travelled / 45.0))
            assert 1 <= age <= 11 and abs(b.dist_travelled - 45.0 * age) < 1e-9
            s = (ptick - age + 1) % K
            assert not live[i, s]
            live[i, s] = True; bx[i, s] = b.rect.centerx; by[i, s] = b.rect.centery; bd[i, s] = float(b.direction)
        r["bl_live"].append(live); r["bl_x"].append(bx); r["bl_y"].append(by); r["bl_dir"].append(bd)
        r["total_games"].append(env.total_games); r["ties"].append(env.ties)
        r["wins_red"].append(env.team["red"]["wins"]); r["wins_blue"].append(env.team["blue"]["wins"])

    def save(self, name, meta):
        self.end_episode()
        r = self.rows
        out = {
            "ep_ptr": np.asarray(self.ep_ptr, np.int64),
            "spawn": np.asarray(self.spawn, np.int32),
            "obs0": np.asarray(self.obs0, np.float32),
            "actions": np.asarray(r["actions"], np.float64 if self.cont else np.int32),
            "empty_call": np.asarray(r["empty_call"], bool),
            "u": np.asarray(r["u"], np.float64),
            "obs": np.asarray(r["obs"], np.float32),
            "rew": np.asarray(r["rew"], np.float64),
            "done": np.asarray(r["done"], bool),
            "env_done": np.asarray(r["env_done"], bool),
            "winner": np.asarray(r["winner"], np.int8),
            "px": np.asarray(r["px"], np.int32), "py": np.asarray(r["py"], np.int32),
            "pdir": np.asarray(r["pdir"], np.float64), "php": np.asarray(r["php"], np.int32),
            "palive": np.asarray(r["palive"], bool), "bhp": np.asarray(r["bhp"], np.int32),
            "tick": np.asarray(r["tick"], np.int32), "total_time": np.asarray(r["total_time"], np.float64),
            "bl_live": np.asarray(r["bl_live"], bool), "bl_x": np.asarray(r["bl_x"], np.int16),
            "bl_y": np.asarray(r["bl_y"], np.int16), "bl_dir": np.asarray(r["bl_dir"], np.float64),
            "total_games": np.asarray(r["total_games"], np.int32), "ties": np.asarray(r["ties"], np.int32),
            "wins_red": np.asarray(r["wins_red"], np.int32), "wins_blue": np.asarray(r["wins_blue"], np.int32),
        }
        if self.has_logits:
            out["logits"] = np.asarray(r["logits"], np.float64)
        m = dict(meta)
        m.update(cfg=self.cfg, n=self.n, A=self.A, D=self.D, continuous=self.cont,
                 python=sys.version.split()[0], numpy=np.__version__,
                 libm="glibc " + os.confstr("CS_GNU_LIBC_VERSION").split()[-1],
                 generator="tests/golden/make_golden.py", reference="WilliamFlinchbaugh/Deep-RL-Battlespace @ v1")
        out["meta"] = np.asarray(json.dumps(m))
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        ne = len(self.ep_ptr) - 1
        w = np.asarray(r["winner"])[np.asarray(self.ep_ptr[1:]) - 1] if ne else []
        print(f"{name}: {ne} episodes, {len(r['obs'])} steps, {os.path.getsize(path)/1024:.0f} kB, "
              f"final winners red/blue/tie/none = {[int((np.asarray(w)==c).sum()) for c in (1,2,3,0)]}, "
              f"deaths = {int((~np.asarray(r['palive'])[np.asarray(self.ep_ptr[1:])-1]).sum()) if ne else 0}")


# ----------------------------------------------------------------------------- drivers
def play(rec, policy, episodes, max_calls=400, extra_calls_after_done=0, forced_spawn=None):
    for ep in range(episodes):
        obs = rec.reset(forced_spawn[ep] if forced_spawn else None)
        calls = 0
        while not rec.env.env_done and calls < max_calls:
            obs, _, _ = rec.step(policy(obs))
            calls += 1
        for _ in range(extra_calls_after_done):
            obs, _, _ = rec.step(policy(obs))
        rec.end_episode()


def make_instinct(rec, instinct_mod):
    env = rec.env
    red = instinct_mod.Team(env.possible_red, env.possible_blue, env)
    blue = instinct_mod.Team(env.possible_blue, env.possible_red, env)

    def policy(obs):
        acts = dict(blue.choose_actions({a: obs[a] for a in env.possible_blue}))
        acts.update(red.choose_actions({a: obs[a] for a in env.possible_red}))
        return [acts[a] for a in rec.ids]
    return policy


def main():
    os.chdir(REF)
    sys.dont_write_bytecode = True
    sys.path[:0] = [os.path.join(HERE, "standins"), REF]
    import envs.battle_env as be  # the reference, unmodified
    import instinct.team as instinct_mod
    tap = RngTap()
    tap.install()
    try:
        if "--instinct-only" in sys.argv:
            instinct_fixture(be, instinct_mod, tap)
        else:
            generate(be, instinct_mod, tap)
            instinct_fixture(be, instinct_mod, tap)
    finally:
        tap.uninstall()


def generate(be, instinct_mod, tap):
    rcfg = dict(hit_base_reward=1.0, hit_plane_reward=0.9, miss_punishment=-0.02,
                die_punishment=-0.03, lose_punishment=-0.05)  # main.py:29-39 (the trained config)

    # ---- G1: 1v1 discrete, instinct vs instinct (wins / deaths / ties)
    random.seed(101)
    rec = Recorder(be, tap, dict(n_agents=1))
    play(rec, make_instinct(rec, instinct_mod), 24, extra_calls_after_done=2)
    rec.save("g1_1v1_instinct", dict(driver="instinct-vs-instinct", seed=101))

    # ---- G2: 1v1 discrete, uniform random actions, 1000 steps across resets (config C1 itself)
    random.seed(1234)
    rng = np.random.default_rng(1234)
    rec = Recorder(be, tap, dict(n_agents=1))
    calls = 0
    while calls < 1000:
        rec.reset()
        while not rec.env.env_done and calls < 1000:
            rec.step([int(v) for v in rng.integers(0, 4, size=rec.A)]); calls += 1
        rec.end_episode()
    rec.save("g2_1v1_random", dict(driver="uniform-random", seed=1234, np_seed=1234))

    # ---- G3: 2v2 / 3v3 / 4v4 instinct and random; float rewards of main.py on the 2v2 case
    for n, eps, cfg, tag in ((2, 10, rcfg, "2v2_instinct_floatrew"), (4, 6, {}, "4v4_instinct"), (3, 4, {}, "3v3_instinct")):
        random.seed(300 + n)
        rec = Recorder(be, tap, dict(n_agents=n, **cfg))
        play(rec, make_instinct(rec, instinct_mod), eps, extra_calls_after_done=1)
        rec.save("g3_" + tag, dict(driver="instinct-vs-instinct", seed=300 + n))
    random.seed(344)
    rng = np.random.default_rng(344)
    rec = Recorder(be, tap, dict(n_agents=4))
    play(rec, lambda obs: [int(v) for v in rng.integers(0, 4, size=8)], 3)
    rec.save("g3_4v4_random", dict(driver="uniform-random", seed=344, np_seed=344))
    # all-shoot stress: every slot of every ring in use, frequent same-step multi-hits
    random.seed(345)
    rng = np.random.default_rng(345)
    rec = Recorder(be, tap, dict(n_agents=2))
    play(rec, lambda obs: [1 if rng.random() < 0.8 else int(rng.integers(0, 4)) for _ in range(4)], 6)
    rec.save("g3_2v2_mostly_shoot", dict(driver="80% shoot", seed=345, np_seed=345))

    # ---- G4: continuous actions (float64 arrays: the pinned numpy 1.23.1 promotes float32 action
    #      scalars to float64 on the first Python-number op, so float64 arithmetic IS the reference's)
    random.seed(401); np.random.seed(401)
    rec = Recorder(be, tap, dict(n_agents=1, continuous_actions=True))
    play(rec, make_instinct(rec, instinct_mod), 12, extra_calls_after_done=1)
    rec.save("g4_1v1_cont_instinct", dict(driver="instinct-vs-instinct", seed=401, np_seed=401))
    random.seed(402); np.random.seed(402)
    rec = Recorder(be, tap, dict(n_agents=2, continuous_actions=True, **rcfg))
    play(rec, make_instinct(rec, instinct_mod), 8)
    rec.save("g4_2v2_cont_instinct_floatrew", dict(driver="instinct-vs-instinct", seed=402, np_seed=402))
    random.seed(403)
    rng = np.random.default_rng(403)
    rec = Recorder(be, tap, dict(n_agents=1, continuous_actions=True))
    # float32-representable values, incl. out-of-range ones that step() must clip (battle_env.py:295-297)
    play(rec, lambda obs: [rng.uniform(-1.3, 1.3, size=3).astype(np.float32).astype(np.float64) for _ in range(2)], 4)
    rec.save("g4_1v1_cont_random_f32vals", dict(driver="uniform(-1.3,1.3) float32-representable", seed=403, np_seed=403))

    # ---- G5: hand-scripted edge cases
    scripted(be, tap)

    # ---- G6: rel_angle / dist table on the integer lattice, incl. exact +-180 / 0 cases
    pts = []
    rng = np.random.default_rng(6)
    for dx in (-300, -45, -1, 0, 1, 45, 300):
        for dy in (-300, -45, -1, 0, 1, 45, 300):
            if dx or dy:
                pts.append((600, 400, 600 - dx, 400 - dy))
    for _ in range(300):
        pts.append(tuple(int(v) for v in (rng.integers(25, 1176), rng.integers(24, 777), rng.integers(25, 1176), rng.integers(24, 777))))
    dirs = list(range(0, 361, 15)) + [1, 44, 46, 89, 91, 179, 181, 359] + [float(v) for v in rng.uniform(0, 360, 12)]
    rows = []
    for (x0, y0, x1, y1) in pts:
        for a0 in dirs:
            rows.append((x0, y0, x1, y1, a0, be.rel_angle((x0, y0), a0, (x1, y1)), be.dist((x0, y0), (x1, y1))))
    rows = np.asarray(rows, np.float64)
    np.savez_compressed(os.path.join(HERE, "g6_rel_angle_table.npz"), table=rows,
                        meta=np.asarray(json.dumps(dict(cols="x0,y0,x1,y1,a0,rel_angle,dist", fn="battle_env.py:38-58"))))
    print(f"g6_rel_angle_table: {len(rows)} rows")

    # ---- G7: spawn ranges from 20000 resets per team size (min/max/histogram support)
    random.seed(7)
    env = be.parallel_env(n_agents=2)
    mark = len(tap.ints)
    for _ in range(20000):
        env.reset()
    d = np.asarray(tap.ints[mark:], np.int32).reshape(20000, 4 + 3 * 4)
    # fold red dirs as Plane.reset does
    for i in range(2):
        c = 4 + 3 * i + 2
        d[:, c] = np.where(d[:, c] >= 360, d[:, c] - 360, d[:, c])
    np.savez_compressed(os.path.join(HERE, "g7_spawn_stats.npz"), lo=d.min(0), hi=d.max(0),
                        mean=d.mean(0), n=np.asarray(20000),
                        red_dir_hist=np.bincount(d[:, 6], minlength=361), blue_dir_hist=np.bincount(d[:, 12], minlength=361),
                        meta=np.asarray(json.dumps(dict(cols="base_red x,y, base_blue x,y, plane0..3 x,y,dir (2v2)", seed=7))))
    print("g7_spawn_stats: lo", d.min(0).tolist(), "hi", d.max(0).tolist())


def report(rec, left):
    """One line per scripted episode so the author can see that the scenario did what it was written for."""
    a, b = rec.ep_ptr[-2], rec.ep_ptr[-1]
    r = rec.rows
    rew = np.asarray(r["rew"][a:b]).sum(0)
    print(f"   ep{len(rec.ep_ptr)-2:3d}: calls={b-a:3d} winner={r['winner'][b-1]} tick={r['tick'][b-1]:3d} "
          f"rew={rew.tolist()} hp={r['php'][b-1]} bhp={r['bhp'][b-1]} alive={[int(v) for v in r['palive'][b-1]]} "
          f"pos={list(zip(r['px'][b-1], r['py'][b-1]))} dir={r['pdir'][b-1]} unused_u={left}")


def scripted(be, tap):
    """G5: every episode forces its spawn; planes far apart unless the case needs contact."""
    FAR = [100, 700, 1100, 100]  # base_red (100,700), base_blue (1100,100)

    def sp(p0, p1, bases=FAR):
        return [*bases, *p0, *p1]

    rec = Recorder(be, tap, dict(n_agents=1))

    def run(spawn, script, us=None, post=0):
        """script: list of [a0, a1] per call; us: forced random() values in draw order."""
        rec.reset(spawn)
        if us:
            tap.force_floats.extend(us)
        for acts in script:
            rec.step(list(acts))
        for _ in range(post):
            rec.step([0, 0])
        left = len(tap.force_floats); tap.force_floats.clear()
        rec.end_episode()
        report(rec, left)

    # (1) truncation / clamp lattice: dirs 0/90/180/270 (and 360 reached by turning) at x in {50,63,64,100,383}
    for d0 in (0, 90, 270, 285, 345):
        for x in (50, 63, 64, 100, 383):
            run(sp((x, 400, d0), (1100, 400, 180)), [[0, 0]] * 6)
    # blue side incl. right/top/bottom clamps
    for d1, y in ((90, 48), (270, 752), (180, 400), (105, 60), (255, 740)):
        run(sp((100, 400, 0), (1150, y, d1)), [[0, 0]] * 5)
    # (2) direction wrap: 350 -> +15 -> 365 -> 5 ; 5 -> -15 -> -10 -> 350 ; 345+15 = 360 stays 360 ; 360+15 -> 15
    run(sp((200, 400, 350), (1000, 400, 100)), [[2, 3]] * 3 + [[3, 2]] * 4)
    run(sp((200, 400, 345), (1000, 400, 90)), [[2, 0], [0, 0], [1, 0], [2, 0], [3, 0], [3, 0]], us=[0.5])
    # (3) invalid discrete actions: no movement, no shot (battle_env.py:399-417)
    run(sp((200, 400, 0), (1000, 400, 180)), [[4, -1], [7, 100], [0, 0], [5, 1]], us=[0.25])
    # (4) bullet exactly axis-aligned with zero jitter (u = 0.5 -> jitter 0): dir 270 from x=40 cannot
    #     happen for a plane (x>=25 ok): shooter at x=63, dir 270 / 90 / 180 / 0
    for d0 in (0, 90, 180, 270):
        run(sp((63, 400, d0 if d0 in (0, 90, 270) else 0), (1100, 400, 180 if d0 != 180 else 180)),
            [[1, 1]] + [[0, 0]] * 12, us=[0.5, 0.5])
    # (5) range miss on the 12th update (bullet flies along the field, nothing in the way)
    run(sp((50, 48, 0), (1150, 752, 180), bases=[62, 738, 1138, 62]), [[1, 1]] + [[0, 0]] * 13, us=[0.5, 0.5])
    # (6) off-screen miss incl. truncation toward zero near the left/top edge
    run(sp((50, 48, 90), (1150, 752, 270), bases=[379, 400, 758, 400]), [[1, 1]] + [[0, 0]] * 4, us=[0.5, 0.5])
    run(sp((50, 400, 0), (766, 400, 180), bases=[379, 700, 1138, 700]), [[3, 2]] * 6 + [[1, 1]] + [[0, 0]] * 6, us=[0.4375, 0.5625])
    # (7) base hits: blue plane next to red base shoots it repeatedly -> red base dies -> blue wins
    run(sp((383, 100, 0), (766, 400, 180), bases=[300, 400, 1138, 700]),
        [[0, 0]] * 14 + [[0, 1]] * 14, us=[0.5] * 14, post=3)
    # (8) red kills blue base; winner receives lose_punishment
    run(sp((383, 400, 0), (1150, 60, 90), bases=[62, 62, 800, 400]), [[0, 0]] * 12 + [[1, 0]] * 12, us=[0.5] * 12, post=2)
    # (9) plane kill: head-on planes, both shooting (4 hp each); the first to lose 4 hp dies, its bullets fly on
    run(sp((383, 400, 0), (766, 400, 180), bases=[62, 62, 1138, 738]), [[1, 1]] * 14 + [[0, 0]] * 6,
        us=[0.5] * 28, post=0)
    run(sp((383, 400, 0), (766, 400, 180), bases=[62, 62, 1138, 738]), [[1, 0]] * 16 + [[0, 0]] * 6, us=[0.5] * 16)
    # (10) empty-action call -> tie (battle_env.py:309-313), then inert calls
    rec.reset(sp((200, 400, 0), (1000, 400, 180)))
    rec.step([0, 1]); tap.force_floats.clear()
    rec.step([0, 0], empty=True)
    rec.step([0, 0]); rec.step([1, 1])
    rec.end_episode()
    # (11) time-limit tie on call #121 with nothing happening, then two inert calls
    run(sp((200, 400, 90), (1000, 400, 90)), [[2, 3]] * 121, post=2)
    rec.save("g5_scripted_1v1", dict(driver="hand-scripted edge cases"))

    # (12) ndarray actions -> flat argmax (battle_env.py:327-328)
    rec = Recorder(be, tap, dict(n_agents=1))
    rng = np.random.default_rng(512)
    random.seed(512)
    rec.reset()
    for _ in range(60):
        lg = rng.normal(size=(2, 4))
        if rng.random() < 0.2:
            lg[0, 1] = lg[0, 2] = lg[0].max() + 1.0  # tie -> first index wins
        rec.step([int(np.argmax(l)) for l in lg], logits=lg)
        if rec.env.env_done:
            break
    rec.save("g5_ndarray_actions_1v1", dict(driver="random logits, argmax", seed=512, np_seed=512))

    # (13) 2v2 scripted: both bases dying in ONE step; same-step kill chain; shot whose first target died
    rec = Recorder(be, tap, dict(n_agents=2))

    def run2(spawn, script, us=None, post=0):
        rec.reset(spawn)
        if us:
            tap.force_floats.extend(us)
        for acts in script:
            rec.step(list(acts))
        for _ in range(post):
            rec.step([0] * 4)
        left = len(tap.force_floats); tap.force_floats.clear()
        rec.end_episode()
        report(rec, left)

    # both teams hammer the other's base symmetrically (base hp 10): mirrored geometry -> same-step double kill
    run2([200, 400, 1000, 400, 383, 380, 0, 383, 420, 0, 817, 380, 180, 817, 420, 180],
         [[0] * 4] * 11 + [[1, 1, 1, 1]] * 12, us=[0.5] * 48, post=2)
    # two red planes stacked on the same line firing at one blue plane: kill chain + pass-through to the 2nd blue
    run2([62, 62, 1138, 738, 300, 400, 0, 383, 400, 0, 766, 400, 180, 900, 400, 180],
         [[1, 1, 0, 0]] * 14 + [[0] * 4] * 8, us=[0.5] * 28)
    rec.save("g5_scripted_2v2", dict(driver="hand-scripted 2v2 edge cases"))


def instinct_fixture(be, instinct_mod, tap):
    """G8: (observation row -> action) pairs of the reference's scripted opponent (instinct/agent.py:10-62).
    The agent is called with a float64 COPY of each float32 observation: with the reference's pinned numpy 1.23.1 a
    float32 scalar is promoted to float64 by its first operation with a Python number, so float64 arithmetic on the
    float32 values is what the reference computes (numpy 2.x here would otherwise stay in float32).  The continuous
    branch draws from numpy's global generator; np.random.rand / np.random.uniform are wrapped to record the draws."""
    real_rand, real_uniform = np.random.rand, np.random.uniform
    draws = {"rand": [], "noise": []}

    def rand_tap():
        v = real_rand(); draws["rand"].append(v); return v

    def uniform_tap(lo, hi, size=None):
        v = real_uniform(lo, hi, size=size); draws["noise"].append(np.asarray(v, np.float64).copy()); return v
    out = {}
    for cont in (False, True):
        for n in (1, 2, 4):
            random.seed(800 + n + 10 * cont); np.random.seed(800 + n + 10 * cont)
            env = be.parallel_env(n_agents=n, continuous_actions=cont)
            red = instinct_mod.Team(env.possible_red, env.possible_blue, env)
            blue = instinct_mod.Team(env.possible_blue, env.possible_red, env)
            rows = []
            np.random.rand, np.random.uniform = rand_tap, uniform_tap
            try:
                for ep in range(6 if n < 4 else 3):
                    obs = env.reset()
                    while not env.env_done:
                        acts = {}
                        for team in (red, blue):
                            for aid, agent in team.agents.items():
                                o32 = obs[aid]
                                draws["rand"].clear(); draws["noise"].clear()
                                a = agent.choose_action(np.asarray(o32, np.float64))
                                acts[aid] = a
                                if cont:
                                    rows.append((o32.copy(), env.possible_agents.index(aid), np.asarray(a, np.float64),
                                                 draws["rand"][0] if draws["rand"] else np.nan, draws["noise"][0].copy()))
                                else:
                                    rows.append((o32.copy(), env.possible_agents.index(aid), int(a)))
                        obs, _, _, _ = env.step(acts)
                    # a few post-mortem rows: dead observers see all -1
            finally:
                np.random.rand, np.random.uniform = real_rand, real_uniform
            tag = f"{'cont' if cont else 'disc'}_{n}v{n}"
            out[f"{tag}/obs"] = np.stack([r[0] for r in rows]).astype(np.float32)
            out[f"{tag}/agent"] = np.asarray([r[1] for r in rows], np.int32)
            if cont:
                out[f"{tag}/action"] = np.stack([r[2] for r in rows]).astype(np.float64)
                out[f"{tag}/rand"] = np.asarray([r[3] for r in rows], np.float64)
                out[f"{tag}/noise"] = np.stack([r[4] for r in rows]).astype(np.float64)
            else:
                out[f"{tag}/action"] = np.asarray([r[2] for r in rows], np.int32)
            print(f"g8 {tag}: {len(rows)} rows", np.bincount(out[f"{tag}/action"]) if not cont else "")
    np.savez_compressed(os.path.join(HERE, "g8_instinct_pairs.npz"), **out)


def actor_fixture():
    """G9: the reference ActorNetwork (maddpg/networks.py:54-85), seeded random init, on random observation rows:
    weights + inputs + outputs, to pin the stacked on-device actor's forward pass.  (torch only; no stand-ins.)"""
    import torch
    sys.path[:0] = [REF]
    import maddpg.networks as nets
    torch.manual_seed(9)
    out = {}
    for tag, obs_len in (("1v1", 5), ("2v2", 8), ("4v4", 14)):
        net = nets.ActorNetwork(obs_len, 4, chkpt_dir="/tmp", name="g9").to("cpu")
        with torch.no_grad():
            net.bn1.weight.uniform_(0.5, 1.5); net.bn1.bias.uniform_(-0.2, 0.2)      # non-trivial LayerNorm affine
            net.bn2.weight.uniform_(0.5, 1.5); net.bn2.bias.uniform_(-0.2, 0.2)
            net.pi.weight.uniform_(-0.5, 0.5)                                           # make tanh outputs distinguishable
        net.eval()
        x = torch.rand(64, obs_len) * 2 - 1
        with torch.no_grad():
            y = net.forward(x)
        for k, v in net.state_dict().items():
            out[f"{tag}/{k}"] = v.numpy()
        out[f"{tag}/x"] = x.numpy(); out[f"{tag}/y"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "g9_actor_forward.npz"), **out)
    print("g9_actor_forward:", sorted(out)[:6], "...")


def learner_side_fixture():
    """G10 / G11: the two numpy-only pieces next to the step path that the rollout feeds (SURVEY.md section 8f-3 and the OU
    exploration noise of the caller's loop), run UNMODIFIED from /root/reference:
      g10  maddpg/buffer.py ReplayBuffer -- store_transition layouts after a partial fill and after the ring wrapped, and what
           sample() returns for the indices np.random.choice drew (recorded by wrapping that numpy function);
      g11  utils/noise.py OUNoise -- a trajectory of noise() values and process states for recorded np.random.randn draws,
           with a reset() (main.py:155 restarts the process per game) and a re-scale (main.py:154) on the way."""
    sys.path[:0] = [REF]
    import maddpg.buffer as rbuf
    import utils.noise as rnoise
    rng = np.random.default_rng(10)
    out = {}
    agents, obs_size, n_act, M, B = ["plane0", "plane1"], 8, 4, 6, 5              # the red team of a 2v2 game (main.py:113-118)
    buf = rbuf.ReplayBuffer(M, B, agents, obs_size, obs_size * len(agents), n_act)
    S = 10                                                                          # > M: rows 0..3 are overwritten
    tr = dict(states=rng.uniform(-1, 1, (S, 2, obs_size)).astype(np.float32), actions=rng.uniform(-1, 1, (S, 2, n_act)).astype(np.float32),
              rewards=rng.integers(-20, 100, (S, 2)).astype(np.float64), states_=rng.uniform(-1, 1, (S, 2, obs_size)).astype(np.float32),
              dones=rng.random((S, 2)) < 0.3)
    picks = []
    real_choice = np.random.choice

    def choice_tap(a, size=None, *args, **kw):
        r = real_choice(a, size, *args, **kw)
        picks.append(np.asarray(r).copy())
        return r

    def snap(tag):
        for k in ("state_mem", "new_state_mem", "rew_mem", "done_mem"):
            out[f"{tag}/{k}"] = getattr(buf, k).copy()
        for k in ("actor_states", "actor_new_states", "action_mem"):
            out[f"{tag}/{k}"] = np.stack(getattr(buf, k)).copy()                   # [agent, M, ...]
        out[f"{tag}/mem_cntr"] = np.int64(buf.mem_cntr)
        out[f"{tag}/is_ready"] = np.bool_(buf.is_ready())
        if buf.is_ready():
            np.random.seed(100 + buf.mem_cntr)
            np.random.choice = choice_tap
            try:
                res = buf.sample()
            finally:
                np.random.choice = real_choice
            out[f"{tag}/idx"] = picks[-1].astype(np.int64)
            for name, v in zip(("actor_states", "states", "actions", "rewards", "actor_new_states", "states_", "dones"), res):
                out[f"{tag}/sample/{name}"] = np.asarray(v).copy()
    for k in range(S):
        d = lambda key: {a: tr[key][k, i] for i, a in enumerate(agents)}            # noqa: E731
        buf.store_transition(d("states"), d("actions"), d("rewards"), d("states_"), d("dones"))
        if k + 1 in (3, 5, S):
            snap(f"after{k + 1}")
    for k, v in tr.items():
        out[f"in/{k}"] = v
    out["cfg"] = np.asarray([M, B, obs_size, n_act, len(agents)], np.int64)
    np.savez_compressed(os.path.join(HERE, "g10_replay_buffer.npz"), **out)
    print("g10_replay_buffer:", len(out), "arrays; sampled idx", [p.tolist() for p in picks])

    zs = []
    real_randn = np.random.randn

    def randn_tap(*shape):
        z = real_randn(*shape)
        zs.append(np.asarray(z, np.float64).copy())
        return z
    np.random.seed(11)
    np.random.randn = randn_tap
    try:
        ou = rnoise.OUNoise(4)                                                      # maddpg/agent.py:15
        vals, states, events = [], [], []
        for t in range(40):
            if t == 13:
                ou.reset(); events.append((t, 1, 0.0))                             # reset_noise per game (main.py:155)
            if t == 20:
                ou.scale = 0.37; events.append((t, 2, 0.37))                       # scale_noise (main.py:154)
            vals.append(np.asarray(ou.noise(), np.float64).copy())
            states.append(np.asarray(ou.state, np.float64).copy())
    finally:
        np.random.randn = real_randn
    np.savez_compressed(os.path.join(HERE, "g11_ou_noise.npz"), z=np.stack(zs), noise=np.stack(vals), state=np.stack(states),
                        events=np.asarray(events, np.float64), params=np.asarray([0.1, 0.0, 0.15, 0.2]))   # scale0, mu, theta, sigma
    print("g11_ou_noise:", np.stack(vals).shape, "last", vals[-1])


def evaluation_fixture(games):
    """G12: the reference's own evaluation workload (evaluate.py:14-109; README.md:30 publishes "~80 %" for it): 2v2 with the shipped
    reward config models/completed_model/cf.json, red = the shipped checkpoints actor_plane0 / actor_plane1 through
    maddpg.Team.load_models + NetworkedAgent.choose_action (OU noise, scale 0.1, never restarted), blue = instinct.Team.
    evaluate.main() is run UNMODIFIED; what is injected are names around it: input() answers the model-name prompt, the module
    sees a `range` that plays `games` games instead of 10 000 and none of the 10 recorded ones (those need a display), torch.load
    maps the checkpoints to the CPU, and a subclass of the reference env counts calls -- main() keeps its win tally in a local
    dict and never prints it, so the tally is read from the env's own counters (battle_env.py:102-103,169-170) afterwards.
    Recorded: the two actors' weights, their forward outputs on 96 observation rows seen in play, cf.json, and the tally."""
    import builtins
    import torch
    os.chdir(REF)
    sys.path[:0] = [os.path.join(HERE, "standins"), REF]
    import evaluate as ev                                   # imports envs.battle_env, instinct.team, maddpg.team as they lie
    import maddpg.networks as nets
    random.seed(12); np.random.seed(12); torch.manual_seed(12)
    created, seen_obs, lens = [], [], []
    TRACE_GAMES = 40                                        # the first games also as a step trace (g3_2v2_evaluation_games)
    tap = RngTap(); tap.install()                           # records the stdlib draws (spawns, bullet jitter); does not change them
    rec = Recorder.__new__(Recorder)

    class CountingEnv(ev.battle_env.parallel_env):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.calls_this_game, created[:] = 0, [self]
            self.pending, self.recorded, self.recording = None, 0, False
            rec.be, rec.tap, rec.cfg, rec.env = ev.battle_env, tap, dict(k), self
            rec.n, rec.A, rec.D, rec.cont, rec.ids = self.n_agents, 2 * self.n_agents, self.obs_size, False, list(self.possible_agents)
            rec.rows = {key: [] for key in (
                "actions", "logits", "empty_call", "u", "obs", "rew", "done", "env_done", "winner",
                "px", "py", "pdir", "php", "palive", "bhp", "tick", "total_time",
                "bl_live", "bl_x", "bl_y", "bl_dir", "total_games", "ties", "wins_red", "wins_blue")}
            rec.spawn, rec.obs0, rec.ep_ptr, rec.has_logits = [], [], [0], False

        def reset(self, *a, **k):
            if self.calls_this_game:
                lens.append(self.calls_this_game)
            self.calls_this_game = 0
            mark = len(tap.ints)
            obs = super().reset(*a, **k)
            self.pending = (obs, tap.ints[mark:])           # evaluate.py resets twice per game: the LAST reset before a step starts the episode
            return obs

        def step(self, actions):
            if self.pending is not None:                    # first call of a game: is it one of the recorded ones?
                self.recording = self.recorded < TRACE_GAMES
                if self.recording:
                    rec.adopt_reset(*self.pending)
                    self.recorded += 1
                self.pending = None
            if self.recording:
                ints = [int(np.argmax(actions[a])) if isinstance(actions[a], np.ndarray) else int(actions[a]) for a in rec.ids]
                o, r, d = rec.step(ints, call=actions, stepper=super().step)
                out = (o, r, d, {a: {} for a in rec.ids})
            else:
                out = super().step(actions)
            self.calls_this_game += 1
            if len(seen_obs) < 96 and self.calls_this_game % 7 == 3:
                seen_obs.append(np.stack([out[0][a] for a in self.possible_red]))
            return out

    real_input, real_load, real_cls = builtins.input, torch.load, ev.battle_env.parallel_env
    builtins.input = lambda prompt="": "completed_model"
    torch.load = lambda f, *a, **k: real_load(f, *a, **{**k, "map_location": "cpu"})
    ev.battle_env.parallel_env = CountingEnv
    ev.range = lambda n: builtins.range(games if n == 10000 else 0)
    try:
        try:
            ev.main()
        except AttributeError as exc:                     # evaluate.py:109 calls env.stop_recording(), which the env does not have
            assert "stop_recording" in str(exc), exc
    finally:
        builtins.input, torch.load, ev.battle_env.parallel_env = real_input, real_load, real_cls
        del ev.range
        tap.uninstall()
    env = created[0]
    rec.save("g3_2v2_evaluation_games", {"policy": "evaluate.py main() unmodified: red = shipped checkpoints + OU noise 0.1 (maddpg.Team.load_models), "
                                                   "blue = instinct.Team; the first %d games" % TRACE_GAMES, "seed": 12})
    lens.append(env.calls_this_game)
    cf = json.load(open(os.path.join(REF, "models/completed_model/cf.json")))
    out = {"games": np.int64(env.total_games), "ties": np.int64(env.ties), "red_wins": np.int64(env.team["red"]["wins"]),
           "blue_wins": np.int64(env.team["blue"]["wins"]), "calls_per_game": np.asarray(lens[-games:], np.int32),
           "cf": np.asarray([cf[k] for k in ("hit_base_reward", "hit_plane_reward", "miss_punishment", "die_punishment", "lose_punishment")], np.float64),
           "n_agents": np.int64(cf["n_agents"])}
    x = torch.tensor(np.stack(seen_obs)[:96])              # [rows, 2, 8]: red plane0 / plane1 observations met in play
    for i in range(2):
        net = nets.ActorNetwork(8, 4, 64, 64, 0.001, os.path.join(REF, "models/completed_model"), f"actor_plane{i}").to("cpu")
        net.load_checkpoint()
        net.eval()
        with torch.no_grad():
            y = net.forward(x[:, i])
        for k, v in net.state_dict().items():
            out[f"plane{i}/{k}"] = v.numpy()
        out[f"plane{i}/x"], out[f"plane{i}/y"] = x[:, i].numpy(), y.numpy()
    out["meta"] = np.asarray(json.dumps({"source": "evaluate.py main() unmodified, %d games" % games, "seed": 12,
                                         "win_rate_red": float(out["red_wins"]) / float(out["games"])}))
    np.savez_compressed(os.path.join(HERE, "g12_evaluation.npz"), **out)
    print("g12_evaluation:", {k: int(out[k]) for k in ("games", "ties", "red_wins", "blue_wins")}, "mean calls per game", float(np.mean(lens)))


if __name__ == "__main__":
    if "--actor-only" in sys.argv:
        actor_fixture()
        sys.exit(0)
    if "--learner-side-only" in sys.argv:
        learner_side_fixture()
        sys.exit(0)
    if "--evaluation-only" in sys.argv:                    # python tests/golden/make_golden.py --evaluation-only [games]
        cwd = os.getcwd()
        try:
            evaluation_fixture(int(sys.argv[sys.argv.index("--evaluation-only") + 1]) if sys.argv[-1].isdigit() else 2000)
        finally:
            os.chdir(cwd)
        sys.exit(0)
    cwd = os.getcwd()
    try:
        main()
    finally:
        os.chdir(cwd)
    if "--instinct-only" not in sys.argv:
        actor_fixture()
        learner_side_fixture()
