"""CPU oracle: scalar restatement of the reference Battlespace step() path.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module; the
product (deep-rl-battlespace_amd/) never does and has no CPU fallback.

Parity pinning: the reference has no tests or golden vectors of its own, so this
restatement is pinned against outputs of the reference itself, run unmodified in the build container
by tests/golden/make_golden.py (fixtures tests/golden/g*.npz; checked by tests/test_oracle_golden.py).
pygame 2.1.2 (third-party, absent from /root/reference and from this image) supplies the integer Rect
arithmetic on the path; its published semantics are restated here as explicit integer math:
float -> int by truncation toward zero, centre = top-left + (size >> 1), strict-overlap colliderect.

What is restated, with the reference lines each part follows (paths relative to /root/reference):
  RefEnv.__init__      envs/battle_env.py:73-184   constants, ids, spaces metadata
  RefEnv.reset         envs/battle_env.py:246-279, envs/sprites.py:74-91 (Plane.reset), :238-252 (Base.reset)
  RefEnv.step          envs/battle_env.py:281-381
  RefEnv._act          envs/battle_env.py:383-424  (process_action)
  RefEnv._forward      envs/sprites.py:123-141 + :35-42 (calc_new_xy)
  RefEnv._rotate       envs/sprites.py:93-113
  RefEnv._bullet_update envs/sprites.py:321-351    (Bullet.update) + :293-318 (Bullet.__init__)
  RefEnv.observe       envs/battle_env.py:202-244
  rel_angle / dist     envs/battle_env.py:38-58
  RefEnv._tie/_win     envs/battle_env.py:469-496

Arithmetic notes.  All float math is Python float (IEEE binary64) through the stdlib `math` module,
i.e. the host libm -- the same functions the reference calls.  Continuous actions are converted with
float(): the reference's pinned numpy 1.23.1 promotes a float32 action scalar to float64 on its first
operation with a Python number, so float64 arithmetic on the (possibly float32-representable) action
values is the reference's arithmetic.
"""
import math
import random as _stdlib_random

import numpy as np

FIELD_W = 1200          # sprites.py:9
FIELD_H = 800           # sprites.py:10
PLANE_W, PLANE_H = 50, 48   # assets/{red,blue}_plane.png (size read at sprites.py:61-62)
BASE_W, BASE_H = 62, 62     # assets/{red,blue}_base.png  (sprites.py:226-227)
BULLET_W, BULLET_H = 6, 3   # sprites.py:306
BULLET_SLOTS = 12           # a bullet is removed at the latest on its 12th update (12*45 >= 500)
WINNER_CODE = {"none": 0, "red": 1, "blue": 2, "tie": 3}


def rel_angle(p0, a0, p1):
    """battle_env.py:38-52."""
    rads = math.atan2(p0[1] - p1[1], p0[0] - p1[0])
    rads %= 2 * math.pi
    degs = math.degrees(rads)
    r = 180 + a0 - (360 - degs)
    if r < -180:
        r += 360
    if r > 180:
        r -= 360
    return r


def dist(p1, p0):
    """battle_env.py:54-58."""
    return math.sqrt((p1[0] - p0[0]) ** 2 + (p1[1] - p0[1]) ** 2)


def tie_tick(n_agents, time_step=0.1):
    """Number of the step() call on which the time-limit tie fires: the reference accumulates
    total_time += 0.1 in binary64 and compares >= max_time (battle_env.py:316-323,168)."""
    max_time = 10 + n_agents * 2
    t, k = 0, 0
    while True:
        t += time_step
        k += 1
        if t >= max_time:
            return k


class _Box:
    def __init__(self, low, high, shape, dtype):
        self.low = np.full(shape, low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)


class _Discrete:
    def __init__(self, n):
        self.n = n
        self.shape = ()
        self.dtype = np.dtype(np.int64)


class RefEnv:
    """One Battlespace game, advanced the way the reference advances it."""

    metadata = {"render_modes": ["human"], "name": "battle_env_v1"}  # battle_env.py:68-71

    def __init__(self, n_agents=1, show=False, hit_base_reward=100, hit_plane_reward=10,
                 miss_punishment=-1, die_punishment=-5, lose_punishment=-20, fps=20,
                 continuous_actions=False, rng=None):
        self.n_agents = n = n_agents
        self.rng = rng if rng is not None else _stdlib_random   # .randint(a, b) inclusive, .random()
        self.base_hp = 5 * n                   # :91
        self.plane_hp = 4                      # :92
        self.possible_agents = [f"plane{r}" for r in range(2 * n)]       # :106
        self.possible_red = self.possible_agents[:n]
        self.possible_blue = self.possible_agents[n:]
        self.agents = self.possible_agents[:]
        self.team_map = {a: ("red" if i < n else "blue") for i, a in enumerate(self.possible_agents)}
        self._idx = {a: i for i, a in enumerate(self.possible_agents)}
        self.team = {"red": {"wins": 0}, "blue": {"wins": 0}}
        self.obs_size = 3 * n + 2              # :132
        # :133-134  Box(high, -high): low = +1, high = -1 (swapped in the reference; reproduced)
        obs_space = _Box(1.0, -1.0, (self.obs_size,), np.float32)
        self.observation_spaces = {a: obs_space for a in self.possible_agents}
        self.continuous_actions = bool(continuous_actions)
        if self.continuous_actions:            # :145-155
            self.n_actions = 3
            self.max_turn = 35
            self.max_speed = 275
            self.min_speed = 200
            act_space = _Box(-1.0, 1.0, (3,), np.float32)
        else:                                  # :156-160
            self.n_actions = 4
            self.step_turn = 15
            self.speed = 215
            act_space = _Discrete(4)
        self.action_spaces = {a: act_space for a in self.possible_agents}
        self.width, self.height = FIELD_W, FIELD_H
        self.max_time = 10 + n * 2             # :168
        self.total_games = 0
        self.ties = 0
        self.bullet_speed = 450
        self.shot_dist = 500
        self.total_time = 0
        self.time_step = 0.1
        self.show = show
        self.hit_base_reward = hit_base_reward
        self.hit_plane_reward = hit_plane_reward
        self.miss_punishment = miss_punishment
        self.die_punishment = die_punishment
        self.lose_punishment = lose_punishment
        self.fps = fps
        self._diag = math.sqrt(math.pow(self.width, 2) + math.pow(self.height, 2))   # :230
        # spawn ranges (sprites.py:63-66,228-231): xmin=w, xmax=W-w, ymin=h, ymax=H-h
        self._pl_rng = (PLANE_W, FIELD_W - PLANE_W, PLANE_H, FIELD_H - PLANE_H)
        self._bs_rng = (BASE_W, FIELD_W - BASE_W, BASE_H, FIELD_H - BASE_H)
        # the constructor builds bases and planes once, drawing 4 + 3A values (:98-118)
        self._spawn(None)
        self.bullets = []
        self.dones = {a: False for a in self.possible_agents}
        self.env_done = False
        self.winner = "none"
        self.tick = 0

    # ---- spaces (battle_env.py:186-200)
    def observation_space(self, agent):
        return self.observation_spaces[agent]

    def action_space(self, agent):
        return self.action_spaces[agent]

    # ---- spawn: Base.reset x2 then Plane.reset per plane, red ids first (:257-268)
    def _spawn(self, forced):
        n, rint = self.n_agents, self.rng.randint
        xmin, xmax, ymin, ymax = self._bs_rng
        if forced is None:
            brx = rint(xmin, xmax // 3); bry = rint(ymin, ymax)            # sprites.py:246-247
            bbx = rint(xmax // 3 * 2, xmax); bby = rint(ymin, ymax)        # sprites.py:250-251
        else:
            brx, bry, bbx, bby = (int(v) for v in forced[:4])
        self.base_x = [brx, bbx]
        self.base_y = [bry, bby]
        self.base_hp_now = [self.base_hp, self.base_hp]
        xmin, xmax, ymin, ymax = self._pl_rng
        self.px, self.py, self.pdir = [], [], []
        for i in range(2 * n):
            if forced is not None:
                x, y, d = forced[4 + 3 * i: 7 + 3 * i]
                x, y = int(x), int(y)
                d = int(d) if float(d).is_integer() else float(d)
            elif i < n:                                                   # sprites.py:81-86
                x = rint(xmin, xmax // 3); y = rint(ymin, ymax)
                d = rint(270, 450)
                if d >= 360:
                    d -= 360
            else:                                                         # sprites.py:87-91
                x = rint(xmax // 3 * 2, xmax); y = rint(ymin, ymax)
                d = rint(90, 270)
            self.px.append(x); self.py.append(y); self.pdir.append(d)
        self.php = [self.plane_hp] * (2 * n)
        self.palive = [True] * (2 * n)

    def reset(self, seed=None, return_info=False, options=None, spawn=None):
        """battle_env.py:246-279.  seed/return_info/options are accepted and ignored, as there.
        `spawn` (oracle-only) forces the 4+3A spawn values instead of drawing them."""
        self.winner = "none"
        self._spawn(spawn)
        self.total_time = 0
        self.tick = 0
        self.bullets = []
        self.agents = self.possible_agents[:]
        self.dones = {a: False for a in self.possible_agents}
        self.env_done = False
        return {a: self.observe(a) for a in self.possible_agents}

    # ---- observation (battle_env.py:202-244)
    def observe(self, agent):
        i = self._idx[agent]
        n = self.n_agents
        obs = -np.ones(self.obs_size, dtype=np.float32)
        if not self.palive[i]:
            return obs
        me = (self.px[i], self.py[i])
        a0 = self.pdir[i]
        eb = 1 if i < n else 0                       # enemy base index
        base = (self.base_x[eb], self.base_y[eb])
        obs[0] = dist(me, base) / self._diag * 2 - 1
        obs[1] = rel_angle(me, a0, base) / 360
        k = 2
        for j in (range(n, 2 * n) if i < n else range(0, n)):
            if self.palive[j]:
                p = (self.px[j], self.py[j])
                obs[k] = 1
                obs[k + 1] = dist(me, p) / self._diag * 2 - 1
                obs[k + 2] = rel_angle(me, a0, p) / 360
            k += 3
        return obs

    def _all_obs(self):
        return {a: self.observe(a) for a in self.possible_agents}

    # ---- plane kinematics
    def _clamp(self, i):
        """sprites.py:134-141 on the un-rotated 50x48 rect: left = cx-25, right = cx+25, top = cy-24, bottom = cy+24."""
        if self.px[i] - (PLANE_W >> 1) < 0:
            self.px[i] = PLANE_W >> 1
        if self.px[i] - (PLANE_W >> 1) + PLANE_W > FIELD_W:
            self.px[i] = FIELD_W - PLANE_W + (PLANE_W >> 1)
        if self.py[i] - (PLANE_H >> 1) <= 0:
            self.py[i] = PLANE_H >> 1
        if self.py[i] - (PLANE_H >> 1) + PLANE_H >= FIELD_H:
            self.py[i] = FIELD_H - PLANE_H + (PLANE_H >> 1)

    def _forward(self, i, speed, time):
        """sprites.py:123-141; calc_new_xy :35-42; Rect centre store truncates toward zero."""
        ang = -math.radians(self.pdir[i])
        nx = self.px[i] + (speed * time * math.cos(ang))
        ny = self.py[i] + (speed * time * math.sin(ang))
        self.px[i] = int(nx)
        self.py[i] = int(ny)
        self._clamp(i)

    def _rotate(self, i, angle):
        """sprites.py:93-113: direction stays in [0, 360] inclusive."""
        d = self.pdir[i] + angle
        while d > 360:
            d -= 360
        while d < 0:
            d += 360
        self.pdir[i] = d
        self._clamp(i)

    def _shoot(self, i, x, y, d, u):
        """sprites.py:293-318: one random() per shot; direction = angle + (random()*8 - 4)."""
        if u is None:
            u = self.rng.random()
        self.bullets.append([i, x, y, d + (u * 8 - 4), 0])   # shooter, cx, cy, direction, updates so far

    def _act(self, action, i, u):
        """battle_env.py:383-424."""
        if not self.palive[i]:
            return
        x0, y0, d0 = self.px[i], self.py[i], self.pdir[i]      # pose before the move (:397-398)
        if not self.continuous_actions:
            if action == 0:
                self._forward(i, self.speed, self.time_step)
            elif action == 1:
                self._shoot(i, x0, y0, d0, u)
                self._forward(i, self.speed, self.time_step)
            elif action == 2:
                self._rotate(i, self.step_turn)
                self._forward(i, self.speed, self.time_step)
            elif action == 3:
                self._rotate(i, -self.step_turn)
                self._forward(i, self.speed, self.time_step)
            # any other value: nothing happens
        else:
            a0, a1, a2 = float(action[0]), float(action[1]), float(action[2])
            speed = ((a0 + 1) / 2) * (self.max_speed - self.min_speed) + self.min_speed
            self._forward(i, speed, self.time_step)
            self._rotate(i, a1 * self.max_turn)
            if a2 > 0:
                self._shoot(i, x0, y0, d0, u)

    # ---- bullets
    def _bullet_update(self, b):
        """sprites.py:321-351.  Returns 'miss', ('base', idx), ('plane', idx) or 'none'."""
        shooter = b[0]
        ang = -math.radians(b[3])
        step = self.bullet_speed * self.time_step
        b[1] = int(b[1] + (step * math.cos(ang)))
        b[2] = int(b[2] + (step * math.sin(ang)))
        b[4] += 1
        if step * b[4] >= self.shot_dist:              # dist_travelled >= max_dist
            return "miss"
        cx, cy = b[1], b[2]
        if cx > FIELD_W or cx < 0 or cy > FIELD_H or cy < 0:
            return "miss"
        # bullet rect 6x3 around its centre: left = cx-3, right = cx+3, top = cy-1, bottom = cy+2
        bl, br, bt, bb = cx - 3, cx + 3, cy - 1, cy + 2
        n = self.n_agents
        eb = 1 if shooter < n else 0
        ex, ey = self.base_x[eb], self.base_y[eb]
        if bl < ex + 31 and bt < ey + 31 and br > ex - 31 and bb > ey - 31:   # base hit even if the base is dead
            return ("base", eb)
        for j in (range(n, 2 * n) if shooter < n else range(0, n)):           # live enemy planes, id order
            if not self.palive[j]:
                continue
            qx, qy = self.px[j], self.py[j]
            if bl < qx + 25 and bt < qy + 24 and br > qx - 25 and bb > qy - 24:
                return ("plane", j)
        return "none"

    # ---- termination (battle_env.py:469-496)
    def _tie(self):
        self.winner = "tie"
        self.total_games += 1
        self.ties += 1
        self.env_done = True
        self.dones = {a: True for a in self.possible_agents}

    def _win(self, winner):
        self.winner = winner
        self.total_games += 1
        self.team[winner]["wins"] += 1
        self.env_done = True
        self.dones = {a: True for a in self.possible_agents}

    # ---- one tick (battle_env.py:281-381)
    def step(self, actions, u=None):
        """`u` (oracle-only): forced random() value per agent index (sequence; None/NaN = draw)."""
        ids = self.possible_agents
        if self.continuous_actions:                       # :295-297
            for key, value in actions.items():
                actions[key] = np.clip(value, -1.0, 1.0)
        rewards = {a: 0 for a in ids}
        infos = {a: {} for a in ids}
        if self.env_done:                                 # :303-306
            return self._all_obs(), rewards, self.dones, infos
        if len(actions) == 0 or len(self.agents) == 0:    # :309-313
            self._tie()
            return self._all_obs(), rewards, self.dones, infos
        self.total_time += self.time_step                 # :316
        self.tick += 1
        if self.total_time >= self.max_time:              # :319-323
            self._tie()
            return self._all_obs(), rewards, self.dones, infos
        for agent_id in self.agents:                      # :325-329
            action = actions[agent_id]
            if type(action) == np.ndarray and not self.continuous_actions:
                action = np.argmax(action)
            i = self._idx[agent_id]
            ui = None
            if u is not None and u[i] is not None and not (isinstance(u[i], float) and math.isnan(u[i])):
                ui = float(u[i])
            self._act(action, i, ui)
        for b in self.bullets[:]:                         # :332-360
            out = self._bullet_update(b)
            shooter = ids[b[0]]
            if out == "miss":
                rewards[shooter] += self.miss_punishment
                self.bullets.remove(b)
            elif out == "none":
                pass
            elif out[0] == "base":
                self.base_hp_now[out[1]] -= 1             # Base.hit sprites.py:254-263 (hp may go negative)
                rewards[shooter] += self.hit_base_reward
                self.bullets.remove(b)
            else:
                j = out[1]
                self.php[j] -= 1                          # Plane.hit sprites.py:143-153
                rewards[shooter] += self.hit_plane_reward
                self.bullets.remove(b)
                if self.php[j] <= 0:
                    self.palive[j] = False
                    self.agents.remove(ids[j])
                    rewards[ids[j]] += self.die_punishment
                    self.dones[ids[j]] = True
        if self.base_hp_now[1] <= 0:                      # blue base dead -> red "wins" (:363-366)
            for a in self.possible_red:
                rewards[a] += self.lose_punishment
            self._win("red")
        if self.base_hp_now[0] <= 0:                      # :369-372
            for a in self.possible_blue:
                rewards[a] += self.lose_punishment
            self._win("blue")
        return self._all_obs(), rewards, self.dones, infos

    # ---- misc surface (battle_env.py:449-467)
    def wins(self):
        return "Wins by red: {}\nWins by blue: {}\nTied games: {}\nWin rate: {}".format(
            self.team["red"]["wins"], self.team["blue"]["wins"], self.ties,
            self.team["red"]["wins"] / self.total_games)

    def make_discrete(self, actions_dict):
        return {a: np.argmax(v) for a, v in actions_dict.items()}

    def close(self):
        pass

    def render(self, mode="human"):
        pass

    # ---- trace helpers for the tests (same schema as tests/golden/make_golden.py)
    def snapshot(self):
        A = 2 * self.n_agents
        live = np.zeros((A, BULLET_SLOTS), bool)
        bx = np.zeros((A, BULLET_SLOTS), np.int32)
        by = np.zeros((A, BULLET_SLOTS), np.int32)
        bd = np.zeros((A, BULLET_SLOTS), np.float64)
        # the time-limit tie call advances the clock but not the bullets: label slots by physics ticks
        ptick = self.tick - 1 if (self.env_done and self.winner == "tie" and self.total_time >= self.max_time) else self.tick
        for sh, x, y, d, age in self.bullets:
            s = (ptick - age + 1) % BULLET_SLOTS
            assert not live[sh, s]
            live[sh, s] = True; bx[sh, s] = x; by[sh, s] = y; bd[sh, s] = d
        return dict(px=np.asarray(self.px, np.int32), py=np.asarray(self.py, np.int32),
                    pdir=np.asarray(self.pdir, np.float64), php=np.asarray(self.php, np.int32),
                    palive=np.asarray(self.palive, bool), bhp=np.asarray(self.base_hp_now, np.int32),
                    tick=self.tick, total_time=float(self.total_time),
                    bl_live=live, bl_x=bx, bl_y=by, bl_dir=bd,
                    total_games=self.total_games, ties=self.ties,
                    wins_red=self.team["red"]["wins"], wins_blue=self.team["blue"]["wins"])
