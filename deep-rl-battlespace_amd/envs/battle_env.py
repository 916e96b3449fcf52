"""`parallel_env`: the reference's PettingZoo/Gym ParallelEnv surface (reference envs/battle_env.py:61-496), with the
whole step() path -- kinematics, bullets, hits, rewards, dones, observations -- executed for `n_envs` independent
games by one fused HIP kernel launch on an MI355X (csrc/bsx_kernels.hip through the C ABI of
include/battlespace_hip.h).  Game state lives in one device buffer owned by this object; nothing on the step path
runs on the CPU and there is no CPU fallback.

Two ways to use it

  * drop-in (``n_envs=None``, the default): one game, and every return value has the reference's type -- observation
    dicts of float32 numpy rows, Python-number rewards, bool dones, ``env_done`` a bool, ``winner`` a str, ``agents``
    the list of live ids.  Randomness comes from the stdlib ``random`` module in the reference's draw order
    (sprites.py:82-91,246-252,314), so after ``random.seed(s)`` the game is the reference's game.  Each call
    synchronises with the device; this mode is for parity and for callers that cannot batch.

  * batched (``n_envs=E``): every per-agent value gains a leading E axis and stays on the device:
    ``step(actions) -> (obs, rewards, dones, infos)`` with dicts ``{"plane{i}": tensor[E, ...]}`` that are views of
    column i of the env-owned output tensors (overwritten by the next call), or ``step_batch(actions[E, A])`` which
    returns the ``[E, A, ...]`` tensors themselves and builds no dicts.  Randomness is in-kernel Philox4x32-10 keyed
    by (seed, global env index, game number, tick, agent): results do not depend on how envs are sharded over GPUs.

Reference quirks are reproduced, not fixed: observation Box has low=+1/high=-1; the winning
team receives ``lose_punishment``; headings live in [0, 360] inclusive; a discrete action outside 0..3 means "do not
move"; the time-limit tie fires on call number 10*max_time + 1; a finished game ignores step() until reset().
"""
import contextlib
import ctypes
import random as _stdlib_random
import warnings

import numpy as np
import torch

from .. import _lib
from ..spaces import Box, Discrete

DISP_WIDTH = 1200    # reference envs/sprites.py:9
DISP_HEIGHT = 800    # reference envs/sprites.py:10


def draw_spawn(n_agents, randint=None):
    """4 + 3A stdlib draws in the reference's order: base red x,y, base blue x,y (sprites.py:238-252), then x,y,dir per
    plane, red ids first (sprites.py:74-91; battle_env.py:257-268).  Red headings are randint(270,450) folded."""
    n, ri = n_agents, (randint if randint is not None else _stdlib_random.randint)
    out = [ri(62, 1138 // 3), ri(62, 738), ri(1138 // 3 * 2, 1138), ri(62, 738)]
    for i in range(2 * n):
        if i < n:
            x, y, d = ri(50, 1150 // 3), ri(48, 752), ri(270, 450)
            if d >= 360:
                d -= 360
        else:
            x, y, d = ri(1150 // 3 * 2, 1150), ri(48, 752), ri(90, 270)
        out += [x, y, d]
    return out


_current_device = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)   # the raw binding: no Python frames per call
_NULL_CONTEXT = contextlib.nullcontext()


class _TeamCounters:
    """`env.team[colour]['wins']` (reference battle_env.py:102-103,492) backed by the device counters."""

    def __init__(self, env, col):
        self._env, self._col = env, col

    def __getitem__(self, key):
        if key != "wins":
            raise KeyError(key)
        c = self._env.counters()
        v = c[:, self._col]
        return int(v[0]) if self._env._compat else v


class parallel_env:
    metadata = {"render_modes": ["human"], "name": "battle_env_v1"}   # battle_env.py:68-71

    def __init__(self, n_agents=1, show=False, hit_base_reward=100, hit_plane_reward=10, miss_punishment=-1,
                 die_punishment=-5, lose_punishment=-20, fps=20, continuous_actions=False,
                 n_envs=None, device=None, seed=0, auto_reset=False, env_offset=0, rng=None, wide_offsets=False, one_wave=False):
        """First nine arguments: exactly the reference constructor (battle_env.py:73).

        n_envs      None = drop-in single game (reference return types); int = batched on-device tensors
        device      torch device of the MI355X to run on (default: current cuda device)
        seed        Philox key for in-kernel spawn / bullet-jitter draws
        auto_reset  batched mode: a step() on a finished game re-spawns it instead of the reference's inert call
        env_offset  global index of this object's env 0 (sharding: rank r owns [r*E, (r+1)*E))
        rng         "python" (stdlib random, reference draw order; default when n_envs is None) or "philox"
        wide_offsets  take the 64-bit-offset kernels (BSX_F_WIDE_OFFSETS) although the job is small enough for 32-bit offsets;
                    the library switches by itself above 4 GB per array -- same results, for tests
        one_wave    BSX_F_ONE_WAVE: keep the one-wave kernel where the library would take a two-wave form (1v1: step_many of up to 65 536 games,
                    step / step_batch of up to 114 688 games, 81 920 with continuous actions) -- same results, for the tests that compare the two and for A/B runs
        """
        if not isinstance(n_agents, (int, np.integer)) or not 1 <= n_agents <= _lib.MAX_N:
            raise ValueError(f"n_agents must be an int in 1..{_lib.MAX_N}, got {n_agents!r}")
        self._lib = _lib.load()                       # raises if the HIP extension has not been built
        if not torch.cuda.is_available():
            raise RuntimeError("parallel_env needs an MI355X visible to torch (torch.cuda.is_available() is False); "
                               "the step() path has no CPU implementation")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type == "cuda" and self.device.index is None:           # "cuda" -> the current device, by number
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._dev_index = self.device.index
        if self.device.type != "cuda":
            raise ValueError(f"device must be a cuda (HIP) device, got {self.device}")
        self._compat = n_envs is None
        self.n_envs = E = 1 if n_envs is None else int(n_envs)
        if E < 1:
            raise ValueError("n_envs must be >= 1")
        self.n_agents = n = int(n_agents)
        self._A = A = 2 * n
        self.base_hp = 5 * n
        self.plane_hp = 4
        self.possible_agents = [f"plane{r}" for r in range(A)]          # battle_env.py:106-109
        self.possible_red = self.possible_agents[:n]
        self.possible_blue = self.possible_agents[n:]
        self.team_map = {a: ("red" if i < n else "blue") for i, a in enumerate(self.possible_agents)}
        self._idx = {a: i for i, a in enumerate(self.possible_agents)}
        self.team = {"red": _TeamCounters(self, 2), "blue": _TeamCounters(self, 3)}
        self.obs_size = D = 3 * n + 2                                   # battle_env.py:132
        high = np.ones(D, dtype=np.float32)
        obs_space = Box(high, -high)                                    # battle_env.py:134 (low=+1, high=-1: as there)
        self.observation_spaces = {a: obs_space for a in self.possible_agents}
        self.continuous_actions = bool(continuous_actions)
        if self.continuous_actions:                                     # battle_env.py:145-155
            self.n_actions, self.max_turn, self.max_speed, self.min_speed = 3, 35, 275, 200
            action_space = Box(-1.0, 1.0, shape=(3,), dtype=np.float32)
        else:                                                           # battle_env.py:156-160
            self.n_actions, self.step_turn, self.speed = 4, 15, 215
            action_space = Discrete(4)
        self.action_spaces = {a: action_space for a in self.possible_agents}
        self.width, self.height = DISP_WIDTH, DISP_HEIGHT
        self.max_time = 10 + n * 2
        self.bullet_speed, self.shot_dist, self.time_step = 450, 500, 0.1
        self.show = show
        self.hit_base_reward, self.hit_plane_reward = hit_base_reward, hit_plane_reward
        self.miss_punishment, self.die_punishment, self.lose_punishment = miss_punishment, die_punishment, lose_punishment
        self.fps = fps
        self.recording = False
        self._int_rewards = all(isinstance(v, (int, np.integer)) and not isinstance(v, bool) for v in
                                (hit_base_reward, hit_plane_reward, miss_punishment, die_punishment, lose_punishment))
        self._cfg = _lib.BsxRewards(float(hit_base_reward), float(hit_plane_reward), float(miss_punishment),
                                    float(die_punishment), float(lose_punishment))
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.auto_reset = bool(auto_reset)
        self._base_flags = (_lib.F_AUTO_RESET if self.auto_reset else 0) | (_lib.F_WIDE_OFFSETS if wide_offsets else 0) | \
            (_lib.F_ONE_WAVE if one_wave else 0)
        self.env_offset = int(env_offset)
        self.rng = rng if rng is not None else ("python" if self._compat else "philox")
        if self.rng not in ("python", "philox"):
            raise ValueError("rng must be 'python' or 'philox'")
        if self.auto_reset and self.rng == "python":
            raise ValueError("auto_reset draws spawns in-kernel: use rng='philox'")
        if self.auto_reset and self._compat:
            # the drop-in surface mirrors `agents`, `dones`, `env_done` on the host by the reference's rules, in which a finished game
            # stays finished until reset() (battle_env.py:303-306); an in-kernel re-spawn would leave those mirrors stale
            raise ValueError("auto_reset needs a batched env (n_envs=...): the drop-in surface keeps the reference's reset() contract")
        self.tie_tick = int(self._lib.bsx_tie_tick(n))

        # ---- device buffers
        nbytes = ctypes.c_size_t()
        _lib.check(self._lib.bsx_state_bytes(E, n, ctypes.byref(nbytes)), "bsx_state_bytes")
        dev = self.device
        self._state = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        self._obs = torch.empty((E, A, D), dtype=torch.float32, device=dev)
        self._rew = torch.empty((E, A), dtype=torch.float32, device=dev)
        self._done = torch.empty((E, A), dtype=torch.uint8, device=dev)
        self._env_done = torch.empty(E, dtype=torch.uint8, device=dev)
        self._winner = torch.empty(E, dtype=torch.uint8, device=dev)
        self._u = None
        self._reset_nonce = 0
        if self._compat:
            self._init_compat_io()
        # raw pointers of the env-owned buffers (never re-allocated): one attribute read per call instead of a data_ptr() each
        self._p_state, self._p_obs, self._p_rew, self._p_done = (t.data_ptr() for t in (self._state, self._obs, self._rew, self._done))
        self._p_env_done, self._p_winner = self._env_done.data_ptr(), self._winner.data_ptr()
        if self._compat:                                  # drop-in mode: the outputs live in pinned host memory; the kernels take its device address
            cp = self._compat_dev_ptrs
            self._p_obs, self._p_rew, self._p_done, self._p_env_done, self._p_winner = cp["obs"], cp["rew"], cp["done"], cp["env_done"], cp["winner"]
        self._cfg_ref = ctypes.byref(self._cfg)
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the current stream's handle without building a Stream object
        with self._guard():
            _lib.check(self._lib.bsx_state_init(self._state.data_ptr(), E, n, self._stream()), "bsx_state_init")
        self._done_bool = self._done.view(torch.bool)
        self._env_done.fill_(1)
        self._winner.zero_()
        self._done.fill_(1)
        self._rew.zero_()
        # host mirrors (only maintained when rng == "python" or in drop-in mode)
        self._mirror = self._compat or self.rng == "python"
        self._h_alive = np.ones((E, A), bool)
        self._h_done = np.ones(E, bool)
        self._h_tick = np.zeros(E, np.int64)
        self.dones = {a: False for a in self.possible_agents} if self._compat else None
        self._winner_name = "none"
        # the reference constructor builds bases and planes once, consuming 4 + 3A draws (battle_env.py:98-118)
        if self.rng == "python":
            for _ in range(E):
                self._draw_spawn()

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        if self._raw_stream is not None:
            return self._raw_stream(self._dev_index)
        return torch.cuda.current_stream(self.device).cuda_stream

    def _sync(self):
        """Drop-in mode: wait for the launches of this call (their outputs sit in pinned host memory, read by the host next)."""
        _lib.check(self._lib.bsx_stream_synchronize(self._stream()), "bsx_stream_synchronize")

    def _guard(self):
        """HIP launches go to the calling thread's current device: make that the env's device for the duration of a call
        (a no-op context when it already is, which is the normal one-process-per-GPU case)."""
        if _current_device() == self._dev_index:
            return _NULL_CONTEXT
        return torch.cuda.device(self.device)

    def _draw_spawn(self):
        return draw_spawn(self.n_agents)

    def _agent_views(self, t):
        """{agent id: column i of t}.  The env-owned output tensors never change identity, so their column views are built once and
        every call returns a fresh dict of the same views (the reference hands out a new dict per step, battle_env.py:374-381)."""
        cache = self.__dict__.setdefault("_view_cache", {})
        key = (t.data_ptr(), t.dtype, tuple(t.shape))
        views = cache.get(key)
        if views is None:
            views = {a: t[:, i] for i, a in enumerate(self.possible_agents)}
            if any(t is own for own in (self._obs, self._rew, self._done_bool)):
                cache[key] = views                                   # only the persistent tensors are cached (copies come and go)
        return dict(views)

    # ------------------------------------------------------------------ spaces (battle_env.py:186-200)
    def observation_space(self, agent):
        return self.observation_spaces[agent]

    def action_space(self, agent):
        return self.action_spaces[agent]

    # ------------------------------------------------------------------ reset (battle_env.py:246-279)
    def reset(self, seed=None, return_info=False, options=None, spawn=None, mask=None):
        """`seed`, `return_info`, `options` are accepted and ignored, as in the reference.
        spawn: optional int32 [E, 4+3A] (base_red x,y, base_blue x,y, then x,y,dir per plane): forced spawn states.
        mask:  optional bool/uint8 [E]: reset only these games (batched mode)."""
        E, A = self.n_envs, self._A
        mask_t = None
        if mask is not None:
            mask_t = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            if mask_t.shape != (E,):
                raise ValueError(f"mask must have shape ({E},)")
        if spawn is None and self.rng == "python":
            sel = range(E) if mask is None else [i for i in range(E) if bool(mask_t[i])]
            rows = np.zeros((E, 4 + 3 * A), np.int32)
            for i in sel:
                rows[i] = self._draw_spawn()
            spawn = rows
        spawn_t = None
        if spawn is not None:
            spawn_t = torch.as_tensor(np.asarray(spawn) if not torch.is_tensor(spawn) else spawn)
            spawn_t = spawn_t.to(device=self.device, dtype=torch.int32).contiguous()
            if spawn_t.shape != (E, 4 + 3 * A):
                raise ValueError(f"spawn must have shape ({E}, {4 + 3 * A}), got {tuple(spawn_t.shape)}")
        self._reset_nonce += 1
        with self._guard():
            _lib.check(self._lib.bsx_reset(self._state.data_ptr(), E, self.n_agents,
                                           mask_t.data_ptr() if mask_t is not None else None,
                                           spawn_t.data_ptr() if spawn_t is not None else None,
                                           self.seed, self._reset_nonce, self.env_offset, self._p_obs,
                                           self._stream()), "bsx_reset")
        if mask_t is None:
            self._env_done.zero_(); self._winner.zero_(); self._done.zero_()
        else:
            m = mask_t.bool().to(self._env_done.device)       # (drop-in mode: these rows are views of pinned HOST memory)
            self._env_done[m] = 0; self._winner[m] = 0; self._done[m] = 0
        if self._mirror:
            self._sync_mirror()
        if self._compat:
            self.dones = {a: False for a in self.possible_agents}
            self._winner_name = "none"
            self._sync()
            o = self._ho_obs
            return {a: o[i].copy() for i, a in enumerate(self.possible_agents)}
        return self._agent_views(self._obs)

    # ------------------------------------------------------------------ step (battle_env.py:281-381)
    def _pack_actions(self, actions):
        """-> (tensor or None, kind, empty).  Accepts a dict {agent: per-env values} or one [E, A(, k)] tensor."""
        E, A, dev = self.n_envs, self._A, self.device
        if isinstance(actions, dict):
            if len(actions) == 0:
                return None, 0, True
            cols = []
            for a in self.possible_agents:
                v = actions.get(a)
                if v is None:       # the reference only reads actions of live agents; absent = "no movement"
                    v = torch.zeros((E, 3)) if self.continuous_actions else torch.full((E,), -1)
                if not torch.is_tensor(v):
                    v = torch.as_tensor(np.asarray(v))
                if v.dim() == 0:
                    v = v.reshape(1)
                if v.shape[0] != E:
                    raise ValueError(f"action for {a}: leading dimension must be n_envs={E}, got shape {tuple(v.shape)}")
                cols.append(v.to(dev))
            if len({c.dim() for c in cols}) != 1:
                # mixed per-agent encodings (indices and score vectors): arg-max the vectors (battle_env.py:327-328)
                cols = [c.argmax(-1) if c.dim() == 2 else c for c in cols]
            t = torch.stack(cols, dim=1)
        else:
            t = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions))
            t = t.to(dev)
        if self.continuous_actions:
            if t.shape != (E, A, 3):
                raise ValueError(f"continuous actions must have shape ({E}, {A}, 3), got {tuple(t.shape)}")
            if t.dtype == torch.float64:
                return t.contiguous(), _lib.ACT_F64, False
            return t.to(torch.float32).contiguous(), _lib.ACT_F32, False
        if t.dim() == 3:
            if t.shape != (E, A, 4):
                raise ValueError(f"discrete action vectors must have shape ({E}, {A}, 4), got {tuple(t.shape)}")
            return t.to(torch.float32).contiguous(), _lib.ACT_LOGITS_F32, False
        if t.shape != (E, A):
            raise ValueError(f"discrete actions must have shape ({E}, {A}), got {tuple(t.shape)}")
        if t.is_floating_point():
            raise TypeError("discrete actions must be integers (or [E, A, 4] score vectors)")
        return t.to(torch.int32).contiguous(), _lib.ACT_I32, False

    def _draw_jitter(self, act_t, kind, empty):
        """rng='python': one stdlib random() per shot, live agents in id order, only on calls that reach the physics
        (battle_env.py:303-329, sprites.py:314).  Returns a float64 [E, A] tensor (NaN = no shot)."""
        E, A = self.n_envs, self._A
        u = np.full((E, A), np.nan)
        if empty:
            return u
        a = act_t.detach().cpu().numpy()
        if self.continuous_actions:
            shoot = np.clip(a[..., 2].astype(np.float64), -1, 1) > 0
        elif kind == _lib.ACT_LOGITS_F32:
            shoot = a.argmax(-1) == 1
        else:
            shoot = a == 1
        for e in range(E):
            if self._h_done[e] or not self._h_alive[e].any() or self._h_tick[e] + 1 >= self.tie_tick:
                continue
            for i in range(A):
                if self._h_alive[e, i] and shoot[e, i]:
                    u[e, i] = _stdlib_random.random()
        return u

    def step_batch(self, actions, u=None, copy=False):
        """One tick for all games.  actions: int [E, A] | float [E, A, 4] score vectors (arg-maxed in-kernel) |
        float [E, A, 3] when continuous | {} (every running game ties).  u: optional float64 [E, A] of random()
        values to use for this call's shots instead of the generator.  Returns the env-owned tensors
        (obs [E, A, D] f32, rew [E, A] f32, done [E, A] bool), overwritten by the next call; copy=True returns fresh
        tensors instead (the reference hands out fresh arrays every step, battle_env.py:374-381)."""
        if (u is None and not self._mirror and not copy and torch.is_tensor(actions) and actions.dtype == torch.int32
                and not self.continuous_actions and actions.device == self.device and actions.shape == self._obs.shape[:2]
                and actions.is_contiguous()):
            # the common batched call: int32 [E, A] on the env's device -- nothing to convert, nothing to mirror
            self._launch(actions.data_ptr(), _lib.ACT_I32, False, None, self._p_obs, self._p_rew, self._p_done)
            return self._obs, self._rew, self._done_bool
        act_t, kind, empty = self._pack_actions(actions)
        if u is None and self.rng == "python":
            u = self._draw_jitter(act_t, kind, empty)
        u_ptr = None
        if u is not None:
            ut = torch.as_tensor(np.asarray(u, dtype=np.float64) if not torch.is_tensor(u) else u)
            self._u = ut.to(device=self.device, dtype=torch.float64).contiguous()
            if self._u.shape != (self.n_envs, self._A):
                raise ValueError(f"u must have shape ({self.n_envs}, {self._A})")
            u_ptr = self._u.data_ptr()
        self._launch(act_t.data_ptr() if act_t is not None else None, kind, empty, u_ptr,
                     self._p_obs, self._p_rew, self._p_done)
        if self._mirror:
            self._sync_mirror()
        if copy:
            return self._obs.clone(), self._rew.clone(), self._done_bool.clone()
        return self._obs, self._rew, self._done_bool

    def _launch(self, act_ptr, kind, empty, u_ptr, obs_ptr, rew_ptr, done_ptr, env_done_ptr=None, games=None):
        """Enqueue one fused step kernel on the current stream (no sync, no allocation: graph-capturable).
        env_done_ptr: where this call's env_done [E] goes instead of the env-owned tensor (a rollout's per-tick record; the
        caller copies the last one back into `self._env_done`).
        games: (first, count) = step only that range of the games (bsx_step_*_range; every pointer still the full array's)."""
        flags = self._base_flags | (_lib.F_EMPTY_CALL if empty else 0)
        if games is not None:
            fn = self._lib.bsx_step_continuous_range if self.continuous_actions else self._lib.bsx_step_discrete_range
            with self._guard():
                _lib.check(fn(self._p_state, self.n_envs, self.n_agents, int(games[0]), int(games[1]), act_ptr, kind, u_ptr, obs_ptr,
                              rew_ptr, done_ptr, env_done_ptr if env_done_ptr is not None else self._p_env_done, self._p_winner,
                              self._cfg_ref, flags, self.seed, self.env_offset, self._stream()), "bsx_step_range")
            return
        fn = self._lib.bsx_step_continuous if self.continuous_actions else self._lib.bsx_step_discrete
        if _current_device() == self._dev_index:           # the normal one-process-per-GPU case: no context object at all on the per-call path
            rc = fn(self._p_state, self.n_envs, self.n_agents, act_ptr, kind, u_ptr, obs_ptr, rew_ptr, done_ptr,
                    env_done_ptr if env_done_ptr is not None else self._p_env_done, self._p_winner,
                    self._cfg_ref, flags, self.seed, self.env_offset, self._stream())
        else:
            with torch.cuda.device(self.device):
                rc = fn(self._p_state, self.n_envs, self.n_agents, act_ptr, kind, u_ptr, obs_ptr, rew_ptr, done_ptr,
                        env_done_ptr if env_done_ptr is not None else self._p_env_done, self._p_winner,
                        self._cfg_ref, flags, self.seed, self.env_offset, self._stream())
        if rc:
            _lib.check(rc, "bsx_step")

    def _check_action_series(self, actions):
        """actions for T calls: [T, E, A] int32 | [T, E, A, 4] float32 | [T, E, A, 3] float32/float64 (continuous) |
        [T, E, A, 4] float32 (continuous: speed, turn, shoot + one ignored column, as an actor writes them)."""
        E, A = self.n_envs, self._A
        if not torch.is_tensor(actions) or actions.device != self.device or not actions.is_contiguous() or actions.dim() < 3:
            raise ValueError("actions must be a contiguous [T, E, A, ...] tensor on the env's device")
        T = actions.shape[0]
        if self.continuous_actions and actions.dim() == 4 and actions.shape[3] == 4:     # an actor's 4-wide rows: 3 actions + padding
            ok, kind = actions.shape == (T, E, A, 4) and actions.dtype == torch.float32, _lib.ACT_F32X4
        elif self.continuous_actions:
            ok, kind = actions.shape == (T, E, A, 3) and actions.dtype in (torch.float32, torch.float64), \
                (_lib.ACT_F64 if actions.dtype == torch.float64 else _lib.ACT_F32)
        elif actions.dim() == 4:
            ok, kind = actions.shape == (T, E, A, 4) and actions.dtype == torch.float32, _lib.ACT_LOGITS_F32
        else:
            ok, kind = actions.shape == (T, E, A) and actions.dtype == torch.int32, _lib.ACT_I32
        if not ok or T < 1:
            raise ValueError(f"bad actions tensor for {T} calls: shape {tuple(actions.shape)}, dtype {actions.dtype}")
        return T, kind

    def step_many(self, actions, store=False, out=None, u=None, env_done_out=None):
        """T consecutive step() calls in ONE kernel launch (`for t: step(actions[t])` when all actions are known up
        front: random or scripted play, replays).  Results are those of T step_batch() calls, bit for bit; each
        wavefront walks its own games through the T ticks, so the state stays in the L2 between ticks instead of
        crossing a kernel boundary.

        actions: device tensor [T, E, A] int32 | [T, E, A, 4] float32 scores | [T, E, A, 3] float32/64 (continuous)
        store:   False = obs/rew/done are the env-owned [E, A, ...] tensors and hold the last tick's results;
                 True  = new (or `out`) [T, E, A, ...] tensors hold every tick's results
        out:     optional (obs f32 [T,E,A,D], rew f32 [T,E,A], done uint8 [T,E,A]) to fill when store=True
        u:       optional float64 [T, E, A] random() values for the shots (parity runs); needs rng='philox' otherwise
        env_done_out: optional uint8 [T, E] tensor that receives env_done after every tick (row t-1 set = call t found the game
                 finished: the inert / re-spawn call of battle_env.py:303-306, not a transition)"""
        if self._compat:
            raise ValueError("step_many needs a batched env (n_envs=...)")
        if self.rng != "philox" and u is None:
            raise ValueError("step_many needs rng='philox' (or injected u)")
        T, kind = self._check_action_series(actions)
        E, A, D = self.n_envs, self._A, self.obs_size
        if T > _lib.MAX_T:
            raise ValueError(f"at most {_lib.MAX_T} ticks per launch")
        if store:
            if out is None:
                out = (torch.empty((T, E, A, D), dtype=torch.float32, device=self.device),
                       torch.empty((T, E, A), dtype=torch.float32, device=self.device),
                       torch.empty((T, E, A), dtype=torch.uint8, device=self.device))
            obs, rew, done = out
            if (obs.shape != (T, E, A, D) or rew.shape != (T, E, A) or done.shape != (T, E, A) or obs.dtype != torch.float32
                    or rew.dtype != torch.float32 or done.dtype != torch.uint8
                    or not (obs.is_contiguous() and rew.is_contiguous() and done.is_contiguous())
                    or any(t.device != self.device for t in out)):
                raise ValueError("out must be contiguous (f32 [T,E,A,D], f32 [T,E,A], uint8 [T,E,A]) on the env's device")
        else:
            obs, rew, done = self._obs, self._rew, self._done
        u_ptr = None
        if u is not None:
            self._u = torch.as_tensor(u).to(device=self.device, dtype=torch.float64).contiguous()
            if self._u.shape != (T, E, A):
                raise ValueError(f"u must have shape ({T}, {E}, {A})")
            u_ptr = self._u.data_ptr()
        if env_done_out is not None and (env_done_out.shape != (T, E) or env_done_out.dtype != torch.uint8
                                         or not env_done_out.is_contiguous() or env_done_out.device != self.device):
            raise ValueError(f"env_done_out must be a contiguous uint8 [{T}, {E}] tensor on the env's device")
        flags = self._base_flags
        fn = self._lib.bsx_step_many_continuous if self.continuous_actions else self._lib.bsx_step_many_discrete
        with self._guard():
            _lib.check(fn(self._state.data_ptr(), E, self.n_agents, T, actions.data_ptr(), kind, u_ptr, obs.data_ptr(),
                          rew.data_ptr(), done.data_ptr(), self._p_env_done, self._p_winner,
                          env_done_out.data_ptr() if env_done_out is not None else None,
                          ctypes.byref(self._cfg), flags, 1 if store else 0, self.seed, self.env_offset, self._stream()),
                       "bsx_step_many")
        if self._mirror:
            self._sync_mirror()
        return obs, rew, done.view(torch.bool)

    def _launch_rollout(self, T, weights_ptr, precision, scripted_team, obs_ptr, scores_ptr, rew_ptr, done_ptr, noise, actor_seed, seq,
                        seq_base_ptr, env_done_t_ptr=None, scripted_seed=0):
        """Enqueue bsx_rollout_discrete / bsx_rollout_continuous: T ticks of (actor -> step) in one launch (rollout.PolicyRollout)."""
        if self.n_agents > 4:
            raise ValueError("the one-launch rollout is built for 1v1 ... 4v4")
        flags = self._base_flags
        common = (self._p_env_done, self._p_winner, env_done_t_ptr, ctypes.byref(self._cfg), flags,
                  ctypes.byref(noise) if noise is not None else None, int(actor_seed), int(seq), seq_base_ptr, self.seed,
                  self.env_offset, self._stream())
        with self._guard():
            if self.continuous_actions:
                _lib.check(self._lib.bsx_rollout_continuous(self._state.data_ptr(), self.n_envs, self.n_agents, int(T), weights_ptr,
                                                            int(precision), int(scripted_team), int(scripted_seed) & 0xFFFFFFFFFFFFFFFF,
                                                            obs_ptr, scores_ptr, rew_ptr, done_ptr, *common),
                           "bsx_rollout_continuous")
            else:
                _lib.check(self._lib.bsx_rollout_discrete(self._state.data_ptr(), self.n_envs, self.n_agents, int(T), weights_ptr,
                                                          int(precision), int(scripted_team), obs_ptr, scores_ptr, rew_ptr, done_ptr,
                                                          *common), "bsx_rollout_discrete")

    def chain_ranges(self, chains):
        """The batch as `chains` contiguous game ranges [(first, count), ...] in whole 256-game blocks (sharding.chain_ranges)."""
        from ..sharding import chain_ranges
        return chain_ranges(self.n_envs, self.n_agents, chains)

    def capture_steps(self, actions, store=False, chains="auto"):
        """Capture T consecutive step() launches into ONE HIP graph (the launch-bound inner loop of a rollout).

        actions: static device tensor [T, E, A] int32 (discrete), [T, E, A, 4] float32 score vectors, or
                 [T, E, A, 3] float32/float64 (continuous); step t of every replay reads actions[t] -- refill the
                 tensor in place between replays.
        store:   False = every step overwrites the env-owned obs/rew/done tensors (as step_batch does);
                 True  = step t writes slice t of new [T, E, A, ...] tensors (a rollout buffer in HBM).
        chains:  > 1 = the batch as that many contiguous game ranges, each a chain of T launches on its own branch of the graph (forked
                 from and joined into the capture stream).  The reference's games share nothing (battle_env.py:281-381), so the results
                 are those of chains=1 bit for bit; what changes is that a range's step t+1 waits for ITS step t only, and one chain's
                 kernel boundary (launch, first loads, store drain) runs under the other chains' arithmetic.  Measured with replays
                 queued back to back: two chains take 13 ... 26 % off a step for 2v2, 3v3, 6v6 ... 16v16, three chains 17 % at 4v4
                 (20.3 -> 16.8 us at 65 536 games); at 1v1 nothing at 65 536 games, -5 ... -14 % from 196 608 games up
                 (profiles/r06_chains_1v1.json).  A multi-branch graph is launched node by node at ~6 us each, though, and a replay
                 that is synchronised before the next one pays for that: there 4v4 is a tie, 2v2 and 1v1 below 1 M games LOSE.
                 "auto" (the default) takes chains only where neither use loses (sharding.chain_ranges: 4v4, 3v3, 5v5 and larger,
                 1v1 from 1 M games); chains=2 / 3 is for the caller who queues replays; chains=1 is ONE launch per step over
                 the whole batch.
        Returns (graph, outputs): graph.replay() runs the T steps; outputs = (obs, rew, done) tensors.
        Needs rng='philox' (no host draws inside a graph)."""
        if self.rng != "philox" or self._compat:
            raise ValueError("capture_steps needs a batched env with rng='philox'")
        T, kind = self._check_action_series(actions)
        ranges = self.chain_ranges(chains)
        E, A, D = self.n_envs, self._A, self.obs_size
        if store:
            obs = torch.empty((T, E, A, D), dtype=torch.float32, device=self.device)
            rew = torch.empty((T, E, A), dtype=torch.float32, device=self.device)
            done = torch.empty((T, E, A), dtype=torch.uint8, device=self.device)
        else:
            obs, rew, done = self._obs, self._rew, self._done
        step_bytes = actions[0].numel() * actions.element_size()
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(self.device)
        side = [torch.cuda.Stream(self.device) for _ in ranges[1:]]
        with torch.cuda.graph(graph):
            main = torch.cuda.current_stream(self.device)
            for s in side:
                s.wait_stream(main)
            for r, games in enumerate(ranges):
                with torch.cuda.stream(main if r == 0 else side[r - 1]):
                    for t in range(T):
                        o, w, d = (obs[t], rew[t], done[t]) if store else (obs, rew, done)
                        self._launch(actions.data_ptr() + t * step_bytes, kind, False, None, o.data_ptr(), w.data_ptr(), d.data_ptr(),
                                     games=games if len(ranges) > 1 else None)
            for s in side:
                main.wait_stream(s)
        self._graph_keepalive = (actions, obs, rew, done, side)
        return graph, (obs, rew, done.view(torch.bool))

    def _sync_mirror(self):
        st = self.export_state(("palive", "tick", "env_done"))
        self._h_alive = st["palive"].cpu().numpy().astype(bool)
        self._h_tick = st["tick"].cpu().numpy().astype(np.int64)
        self._h_done = st["env_done"].cpu().numpy().astype(bool)

    def step(self, actions, u=None, copy=False):
        """The reference's 4-tuple: (observations, rewards, dones, infos), each a dict keyed by agent id.  Batched mode:
        the dict values are views of the env-owned tensors (overwritten by the next call) unless copy=True."""
        if self._compat:
            return self._step_compat(actions, u)
        obs, rew, done = self.step_batch(actions, u, copy=copy)
        if not copy:
            done = self._done_bool                                       # the one persistent bool view (a fresh .view() each call would defeat the cache)
        return (self._agent_views(obs), self._agent_views(rew), self._agent_views(done),
                {a: {} for a in self.possible_agents})

    def _init_compat_io(self):
        """Drop-in mode exchanges a few dozen bytes with the device per call, so nothing is copied: all outputs (obs, rew, done,
        env_done, winner) are views of ONE pinned host buffer and all inputs (the shots' random() values, the actions) of another;
        hipHostMalloc memory is mapped into the device's address space at the same address, the kernels read and write it
        directly over the link -- one launch and one synchronisation per step(), no upload, no download."""
        A, D, dev = self._A, self.obs_size, self.device
        def carve(spec):
            off, o = {}, 0
            for name, nbytes in spec:
                o = (o + 15) // 16 * 16
                off[name] = (o, nbytes)
                o += nbytes
            return off, (o + 15) // 16 * 16
        act_bytes = A * 3 * 8 if self.continuous_actions else A * 4
        out_off, out_n = carve((("obs", A * D * 4), ("rew", A * 4), ("done", A), ("env_done", 1), ("winner", 1)))
        in_off, in_n = carve((("u", A * 8), ("act", act_bytes)))
        self._out_host = torch.zeros(out_n, dtype=torch.uint8).pin_memory()
        self._in_host = torch.zeros(in_n, dtype=torch.uint8).pin_memory()

        def view(buf, off, dtype, shape):
            o, nb = off
            return buf[o:o + nb].view(dtype).view(shape)
        self._obs = view(self._out_host, out_off["obs"], torch.float32, (1, A, D))
        self._rew = view(self._out_host, out_off["rew"], torch.float32, (1, A))
        self._done = view(self._out_host, out_off["done"], torch.uint8, (1, A))
        self._env_done = view(self._out_host, out_off["env_done"], torch.uint8, (1,))
        self._winner = view(self._out_host, out_off["winner"], torch.uint8, (1,))
        hn = self._out_host.numpy()
        self._ho_obs = hn[out_off["obs"][0]:][:A * D * 4].view(np.float32).reshape(A, D)
        self._ho_rew = hn[out_off["rew"][0]:][:A * 4].view(np.float32)
        self._ho_done = hn[out_off["done"][0]:][:A]
        self._ho_flags = hn[out_off["env_done"][0]:]                    # [0] env_done ... winner at its own offset
        self._ho_winner_at = out_off["winner"][0]
        hi = self._in_host.numpy()
        self._hi_u = hi[in_off["u"][0]:][:A * 8].view(np.float64)
        if self.continuous_actions:
            self._hi_act = hi[in_off["act"][0]:][:act_bytes].view(np.float64).reshape(A, 3)
        else:
            self._hi_act = hi[in_off["act"][0]:][:act_bytes].view(np.int32)
        # The kernels get the DEVICE address of the two staging buffers (hipHostGetDevicePointer): with hipHostMalloc'ed memory it equals
        # the host address, with registered (hipHostRegister) memory it need not -- never assumed.
        def device_address(buf):
            dp = ctypes.c_void_p()
            _lib.check(self._lib.bsx_host_device_pointer(buf.data_ptr(), ctypes.byref(dp)), "bsx_host_device_pointer (is the staging buffer pinned?)")
            return dp.value
        d_in, d_out = device_address(self._in_host), device_address(self._out_host)
        self._p_in_u = d_in + in_off["u"][0]
        self._p_in_act = d_in + in_off["act"][0]
        self._compat_dev_ptrs = {k: d_out + out_off[k][0] for k in ("obs", "rew", "done", "env_done", "winner")}

    def _step_compat(self, actions, u):
        ids = self.possible_agents
        if not isinstance(actions, dict):
            raise TypeError("drop-in mode takes the reference's action dict {agent id: action} (battle_env.py:281)")
        if self.continuous_actions:                                     # battle_env.py:295-297 clips the caller's dict
            for k, v in actions.items():
                actions[k] = np.clip(v, -1.0, 1.0)
        A = self._A
        was_done = bool(self._h_done[0])
        alive_before = self._h_alive[0].copy()
        empty = len(actions) == 0
        any_alive = bool(alive_before.any())
        physics = not was_done and not empty and any_alive and self._h_tick[0] + 1 < self.tie_tick
        if physics:
            for i, a in enumerate(ids):
                if alive_before[i] and a not in actions:
                    raise KeyError(a)                                   # battle_env.py:326
        # ---- inputs into the pinned staging buffer: actions (ints; 4-vectors arg-maxed, battle_env.py:327-328; 3-vectors), then the
        #      random() value of every shot, drawn in the reference's order: live agents in id order, only when physics runs
        hact, hu = self._hi_act, self._hi_u
        hu[:] = np.nan
        if not empty:
            for i, a in enumerate(ids):
                v = actions.get(a)
                if self.continuous_actions:
                    hact[i] = 0.0 if v is None else np.asarray(v, np.float64).reshape(3)
                elif v is None:
                    hact[i] = -1                                         # the reference only reads actions of live agents; absent = "no movement"
                elif type(v) is int:
                    hact[i] = v if -2147483648 <= v <= 2147483647 else -1   # any integer outside 0..3 is "no movement" (battle_env.py:399-417)
                else:
                    v = np.asarray(v)
                    hact[i] = int(np.argmax(v)) if (v.ndim >= 1 and v.size > 1) else int(v.reshape(-1)[0])
            if u is not None:
                hu[:] = np.asarray(u, np.float64).reshape(-1)
            elif self.rng == "python" and physics:
                shoot = (np.clip(hact[:, 2], -1.0, 1.0) > 0) if self.continuous_actions else (hact == 1)
                for i in range(A):
                    if alive_before[i] and shoot[i]:
                        hu[i] = _stdlib_random.random()
        use_u = u is not None or self.rng == "python"
        self._launch(None if empty else self._p_in_act, _lib.ACT_F64 if self.continuous_actions else _lib.ACT_I32, empty,
                     self._p_in_u if use_u else None, self._p_obs, self._p_rew, self._p_done)
        self._sync()                                                    # the kernel wrote the pinned output buffer itself
        o, r = self._ho_obs, self._ho_rew
        d = self._ho_done.astype(bool)
        ed = bool(self._ho_flags[0])
        # ---- host mirrors (what `agents`, `env_done` and the next call's draws read)
        if not was_done:
            if not empty and any_alive:
                self._h_tick[0] += 1                                    # physics or the time-limit tie: the clock advanced (battle_env.py:316)
            self._h_done[0] = ed
            if ed:
                self._sync_mirror()                                     # game over (once per game): exact alive flags from the state
                self.dones = {a: True for a in ids}                     # win()/tie() rebind the dict (:478,:494)
                self._winner_name = _lib.WINNER_NAMES[int(self._out_host[self._ho_winner_at])]
            else:
                self._h_alive[0] = alive_before & ~d                    # a plane reported done in a running game has died
                for i, a in enumerate(ids):
                    if d[i]:
                        self.dones[a] = True
        # the reference starts every agent at int 0 and adds the constants of the events that occur (battle_env.py:299,338-362): an agent
        # without an event keeps the int 0 whatever the constants' types; sums come from the kernel's float64 accumulator through float32
        rewards = {a: (int(round(float(r[i]))) if (self._int_rewards or r[i] == 0.0) else float(r[i])) for i, a in enumerate(ids)}
        return ({a: o[i].copy() for i, a in enumerate(ids)}, rewards, self.dones, {a: {} for a in ids})

    # ------------------------------------------------------------------ observe (battle_env.py:202-244)
    def observe(self, agent):
        with self._guard():
            _lib.check(self._lib.bsx_observe(self._state.data_ptr(), self.n_envs, self.n_agents, self._p_obs,
                                             self._stream()), "bsx_observe")
        i = self._idx[agent]
        if self._compat:
            self._sync()
            return self._ho_obs[i].copy()
        return self._obs[:, i]

    # ------------------------------------------------------------------ public attributes callers read
    @property
    def agents(self):
        """Drop-in mode: ids of the planes still alive, in id order (battle_env.py:109,274,357).  Batched: all ids."""
        if self._compat:
            return [a for i, a in enumerate(self.possible_agents) if self._h_alive[0, i]]
        return self.possible_agents[:]

    @property
    def env_done(self):
        return bool(self._h_done[0]) if self._compat else self._env_done.view(torch.bool)

    @property
    def winner(self):
        """Drop-in: 'none' | 'red' | 'blue' | 'tie'.  Batched: uint8 codes 0..3 in that order."""
        return self._winner_name if self._compat else self._winner

    @property
    def total_time(self):
        """Game clock in hours (battle_env.py:175,316): the reference adds 0.1 per tick in binary64; reproduced by the same
        accumulation on the host from the tick counter.  Drop-in: float; batched: float64 numpy array [E]."""
        ticks = self.export_state(("tick",))["tick"].cpu().numpy()
        acc = np.zeros(int(ticks.max()) + 1)
        t = 0
        for k in range(1, len(acc)):
            t += self.time_step
            acc[k] = t
        out = acc[ticks]
        return float(out[0]) if self._compat else out

    def counters(self):
        """int32 [E, 4] host array: games finished, ties, red wins, blue wins (battle_env.py:169-170,102-103)."""
        return self.export_state(("counters",))["counters"].cpu().numpy().reshape(self.n_envs, 4)

    @property
    def total_games(self):
        c = self.counters()[:, 0]
        return int(c[0]) if self._compat else c

    @property
    def ties(self):
        c = self.counters()[:, 1]
        return int(c[0]) if self._compat else c

    def wins(self):
        """battle_env.py:449-455 (summed over games when batched)."""
        c = self.counters().sum(0)
        return "Wins by red: {}\nWins by blue: {}\nTied games: {}\nWin rate: {}".format(c[2], c[3], c[1], c[2] / c[0])

    def make_discrete(self, actions_dict):
        return {a: np.argmax(v) for a, v in actions_dict.items()}        # battle_env.py:463-467

    # ------------------------------------------------------------------ state export / checkpoint
    _EXPORT_SPEC = {  # name -> (dtype, per-env shape builder)
        "px": (torch.int32, lambda A: (A,)), "py": (torch.int32, lambda A: (A,)), "pdir": (torch.float64, lambda A: (A,)),
        "php": (torch.int32, lambda A: (A,)), "palive": (torch.uint8, lambda A: (A,)),
        "base_xy": (torch.int32, lambda A: (4,)), "bhp": (torch.int32, lambda A: (2,)), "tick": (torch.int32, lambda A: ()),
        "env_done": (torch.uint8, lambda A: ()), "winner": (torch.uint8, lambda A: ()),
        "bl_live": (torch.uint8, lambda A: (A, 12)), "bl_x": (torch.int32, lambda A: (A, 12)),
        "bl_y": (torch.int32, lambda A: (A, 12)), "bl_dir": (torch.float64, lambda A: (A, 12)),
        "counters": (torch.int32, lambda A: (4,)),
    }

    def export_state(self, fields=None):
        """Unpacked copy of the game state as device tensors (bsx_export_state): planes, bases, bullets, flags, counters."""
        fields = tuple(fields) if fields is not None else _lib.EXPORT_FIELDS
        out, ex = {}, _lib.BsxExport()
        for f in fields:
            dt, shp = self._EXPORT_SPEC[f]
            out[f] = torch.empty((self.n_envs, *shp(self._A)), dtype=dt, device=self.device)
            setattr(ex, f, out[f].data_ptr())
        with self._guard():
            _lib.check(self._lib.bsx_export_state(self._state.data_ptr(), self.n_envs, self.n_agents, ctypes.byref(ex),
                                                  self._stream()), "bsx_export_state")
        return out

    def state_dict(self):
        """Snapshot of the raw device state (the reference never checkpoints env state; here it is one tensor) plus what
        identifies the job it belongs to."""
        return {"state": self._state.clone(), "env_done": self._env_done.clone(), "winner": self._winner.clone(),
                "done": self._done.clone(), "reset_nonce": self._reset_nonce,
                "meta": {"abi": _lib.ABI_VERSION, "n_envs": self.n_envs, "n_agents": self.n_agents, "seed": self.seed,
                         "env_offset": self.env_offset, "continuous_actions": self.continuous_actions}}

    def load_state_dict(self, sd):
        """Restore a snapshot taken by `state_dict()` of an env of the same shape.  Host mirrors (drop-in / rng='python' mode:
        alive flags, ticks, `dones`, `winner`) are rebuilt from the restored device state."""
        meta = sd.get("meta")
        if meta is not None:
            mine = {"abi": _lib.ABI_VERSION, "n_envs": self.n_envs, "n_agents": self.n_agents,
                    "continuous_actions": self.continuous_actions}
            bad = {k: (meta.get(k), v) for k, v in mine.items() if meta.get(k) != v}
            if bad:
                raise ValueError(f"state_dict does not fit this env (saved, expected): {bad}")
        if sd["state"].shape != self._state.shape:
            raise ValueError(f"state block of {sd['state'].numel()} bytes does not fit this env ({self._state.numel()} bytes)")
        self._state.copy_(sd["state"]); self._env_done.copy_(sd["env_done"]); self._winner.copy_(sd["winner"])
        self._done.copy_(sd["done"]); self._reset_nonce = int(sd["reset_nonce"])
        if self._mirror:
            self._sync_mirror()
        if self._compat:
            # `dones` as step() would have left it: all True once the game is over (win()/tie() rebind the dict, :478,:494),
            # else True for the planes that have died; `winner` from the restored code
            d = self._done[0].cpu().numpy().astype(bool)
            over = bool(self._h_done[0])
            self.dones = {a: bool(over or d[i]) for i, a in enumerate(self.possible_agents)}
            self._winner_name = _lib.WINNER_NAMES[int(self._winner[0])]

    # ------------------------------------------------------------------ rendering: out of scope
    def render(self, mode="human"):
        if self.show:
            warnings.warn("rendering is out of scope for the MI355X step() path; export_state() gives the poses", stacklevel=2)
            self.show = False

    def start_recording(self, path):
        warnings.warn("video recording is out of scope for the MI355X step() path", stacklevel=2)

    def export_video(self):
        pass

    def close(self):
        """battle_env.py:457.  Tells the library that the state block is going away (bsx_state_release: the host-side record of the
        block's action family is dropped, so a later allocation at the same address starts unclaimed)."""
        st, self._released = getattr(self, "_state", None), True
        if st is not None and getattr(self, "_lib", None) is not None:
            self._lib.bsx_state_release(st.data_ptr())

    def __del__(self):
        try:
            if not getattr(self, "_released", False):
                self.close()
        except Exception:                                   # noqa: BLE001 -- interpreter shutdown: the library may be gone already
            pass
