"""`instinct.Team`: the reference's scripted opponent (instinct/team.py:3-15, instinct/agent.py:10-62) evaluated for
every game at once by one elementwise HIP kernel (`bsx_instinct_*`): decode the observation row, pick the target with
the smallest dist * |angle| (base first on ties), shoot when within half the shot distance and 20 degrees, else turn
toward it.  Same constructor and `choose_actions(observations)` as the reference; it reads the env's observation
tensor directly, so nothing crosses the host.

    blue = instinct.Team(env.possible_blue, env.possible_red, env)
    actions = blue.choose_actions(observations)          # {agent: tensor[E]} (discrete) / tensor[E, 3] (continuous)
    blue.write_actions(out=action_tensor)                # batched: fills only this team's columns of [E, A(, 4|3)]
"""
import numpy as np
import torch

from .. import _lib


class Team:
    def __init__(self, agent_list, enemy_list, env, seed=0):
        self.agent_list, self.enemy_list, self.env = list(agent_list), list(enemy_list), env
        self._lib = _lib.load()
        if self.agent_list == env.possible_red and self.enemy_list == env.possible_blue:
            self.team = 0
        elif self.agent_list == env.possible_blue and self.enemy_list == env.possible_red:
            self.team = 1
        else:
            raise ValueError("agent_list / enemy_list must be the env's red and blue id lists (instinct/team.py:4-8)")
        self.agents = {a: self for a in self.agent_list}       # reference attribute: {agent id: InstinctAgent}
        self.seed, self.seq = int(seed), 0
        self._cols = [env._idx[a] for a in self.agent_list]
        self._buf = None

    def write_actions(self, out=None, obs=None, rnd=None, seq_base=None, seq=None):
        """Fill this team's rows of `out` from the observation tensor (default: the env's own, i.e. what the last
        reset()/step() produced).  out: int32 [E, A] or float32 [E, A, 4] (one-hot +-1, for the score-vector step path)
        when discrete; float64 [E, A, 3] when continuous.  Returns `out`."""
        env = self.env
        obs = env._obs if obs is None else obs
        if not obs.is_cuda and not obs.is_pinned():            # (drop-in mode keeps the env's rows in pinned host memory, which the kernel reads directly)
            obs = obs.to(env.device)
        E, A = env.n_envs, env._A
        stream = torch.cuda.current_stream(env.device).cuda_stream
        obs_ptr = obs.data_ptr()
        if not obs.is_cuda:                                    # pinned host rows: the kernel takes their DEVICE address (never assumed equal)
            import ctypes
            dp = ctypes.c_void_p()
            _lib.check(self._lib.bsx_host_device_pointer(obs_ptr, ctypes.byref(dp)), "bsx_host_device_pointer")
            obs_ptr = dp.value
        if out is None:
            if self._buf is None:
                self._buf = (torch.zeros((E, A, 3), dtype=torch.float64, device=env.device) if env.continuous_actions
                             else torch.zeros((E, A), dtype=torch.int32, device=env.device))
            out = self._buf
        if env.continuous_actions:
            if out.dtype != torch.float64 or tuple(out.shape) != (E, A, 3) or not out.is_contiguous():
                raise ValueError("continuous instinct actions need a contiguous float64 [E, A, 3] tensor")
            rnd_t = None
            if rnd is not None:
                rnd_t = torch.as_tensor(rnd, dtype=torch.float64, device=env.device).contiguous()
            if seq is None:                                    # (a caller that captures the call into a graph passes its own number + a device word)
                self.seq += 1
                seq = self.seq
            _lib.check(self._lib.bsx_instinct_continuous(obs_ptr, out.data_ptr(),
                                                         rnd_t.data_ptr() if rnd_t is not None else None, E, env.n_agents,
                                                         self.team, self.seed, int(seq),
                                                         seq_base.data_ptr() if seq_base is not None else None, stream),
                       "bsx_instinct_continuous")
            return out
        if out.dtype == torch.int32 and tuple(out.shape) == (E, A):
            kind = _lib.ACT_I32
        elif out.dtype == torch.float32 and tuple(out.shape) == (E, A, 4):
            kind = _lib.ACT_LOGITS_F32
        else:
            raise ValueError("discrete instinct actions need an int32 [E, A] or float32 [E, A, 4] tensor")
        if not out.is_contiguous():
            raise ValueError("out must be contiguous")
        _lib.check(self._lib.bsx_instinct_discrete(obs_ptr, out.data_ptr(), kind, E, env.n_agents, self.team, stream),
                   "bsx_instinct_discrete")
        return out

    def choose_actions(self, observations=None):
        """instinct/team.py:10-15.  `observations` is accepted for signature parity; the env's observation tensor is
        what is read (in drop-in mode, dict values are uploaded first)."""
        env = self.env
        obs = None
        if env._compat and isinstance(observations, dict):
            o = env._obs.to(env.device)                       # drop-in mode: the env's rows live in pinned host memory
            for a in self.agent_list:
                if a in observations:
                    o[0, env._idx[a]] = torch.as_tensor(np.asarray(observations[a], np.float32), device=env.device)
            obs = o
        out = self.write_actions(obs=obs)
        if env._compat:
            h = out[0].cpu().numpy()
            return {a: (h[i].copy() if env.continuous_actions else int(h[i])) for a, i in zip(self.agent_list, self._cols)}
        return {a: out[:, i] for a, i in zip(self.agent_list, self._cols)}
