"""Device-resident transition store: the step after the path.

Mirror of the reference's team replay buffer (maddpg/buffer.py:3-70): same constructor arguments, `store_transition`,
`sample`, `is_ready`, same tuple layout out of `sample()` -- but every array is a torch tensor on the env's device, one
call stores the transitions of ALL games at once (each game contributes one row), and a whole `PolicyRollout` (T ticks)
can be appended without a host copy.  The reference keeps float64 numpy arrays on the host and converts a sampled batch
to float32 tensors for the learner (maddpg/team.py:33-37); here rows are float32 from the start.

Ring semantics are the reference's: row index = counter % mem_size, oldest rows overwritten; `sample()` draws
`batch_size` indices uniformly WITH replacement among the filled rows (np.random.choice, buffer.py:51)."""
import torch


class ReplayBuffer:
    def __init__(self, mem_size, batch_size, agent_list, obs_size, critic_dims, n_actions, device="cuda", generator=None):
        self.mem_size, self.batch_size = int(mem_size), int(batch_size)
        self.agent_list, self.n_agents = list(agent_list), len(agent_list)
        if critic_dims != obs_size * self.n_agents:
            raise ValueError("critic_dims must be obs_size * len(agent_list) (main.py:118)")
        self.mem_cntr = 0
        kw = dict(device=device, dtype=torch.float32)
        nA, M = self.n_agents, self.mem_size
        self.actor_states = torch.zeros((M, nA, obs_size), **kw)        # state_mem is the [M, nA*obs] view of this
        self.actor_new_states = torch.zeros((M, nA, obs_size), **kw)
        self.n_actions = int(n_actions)
        self.action_mem = torch.zeros((M, nA, n_actions), **kw)
        self.rew_mem = torch.zeros((M, nA), **kw)
        self.done_mem = torch.zeros((M, nA), device=device, dtype=torch.bool)
        self.generator = generator

    @property
    def state_mem(self):
        return self.actor_states.view(self.mem_size, -1)

    @property
    def new_state_mem(self):
        return self.actor_new_states.view(self.mem_size, -1)

    def _put(self, obs, act, rew, obs_, done):
        """rows: obs [R, nA, obs], act [R, nA, n_actions], rew [R, nA], obs_ [R, nA, obs], done [R, nA]."""
        R = obs.shape[0]
        if R > self.mem_size:                                            # only the newest mem_size rows can survive
            obs, act, rew, obs_, done = (t[-self.mem_size:] for t in (obs, act, rew, obs_, done))
            self.mem_cntr += R - self.mem_size
            R = self.mem_size
        start = self.mem_cntr % self.mem_size
        first = min(R, self.mem_size - start)
        for dst, src in ((self.actor_states, obs), (self.action_mem, act), (self.rew_mem, rew),
                         (self.actor_new_states, obs_), (self.done_mem, done)):
            dst[start:start + first].copy_(src[:first])
            if first < R:
                dst[:R - first].copy_(src[first:])
        self.mem_cntr += R

    def store_transition(self, states, actions, rewards, states_, dones):
        """buffer.py:25-47, batched: every dict value carries a leading game axis E (a plain per-game value, as the
        reference passes, is one row).  Discrete actions given as indices are stored one-hot."""
        def col(d, a, width=None):
            v = torch.as_tensor(d[a], device=self.actor_states.device)
            return v.reshape(-1, width) if width else v.reshape(-1)
        obs = torch.stack([col(states, a, self.actor_states.shape[2]) for a in self.agent_list], 1).float()
        obs_ = torch.stack([col(states_, a, self.actor_states.shape[2]) for a in self.agent_list], 1).float()
        acts = []
        for a in self.agent_list:
            v = torch.as_tensor(actions[a], device=obs.device)
            if not v.is_floating_point():
                v = torch.nn.functional.one_hot(v.reshape(-1).long(), self.action_mem.shape[2])
            acts.append(v.reshape(-1, self.action_mem.shape[2]).float())
        self._put(obs, torch.stack(acts, 1), torch.stack([col(rewards, a) for a in self.agent_list], 1).float(),
                  obs_, torch.stack([col(dones, a) for a in self.agent_list], 1).bool())

    def store_rollout(self, rollout, columns):
        """Append the transitions of a PolicyRollout for the agents in `columns` (env column indices of this team, e.g.
        range(n) for red), tick-major, without a host copy of the data.  Only rows of RUNNING games are transitions
        (`rollout.valid`): a tick on a finished game is the reference's inert call (battle_env.py:303-306) or, with
        auto_reset, the re-spawn -- its row would pair the old game's last observation with the new game's first under
        done=False, and the reference's loop (`while not env.env_done`, main.py:177-181) never stores such a row.
        Returns the number of rows stored (one host sync: the count decides where the ring wraps)."""
        c = torch.as_tensor(list(columns), device=rollout.obs.device)
        T = rollout.T
        keep = rollout.valid.reshape(-1).nonzero().squeeze(1)                      # [R] indices into the T*E rows, tick-major
        flat = lambda x: x.index_select(2, c).reshape(-1, len(c), *x.shape[3:]).index_select(0, keep)   # noqa: E731
        self._put(flat(rollout.obs[:T]), flat(rollout.scores)[..., :self.n_actions], flat(rollout.rew), flat(rollout.obs[1:T + 1]), flat(rollout.done))
        return int(keep.numel())

    def sample(self, idx=None):
        """buffer.py:49-67 -> (actor_states [nA, B, obs], states [B, nA*obs], actions [nA, B, n_actions], rewards [B, nA],
        actor_new_states [nA, B, obs], states_ [B, nA*obs], dones [B, nA]), all on the device.
        idx: optional row indices to return instead of drawing them (parity runs with the reference's np.random.choice draw)."""
        max_mem = min(self.mem_cntr, self.mem_size)
        if idx is None:
            idx = torch.randint(0, max_mem, (self.batch_size,), device=self.actor_states.device, generator=self.generator)
        else:
            idx = torch.as_tensor(idx, device=self.actor_states.device).long()
            if idx.shape != (self.batch_size,) or int(idx.max()) >= max_mem or int(idx.min()) < 0:
                raise ValueError("idx must be batch_size indices into the filled rows")
        o, o_, a = self.actor_states[idx], self.actor_new_states[idx], self.action_mem[idx]
        B = self.batch_size
        return (o.transpose(0, 1), o.reshape(B, -1), a.transpose(0, 1), self.rew_mem[idx],
                o_.transpose(0, 1), o_.reshape(B, -1), self.done_mem[idx])

    def is_ready(self):
        return self.mem_cntr >= self.batch_size
