"""On-device policy rollout feeding the fused step() (BASELINE.json configs[4]).

The reference's callers run, per tick and per agent, `actor(obs) + noise -> clamp -> numpy -> env.step -> argmax`
(maddpg/agent.py:25-33, main.py:179-181, envs/battle_env.py:327-328), crossing the host for every agent.  Here the
whole loop `obs -> actor -> score vectors -> step() -> obs'` stays on the MI355X: the actors of all A agents are one
stacked module evaluated with batched GEMMs (hipBLASLt through torch -- plain library GEMMs, 5..14 -> 64 -> 64 -> 4),
the score vectors go straight into `bsx_step_discrete`'s arg-max path, T ticks are captured into one HIP graph and
the transition buffers (`[T, E, A, ...]`) live in HBM for a learner to consume.  No host round trip inside a rollout.

`StackedActor` reproduces the reference ActorNetwork (maddpg/networks.py:54-85): Linear(obs,64) -> LayerNorm -> ReLU ->
Linear(64,64) -> LayerNorm -> ReLU -> Linear(64,n_actions) -> tanh, same initialisation; one set of weights per
agent (MADDPG keeps one actor per plane), loadable from the reference's per-agent `state_dict` checkpoints.
"""
import math

import torch
from torch import nn

from . import _lib


class StackedActor(nn.Module):
    """A independent actors evaluated together: parameters carry a leading agent axis."""

    def __init__(self, n_actors, obs_len, n_actions, fc1_dims=64, fc2_dims=64, device=None, dtype=torch.float32):
        super().__init__()
        A, kw = n_actors, dict(device=device, dtype=dtype)
        self.n_actors, self.obs_len, self.n_actions = A, obs_len, n_actions
        f1, f2, f3 = 1.0 / math.sqrt(fc1_dims), 1.0 / math.sqrt(fc2_dims), 0.003     # networks.py:59,65,71

        def uni(shape, f):
            return nn.Parameter(torch.empty(shape, **kw).uniform_(-f, f))
        self.w1, self.b1 = uni((A, obs_len, fc1_dims), f1), uni((A, 1, fc1_dims), f1)
        self.g1, self.h1 = nn.Parameter(torch.ones((A, 1, fc1_dims), **kw)), nn.Parameter(torch.zeros((A, 1, fc1_dims), **kw))
        self.w2, self.b2 = uni((A, fc1_dims, fc2_dims), f2), uni((A, 1, fc2_dims), f2)
        self.g2, self.h2 = nn.Parameter(torch.ones((A, 1, fc2_dims), **kw)), nn.Parameter(torch.zeros((A, 1, fc2_dims), **kw))
        self.w3, self.b3 = uni((A, fc2_dims, n_actions), f3), uni((A, 1, n_actions), f3)

    @staticmethod
    def _ln(x, g, h, eps=1e-5):
        return torch.nn.functional.layer_norm(x, x.shape[-1:], None, None, eps) * g + h

    def forward(self, obs, squash=True):
        """obs [E, A, D] -> scores [E, A, n_actions] (tanh-squashed, as the reference's actor output; squash=False: the raw head,
        what a value head returns)."""
        x = obs.transpose(0, 1)                                     # [A, E, D]
        x = torch.relu(self._ln(torch.baddbmm(self.b1, x, self.w1), self.g1, self.h1))
        x = torch.relu(self._ln(torch.baddbmm(self.b2, x, self.w2), self.g2, self.h2))
        x = torch.baddbmm(self.b3, x, self.w3)
        if squash:
            x = torch.tanh(x)
        return x.transpose(0, 1)

    @torch.no_grad()
    def pack(self, out=None):
        """Weights in the layout `bsx_actor_forward` reads (csrc/bsx_actor.hip): everything is laid out for the MFMA
        fragments of the transposed product H^T = W^T X^T -- per agent
          W1A[mo 2][s Dp/2][lane 64]            = W1[2s + (lane>>5)][32 mo + (lane&31)]   (0 beyond obs_len; Dp = obs_len rounded up to even)
          W2A[mo 2][mt 2][vq 4][lane 64][t 4]   = W2[nid(mt, 4 vq + t, lane>>5)][32 mo + (lane&31)]
          b1 g1 be1 b2 g2 be2, each [hh 2][mo 2][v 16] = vec[nid(mo, v, hh)]
          W3P[hh 2][mt 2][v 16][4]              = W3[nid(mt, v, hh)][:]        b3[4]
          W2B[mo 2][s 4][term 3][lane 64][i 8]  bfloat16: term 0 = bf16(W2), term 1 = bf16(W2 - term 0), term 2 = bf16(the rest) of
                                                W2[nid(s>>1, 8 (s&1) + i, lane>>5)][32 mo + (lane&31)]  (precision="bf16x3" / "bf16x6")
        with nid(m, v, hh) = 32 m + (v&3) + 8 (v>>2) + 4 hh, the neuron that accumulator register v of 32-neuron tile m
        holds in lane half hh.  float32, contiguous, [n_actors, floats]."""
        if self.w1.shape[2] != 64 or self.w2.shape[2] != 64 or self.n_actions not in (1, 3, 4):
            raise ValueError("the fused actor kernel is built for fc1 = fc2 = 64 (main.py:15-16) and 4 action scores, 3 continuous "
                             "actions or 1 value (padded to 4 head columns: the extra outputs are 0)")
        A, D, dev = self.n_actors, self.obs_len, self.w1.device
        Dp = (D + 1) & ~1
        lane = torch.arange(64, device=dev)
        hh_l, c_l = lane >> 5, lane & 31
        v16 = torch.arange(16, device=dev)
        nid = lambda m, v, hh: 32 * m + (v & 3) + 8 * (v >> 2) + 4 * hh          # noqa: E731
        w1 = torch.zeros((A, Dp, 64), dtype=torch.float32, device=dev)
        w1[:, :D] = self.w1.float()
        # W1A[a, mo, s, lane]
        mo = torch.arange(2, device=dev).view(2, 1, 1); s_ = torch.arange(Dp // 2, device=dev).view(1, -1, 1)
        k1 = (2 * s_ + hh_l.view(1, 1, 64)).expand(2, Dp // 2, 64)
        j1 = (32 * mo + c_l.view(1, 1, 64)).expand(2, Dp // 2, 64)
        W1A = w1[:, k1, j1].reshape(A, -1)
        # W2A[a, mo, mt, vq, lane, t]
        mo5 = torch.arange(2, device=dev).view(2, 1, 1, 1, 1); mt5 = torch.arange(2, device=dev).view(1, 2, 1, 1, 1)
        vq5 = torch.arange(4, device=dev).view(1, 1, 4, 1, 1); t5 = torch.arange(4, device=dev).view(1, 1, 1, 1, 4)
        l5 = lane.view(1, 1, 1, 64, 1)
        k2 = nid(mt5, 4 * vq5 + t5, l5 >> 5).expand(2, 2, 4, 64, 4)
        j2 = (32 * mo5 + (l5 & 31)).expand(2, 2, 4, 64, 4)
        W2A = self.w2.float()[:, k2, j2].reshape(A, -1)
        # per-neuron vectors [hh, mo, v]
        hh3 = torch.arange(2, device=dev).view(2, 1, 1); mo3 = torch.arange(2, device=dev).view(1, 2, 1)
        idx = nid(mo3, v16.view(1, 1, 16), hh3).reshape(-1)                      # [hh][mo][v] -> neuron
        small = [x.float().reshape(A, 64)[:, idx] for x in (self.b1, self.g1, self.h1, self.b2, self.g2, self.h2)]
        w3, b3 = self.w3.float(), self.b3.float()
        if self.n_actions < 4:                      # continuous: [speed, turn, shoot] + a zero column; a value head: [V] + three zero columns
            pad = 4 - self.n_actions
            w3 = torch.cat([w3, torch.zeros_like(w3[:, :, :1]).expand(-1, -1, pad)], dim=2)
            b3 = torch.cat([b3, torch.zeros_like(b3[:, :, :1]).expand(-1, -1, pad)], dim=2)
        W3P = w3[:, idx, :].reshape(A, -1)                                         # [hh][mt][v][4]: same index pattern with mt for mo
        # W2B[a, mo, s, term, lane, i]: the 64 x 64 layer split in three bfloat16 terms (precision="bf16x3" reads two, "bf16x6" all)
        mo6 = torch.arange(2, device=dev).view(2, 1, 1, 1); s6 = torch.arange(4, device=dev).view(1, 4, 1, 1)
        l6 = lane.view(1, 1, 64, 1); i6 = torch.arange(8, device=dev).view(1, 1, 1, 8)
        kb = nid(s6 >> 1, 8 * (s6 & 1) + i6, l6 >> 5).expand(2, 4, 64, 8)
        jb = (32 * mo6 + (l6 & 31)).expand(2, 4, 64, 8)
        wsel = self.w2.float()[:, kb, jb]                                          # [A, mo, s, lane, i]
        wh = wsel.to(torch.bfloat16)
        wm = (wsel - wh.float()).to(torch.bfloat16)
        wl = (wsel - wh.float() - wm.float()).to(torch.bfloat16)
        W2B = torch.stack([wh, wm, wl], dim=3).contiguous().view(torch.int16).reshape(A, -1)  # [A, mo, s, term 3, lane, i]
        W2B = W2B.view(torch.float32)                                              # 2 bf16 per float slot: [A, 6144]
        blob = torch.cat([W1A, W2A, *small, W3P, b3.reshape(A, -1), W2B], dim=1).contiguous()
        if out is not None:
            out.copy_(blob)
            return out
        return blob

    @torch.no_grad()
    def load_reference_actor(self, agent_index, state_dict):
        """Copy one reference ActorNetwork checkpoint (keys fc1/bn1/fc2/bn2/pi .weight/.bias, networks.py:58-75) into
        slot `agent_index`."""
        i = agent_index
        self.w1[i].copy_(state_dict["fc1.weight"].t()); self.b1[i, 0].copy_(state_dict["fc1.bias"])
        self.g1[i, 0].copy_(state_dict["bn1.weight"]); self.h1[i, 0].copy_(state_dict["bn1.bias"])
        self.w2[i].copy_(state_dict["fc2.weight"].t()); self.b2[i, 0].copy_(state_dict["fc2.bias"])
        self.g2[i, 0].copy_(state_dict["bn2.weight"]); self.h2[i, 0].copy_(state_dict["bn2.bias"])
        self.w3[i].copy_(state_dict["pi.weight"].t()); self.b3[i, 0].copy_(state_dict["pi.bias"])


class FusedActor:
    """The same per-agent actor as ONE hand-written HIP kernel on the matrix cores (csrc/bsx_actor.hip,
    `bsx_actor_forward`: f32 MFMA, exact f32): a row never leaves the register file -- 4*D bytes in, 16 bytes out --
    instead of ~20 memory-bound torch passes over [A, E, 64] activations.  Holds a packed copy of a StackedActor's weights; call `refresh()` after the learner updates them.
    Optional exploration noise (Gaussian, then clamp(-1, 1) as maddpg/agent.py:31) is drawn in-kernel.
    precision: "f32" = exact float32 everywhere; "bf16x6" = the 64 x 64 layer as six bf16 matrix products of three-term splits of
    both operands (float32-class accuracy, ~1e-7, 2.7x the matrix rate); "bf16x3" = three bf16 matrix products of two-term
    splits of both operands (about 1e-5 on a score, 16x the matrix rate); everything else stays float32."""

    PRECISIONS = {"f32": _lib.ACTOR_F32, "bf16x3": _lib.ACTOR_BF16X3, "bf16x6": _lib.ACTOR_BF16X6}

    def __init__(self, actor, n_agents_per_team, seed=0, precision="f32", env_offset=0):
        """env_offset: global index of the shard's first game (sharding.make_shard): exploration noise is keyed by the GLOBAL row,
        so a job's noise is the same however its games are split over ranks."""
        self.actor, self.n, self.env_offset = actor, int(n_agents_per_team), int(env_offset)
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)}")
        self.precision = self.PRECISIONS[precision]
        if actor.n_actors != 2 * self.n or actor.obs_len != 3 * self.n + 2:
            raise ValueError("actor shape does not match the env")
        self._lib = _lib.load()
        self.weights = actor.pack()
        nf = _lib.c_int()
        _lib.check(self._lib.bsx_actor_blob_floats(actor.obs_len, _lib.ctypes.byref(nf)), "bsx_actor_blob_floats")
        assert self.weights.shape == (actor.n_actors, nf.value), (self.weights.shape, nf.value)
        self.seed, self.seq = int(seed), 0

    def refresh(self):
        self.actor.pack(out=self.weights)

    def noise_struct(self, E, noise_std=0.0, ou=None, z=None, sample=None, value=None):
        """BsxActorNoise for E games (None = nothing to say); validates the OU state tensor.  z: optional float32 [E, A, 4] standard
        normals to use instead of the in-kernel draws (parity runs against the reference's np.random.randn values).
        sample: dict(temperature[, logp, u]) -- categorical policy head: the action is drawn from softmax(scores / temperature)
        (Gumbel-max; the score rows that go out are the perturbed ones), `logp` a float32 tensor ([E, A], or [T, E, A] for a one-launch
        rollout) that receives the drawn action's log-probability, `u` float32 [E, A, 4] uniforms to use instead of the draws (tests).
        value: dict(weights, out) -- a value head: `weights` the packed blob of a StackedActor with ONE output per agent, `out` a
        float32 tensor ([E, A] / [T, E, A]) that receives V(obs)."""
        if not (noise_std > 0.0 or ou is not None or sample is not None or value is not None):
            return None
        nz = _lib.BsxActorNoise(float(noise_std), 0.0, 0.15, 0.2, 0.0, None, None, None, 0, 0, 0.0, None, None, None, None)
        if sample is not None:
            nz.sample_mode, nz.temperature = 1, float(sample.get("temperature", 1.0))
            if not nz.temperature > 0.0:
                raise ValueError("temperature must be > 0")
            lp, u = sample.get("logp"), sample.get("u")
            if lp is not None:
                if lp.dtype != torch.float32 or not lp.is_contiguous() or tuple(lp.shape[-2:]) != (E, 2 * self.n):
                    raise ValueError("sample['logp'] must be a contiguous float32 [.., E, A] tensor")
                nz.logp = lp.data_ptr()
            if u is not None:
                if u.dtype != torch.float32 or tuple(u.shape) != (E, 2 * self.n, 4) or not u.is_contiguous():
                    raise ValueError("sample['u'] must be a contiguous float32 [E, A, 4] tensor")
                nz.u_inject = u.data_ptr()
        if value is not None:
            w, out = value["weights"], value["out"]
            if tuple(w.shape) != tuple(self.weights.shape) or w.dtype != torch.float32 or not w.is_contiguous():
                raise ValueError("value['weights'] must be the packed blob of a StackedActor(n_actors, obs_len, 1)")
            if out.dtype != torch.float32 or not out.is_contiguous() or tuple(out.shape[-2:]) != (E, 2 * self.n):
                raise ValueError("value['out'] must be a contiguous float32 [.., E, A] tensor")
            nz.value_weights, nz.value = w.data_ptr(), out.data_ptr()
        if z is not None:
            if z.dtype != torch.float32 or tuple(z.shape) != (E, 2 * self.n, 4) or not z.is_contiguous():
                raise ValueError("z must be a contiguous float32 [E, A, 4] tensor")
            nz.z_inject = z.data_ptr()
        if ou is not None:
            st = ou["state"]
            if st.dtype != torch.float32 or tuple(st.shape) != (E, 2 * self.n, 4) or not st.is_contiguous():
                raise ValueError("ou['state'] must be a contiguous float32 [E, A, 4] tensor")
            nz.ou_scale, nz.ou_theta = float(ou["scale"]), float(ou.get("theta", 0.15))
            nz.ou_sigma, nz.ou_mu = float(ou.get("sigma", 0.2)), float(ou.get("mu", 0.0))
            nz.ou_state = st.data_ptr()
            nz.env_done = ou["env_done"].data_ptr() if ou.get("env_done") is not None else None
            nz.ou_keep = 0 if ou.get("restart", True) else 1
        return nz

    def forward_into(self, obs, scores, noise_std=0.0, seq=None, seq_base=None, ou=None, z=None, sample=None, value=None, games=None):
        """obs f32 [E, A, D] (contiguous) -> scores f32 [E, A, 4] (contiguous, 16-byte aligned), on the current stream.
        noise_std: Gaussian exploration noise.  ou: optional dict(scale, state[, theta, sigma, mu, env_done]) for the
        reference's Ornstein-Uhlenbeck noise (utils/noise.py): `state` is a float32 [E, A, 4] tensor updated in place,
        `env_done` a uint8 [E] tensor whose set rows restart the process.
        seq_base: optional int64 device tensor (1 element) added to `seq` in-kernel (for captured graphs).
        games: (first, count) = evaluate only that range of the games (every tensor still the full one; the noise is keyed by the
        global row, so the draws are those of the full call)."""
        off = self.env_offset
        if games is not None:
            sl = slice(int(games[0]), int(games[0]) + int(games[1]))
            cut = lambda d, keys: None if d is None else dict(d, **{k: d[k][sl] for k in keys if d.get(k) is not None})   # noqa: E731
            obs, scores, z = obs[sl], scores[sl], (z[sl] if z is not None else None)
            ou, sample, value = cut(ou, ("state", "env_done")), cut(sample, ("logp", "u")), cut(value, ("out",))
            off += int(games[0])
        E = obs.shape[0]
        if seq is None:
            self.seq += 1
            seq = self.seq
        nz = self.noise_struct(E, noise_std, ou, z, sample, value)
        _lib.check(self._lib.bsx_actor_forward(self.weights.data_ptr(), obs.data_ptr(), scores.data_ptr(), E, self.n, self.precision,
                                               _lib.ctypes.byref(nz) if nz is not None else None, self.seed, int(seq),
                                               seq_base.data_ptr() if seq_base is not None else None, off,
                                               torch.cuda.current_stream(obs.device).cuda_stream), "bsx_actor_forward")
        return scores

    def __call__(self, obs, noise_std=0.0, seq=None):
        scores = torch.empty((*obs.shape[:2], 4), dtype=torch.float32, device=obs.device)
        return self.forward_into(obs.contiguous(), scores, noise_std, seq)


class PolicyRollout:
    """T ticks of (actor -> step) for all games, captured once into a HIP graph and replayed.

        ro = PolicyRollout(env, actor, T=32, noise_std=0.1)
        ro.start(); ro.capture()
        ro.run()                       # one replay = T ticks of every game
        ro.obs[t], ro.scores[t], ro.rew[t], ro.done[t], ro.obs[t+1]   # transition t, buffers in HBM
        ro.valid[t]                    # bool [E]: tick t found the game running.  A tick on a FINISHED game is the reference's
                                       # inert call (battle_env.py:303-306) or, with auto_reset, the re-spawn: its row pairs the old
                                       # game's last observation with the new game's first and is not a transition

    The env must be batched, discrete, rng='philox'; auto_reset is recommended (finished games re-spawn in place).
    Exploration noise on the score vectors is Gaussian (noise_std) and/or the reference's Ornstein-Uhlenbeck process
    (ou_scale = main.py's curr_noise; utils/noise.py), whose state is one more [E, A, 4] tensor updated inside the actor
    kernel and restarted per game; then clamp(-1, 1) as maddpg/agent.py:31 does."""

    def __init__(self, env, actor, T, noise_std=0.0, fused=True, seed=0, opponent=None, ou_scale=0.0, one_launch=False,
                 precision="f32", ou_restart=True, sample=None, temperature=1.0, value_actor=None, chains="auto"):
        """actor: a StackedActor.  fused=True evaluates it with the hand-written HIP kernel (FusedActor), False with
        torch ops (the fp32 reference of the same op).  opponent: an `instinct.Team` that plays its team's planes
        instead of the actor (the reference's training setup, main.py:119-122: learned red vs scripted blue): its
        one-hot scores overwrite that team's rows of the score tensor each tick, on device."""
        # chains > 1 (graph form, fused actor, no scripted opponent): the games as that many ranges, each its own chain of
        # (actor -> step) launch pairs on a branch of the graph (battle_env.capture_steps(chains=)): one range's matrix-core actor
        # pass runs under another range's step kernel.  Same transitions bit for bit.  chains="auto" (the default) takes two chains where
        # they pay in EITHER use -- replays queued back to back, or every replay synchronised (a rollout, then the learner): 2v2 and larger
        # from ~260 k agents per tick (65 536 x 2v2: 43.1 -> 38.4 us per tick back to back, 42.8 -> 40.3 synchronised).  Not 1v1: back to
        # back 23.0 -> 21.3, but a multi-branch graph is launched node by node at ~6 us each and a synchronised replay pays for that --
        # 23.5 -> 26.3 (profiles/r06_chains_1v1.json); chains=2 is there for the caller who queues replays.
        # one_launch beyond 4v4 (an MFMA tile per plane id and a workgroup of 32 games stop fitting: the fused kernels exist for 1v1 ... 4v4):
        # the rollout falls through to the graph of (actor -> step) pairs, as chains over game ranges where those pay -- the same
        # transitions; `form_note` says so
        self.form_note = None
        if one_launch and fused and env.n_agents > 4:
            one_launch = False
            if chains in (1, "auto") and opponent is None:
                chains = "auto"
            self.form_note = (f"one_launch asked for {env.n_agents}v{env.n_agents}: the fused kernels cover 1v1 ... 4v4, "
                              "this rollout runs as the two-kernel graph (chains over game ranges where they pay)")
            import warnings
            warnings.warn(self.form_note, RuntimeWarning, stacklevel=2)      # the caller asked for a form that is not the one running
        if chains == "auto":                           # two chains where they were measured to pay (above)
            agents = env.n_envs * 2 * env.n_agents
            chains = 2 if (env.n_agents >= 2 and agents >= (1 << 18) and not one_launch and fused and opponent is None) else 1
        self.chains = int(chains)
        if self.chains > 1 and (one_launch or not fused or opponent is not None):
            raise ValueError("chains > 1 is for the graph form with the fused actor and no scripted opponent")
        if env._compat or env.rng != "philox":
            raise ValueError("PolicyRollout needs a batched env with rng='philox'")
        self.continuous = bool(env.continuous_actions)
        if actor.n_actions != (3 if self.continuous else 4):
            raise ValueError("the actor must have 4 outputs for a discrete env (action scores), 3 for a continuous one")
        self.env, self.actor, self.T, self.noise_std = env, actor, int(T), float(noise_std)
        self.fused = FusedActor(actor, env.n_agents, seed=seed, precision=precision, env_offset=env.env_offset) if fused else None
        self.opponent = opponent
        # ou_scale > 0: the reference's Ornstein-Uhlenbeck exploration noise (utils/noise.py; main.py:151-155 scales it per
        # game and restarts it at every game start) -- fused path only; the process state is one more [E, A, 4] tensor
        self.ou = None
        if ou_scale > 0.0:
            if not fused:
                raise ValueError("ou_scale needs the fused actor")
            # ou_restart=False: the process is never restarted -- the reference's evaluation loop (evaluate.py:52-76) never calls
            # reset_noise, whereas training restarts it at every game start (main.py:155)
            self.ou = dict(scale=float(ou_scale), restart=bool(ou_restart),
                           state=torch.zeros((env.n_envs, env._A, 4), dtype=torch.float32, device=env.device))
        # one_launch: all T ticks (actor -> step) in ONE kernel (bsx_rollout_discrete / _continuous) instead of 2T launches in a
        # graph: observation rows stay in LDS between the step and the actor, game state in registers / L2.  Up to 4v4; discrete: a
        # scripted opponent (instinct.Team of one side) is played in-kernel and its actor is skipped; same transitions, bit for bit.
        self.one_launch = bool(one_launch)
        if self.one_launch and not fused:                  # (with the fused actor, teams beyond 4v4 fell through to the graph above)
            raise ValueError("one_launch needs the fused actor")
        if self.one_launch and opponent is not None and getattr(opponent, "team", None) not in (0, 1):
            raise ValueError("one_launch plays a scripted opponent in-kernel: it must be an instinct.Team of one side")
        self._seq_base = torch.zeros(1, dtype=torch.int64, device=env.device)
        E, A, D, dev = env.n_envs, env._A, env.obs_size, env.device
        # sample="categorical": a stochastic policy head for policy-gradient learners (BASELINE.json configs[4] words C5 as a PPO rollout):
        # the action is DRAWN from softmax(scores / temperature) in-kernel, `logp[t]` holds its log-probability; value_actor (a
        # StackedActor with one output per agent) adds `value[t]` = V(obs[t]).  Fused actor only; the one-launch form takes the value
        # head at 1v1.
        if sample not in (None, "categorical"):
            raise ValueError("sample must be None or 'categorical'")
        if (sample is not None or value_actor is not None) and (not fused or self.continuous):
            raise ValueError("the categorical / value heads need the fused actor and a discrete env")
        self.logp = torch.zeros((T, E, A), dtype=torch.float32, device=dev) if sample is not None else None
        self._sample = dict(temperature=float(temperature)) if sample is not None else None
        self.value, self._value_w = None, None
        if value_actor is not None:
            if value_actor.n_actions != 1 or value_actor.n_actors != A or value_actor.obs_len != D:
                raise ValueError("value_actor must be a StackedActor(n_actors=A, obs_len=D, n_actions=1)")
            if self.one_launch and env.n_agents != 1:
                raise ValueError("the one-launch rollout takes a value head at 1v1 only (use the graph form)")
            self.value_actor, self._value_w = value_actor, value_actor.pack()
            self.value = torch.zeros((T, E, A), dtype=torch.float32, device=dev)
        self.obs = torch.empty((T + 1, E, A, D), dtype=torch.float32, device=dev)
        # discrete: 4 action scores, arg-maxed in the step kernel; continuous: [speed, turn, shoot] + one unused column
        self.scores = torch.zeros((T, E, A, 4), dtype=torch.float32, device=dev)
        self._kind = _lib.ACT_F32X4 if self.continuous else _lib.ACT_LOGITS_F32
        # continuous + scripted opponent, per-tick form: the reference hands the env float32 rows from the actors and float64 rows from
        # instinct/agent.py:41-54 side by side; here the actors' rows are widened (exactly) into one float64 [E, A, 3] array next to them
        self._act64 = torch.zeros((E, A, 3), dtype=torch.float64, device=dev) if (self.continuous and opponent is not None) else None
        self.rew = torch.empty((T, E, A), dtype=torch.float32, device=dev)
        self._done = torch.empty((T, E, A), dtype=torch.uint8, device=dev)
        self.done = self._done.view(torch.bool)
        # env_done[t] = the games' env_done BEFORE tick t (row 0: when the rollout starts; row t+1 is written by tick t)
        self.env_done = torch.ones((T + 1, E), dtype=torch.uint8, device=dev)
        self.graph, self._side = None, None

    @property
    def valid(self):
        """bool [T, E]: tick t started on a running game, i.e. row t of that game is a transition of the reference's loop
        (`while not env.env_done: step`, main.py:177-181)."""
        return self.env_done[:self.T] == 0

    def _tick(self, t, games=None):
        if games is not None:                          # a chain's tick: the fused actor and the step over one range of the games
            ou = dict(self.ou, env_done=self.env_done[t] if self.ou["restart"] else None) if self.ou is not None else None
            sample = dict(self._sample, logp=self.logp[t]) if self._sample is not None else None
            value = dict(weights=self._value_w, out=self.value[t]) if self._value_w is not None else None
            self.fused.forward_into(self.obs[t], self.scores[t], self.noise_std, seq=t, seq_base=self._seq_base, ou=ou, sample=sample,
                                    value=value, games=games)
            self.env._launch(self.scores[t].data_ptr(), self._kind, False, None,
                             self.obs[t + 1].data_ptr(), self.rew[t].data_ptr(), self._done[t].data_ptr(),
                             env_done_ptr=self.env_done[t + 1].data_ptr(), games=games)
            return
        if self.fused is not None:
            # graph arguments are frozen: the noise key is (seed, seq_base + t, row) with seq_base a device word that the
            # graph advances by T once per replay (_body)
            ou = dict(self.ou, env_done=self.env_done[t] if self.ou["restart"] else None) if self.ou is not None else None
            sample = dict(self._sample, logp=self.logp[t]) if self._sample is not None else None
            value = dict(weights=self._value_w, out=self.value[t]) if self._value_w is not None else None
            self.fused.forward_into(self.obs[t], self.scores[t], self.noise_std, seq=t, seq_base=self._seq_base, ou=ou, sample=sample, value=value)
            if self._act64 is not None:
                cols = slice(self.opponent._cols[0], self.opponent._cols[-1] + 1)   # a team's planes are adjacent columns (a slice: nothing is uploaded while capturing)
                self._act64.copy_(self.scores[t][..., :3])
                self.opponent.write_actions(out=self._act64, obs=self.obs[t], seq=t, seq_base=self._seq_base)
                self.scores[t][:, cols, :3] = self._act64[:, cols].float()      # the record: the scripted rows rounded to float32, as the one-launch form writes them
                self.scores[t][:, cols, 3] = 0.0
                self.env._launch(self._act64.data_ptr(), _lib.ACT_F64, False, None,
                                 self.obs[t + 1].data_ptr(), self.rew[t].data_ptr(), self._done[t].data_ptr(),
                                 env_done_ptr=self.env_done[t + 1].data_ptr())
                return
            if self.opponent is not None:
                self.opponent.write_actions(out=self.scores[t], obs=self.obs[t])
            self.env._launch(self.scores[t].data_ptr(), self._kind, False, None,
                             self.obs[t + 1].data_ptr(), self.rew[t].data_ptr(), self._done[t].data_ptr(),
                             env_done_ptr=self.env_done[t + 1].data_ptr())
            return
        with torch.no_grad():
            s = self.actor(self.obs[t])
            if self.noise_std > 0.0:
                s = (s + self.noise_std * torch.randn_like(s)).clamp_(-1.0, 1.0)
            self.scores[t][..., :s.shape[-1]].copy_(s)
        if self.opponent is not None:
            self.opponent.write_actions(out=self.scores[t], obs=self.obs[t])
        self.env._launch(self.scores[t].data_ptr(), self._kind, False, None,
                         self.obs[t + 1].data_ptr(), self.rew[t].data_ptr(), self._done[t].data_ptr(),
                         env_done_ptr=self.env_done[t + 1].data_ptr())

    def start(self):
        """Begin from the env's current observations (call after env.reset())."""
        self.obs[self.T].copy_(self.env._obs)          # every run starts by moving obs[T] to obs[0]
        self.env_done[self.T].copy_(self.env._env_done)

    def capture(self):
        """Record the T ticks into a HIP graph (one warm-up pass runs first, on a side stream, as torch requires)."""
        s = torch.cuda.Stream(device=self.env.device)
        s.wait_stream(torch.cuda.current_stream(self.env.device))
        with torch.cuda.stream(s):
            with torch.no_grad():
                self.actor(self.obs[self.T])           # library warm-up (hipBLASLt handles, workspaces) outside the capture
        torch.cuda.current_stream(self.env.device).wait_stream(s)
        torch.cuda.synchronize(self.env.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body()
        return self

    def _body(self):
        self.obs[0].copy_(self.obs[self.T])            # continue where the previous rollout ended
        self.env_done[0].copy_(self.env_done[self.T])
        if self.one_launch:
            sample = dict(self._sample, logp=self.logp) if self._sample is not None else None
            value = dict(weights=self._value_w, out=self.value) if self._value_w is not None else None
            nz = self.fused.noise_struct(self.env.n_envs, self.noise_std, self.ou, sample=sample, value=value)
            self.env._launch_rollout(self.T, self.fused.weights.data_ptr(), self.fused.precision,
                                     -1 if self.opponent is None else self.opponent.team, self.obs.data_ptr(), self.scores.data_ptr(),
                                     self.rew.data_ptr(), self._done.data_ptr(), nz, self.fused.seed, 0, self._seq_base.data_ptr(),
                                     env_done_t_ptr=self.env_done[1].data_ptr(),
                                     scripted_seed=0 if self.opponent is None else self.opponent.seed)
        else:
            ranges = self.env.chain_ranges(self.chains)
            if len(ranges) == 1:
                for t in range(self.T):
                    self._tick(t)
            else:
                main = torch.cuda.current_stream(self.env.device)
                if self._side is None:
                    self._side = [torch.cuda.Stream(self.env.device) for _ in ranges[1:]]
                for s in self._side:
                    s.wait_stream(main)
                for r, games in enumerate(ranges):
                    with torch.cuda.stream(main if r == 0 else self._side[r - 1]):
                        for t in range(self.T):
                            self._tick(t, games)
                for s in self._side:
                    main.wait_stream(s)
            self.env._env_done.copy_(self.env_done[self.T])        # the per-tick launches wrote their flags into the record
        self._seq_base.add_(self.T)                    # fresh exploration-noise keys for the next run

    def run(self):
        """T ticks (asynchronous).  Afterwards obs[0..T], scores, rew, done hold this rollout's transitions."""
        if self.graph is None:
            self._body()
        else:
            self.graph.replay()


def reference_checkpoint_actor(fixture, n_agents_per_team=2, device="cuda"):
    """StackedActor holding the reference's SHIPPED policy (models/completed_model/actor_plane0, actor_plane1: 8 -> 64 -> 64 -> 4,
    trained for 217 651 games) in its red slots, read from a recorded copy of the two state_dicts (tests/golden/g12_evaluation.npz,
    written by make_golden.py from the checkpoint files: arrays `plane{i}/fc1.weight` ...).  The blue slots repeat plane 0 / 1: in the
    evaluation workload blue is the scripted opponent and those slots are never evaluated."""
    import numpy as np
    z = np.load(fixture) if isinstance(fixture, str) else fixture
    n = int(n_agents_per_team)
    actor = StackedActor(2 * n, 3 * n + 2, 4, device=device)
    for slot in range(2 * n):
        src = slot % n
        sd = {k.split("/", 1)[1]: torch.from_numpy(np.asarray(z[k])).to(device) for k in z.files
              if k.startswith(f"plane{src}/") and k.split("/", 1)[1] not in ("x", "y")}
        actor.load_reference_actor(slot, sd)
    return actor


class _FirstGamesTally:
    """The outcome of exactly the FIRST k games of every slot (the env's own per-slot counters, frozen when a slot's k-th game ends).
    Stopping all slots at one moment and counting what has finished by then over-counts short games -- each slot's unfinished game is
    dropped, and the longer a game the likelier it is the dropped one -- whereas the reference plays N whole games one after another
    (evaluate.py:52).  Checked every few ticks (fewer than a game can last), so a slot passes k by one game at a time; the rare call
    that ends two games at once (both bases in one step, battle_env.py:363-372) is taken whole."""

    def __init__(self, env, k):
        import numpy as np
        self.k, self.c0 = int(k), env.counters().astype(np.int64)          # (host arrays: [E, 4] = games, ties, red wins, blue wins)
        self.frozen = np.zeros_like(self.c0)
        self.closed = np.zeros(env.n_envs, dtype=bool)

    def update(self, env):
        c = env.counters() - self.c0
        hit = (~self.closed) & (c[:, 0] >= self.k)
        self.frozen[hit] = c[hit]
        self.closed |= hit
        return bool(self.closed.all())

    def totals(self):
        return self.frozen.sum(0)


def play_reference_evaluation(env, actor, games, T=32, one_launch=True, precision="f32", seed=0, ou_scale=0.1, first_tick_stale_obs=False,
                              games_per_slot=None):
    """The reference's evaluation workload (evaluate.py:32-76; README.md:30 quotes "~80 %" for it) for every game slot of `env` at
    once: red = `actor` through maddpg/agent.py:25-33 -- tanh scores + Ornstein-Uhlenbeck noise of scale 0.1 (utils/noise.py's
    default, which evaluate.py never rescales) that is NEVER restarted (evaluate.py never calls reset_noise) -> clamp -> arg-max --,
    blue = the scripted instinct.Team, finished games re-spawned in place.  Plays whole rollouts of T ticks until at least `games`
    games are over and returns the tally from the env's own counters (battle_env.py:102-103,169-170,449-455).
    One difference to the script is kept out by default: evaluate.py resets the env twice per game and feeds the FIRST reset's
    observations to the first tick (evaluate.py:53-66), i.e. every plane's first action of a game -- one in ~40 -- is chosen on
    another game's spawn.  games_per_slot=k: play until EVERY slot has finished k games and tally exactly each slot's first k (an
    unbiased sample of k * n_envs whole games: _FirstGamesTally; use a small T, e.g. 4 -- the tally looks at the counters once per T
    ticks); `games` is then ignored.  first_tick_stale_obs=True reproduces the script's quirk: a second env of the same shape does nothing but draw spawns, and
    the first tick of every game sees ITS observations (both teams, as in the script); that form runs tick by tick (actor launch,
    scripted team, step launch -- the two-kernel form, eagerly), since the substitution sits between a step and the next actor pass."""
    from . import instinct
    if env.n_envs < 1 or not env.auto_reset or env.continuous_actions:
        raise ValueError("the evaluation workload needs a batched discrete env with auto_reset=True")
    opp = instinct.Team(env.possible_blue, env.possible_red, env)
    if first_tick_stale_obs:
        from .envs.battle_env import parallel_env
        decoy = parallel_env(n_agents=env.n_agents, n_envs=env.n_envs, device=env.device, seed=env.seed + 0x5EED, env_offset=env.env_offset)
        ro = PolicyRollout(env, actor, 1, opponent=opp, ou_scale=ou_scale, ou_restart=False, one_launch=False, precision=precision, seed=seed)
        c0 = env.counters().sum(0)
        env.reset()
        ro.start()
        fresh = torch.ones(env.n_envs, dtype=torch.bool, device=env.device)       # every slot's first game starts on the discarded reset's rows too
        ticks = 0
        first = _FirstGamesTally(env, games_per_slot) if games_per_slot else None
        while True:
            for _ in range(T):
                ro.obs[0].copy_(ro.obs[1]); ro.env_done[0].copy_(ro.env_done[1])  # (what PolicyRollout._body does around its ticks)
                decoy.reset()                                                     # evaluate.py:53: the reset whose observations the first tick gets
                ro.obs[0][fresh] = decoy._obs[fresh]
                ro._tick(0)
                env._env_done.copy_(ro.env_done[1]); ro._seq_base.add_(1)
                fresh = (ro.env_done[0] != 0) & (ro.env_done[1] == 0)             # this call re-spawned the game (evaluate.py:64): its next tick is a first tick
            ticks += T
            if first is not None:
                if first.update(env):
                    c = first.totals()
                    break
                continue
            c = env.counters().sum(0) - c0
            if c[0] >= games:
                break
        return {"games": int(c[0]), "ties": int(c[1]), "red_wins": int(c[2]), "blue_wins": int(c[3]), "win_rate_red": float(c[2]) / float(c[0]),
                "ticks": ticks, "rollout": ro}
    ro = PolicyRollout(env, actor, T, opponent=opp, ou_scale=ou_scale, ou_restart=False, one_launch=one_launch, precision=precision, seed=seed)
    c0 = env.counters().sum(0)
    env.reset()
    ro.start(); ro.capture()
    ticks = 0
    first = _FirstGamesTally(env, games_per_slot) if games_per_slot else None
    while True:
        ro.run()
        ticks += T
        if first is not None:
            if first.update(env):
                c = first.totals()
                break
            continue
        c = env.counters().sum(0) - c0
        if c[0] >= games:
            break
    return {"games": int(c[0]), "ties": int(c[1]), "red_wins": int(c[2]), "blue_wins": int(c[3]), "win_rate_red": float(c[2]) / float(c[0]),
            "ticks": ticks, "rollout": ro}
