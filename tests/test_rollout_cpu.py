"""The stacked on-device actor against the reference ActorNetwork's forward pass (fixture g9, CPU torch)."""
import os

import numpy as np
import torch

from trace_util import GOLDEN


def test_stacked_actor_matches_reference_actor_forward():
    from deep_rl_battlespace_amd.rollout import StackedActor
    z = np.load(os.path.join(GOLDEN, "g9_actor_forward.npz"))
    for tag, obs_len in (("1v1", 5), ("2v2", 8), ("4v4", 14)):
        sd = {k.split("/", 1)[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/") and k[-1] not in "xy"}
        x, y = torch.from_numpy(z[f"{tag}/x"]), z[f"{tag}/y"]
        actor = StackedActor(3, obs_len, 4)
        actor.load_reference_actor(1, sd)                      # slot 1 of 3; the others keep their random init
        obs = torch.rand(64, 3, obs_len)
        obs[:, 1] = x
        with torch.no_grad():
            out = actor(obs)
        assert out.shape == (64, 3, 4)
        np.testing.assert_allclose(out[:, 1].numpy(), y, rtol=1e-5, atol=1e-6)
        assert float(out.abs().max()) <= 1.0


def test_replay_buffer_ring_and_sample_layout_follow_the_reference():
    """maddpg/buffer.py semantics on tensors: ring index, overwrite of the oldest rows, sample() tuple layout."""
    from deep_rl_battlespace_amd.replay import ReplayBuffer
    agents, obs_size, nact = ["plane0", "plane1"], 8, 4
    buf = ReplayBuffer(10, 4, agents, obs_size, obs_size * 2, nact, device="cpu", generator=torch.Generator().manual_seed(0))
    ref_rows = []

    def push(E, k):
        s = {a: torch.full((E, obs_size), float(k + i)) for i, a in enumerate(agents)}
        s2 = {a: torch.full((E, obs_size), float(k + i) + 0.5) for i, a in enumerate(agents)}
        act = {a: torch.full((E,), (k + i) % 4, dtype=torch.int64) for i, a in enumerate(agents)}
        rw = {a: torch.full((E,), float(k)) for a in agents}
        dn = {a: torch.full((E,), bool(k % 2)) for a in agents}
        buf.store_transition(s, act, rw, s2, dn)
        ref_rows.extend([k] * E)
    assert not buf.is_ready()
    push(3, 1); push(3, 2); assert buf.is_ready() and buf.mem_cntr == 6
    push(3, 3); push(3, 4)                                   # wraps: 12 rows into a 10-row ring
    assert buf.mem_cntr == 12
    want = np.zeros(10)
    for i, k in enumerate(ref_rows):
        want[i % 10] = k                                     # reference: index = mem_cntr % mem_size
    assert np.array_equal(buf.rew_mem[:, 0].numpy(), want)
    assert np.array_equal(buf.state_mem[:, 0].numpy(), want) and np.array_equal(buf.state_mem[:, obs_size].numpy(), want + 1)
    assert buf.action_mem.sum(-1).eq(1).all()                 # indices stored one-hot
    a_s, s, a, r, a_s2, s2, d = buf.sample()
    assert a_s.shape == (2, 4, obs_size) and s.shape == (4, 2 * obs_size) and a.shape == (2, 4, nact)
    assert r.shape == (4, 2) and a_s2.shape == (2, 4, obs_size) and s2.shape == (4, 2 * obs_size) and d.shape == (4, 2)
    assert torch.equal(s[:, :obs_size], a_s[0]) and torch.equal(s2[:, obs_size:], a_s2[1]) and torch.equal(a_s2[0], a_s[0] + 0.5)
    big = ReplayBuffer(5, 2, agents, obs_size, obs_size * 2, nact, device="cpu")
    big._put(torch.arange(12.).reshape(12, 1, 1).expand(12, 2, obs_size), torch.zeros(12, 2, nact), torch.arange(12.).reshape(12, 1).expand(12, 2),
             torch.zeros(12, 2, obs_size), torch.zeros(12, 2, dtype=torch.bool))
    assert sorted(big.rew_mem[:, 0].tolist()) == [7., 8., 9., 10., 11.] and big.mem_cntr == 12
