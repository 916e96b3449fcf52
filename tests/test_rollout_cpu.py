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
