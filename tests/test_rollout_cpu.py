"""The stacked on-device actor against the reference ActorNetwork's forward pass (fixture g9, CPU torch)."""
import os

import numpy as np
import pytest
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


def _mfma_32x32x2(avec, bvec, acc):
    """v_mfma_f32_32x32x2_f32 on host arrays indexed by lane: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31],
    D[i = (v&3) + 8(v>>2) + 4(l>>5)][j = l&31] in register v of lane l (cdna_hip_programming.md section 3)."""
    lanes = np.arange(64)
    Am = np.zeros((32, 2)); Bm = np.zeros((2, 32))
    Am[lanes & 31, lanes >> 5] = avec
    Bm[lanes >> 5, lanes & 31] = bvec
    Dm = Am @ Bm
    for v in range(16):
        acc[v] += Dm[(v & 3) + 8 * (v >> 2) + 4 * (lanes >> 5), lanes & 31]


def _mfma_32x32x16(a8, b8, acc):
    """v_mfma_f32_32x32x16_bf16 on host arrays [lane, 8]: A[i = l&31][k = 8(l>>5) + e], B[k = 8(l>>5) + e][j = l&31], same D map."""
    lanes = np.arange(64)
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for e in range(8):
        Am[lanes & 31, 8 * (lanes >> 5) + e] = a8[:, e]
        Bm[8 * (lanes >> 5) + e, lanes & 31] = b8[:, e]
    Dm = Am @ Bm
    for v in range(16):
        acc[v] += Dm[(v & 3) + 8 * (v >> 2) + 4 * (lanes >> 5), lanes & 31]


def _bf16(x):
    """round-to-nearest-even float32 -> bfloat16, returned as float64 values"""
    return torch.as_tensor(np.asarray(x, np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)


@pytest.mark.parametrize("n", [1, 2])
def test_packed_actor_blob_drives_the_mfma_fragment_arithmetic(n):
    """StackedActor.pack() + the kernel's index arithmetic (csrc/bsx_actor.hip), emulated lane by lane on the host with
    the documented MFMA fragment maps, reproduce the plain forward: pins the blob layout of include/battlespace_hip.h."""
    from deep_rl_battlespace_amd.rollout import StackedActor
    torch.manual_seed(n)
    A, D = 2 * n, 3 * n + 2
    act = StackedActor(A, D, 4)
    with torch.no_grad():
        act.g1.uniform_(0.5, 1.5); act.h1.uniform_(-.3, .3); act.g2.uniform_(0.5, 1.5); act.h2.uniform_(-.3, .3); act.w3.mul_(50)
    blob = act.pack().numpy()
    Dp = (D + 1) & ~1
    ow2 = 64 * Dp; osm = ow2 + 4096; ow3 = osm + 384; ob3 = ow3 + 256
    assert blob.shape == (A, ob3 + 4 + 6144)
    obs = torch.rand(64, A, D) * 2 - 1
    want = act(obs).detach().numpy()
    lanes = np.arange(64); hh = lanes >> 5
    for a in range(A):
        W = blob[a]
        sm = W[osm:osm + 384].reshape(6, 2, 2, 16)                    # [vector][hh][mo][v]
        x = obs[:, a, :].numpy()
        acc1 = np.zeros((2, 2, 16, 64))
        for mo in range(2):
            for nt in range(2):
                acc1[mo, nt] = sm[0][hh, mo].T
        for s in range(Dp // 2):
            k = 2 * s + hh
            for mo in range(2):
                avec = W[(mo * (Dp // 2) + s) * 64:(mo * (Dp // 2) + s) * 64 + 64]
                for nt in range(2):
                    bvec = np.where(k < D, x[32 * nt + (lanes & 31), np.minimum(k, D - 1)], 0.0)
                    _mfma_32x32x2(avec, bvec, acc1[mo, nt])

        def ln(acc, gi, bi):
            for nt in range(2):
                for c in range(32):
                    cols = [c, c + 32]
                    vals = acc[:, nt][:, :, cols].reshape(-1)
                    mean = vals.mean(); rstd = 1 / np.sqrt(((vals - mean) ** 2).mean() + 1e-5)
                    for h2 in range(2):
                        for mo in range(2):
                            acc[mo, nt, :, c + 32 * h2] = np.maximum((acc[mo, nt, :, c + 32 * h2] - mean) * rstd * sm[gi, h2, mo] + sm[bi, h2, mo], 0)
        ln(acc1, 1, 2)
        acc2 = np.zeros((2, 2, 16, 64))
        for mo in range(2):
            for nt in range(2):
                acc2[mo, nt] = sm[3][hh, mo].T
        w2 = W[ow2:ow2 + 4096].reshape(2, 2, 4, 64, 4)
        for mt in range(2):
            for v in range(16):
                for mo in range(2):
                    for nt in range(2):
                        _mfma_32x32x2(w2[mo, mt, v >> 2, :, v & 3], acc1[mt, nt, v], acc2[mo, nt])
        # the same layer from the bfloat16 section: W2B[mo][s][term 3][lane][i]; "bf16x3" = three products of the first two terms per
        # K step, "bf16x6" = six products of all three terms (float32-class accuracy)
        w2b = (W[ob3 + 4:ob3 + 4 + 6144].view(np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64).reshape(2, 4, 3, 64, 8)
        w2ref = np.zeros((2, 4, 64, 8))                                                # the float32 weights the terms sum to
        nid = lambda m, v, h2: 32 * m + (v & 3) + 8 * (v >> 2) + 4 * h2              # noqa: E731
        acc2b = np.zeros((2, 2, 16, 64)); acc2c = np.zeros((2, 2, 16, 64))
        for mo in range(2):
            for nt in range(2):
                acc2b[mo, nt] = sm[3][hh, mo].T; acc2c[mo, nt] = sm[3][hh, mo].T
        for s4 in range(4):
            for nt in range(2):
                x = np.stack([acc1[s4 >> 1, nt, 8 * (s4 & 1) + i] for i in range(8)], axis=1)          # [lane, 8]
                x = x.astype(np.float32).astype(np.float64)                              # the kernel splits float32 accumulator values
                xh = _bf16(x); xm = _bf16(x - xh); xl = _bf16(x - xh - xm)
                for mo in range(2):
                    wh, wm, wl = w2b[mo, s4, 0], w2b[mo, s4, 1], w2b[mo, s4, 2]
                    _mfma_32x32x16(wm, xh, acc2b[mo, nt]); _mfma_32x32x16(wh, xm, acc2b[mo, nt]); _mfma_32x32x16(wh, xh, acc2b[mo, nt])
                    for wa, xa in ((wl, xh), (wm, xm), (wh, xl), (wm, xh), (wh, xm), (wh, xh)):
                        _mfma_32x32x16(wa, xa, acc2c[mo, nt])
        assert np.abs(acc2c - acc2).max() < 3e-6 * max(1.0, np.abs(acc2).max())   # six products: float32-class (the emulation's acc1 is float64)
        err_split = np.abs(acc2b - acc2).max()
        assert err_split < 2e-4 * max(1.0, np.abs(acc2).max()), err_split         # ~2^-16 relative; plain bf16 would be ~4e-3
        ln(acc2, 4, 5); ln(acc2b, 4, 5); ln(acc2c, 4, 5)
        w3 = W[ow3:ow3 + 256].reshape(2, 2, 16, 4); b3 = W[ob3:ob3 + 4]
        for acc, tol in ((acc2, 2e-5), (acc2b, 1e-4), (acc2c, 2e-5)):
            out = np.zeros((64, 4))
            for nt in range(2):
                for c in range(32):
                    o = sum(acc[mt, nt, v, c + 32 * h2] * w3[h2, mt, v] for h2 in range(2) for mt in range(2) for v in range(16))
                    out[32 * nt + c] = np.tanh(o + b3)
            np.testing.assert_allclose(out, want[:, a, :], atol=tol)


def test_replay_ring_reproduces_the_reference_buffer_on_cpu():
    """g10: the memory layout and sample() tuple of maddpg/buffer.py (run unmodified by tests/golden/make_golden.py), with the
    ring on the CPU device -- the same check runs with the ring in HBM under -m gpu."""
    from trace_util import check_replay_against_reference
    check_replay_against_reference("cpu")


def test_ou_fixture_is_the_reference_recursion():
    """g11 sanity (no GPU): the recorded trajectory obeys x += theta*(mu - x) + sigma*z, noise = scale*x (utils/noise.py:17-21)
    with the recorded normals, the reset and the re-scale -- what the -m gpu test asks of the in-kernel process."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_ou_noise.npz"))
    scale, mu, theta, sigma = (float(v) for v in g["params"])
    ev = {int(e[0]): (int(e[1]), float(e[2])) for e in g["events"]}
    x = np.full(4, mu)
    for t in range(g["z"].shape[0]):
        if t in ev:
            if ev[t][0] == 1:
                x = np.full(4, mu)
            else:
                scale = ev[t][1]
        x = x + theta * (mu - x) + sigma * g["z"][t]
        np.testing.assert_allclose(g["state"][t], x, rtol=0, atol=1e-15)
        np.testing.assert_allclose(g["noise"][t], x * scale, rtol=0, atol=1e-15)


def test_shipped_checkpoints_forward_and_evaluation_tally_fixture():
    """Fixture g12 (the reference's evaluation workload, evaluate.py:14-109, run unmodified by make_golden.py): the stacked module
    loaded with the SHIPPED checkpoints' weights (models/completed_model/actor_plane0, actor_plane1) reproduces the reference
    ActorNetwork.forward on the observation rows met in the reference's own play; the recorded tally is self-consistent and the
    reward config is cf.json's."""
    from deep_rl_battlespace_amd.rollout import reference_checkpoint_actor
    z = np.load(os.path.join(GOLDEN, "g12_evaluation.npz"))
    n = int(z["n_agents"])
    assert n == 2 and z["cf"].tolist() == [1.0, 0.9, -0.02, -0.03, -0.05]
    actor = reference_checkpoint_actor(z, n, device="cpu")
    x = torch.stack([torch.from_numpy(z[f"plane{i % n}/x"]) for i in range(2 * n)], 1)
    with torch.no_grad():
        out = actor(x)
    for i in range(2 * n):
        np.testing.assert_allclose(out[:, i].numpy(), z[f"plane{i % n}/y"], rtol=0, atol=2e-5)      # float32, another summation order
    games, ties, red, blue = (int(z[k]) for k in ("games", "ties", "red_wins", "blue_wins"))
    assert games == ties + red + blue and games >= 3000 and 0.75 < red / games < 0.90        # README.md:30: "~80%"
    assert z["calls_per_game"].min() >= 1 and z["calls_per_game"].max() <= 141               # a 2v2 game ends at the latest on call 141


def test_value_head_packs_like_an_actor_with_one_output():
    """A value head is a StackedActor with ONE output per agent: pack() pads its head to the 4-wide rows the kernels read (columns
    1-3 and their biases zero), and forward(squash=False) returns the raw head."""
    from deep_rl_battlespace_amd.rollout import StackedActor
    torch.manual_seed(2)
    A, D = 4, 8
    critic, actor = StackedActor(A, D, 1), StackedActor(A, D, 4)
    with torch.no_grad():
        actor.w3[:, :, 0] = critic.w3[:, :, 0]; actor.b3[:, :, 0] = critic.b3[:, :, 0]
        actor.w3[:, :, 1:] = 0; actor.b3[:, :, 1:] = 0
        for k in ("w1", "b1", "g1", "h1", "w2", "b2", "g2", "h2"):
            getattr(actor, k).copy_(getattr(critic, k))
    assert torch.equal(critic.pack(), actor.pack())
    obs = torch.rand(16, A, D) * 2 - 1
    with torch.no_grad():
        assert torch.allclose(critic(obs, squash=False)[..., 0], actor(obs, squash=False)[..., 0])
        assert torch.allclose(torch.tanh(critic(obs, squash=False)), critic(obs))


def test_roofline_claim_and_live_aware_bytes():
    """bench.py's roofline bookkeeping: the claimed fraction is the smaller of the contract fraction and the one on measured traffic,
    a contract fraction above 1 is never printed, and the live-bullet-aware byte count of this layout sits between the API-only bound
    and the 12-slot contract formula."""
    import bench
    assert bench.claim(0.59, 0.29) == (0.29, 0.59) and bench.claim(0.3, 0.5) == (0.3, 0.3)
    assert bench.claim(1.15, 0.21) == (0.21, None) and bench.claim(0.45, None) == (0.45, 0.45)
    for n in (1, 4):
        lo, hi = bench.b_io(n), bench.b_alg(n)
        assert lo < bench.b_live(n, 0.57, 0.25) < hi
        assert abs(bench.b_live(n, 1.0, 0.0) - bench.b_live(n, 0.0, 0.0) - 16.0) < 1e-9      # layout v2: a pool entry is read and rewritten whole, 8 B each way
    assert abs(bench.b_alg(1) - 260.0) < 1e-9 and abs(bench.b_alg(4) - 289.25) < 1e-9      # (SURVEY.md section 8d rounds it to 289.3)


def test_first_games_tally_counts_whole_games_only():
    """rollout._FirstGamesTally (the evaluation workload's sampling, evaluate.py:52: N whole games one after another): per slot the
    counters are frozen when its k-th game ends -- later games of fast slots and the unfinished games of slow ones do not enter, so the
    tally is not tilted toward short games as `stop everybody now and count` is."""
    from deep_rl_battlespace_amd.rollout import _FirstGamesTally

    class FakeEnv:
        n_envs = 3

        def __init__(self):
            self.c = np.zeros((3, 4), np.int32)

        def counters(self):
            return self.c.copy()
    env = FakeEnv()
    env.c[:] = [[5, 1, 3, 1], [2, 0, 2, 0], [0, 0, 0, 0]]                   # counters persist across rollouts: the tally starts from here
    t = _FirstGamesTally(env, 2)
    assert not t.update(env)
    env.c[0] += [1, 0, 1, 0]                                                # slot 0: a quick red win
    assert not t.update(env)
    env.c[0] += [1, 0, 1, 0]; env.c[1] += [1, 1, 0, 0]                      # slot 0 reaches 2 (red, red); slot 1: a tie
    assert not t.update(env)
    env.c[0] += [3, 0, 3, 0]; env.c[1] += [1, 0, 0, 1]                      # slot 0 races on (ignored); slot 1 reaches 2 (tie, blue)
    assert not t.update(env)
    env.c[2] += [2, 2, 0, 0]                                                # the slow slot: two time-limit ties, at last
    env.c[0] += [4, 0, 4, 0]
    assert t.update(env)
    assert t.totals().tolist() == [6, 3, 2, 1]                              # 2 games per slot: red+red, tie+blue, tie+tie
