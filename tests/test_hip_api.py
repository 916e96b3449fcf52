"""Boundary behaviour of the batched parallel_env surface (GPU): dict API, masked reset, checkpoint, observe(), argument
errors, inert-after-done semantics inside a batch, counters."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _env(**kw):
    import deep_rl_battlespace_amd as bsx
    return bsx.parallel_env(**kw)


def _acts(E, A, seed, p_shoot=0.5):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    a = torch.randint(0, 4, (E, A), generator=g, device="cuda", dtype=torch.int32)
    return torch.where(torch.rand((E, A), generator=g, device="cuda") < p_shoot, torch.ones_like(a), a)


def test_dict_api_is_a_view_of_the_batched_tensors():
    E, n = 300, 2
    a, b = _env(n_agents=n, n_envs=E, seed=3), _env(n_agents=n, n_envs=E, seed=3)
    oa, ob = a.reset(), b.reset()
    assert set(oa) == set(a.possible_agents) and oa["plane1"].shape == (E, a.obs_size) and oa["plane1"].is_cuda
    for t in range(40):
        act = _acts(E, 2 * n, t)
        obs, rew, dones, infos = a.step({ag: act[:, i].to(torch.int64) for i, ag in enumerate(a.possible_agents)})
        o2, r2, d2 = b.step_batch(act)
        for i, ag in enumerate(a.possible_agents):
            assert torch.equal(obs[ag], o2[:, i]) and torch.equal(rew[ag], r2[:, i]) and torch.equal(dones[ag], d2[:, i])
        assert infos == {ag: {} for ag in a.possible_agents}
    assert torch.equal(a.observe("plane2"), o2[:, 2])                      # observe() recomputes the same rows
    assert a.env_done.dtype == torch.bool and a.winner.dtype == torch.uint8 and a.agents == a.possible_agents


def test_copy_option_returns_tensors_that_survive_the_next_call():
    env = _env(n_agents=1, n_envs=64, seed=2); env.reset()
    a = torch.ones((64, 2), dtype=torch.int32, device="cuda")
    o1, r1, d1 = env.step_batch(a, copy=True)
    keep = (o1.clone(), r1.clone(), d1.clone())
    o2, _, _ = env.step_batch(a)
    assert o1.data_ptr() != o2.data_ptr() and torch.equal(o1, keep[0]) and torch.equal(r1, keep[1]) and torch.equal(d1, keep[2])
    od, _, _, _ = env.step({f"plane{i}": a[:, i] for i in range(2)}, copy=True)
    assert od["plane0"].data_ptr() != env._obs.data_ptr()


def test_inert_after_done_inside_a_batch_and_masked_reset():
    """auto_reset=False: finished games ignore step() (battle_env.py:303-306) while the others keep playing; a masked
    reset re-spawns only the selected games and keeps every counter (battle_env.py:246-279)."""
    E, n = 2000, 1
    env = _env(n_agents=n, n_envs=E, seed=11)
    env.reset()
    for t in range(125):
        obs, rew, done = env.step_batch(_acts(E, 2, 100 + t, p_shoot=0.7))
    assert bool(env.env_done.all())                                        # the time-limit tie caught every game by call 121
    st0 = env.export_state()
    obs, rew, done = env.step_batch(_acts(E, 2, 999))
    st1 = env.export_state()
    assert all(torch.equal(st0[k], st1[k]) for k in st0) and float(rew.abs().max()) == 0.0 and bool(done.all())
    c0 = env.counters()
    assert (c0[:, 0] >= 1).all() and (c0[:, 0] == c0[:, 1] + c0[:, 2] + c0[:, 3]).all()
    mask = torch.zeros(E, dtype=torch.bool, device="cuda"); mask[::3] = True
    env.reset(mask=mask)
    assert torch.equal(env.env_done, ~mask)
    st2 = env.export_state()
    assert bool((st2["tick"][mask] == 0).all()) and torch.equal(st2["tick"][~mask], st1["tick"][~mask])
    assert np.array_equal(env.counters(), c0)                              # win / tie counters persist across resets
    obs, rew, done = env.step_batch(_acts(E, 2, 5))
    st3 = env.export_state()
    assert bool((st3["tick"][mask] == 1).all()) and torch.equal(st3["px"][~mask], st1["px"][~mask])


def test_checkpoint_roundtrip_replays_the_same_games():
    E, n = 512, 2
    env = _env(n_agents=n, n_envs=E, seed=8, auto_reset=True); env.reset()
    for t in range(30):
        env.step_batch(_acts(E, 4, t))
    sd = env.state_dict()
    ref = [tuple(x.clone() for x in env.step_batch(_acts(E, 4, 100 + t))) for t in range(40)]
    env.load_state_dict(sd)
    for t in range(40):
        o, r, d = env.step_batch(_acts(E, 4, 100 + t))
        assert torch.equal(o, ref[t][0]) and torch.equal(r, ref[t][1]) and torch.equal(d, ref[t][2])


def test_bad_arguments_raise_before_launch():
    env = _env(n_agents=2, n_envs=16)
    env.reset()
    with pytest.raises(ValueError):
        env.step_batch(torch.zeros((16, 3), dtype=torch.int32, device="cuda"))            # wrong agent count
    with pytest.raises(TypeError):
        env.step_batch(torch.zeros((16, 4), dtype=torch.float32, device="cuda"))          # float indices
    with pytest.raises(ValueError):
        env.step_batch(torch.zeros((16, 4, 5), dtype=torch.float32, device="cuda"))       # score vectors must be 4 wide
    with pytest.raises(ValueError):
        env.reset(spawn=np.zeros((16, 5), np.int32))
    with pytest.raises(ValueError):
        _env(n_agents=0)
    with pytest.raises(ValueError):
        _env(n_agents=1, n_envs=4, auto_reset=True, rng="python")
    c = _env(n_agents=1, n_envs=4, continuous_actions=True); c.reset()
    with pytest.raises(ValueError):
        c.step_batch(torch.zeros((4, 2), dtype=torch.float32, device="cuda"))
    # int64 indices (torch's default) and CPU tensors are accepted and converted
    env.step_batch(torch.zeros((16, 4), dtype=torch.int64))


def test_empty_call_ties_every_running_game_of_the_batch():
    E = 64
    env = _env(n_agents=1, n_envs=E); env.reset()
    env.step_batch(_acts(E, 2, 1))
    obs, rew, dones, _ = env.step({})
    assert bool(env.env_done.all()) and bool((env.winner == 3).all()) and float(rew["plane0"].abs().max()) == 0.0
    assert (env.counters()[:, :2] == 1).all()


def test_bench_prints_one_json_line_with_the_contract_fields():
    """bench.py's contract with the driver: ONE JSON line on stdout with the agreed keys (a short run, no CPU baseline)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "300", "--warmup", "30", "--no-cpu-baseline",
                          "--no-other-workloads", "--no-live-traffic"], capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "agent-steps/sec" and d["n_gpus"] == 1 and d["steps"] == 300 and d["warmup"] == 30
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # "hbm" only where the measured traffic runs at half of the peak or more; 65 536 x 1v1 is bound by instruction issue and the kernel boundary
    assert r["bound"] == ("hbm" if (r["frac_on_traffic"] or 0) >= 0.5 else "issue/latency")
    # the LAST key is the compact numeric summary of every BASELINE.json config (it must survive a truncated record of the line)
    assert list(d)[-1] == "baseline_configs" and d["baseline_configs"]["C2"]["agent_steps_per_s"] == round(d["value"])
    assert len(json.dumps(d["baseline_configs"])) < 1800                       # (the driver keeps the last 2 000 characters of the line)
    # the figure to quote is the smaller of the contract fraction and the one on measured traffic; no fraction above 1 anywhere
    assert r["frac_claimed"] == min(v for v in (r["frac"], r["frac_on_traffic"]) if v is not None) and r["frac_claimed"] <= 1.0
    assert r["live_aware_bytes_per_launch"] < r["algorithmic_bytes_per_launch"]
    assert d["value"] > 1e9 and abs(d["value"] - 65536 * 2 * 300 / (d["ms_per_step"] * 1e-3 * 300)) / d["value"] < 1e-3


def test_bench_measures_the_headline_hbm_traffic_live():
    """roofline.traffic is measured by the run itself: bench.py re-runs its workload in two child passes under `rocprofv3
    --kernel-trace --pmc` (FETCH_SIZE, WRITE_SIZE separately) and reports (2 x fetch + write) KiB per launch, with the source
    spelled out; frac_on_traffic follows from it."""
    import json, os, shutil, subprocess, sys
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "200", "--warmup", "20", "--repeats", "2", "--no-cpu-baseline",
                          "--no-other-workloads"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    r = d["roofline"]
    assert r["traffic_detail"] is not None and r["traffic_source"].startswith("this run"), r["traffic_source"]
    assert 5e6 < r["traffic"] < r["algorithmic_bytes_per_launch"]          # sparse bullets: less than the 12-slot algorithmic count
    assert abs(r["traffic"] - (2 * r["traffic_detail"]["fetch_size_kib_raw"] + r["traffic_detail"]["write_size_kib_raw"]) * 1024) < 2048
    assert abs(r["frac_on_traffic"] - r["traffic"] / (r["avg_launch_us"] * 1e-6) / 1e9 / r["peak"]) < 1e-3 and r["frac_on_traffic"] < r["frac"]
    assert r["frac_claimed"] == r["frac_on_traffic"]



def test_bench_bullet_heavy_workload_really_holds_many_bullets():
    """`--action-mix dense`: the recorded closed-loop keep-shooting play (bench.py dense_policy) is replayed from the rewound state as
    one graph; the line reports the MEASURED mean of live bullets per plane over the recorded calls -- around 7, against 0.6 under
    uniform random play and 1.7 under the old "action 1 every tick" stress."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--action-mix", "dense", "--steps", "150", "--warmup", "10", "--repeats", "2",
                          "--ramp-ms", "0", "--envs-per-gpu", "16384", "--no-cpu-baseline", "--no-other-workloads", "--no-live-traffic"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["roofline"]["live_bullets_per_agent"] > 6.0, d["roofline"]["live_bullets_per_agent"]
    assert "keep-shooting" in d["config"]["workload"] and d["config"]["graph_len"] == 150


def test_dropin_io_lives_in_pinned_host_memory_and_custom_rows_still_reach_the_device():
    """Drop-in mode: the kernels read the action / random() block from and write obs / rewards / dones to pinned host memory
    (no staging copies); a tensor that is neither device nor pinned memory must never be handed to a kernel -- the scripted
    team uploads such rows first (instinct/team.py:10-15 surface)."""
    import random as _r
    from deep_rl_battlespace_amd.instinct import Team
    _r.seed(5)
    env = _env(n_agents=2)
    for t in (env._obs, env._rew, env._done, env._env_done, env._winner):
        assert not t.is_cuda and t.is_pinned()
    obs = env.reset()
    blue = Team(env.possible_blue, env.possible_red, env)
    a1 = blue.choose_actions(obs)                                   # dict rows -> uploaded copy
    a2 = blue.choose_actions()                                      # the env's own (pinned) rows
    assert a1 == a2
    acts = {a: 0 for a in env.possible_red}; acts.update(a1)
    o, r, d, _ = env.step(acts)
    assert set(o) == set(env.possible_agents) and all(v.dtype == np.float32 and v.shape == (env.obs_size,) for v in o.values())
    np.testing.assert_array_equal(env.observe("plane0"), o["plane0"])


def test_dropin_masked_reset_and_auto_reset_contract():
    """Drop-in mode keeps its outputs in pinned HOST memory: reset(mask=...) must clear those rows (not index them with a device
    mask), and auto_reset -- whose in-kernel re-spawn the host mirrors of `agents` / `dones` / `env_done` would never see -- is
    refused there."""
    import random as _r
    _r.seed(3)
    env = _env(n_agents=1)
    env.reset()
    env.step({})                                                    # an empty call ties the game (battle_env.py:309-313)
    assert env.env_done and env.winner == "tie"
    obs = env.reset(mask=[True])
    assert not env.env_done and env.winner == "none" and set(obs) == set(env.possible_agents)
    assert int(env._env_done[0]) == 0 and int(env._winner[0]) == 0 and not bool(env._done.any())
    o, r, d, _ = env.step({a: 0 for a in env.possible_agents})
    assert not any(d.values()) and env.agents == env.possible_agents
    with pytest.raises(ValueError):
        _env(n_agents=1, auto_reset=True, rng="philox")
    with pytest.raises(ValueError):
        env.step_many(torch.zeros((2, 1, 2), dtype=torch.int32, device="cuda"))


def test_a_state_block_belongs_to_one_action_family_and_the_abi_says_so():
    """ABI 14 keeps discrete headings as whole degrees inside the plane record and continuous ones as float64 beside it: a discrete
    call on a block the continuous kernels advanced would read truncated headings.  The C ABI refuses it (BSX_E_FAMILY, host-side, no
    device work) until every game has been reset (battle_env.py:73: a parallel_env has one action mode for its life)."""
    from deep_rl_battlespace_amd import _lib
    E = 128
    env = _env(n_agents=1, n_envs=E, seed=5, continuous_actions=True); env.reset()
    act = torch.rand((E, 2, 3), device="cuda") * 2 - 1
    env.step_batch(act)                                                      # the block is now a continuous one
    lib = _lib.load()
    ia = torch.zeros((E, 2), dtype=torch.int32, device="cuda")

    def discrete_call():
        return lib.bsx_step_discrete(env._p_state, E, 1, ia.data_ptr(), 0, None, env._p_obs, env._p_rew, env._p_done, env._p_env_done,
                                     env._p_winner, env._cfg_ref, env._base_flags, env.seed, env.env_offset, env._stream())
    before = env.state_dict()["state"].clone()
    assert discrete_call() == -3                                             # BSX_E_FAMILY
    with pytest.raises(ValueError, match="BSX_E_FAMILY"):
        _lib.check(-3, "bsx_step_discrete")
    torch.cuda.synchronize()
    assert torch.equal(env.state_dict()["state"], before)                    # refused before any launch
    env.step_batch(act)                                                      # its own family goes on
    env.reset(mask=torch.arange(E, device="cuda") < 5)                       # a PARTIAL reset leaves fractional headings elsewhere
    assert discrete_call() == -3
    env.reset()                                                              # every game re-spawned: whole degrees, either family may follow
    assert discrete_call() == 0
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="BSX_E_FAMILY"):                    # ... and now the block is a discrete one
        env.step_batch(act)
    # bsx_state_release (env.close()): the host-side claim goes with the block -- an allocation that lands on the same address starts unclaimed
    env.reset()
    env.step_batch(act)                                                      # continuous again after the full reset
    assert discrete_call() == -3
    env.close()
    env.reset()
    assert discrete_call() == 0                                              # (the address is unclaimed; the full reset made every heading a whole degree)
    assert lib.bsx_state_release(None) == -1
