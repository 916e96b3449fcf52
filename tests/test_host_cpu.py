"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the header declares, argument
validation happens before any device work, host-side draw order equals the reference's, sharding partitions exactly,
and the multi-rank path (world_size 2, gloo) reduces counters correctly.  No kernel is launched here."""
import ctypes
import json
import os
import random
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from deep_rl_battlespace_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "battlespace_hip.h")).read()
    declared = sorted(set(re.findall(r"^\s*int\s+(bsx_\w+)\s*\(", hdr, flags=re.M)))
    assert declared == sorted(_lib().SIGNATURES), "binding and header disagree on the entry points"
    lib = _lib().load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.bsx_abi_version() == int(re.search(r"#define BSX_ABI_VERSION (\d+)", hdr).group(1))


def test_state_bytes_and_tie_tick_need_no_gpu():
    L = _lib(); lib = L.load()
    sz = ctypes.c_size_t()
    assert lib.bsx_state_bytes(65536, 1, ctypes.byref(sz)) == 0
    per_agent = sz.value / (65536 * 2)
    assert 300 < per_agent < 420                      # 16 plane + 48 bullet xy + 192 displacement + 96 heading + env share
    assert lib.bsx_state_bytes(0, 1, ctypes.byref(sz)) == -1 and lib.bsx_state_bytes(8, 17, ctypes.byref(sz)) == -1
    assert lib.bsx_state_bytes(1 << 30, 1, ctypes.byref(sz)) == 0 and sz.value > (1 << 30) * 600     # BSX_MAX_E: no overflow
    assert lib.bsx_state_bytes((1 << 30) + 1, 1, ctypes.byref(sz)) == -1
    assert [lib.bsx_tie_tick(n) for n in (1, 2, 3, 4, 5, 8)] == [121, 141, 161, 181, 200, 260]


def test_bad_arguments_are_rejected_before_any_launch():
    L = _lib(); lib = L.load()
    cfg = L.BsxRewards(100, 10, -1, -5, -20)
    assert lib.bsx_step_discrete(None, 4, 1, None, 0, None, None, None, None, None, None, ctypes.byref(cfg), 0, 0, 0, None) == -1
    assert lib.bsx_reset(None, 4, 1, None, None, 0, 0, 0, None, None) == -1
    assert lib.bsx_observe(None, 4, 1, None, None) == -1
    assert lib.bsx_state_init(ctypes.c_void_p(4096 + 8), 4, 1, None) == -2      # misaligned state base
    assert lib.bsx_state_release(None) == -1 and lib.bsx_state_release(ctypes.c_void_p(4096)) == 0   # (host-side bookkeeping only: no device call)
    # the multi-tick and rollout entry points: T out of range, team sizes the one-launch rollout is not built for, unknown precision
    ok = ctypes.c_void_p(4096)
    assert lib.bsx_step_many_discrete(ok, 4, 1, 0, ok, 0, None, ok, ok, ok, None, None, None, ctypes.byref(cfg), 0, 1, 0, 0, None) == -1
    assert lib.bsx_step_many_discrete(ok, 4, 1, 70000, ok, 0, None, ok, ok, ok, None, None, None, ctypes.byref(cfg), 0, 1, 0, 0, None) == -1
    assert lib.bsx_step_many_continuous(ok, 4, 1, 5, None, 0, None, ok, ok, ok, None, None, None, ctypes.byref(cfg), 0, 1, 0, 0, None) == -1
    # a game range: whole 256-game blocks from `first`, inside the batch
    rng = lambda first, count: lib.bsx_step_discrete_range(ok, 1024, 1, first, count, ok, 0, None, ok, ok, ok, None, None, ctypes.byref(cfg), 0, 0, 0, None)   # noqa: E731
    assert rng(100, 50) == -1 and rng(512, 513) == -1 and rng(0, 0) == -1 and rng(-256, 256) == -1 and rng(1024, 1) == -1
    assert lib.bsx_step_continuous_range(ok, 1024, 1, 256, 769, ok, 0, None, ok, ok, ok, None, None, ctypes.byref(cfg), 0, 0, 0, None) == -1
    assert lib.bsx_rollout_discrete(ok, 4, 5, 8, ok, 0, -1, ok, ok, ok, ok, None, None, None, ctypes.byref(cfg), 0, None, 0, 0, None, 0, 0, None) == -1
    assert lib.bsx_rollout_discrete(ok, 4, 1, 8, ok, 7, -1, ok, ok, ok, ok, None, None, None, ctypes.byref(cfg), 0, None, 0, 0, None, 0, 0, None) == -1
    assert lib.bsx_rollout_discrete(ok, 4, 1, 8, ok, 0, 2, ok, ok, ok, ok, None, None, None, ctypes.byref(cfg), 0, None, 0, 0, None, 0, 0, None) == -1
    assert lib.bsx_rollout_continuous(ok, 4, 5, 8, ok, 0, -1, 0, ok, ok, ok, ok, None, None, None, ctypes.byref(cfg), 0, None, 0, 0, None, 0, 0, None) == -1
    assert lib.bsx_rollout_continuous(ok, 4, 1, 8, ok, 0, 2, 0, ok, ok, ok, ok, None, None, None, ctypes.byref(cfg), 0, None, 0, 0, None, 0, 0, None) == -1   # scripted_team out of range
    nz = L.BsxActorNoise(0.1, 0.0, 0.15, 0.2, 0.0, None, None, ctypes.c_void_p(4096))     # injected normals are per call: not for a T-tick launch
    assert lib.bsx_rollout_continuous(ok, 4, 1, 8, ok, 0, -1, 0, ok, ok, ok, ok, None, None, None, ctypes.byref(cfg), 0, ctypes.byref(nz), 0, 0, None, 0, 0, None) == -1
    # the policy-gradient heads: a categorical head needs a temperature and discrete actions; a value head rides in the 1v1 one-launch kernel only
    nz = L.BsxActorNoise(); nz.sample_mode = 1
    assert lib.bsx_actor_forward(ok, ok, ok, 4, 1, 0, ctypes.byref(nz), 0, 0, None, 0, None) == -1                  # temperature 0
    nz.temperature = 1.0
    assert lib.bsx_rollout_continuous(ok, 4, 1, 8, ok, 0, -1, 0, ok, ok, ok, ok, None, None, None, ctypes.byref(cfg), 0, ctypes.byref(nz), 0, 0, None, 0, 0, None) == -1
    nz = L.BsxActorNoise(); nz.value_weights = 4096
    assert lib.bsx_actor_forward(ok, ok, ok, 4, 1, 0, ctypes.byref(nz), 0, 0, None, 0, None) == -1                  # value head without an output
    nz.value = 4096
    assert lib.bsx_rollout_discrete(ok, 4, 2, 8, ok, 0, -1, ok, ok, ok, ok, None, None, None, ctypes.byref(cfg), 0, ctypes.byref(nz), 0, 0, None, 0, 0, None) == -1
    dp = ctypes.c_void_p()
    assert lib.bsx_host_device_pointer(None, ctypes.byref(dp)) == -1
    assert lib.bsx_actor_forward(ok, ok, ok, 4, 1, 7, None, 0, 0, None, 0, None) == -1
    assert lib.bsx_actor_forward(ok, ok, ok, 4, 1, 0, None, 0, 0, None, -5, None) == -1   # negative env_offset
    assert lib.bsx_build_flags() == 0                                                      # the in-tree library is the product build


def test_env_refuses_to_run_without_gpu_or_library():
    import torch
    import deep_rl_battlespace_amd as bsx
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        bsx.parallel_env()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: no file of the product imports, loads or links anything under oracle/."""
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle)|libbattlespace_ref|oracle/|oracle\.", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "deep-rl-battlespace_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), os.path.join(dirpath, f)


def test_host_spawn_draw_order_is_the_references():
    """rng='python': the host draws spawns from the stdlib generator exactly as the oracle (hence the reference) does."""
    from deep_rl_battlespace_amd.envs.battle_env import draw_spawn
    from oracle import battlespace_ref as ref
    for n in (1, 2, 4):
        random.seed(50 + n)
        env = ref.RefEnv(n_agents=n)            # constructor draws once
        env.reset()
        want = [env.base_x[0], env.base_y[0], env.base_x[1], env.base_y[1]] + \
               [v for i in range(2 * n) for v in (env.px[i], env.py[i], env.pdir[i])]
        random.seed(50 + n)
        draw_spawn(n)
        assert draw_spawn(n) == want


def test_env_range_partitions_exactly():
    from deep_rl_battlespace_amd.sharding import env_range
    for total, world in ((524288, 8), (65536, 1), (10, 3), (7, 8), (0, 2)):
        spans = [env_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    assert [env_range(524288, r, 8) for r in (0, 7)] == [(0, 65536), (458752, 524288)]
    with pytest.raises(ValueError):
        env_range(8, 2, 2)


def test_chain_ranges_partition_the_batch_in_256_game_blocks():
    """sharding.chain_ranges: what capture_steps(chains=) hands to bsx_step_*_range -- contiguous, exhaustive, every first game a
    multiple of 256, never more ranges than blocks; "auto" by team size and batch size."""
    from deep_rl_battlespace_amd import sharding as sh
    for E in (1, 255, 256, 257, 1000, 4096, 65536, 65537, 1048576):
        for P in (1, 2, 3, 4, 7, 100):
            r = sh.chain_ranges(E, 2, P)
            assert r[0][0] == 0 and sum(c for _, c in r) == E and all(c > 0 and f % 256 == 0 for f, c in r)
            assert all(r[i][0] + r[i][1] == r[i + 1][0] for i in range(len(r) - 1)) and len(r) == min(P, -(-E // 256))
            assert max(c for _, c in r) - min(c for _, c in r[:-1] or r) <= 256 + 255     # balanced to a block (the last range takes the ragged rest)
    auto = lambda E, n: len(sh.chain_ranges(E, n, "auto"))   # noqa: E731
    assert auto(65536, 1) == 1 and auto(196608, 1) == 1 and auto(1048576, 1) == 2   # 1v1: from 1 M games -- below, a synchronised replay of the chained graph loses (profiles/r06_chains_1v1.json)
    assert auto(65536, 2) == 1 and auto(65536, 3) == 2 and auto(65536, 4) == 3 and auto(16384, 16) == 2     # (2v2: gains only when replays are queued back to back)
    assert auto(16384, 4) == 1 and auto(8192, 2) == 1 and auto(100, 8) == 1   # short launches: the branches cost more than they hide


def test_spaces_metadata():
    from deep_rl_battlespace_amd.spaces import Box, Discrete
    b = Box(np.ones(5, np.float32), -np.ones(5, np.float32))
    assert b.shape == (5,) and b.dtype == np.float32 and (b.low == 1).all() and (b.high == -1).all()
    assert b.sample().shape == (5,)
    d = Discrete(4)
    assert d.n == 4 and 0 <= d.sample() < 4


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from deep_rl_battlespace_amd import sharding
rank, world, _ = sharding.rank_world()
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=rank, world_size=world)
lo, hi = sharding.env_range(1000, rank, world)
mine = torch.tensor([hi - lo, rank + 1, 10 * (rank + 1), lo], dtype=torch.int64)
tot = sharding.reduce_counters(mine)
assert tot.tolist() == [1000, 3, 30, 500], tot.tolist()
assert mine.tolist() == [500, rank + 1, 10 * (rank + 1), lo]       # input untouched
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def test_two_rank_counter_reduction_over_gloo(tmp_path):
    """world_size 2 on CPU (gloo): each rank owns a contiguous env range; the only collective is the logging all-reduce."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = 29500 + os.getpid() % 2000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(port)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all(f"ok {r}" in outs[r] for r in range(2))


_PROBE_WORKER = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from deep_rl_battlespace_amd import sharding
rank, world, _ = sharding.rank_world()
case = sys.argv[2]
sharding._rccl_preflight = lambda device: None                     # no card here: the collective half is what is under test
def probe(device, world_size, timeout_s):
    g = dist.new_group(backend="gloo")                             # stands in for the RCCL group (its creation is itself collective)
    if case == "one_rank_fails":
        if rank == 1:
            raise RuntimeError("injected: ncclCommInitRank failed on this rank")
        time.sleep(3600)                                           # the other ranks sit in the collective that rank 1 never joins
    if case == "one_rank_hangs" and rank == 1:
        time.sleep(3600)                                           # nobody reports anything: only the bounded wait ends this
    return g
sharding._rccl_probe = probe
t0 = time.time()
group, backend, note, clean = sharding.init_timing_group("nccl", torch.device("cpu"), probe_wait_s=float(sys.argv[3]))
dt = time.time() - t0
t = torch.tensor([rank + 1.0])
dist.all_reduce(t, group=group)                                    # the group every rank was handed works, and it is the same one
assert t.item() == world * (world + 1) / 2
print(f"RESULT {rank} {backend} {int(clean)} {dt:.2f} {note}", flush=True)
dist.barrier()
os._exit(0) if not clean else dist.destroy_process_group()
'''


def _run_probe_case(tmp_path, case, wait_s, world=3):
    import socket
    script = tmp_path / "probe_worker.py"
    script.write_text(_PROBE_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, case, str(wait_s)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = {}
    for o in outs:
        f = [l for l in o.splitlines() if l.startswith("RESULT ")][0].split(" ", 5)
        res[int(f[1])] = dict(backend=f[2], clean=f[3] == "1", dt=float(f[4]), note=f[5])
    return res


def test_rccl_failure_on_one_rank_moves_every_rank_to_gloo_within_seconds(tmp_path):
    """bench.py's N > 1 process-group setup (sharding.init_timing_group): an RCCL-init failure injected on ONE rank while the other
    ranks sit in the collective probe -- every rank must come out on gloo, together and quickly (the failing rank publishes through
    the control store; nobody waits for a collective timeout), and say why."""
    res = _run_probe_case(tmp_path, "one_rank_fails", wait_s=120)
    assert {r["backend"] for r in res.values()} == {"gloo"}, res
    assert max(r["dt"] for r in res.values()) < 30, res                  # seconds, not the 120 s bound or RCCL's own timeout
    assert "injected" in res[1]["note"] and res[1]["clean"]
    assert all("peer rank reported a failure" in res[r]["note"] and not res[r]["clean"] for r in (0, 2)), res


def test_rccl_probe_that_never_returns_is_bounded_and_agreed(tmp_path):
    """A rank whose probe neither returns nor raises: the other ranks' probes pass, the stuck rank gives up after probe_wait_s, and the
    MIN all-reduce over gloo still puts ALL ranks on gloo -- never a mixture."""
    res = _run_probe_case(tmp_path, "one_rank_hangs", wait_s=5)
    assert {r["backend"] for r in res.values()} == {"gloo"}, res
    assert "no result within" in res[1]["note"] and not res[1]["clean"]
    # (the ranks whose probe passed hold a group nobody will use: they too leave without tearing it down -- clean is False)
    assert all("another rank" in res[r]["note"] and not res[r]["clean"] for r in (0, 2)), res


def test_rccl_probe_passing_everywhere_hands_out_the_probed_group(tmp_path):
    res = _run_probe_case(tmp_path, "all_pass", wait_s=60, world=2)
    assert all(r["backend"] == "nccl" and r["clean"] and r["note"] == "None" for r in res.values()), res


def test_state_block_size_follows_the_documented_layout():
    """bsx_state_bytes (a host function: no GPU needed) against DESIGN.md section 3 / include/battlespace_hip.h, layout v2 (ABI 14):
    heading table, 8-byte game records (constant + dynamic), 16-byte counters, 8-byte plane records + the float64 headings of the
    continuous kernels, one pool per wave block of 64 lanes (a count + 768 entries of 8 bytes), the two birth-tick rings."""
    import ctypes
    from deep_rl_battlespace_amd import _lib
    lib = _lib.load()
    assert lib.bsx_abi_version() == 14

    def expect(E, n):
        al = lambda v: (v + 255) & ~255
        A = 2 * n
        G = 2
        while G < A:
            G *= 2
        NB = -(-E // (64 // G))
        EA = E * A
        o = 0
        for nbytes in (361 * 16, E * 8, E * 8, E * 16, EA * 8, EA * 8, NB * 4, NB * 768 * 8, 12 * EA * 8, 12 * EA * 16):
            o = al(o + nbytes)
        return o
    for E, n in ((1, 1), (31, 1), (65536, 1), (65536, 4), (1000, 3), (7, 16), (1 << 20, 1), (33, 5)):
        got = ctypes.c_size_t()
        assert lib.bsx_state_bytes(E, n, ctypes.byref(got)) == 0
        assert got.value == expect(E, n), (E, n, got.value, expect(E, n))
    assert lib.bsx_state_bytes(0, 1, ctypes.byref(got)) == -1 and lib.bsx_state_bytes(8, 17, ctypes.byref(got)) == -1
    assert [lib.bsx_tie_tick(n) for n in (1, 2, 3, 4)] == [121, 141, 161, 181] and lib.bsx_tie_tick(16) == 420 < 512   # (the game record holds the clock in 9 bits)


def test_render_frame_from_exported_state(tmp_path):
    """f-4: one game's exported state rasterised on the host (no pygame): shapes, colours, dead plane hollow."""
    from deep_rl_battlespace_amd import render
    st = dict(px=np.array([200, 900]), py=np.array([300, 500]), pdir=np.array([0.0, 180.0]), php=np.array([4, 0]),
              base_xy=np.array([100, 400, 1100, 400]), bhp=np.array([5, 5]),
              bl_live=np.zeros((2, 12), bool), bl_x=np.zeros((2, 12), int), bl_y=np.zeros((2, 12), int))
    st["bl_live"][0, 3] = True; st["bl_x"][0, 3] = 400; st["bl_y"][0, 3] = 300
    img = render.frame_from_state(st, 1)
    assert img.shape == (800, 1200, 3) and img.dtype == np.uint8
    assert tuple(img[310, 190]) == render.RED and tuple(img[300, 400]) == render.RED       # plane body (off the heading tick), bullet
    assert tuple(img[300, 215]) == render.BLACK                                             # heading tick of plane0 points +x
    assert tuple(img[500, 900]) != render.BLUE and tuple(img[500 - 24, 900]) == render.BLUE  # dead plane: outline only
    assert tuple(img[400, 1100]) == render.BLUE and tuple(img[400, 100]) == render.RED       # bases
    render.save_ppm(tmp_path / "f.ppm", img)
    assert (tmp_path / "f.ppm").stat().st_size == 800 * 1200 * 3 + len(b"P6\n1200 800\n255\n")


def test_binding_refuses_a_diagnostic_build(tmp_path):
    """Timing-ablation / stamped builds give WRONG results; they are built as separate variant files (tools/build_variant.py),
    report themselves through bsx_build_flags(), and the binding loads one only when BSX_ALLOW_DIAG=1 says so -- the product
    library in the tree is never replaced by one."""
    prod = os.path.join(ROOT, "deep-rl-battlespace_amd", "csrc", "libbattlespace_hip.so")
    before = os.path.getmtime(prod)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "citest_diag2", "-DBSX_DIAG=2"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    variant = out.stdout.strip().splitlines()[-1]
    try:
        assert os.path.getmtime(prod) == before and os.path.dirname(variant).endswith("variants")
        code = "import sys; sys.path.insert(0, %r); from deep_rl_battlespace_amd import _lib; print(_lib.load().bsx_build_flags())" % ROOT
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BSX_LIB_PATH=variant), capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "diagnostic build" in r.stderr
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BSX_LIB_PATH=variant, BSX_ALLOW_DIAG="1"), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout.strip() == "2", r.stderr[-500:]
    finally:
        os.remove(variant)


def test_bench_action_table_is_shard_invariant_and_uniform():
    """bench.py keys its synthetic actions by the GLOBAL game index, so `--gpus N` shards play the games of the unsplit job."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    whole = bench.hashed_bits(40, 0, 3000, 2, 1, 1234, "cpu")[..., 0] >> 29
    part = bench.hashed_bits(40, 1000, 2000, 2, 1, 1234, "cpu")[..., 0] >> 29
    assert torch.equal(whole[:, 1000:2000], part)
    cnt = torch.bincount(whole.flatten(), minlength=4).float()
    assert cnt.numel() == 4 and float((cnt / cnt.sum() - 0.25).abs().max()) < 0.01
    f = bench.hashed_bits(8, 0, 500, 2, 3, 1234, "cpu").float() * (2.0 / 2147483648.0) - 1.0
    assert float(f.min()) >= -1.0 and float(f.max()) < 1.0 and abs(float(f.mean())) < 0.02


def test_bench_multi_gpu_launch_without_gpus_fails_loudly():
    """`python bench.py --gpus 2` launched plainly starts its own ranks; on a box with fewer cards it must say so, not hang."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two cards visible: the launch would really start two ranks")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")})
    assert r.returncode != 0 and "--rehearse-on-device0" in (r.stderr + r.stdout)


def test_rank_affinity_blocks_are_disjoint_and_follow_the_cards_numa_nodes(tmp_path):
    """bench.py --gpus N pins every rank to its own host cores before its first GPU call (sharding.pin_rank_to_its_cores): the table
    is a pure function of (world, allowed cores, the cards' NUMA nodes), so the ranks agree on it without talking."""
    from deep_rl_battlespace_amd import sharding
    allowed = list(range(256))
    blocks, how = sharding.affinity_blocks(8, allowed)
    assert how == "plain" and [len(b) for b in blocks] == [32] * 8 and blocks[3] == list(range(96, 128))
    assert len(set().union(*map(set, blocks))) == 256                                     # disjoint and complete
    # two sockets: cards 0-3 on node 0 (cores 0-63 + SMT siblings 128-191), cards 4-7 on node 1
    node_cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    blocks, how = sharding.affinity_blocks(8, allowed, [0, 0, 0, 0, 1, 1, 1, 1], node_cpus)
    assert how == "numa" and all(len(b) == 32 for b in blocks)
    assert all(set(blocks[r]) <= set(node_cpus[0 if r < 4 else 1]) for r in range(8))
    assert sum(len(b) for b in blocks) == len(set().union(*map(set, blocks))) == 256
    # interleaved card order, an uneven node, a restricted allowed set (a container's cpuset): still disjoint, still inside the node
    blocks, how = sharding.affinity_blocks(5, range(10, 90), [1, 0, 1, 0, 0], {0: list(range(0, 48)), 1: list(range(48, 96))})
    assert how == "numa" and sum(len(b) for b in blocks) == len(set().union(*map(set, blocks)))
    assert set(blocks[0]) | set(blocks[2]) <= set(range(48, 90)) and set(blocks[1]) | set(blocks[3]) | set(blocks[4]) <= set(range(10, 48))
    # a card without a node, or a node with fewer cores than ranks on it: the plain split
    assert sharding.affinity_blocks(4, allowed, [0, None, 0, 0], node_cpus)[1] == "plain"
    assert sharding.affinity_blocks(4, range(8), [0, 0, 0, 0], {0: [0, 1]})[1] == "plain"
    # fewer cores than ranks: nothing to separate
    assert sharding.affinity_blocks(8, range(4))[0] == [[0, 1, 2, 3]] * 8
    assert sharding._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    # the sysfs reader, on a fake tree: two AMD cards (PCI order decides the rank) and a foreign one
    for i, (pci, vendor, node) in enumerate((("0000:c1:00.0", "0x1002", 1), ("0000:05:00.0", "0x1002", 0), ("0000:03:00.0", "0x1a03", 0))):
        d = tmp_path / "devices" / pci
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n"); (d / "numa_node").write_text(f"{node}\n")
        c = tmp_path / "class" / "drm" / f"card{i}"
        c.mkdir(parents=True)
        os.symlink(d, c / "device")
    assert sharding.card_numa_nodes(str(tmp_path)) == [0, 1]
    # and the pinning itself, in child processes (this process keeps its cores): two ranks of a 2-rank job end up on disjoint sets
    code = ("import os, sys, json; sys.path.insert(0, %r); from deep_rl_battlespace_amd import sharding; "
            "r = sharding.pin_rank_to_its_cores(int(sys.argv[1]), 2, sysfs=%r); "
            "print(json.dumps([r, sorted(os.sched_getaffinity(0))]))" % (ROOT, str(tmp_path / "nothing_here")))
    got = [json.loads(subprocess.run([sys.executable, "-c", code, str(r)], capture_output=True, text=True, timeout=120, check=True).stdout) for r in range(2)]
    if len(os.sched_getaffinity(0)) >= 2:
        assert all(g[0]["cores"] == g[1] and g[0]["how"] == "plain" for g in got) and not set(got[0][1]) & set(got[1][1])


def test_bench_line_helpers_survive_failed_variants_and_describe_a_multi_rank_job():
    """(1) ADVICE r4: a rollout variant that failed is recorded as {"error": ...}; the compact summary (the line's last key) must turn
    it into null instead of raising -- the line must always be printed.  (2) the N > 1 fields computed from per-rank samples."""
    sys.path.insert(0, ROOT)
    import bench
    ev_tag = "reference evaluation workload (evaluate.py:32-76): 2v2, shipped checkpoints vs scripted instinct team"
    out = {"value": 1.8e10, "policy_rollouts": {"variants": {ev_tag: {"error": "RuntimeError: boom"},
                                                              "graph of 2 kernels per tick, both teams on actors": {"error": "x"},
                                                              "one launch for all ticks, both teams on actors": {"agent_steps_per_s": 6.4e9, "us_per_tick": 20.3,
                                                                                                                 "roofline": {"frac_mfma": 0.38, "bound": "mfma"}}}},
           "other_workloads": {"configs[2] 65536 x 4v4": {"error": "y"}}, "multi_tick_launch": None, "cpu_baseline": None, "drop_in_one_game": {"error": "z"}}
    s = bench.baseline_summary(out, 1, 65536, 1, 6.1e-3, 0.22)
    assert s["eval_2v2"] is None and s["C5_graph"] is None and s["C3"] is None and s["C2"]["us"] == 6.1
    assert s["C5_one_launch"]["agent_steps_per_s"] == 6400000000 and s["C5_one_launch"]["frac"] == 0.38
    json.dumps(s)
    ok = dict(out["policy_rollouts"]["variants"])
    ok[ev_tag] = {"agent_steps_per_s": 7e9, "us_per_tick": 37.4, "win_rate_red": 0.84612, "win_rate_red_stale_first_obs": 0.8301}
    out["policy_rollouts"]["variants"] = ok
    s = bench.baseline_summary(out, 1, 65536, 1, 6.1e-3, 0.22)
    assert s["eval_2v2"] == {"agent_steps_per_s": 7000000000, "us": 37.4, "red_win_rate": 0.8461, "red_win_rate_stale_first_obs": 0.8301}
    # (2) four ranks, five samples each: rank 2's host was descheduled in its barriered blocks -- value pays, value_device does not
    E, A, K = 65536, 2, 20
    wall = lambda us: [us * K * 1e-6] * 5                                                    # noqa: E731
    per_rank = {"head_walls": [wall(7.0), wall(7.1), wall(12.0), wall(7.0)], "head_kms": [[6.1e-3] * 5] * 4,
                "head_solo_walls": [wall(7.0)] * 4, "head_solo_kms": [[6.1e-3] * 5] * 4}
    value = 4 * E * A * K / (12.0 * K * 1e-6)
    f = bench.multi_rank_fields(per_rank, 4, E, A, K, value, "gloo")
    assert f["per_rank"]["ms_per_step"] == [0.007, 0.0071, 0.012, 0.007] and f["per_rank"]["ms_per_step_min_median_max"]["max"] == 0.012
    assert abs(f["value_device"] - 4 * E * A / 6.1e-6) < 1 and f["process_group"] == "gloo" and "rccl_version" in f
    assert abs(f["scaling_efficiency"] - 7.0 / 12.0) < 1e-3 and f["scaling_efficiency_device"] == 1.0
    assert f["single_shard_reference"]["ms_per_step"] == 0.007
    assert bench._ranges([0, 1, 2, 3, 8, 9, 11]) == "0-3,8-9,11" and bench._ranges(None) is None


def test_no_kernel_of_the_product_build_keeps_registers_in_scratch_memory():
    """round 4's `amdgpu_waves_per_eu(4)` on the multi-tick 2v2 kernels left three of them with 2 ... 7 vector
    registers in scratch memory, touched every tick, and nothing noticed.  The product's translation units are compiled to assembly
    here (hipcc cross-compiles without a GPU; tools/isa_meta.py) and every kernel's metadata must say: no spilled VGPR, no private
    segment.  The committed table (profiles/r06_isa_meta.json) must be the one this tree compiles to."""
    with __import__("tempfile").TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "meta.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_meta.py"), "--json", out], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        meta = json.load(open(out))
    assert len(meta) >= 80 and sum(1 for k in meta if k.startswith("bsx_step_kernel<")) >= 70          # every instance of the step kernel is in the table
    bad = {k: v for k, v in meta.items() if v["vgpr_spill_count"] or v["private_segment_fixed_size"]}
    assert not bad, bad
    assert all(v["vgpr_count"] <= 256 for v in meta.values())
    committed = json.load(open(os.path.join(ROOT, "profiles", "r06_isa_meta.json")))
    assert committed == meta, sorted(k for k in set(meta) | set(committed) if meta.get(k) != committed.get(k))[:6]


def test_phase_files_keep_their_read_write_contract(tmp_path):
    """the step kernel's tick is seven textual phase files that share ~40 kernel locals.  Each opens with what it
    reads, writes, exports and which LDS arrays it stores to, and tools/check_phase_contract.py holds the text to it: a phase that
    assigns a kernel local (or an earlier phase's value) it does not declare as written, or drops an export, fails here."""
    tool = os.path.join(ROOT, "tools", "check_phase_contract.py")
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "7 phase files checked, 0 violation(s)" in r.stdout, r.stdout[-2000:]
    # and it does catch what it is there for: a doctored copy in which the geometry phase quietly zeroes the plane's hit points and
    # the outcome phase no longer declares `alive`
    import shutil
    csrc = os.path.join(ROOT, "deep-rl-battlespace_amd", "csrc")
    for f in os.listdir(csrc):
        if f.startswith("bsx_step_phase_") or f == "bsx_step_kernel.h":
            shutil.copy(os.path.join(csrc, f), tmp_path / f)
    g = tmp_path / "bsx_step_phase_geometry.inl"
    g.write_text(g.read_text() + "\n    hp = 0;\n    s_fl[tid] |= 64u;\n")
    o = tmp_path / "bsx_step_phase_outcome.inl"
    o.write_text(o.read_text().replace("// @exports alive rew", "// @exports rew"))
    r = subprocess.run([sys.executable, tool, "--csrc", str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert "bsx_step_phase_geometry.inl: @writes misses ['hp']" in r.stdout and "bsx_step_phase_geometry.inl: @lds misses ['s_fl']" in r.stdout
    assert "bsx_step_phase_outcome.inl: @exports misses ['alive']" in r.stdout, r.stdout


def test_the_product_sources_carry_no_experiment_switches():
    """Round 5's measured-and-rejected forms (per-call two-wave forms 1 / 2, own loads, de-phased waves, priorities by slot, padding
    instructions, row-store forms ...) are history (profiles/HISTORY_r05.md, profiles/r05_experiments.json), not product source: no
    `X_*` / `-DBSX_X_*` switch is left in csrc/ -- what remains configurable at build time is the measuring instrument (DIAG bits and
    stamps, csrc/bsx_diag.h) -- and the GPU suite builds nothing on the box."""
    import glob
    import re
    csrc = os.path.join(ROOT, "deep-rl-battlespace_amd", "csrc")
    left = {}
    for f in sorted(glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.inl")) + glob.glob(os.path.join(csrc, "*.hip"))):
        hits = re.findall(r"\b(?:BSX_X_\w+|X_[A-Z][A-Z0-9_]+)\b", open(f).read())
        if hits:
            left[os.path.basename(f)] = sorted(set(hits))
    assert not left, left
    for f in glob.glob(os.path.join(ROOT, "tests", "test_hip_*.py")):
        assert "build_variant" not in open(f).read(), f


def test_the_traffic_constants_the_bench_quotes_belong_to_the_series_they_name():
    """bench.py quotes HBM bytes per launch from profiles/traffic.json wherever it does not measure them live (the non-headline rows, the
    N > 1 line, runs under a profiler) and names the entry's `series` beside the number.  The file must be ONE step-kernel series, that
    series' PMC summary must be committed and say the same bytes (2 x FETCH_SIZE + WRITE_SIZE KiB), and the rollout entries must name
    kernels of this tree."""
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    step = {k: v for k, v in t.items() if k.startswith("E")}
    series = {v["series"] for v in step.values()}
    assert len(series) == 1, series
    tag = series.pop()
    pmc = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json")))
    for k, v in step.items():
        f, w = pmc[k]["FETCH_SIZE"]["mean_per_launch"], pmc[k]["WRITE_SIZE"]["mean_per_launch"]
        assert v["hbm_bytes_per_launch"] == int((2 * f + w) * 1024), k
        if "ticks_per_launch" in v:
            assert v["hbm_bytes_per_tick"] == v["hbm_bytes_per_launch"] // v["ticks_per_launch"], k
    b = open(os.path.join(ROOT, "bench.py")).read()
    assert "series {te.get('series')}" in b                  # the line names the series it read
    roll = {k: v for k, v in t.items() if k.startswith("rollout_")}
    assert len({v["series"] for v in roll.values()}) == 1
    names = [kn for kn in roll["rollout_graph"]["kernels"]["FETCH_SIZE"] if "bsx_step_" in kn]
    assert names and all("bsx_step_split_kernel<true, true, 0, false, true>" in kn for kn in names), names   # the graph rollout's step kernel at 65 536 x 1v1, score rows


def test_the_size_limits_of_the_two_wave_kernels_are_the_same_in_the_launcher_the_bench_and_the_profile_tool():
    """csrc/bsx_kernels.hip decides by launch size which 1v1 kernel runs (two-wave per call up to 114 688 games, 81 920 with continuous
    actions; two-wave multi-tick up to 65 536, its form 2 above 32 768).  bench.py and tools/collect_profile.py NAME the kernel a workload
    runs -- for the live PMC passes and the roofline's kernel field -- from the same numbers: they must not drift apart."""
    import re
    csrc = os.path.join(ROOT, "deep-rl-battlespace_amd", "csrc")
    k = open(os.path.join(csrc, "bsx_kernels.hip")).read()
    m = re.search(r"SPLIT_MAX_GAMES = (\d+), SPLIT_MANY_MAX_GAMES = (\d+), SPLIT_CONT_MAX_GAMES = (\d+);", k)
    lim = {"BSX_X_SPLIT_MAX": int(m.group(1)), "BSX_X_SPLIT_MANY_MAX": int(m.group(2)), "BSX_X_SPLIT_CONT_MAX": int(m.group(3))}
    form2_from = int(re.search(r"SPLIT_MANY_FORM2_FROM = (\d+);", k).group(1))
    draw_max = int(re.search(r"SPLIT_DRAW_MAX_GAMES = (\d+);", k).group(1))
    assert lim == {"BSX_X_SPLIT_MAX": 114688, "BSX_X_SPLIT_CONT_MAX": 81920, "BSX_X_SPLIT_MANY_MAX": 65536} and form2_from == 32768 and draw_max == 98304
    b = open(os.path.join(ROOT, "bench.py")).read()
    t = open(os.path.join(ROOT, "tools", "collect_profile.py")).read()
    want = "E <= (%d if many else (%d if %s else %d))" % (lim["BSX_X_SPLIT_MANY_MAX"], lim["BSX_X_SPLIT_CONT_MAX"], "%s", lim["BSX_X_SPLIT_MAX"])
    assert want % "continuous" in b, want
    assert want % "cont" in t, want
    assert "(2 if E > %d else 1) if many else 0" % form2_from in b and "(2 if E > %d else 1) if many else 0" % form2_from in t
    assert "draw = not many and (continuous or E <= %d)" % draw_max in b and "draw = not many and (cont or E <= %d)" % draw_max in t
    h = open(os.path.join(ROOT, "include", "battlespace_hip.h")).read()
    assert "114 688" in h and "81 920" in h and "65 536" in h                     # (BSX_F_ONE_WAVE's description names the three limits)
