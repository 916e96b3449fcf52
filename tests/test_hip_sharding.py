"""configs[3] on one card: the multi-rank path of bench.py exercised across PROCESSES.  `python bench.py --gpus 2
--rehearse-on-device0` starts two ranks itself (one process each, gloo for the barrier / timing reduction, both on cuda:0 --
the only part of the 8-GPU layout a one-GPU box can run); every rank owns a contiguous range of a 131 072-game job and
writes its final game state.  The concatenation must equal, field by field, what ONE process playing all 131 072 games
ends with: independent games, RNG and action table keyed by the global game index, no collective on the step path."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world,per_rank", [(2, 65536), (5, 16384)])     # 5 ranks + this test process = the 6 processes this pool lets share one card
def test_multi_rank_job_equals_the_single_process_job(tmp_path, world, per_rank):
    common = ["--steps", "150", "--warmup", "10", "--repeats", "2", "--ramp-ms", "0", "--no-cpu-baseline", "--no-other-workloads", "--no-live-traffic"]
    d2, d1 = str(tmp_path / "many"), str(tmp_path / "one")
    two = _bench(["--gpus", str(world), "--rehearse-on-device0", "--envs-per-gpu", str(per_rank), "--digest-dir", d2, *common])
    one = _bench(["--gpus", "1", "--envs-per-gpu", str(world * per_rank), "--digest-dir", d1, *common])
    assert two["n_gpus"] == world and one["n_gpus"] == 1 and two["scaling"] == "weak"
    assert two["config"]["rehearsal_all_ranks_on_device0"] is True and two["config"]["process_group"] == "gloo"
    # whole-job throughput of the N-rank line = the games of ALL ranks over the slowest rank's time
    assert abs(two["value"] - world * per_rank * 2 * 150 / (two["ms_per_step"] * 1e-3 * 150)) / two["value"] < 1e-3
    # every rank's own medians travel in the line (imbalance between shards would show here); the job's figure is not below any of them
    pr = two["per_rank"]
    assert len(pr["ms_per_step"]) == world and len(pr["avg_launch_us"]) == world and all(v > 0 for v in pr["ms_per_step"] + pr["avg_launch_us"])
    # the N > 1 line beyond `value`: a device-time aggregate, the same-run single-shard reference and the efficiency
    # against it, every rank's solo figures, min / median / max, the process group and the RCCL version.  On ONE card the ranks share
    # the device, so the efficiency's VALUE says nothing here (it is ~1/N of a card each): only that it is there and consistent.
    assert len(pr["solo_ms_per_step"]) == world and len(pr["solo_avg_launch_us"]) == world and all(v > 0 for v in pr["solo_ms_per_step"])
    mmm = pr["ms_per_step_min_median_max"]
    assert mmm["min"] <= mmm["median"] <= mmm["max"] and abs(mmm["max"] - max(pr["ms_per_step"])) < 1e-5
    assert two["value_device"] > 0 and two["process_group"] == "gloo" and isinstance(two["rccl_version"], str)
    ref = two["single_shard_reference"]
    assert abs(ref["agent_steps_per_s"] - per_rank * 2 / (ref["ms_per_step"] * 1e-3)) / ref["agent_steps_per_s"] < 1e-3
    assert abs(two["scaling_efficiency"] - two["value"] / (world * ref["agent_steps_per_s"])) < 2e-3 and 0 < two["scaling_efficiency"] < 1.5
    assert two["scaling_efficiency_device"] > 0
    assert two["host"]["pinning"] in ("numa", "plain") and two["host"]["cores_of_rank0"]           # rank 0 sits on its own block of cores
    assert "scaling_efficiency" not in one and "value_device" not in one and one["host"]["pinning"].startswith("unpinned")
    assert one["timing"]["unbarriered_ms_per_step"] > 0
    assert "per_rank" not in one and two["baseline_configs"][f"N{world}_x_{per_rank}_1v1"]["agent_steps_per_s"] == round(two["value"])
    metas = [json.load(open(os.path.join(d2, f"rank{r}.json"))) for r in range(world)]
    assert [m["env_offset"] for m in metas] == [r * per_rank for r in range(world)] and all(m["n_envs"] == per_rank and m["world"] == world for m in metas)
    parts = [torch.load(os.path.join(d2, f"rank{r}.pt")) for r in range(world)]
    whole = torch.load(os.path.join(d1, "rank0.pt"))
    for k in sorted(whole):
        cat = torch.cat([p[k] for p in parts])
        if k in ("bl_x", "bl_y", "bl_dir"):                      # slots without a live bullet hold leftovers
            m = whole["bl_live"].bool()
            assert torch.equal(cat[m], whole[k][m]), k
        else:
            assert torch.equal(cat, whole[k]), k
    # both lines count the same finished games: the logging all-reduce summed the two shards' counters
    assert two["games_finished"] == one["games_finished"] > 0 and two["ties"] == one["ties"]
    assert int(whole["counters"][:, 0].sum()) == one["games_finished"]


def test_bench_under_torch_distributed_run_uses_the_ranks_it_is_given():
    """The driver's launch shape -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with N = 2 on this one card (gloo rehearsal): bench.py reads RANK / WORLD_SIZE /
    MASTER_* from the environment instead of starting ranks itself, rank 0 prints the one JSON line for the whole job."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--rehearse-on-device0", "--no-cpu-baseline", "--no-other-workloads"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 65536 * 2 * 20 / (d["ms_per_step"] * 1e-3 * 20)) / d["value"] < 1e-3
    assert d["roofline"]["traffic_source"] is None or "profiles/traffic.json" in d["roofline"]["traffic_source"]     # no live PMC passes at N > 1


def test_rccl_refused_on_one_card_moves_both_ranks_to_gloo_together():
    """The RCCL leg on real hardware, as far as one card goes: two ranks that share cuda:0 ASK for the nccl backend.  RCCL refuses two
    ranks on one device ("invalid usage"), on both ranks, inside the collective probe -- the setup of sharding.init_timing_group must
    bring both ranks out on gloo within seconds (not RCCL's timeout), say why in the line, and the measurement must go through."""
    import time
    t0 = time.time()
    d = _bench(["--gpus", "2", "--rehearse-on-device0", "--backend", "nccl", "--steps", "50", "--warmup", "5", "--repeats", "2", "--ramp-ms", "0",
                "--no-cpu-baseline", "--no-other-workloads", "--no-live-traffic"], timeout=300)
    assert time.time() - t0 < 240                            # (seconds when RCCL refuses at once, as it does; the probe wait bounds it otherwise)
    assert d["n_gpus"] == 2 and len(d["per_rank"]["ms_per_step"]) == 2 and d["value"] > 1e9          # the measurement went through, on both ranks
    if d["config"]["process_group"] == "gloo":              # what this image's RCCL does: refused, both ranks moved together
        assert "nccl (RCCL) did not come up on every rank" in d["config"]["process_group_note"]
    else:                                                   # an RCCL build that accepts two ranks on one device: then both ranks are on it
        assert d["config"]["process_group"] == "nccl" and d["config"]["process_group_note"] is None


@pytest.mark.skipif(__import__("torch").cuda.device_count() < 2, reason="needs two GPUs: the RCCL leg of bench.py --gpus N on distinct devices")
def test_two_ranks_on_two_devices_come_up_on_rccl():
    """`python bench.py --gpus 2` on a node with at least two cards: one rank per GPU, the process group on the nccl backend (= RCCL
    over xGMI) with device_id and the probe all-reduce -- NOT the gloo fallback that a one-card rehearsal takes.  Skipped on the
    one-GPU box; the first multi-GPU node that runs the suite proves the happy path of bench.py's process-group setup."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "nccl", "--steps", "100", "--warmup", "20", "--repeats", "2",
                          "--no-cpu-baseline", "--no-other-workloads", "--no-live-traffic"], capture_output=True, text=True, timeout=900, cwd=root,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["process_group"] == "nccl" and d["config"]["process_group_note"] is None, d["config"]
    assert d["config"]["rehearsal_all_ranks_on_device0"] is None
    assert d["value"] > 1.5 * 65536 * 2 / (d["ms_per_step"] * 1e-3) * 0.5      # two shards' worth of agent-steps in the same wall time
