#!/usr/bin/env python3
"""bench.py -- agent-steps/sec of the fused Battlespace step() on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: ONE step() of every game a rank owns (65 536 games of 1v1 per
GPU = BASELINE.json configs[1]; N GPUs = N independent shards of 65 536, configs[3] at N=8; no collective on the
step path).  Actions are i.i.d. uniform over {0,1,2,3} (torch Philox generator, seed 1234+rank), generated before
the timed region and resident in HBM; bullet jitter and auto-reset spawns are drawn in-kernel (Philox4x32-10).
Finished games are re-spawned by the next step() call (auto_reset), so resets are inside the measurement.

The K timed steps are launched as replays of a HIP graph holding `--graph-len` consecutive step() launches (the
launch-bound inner loop of a rollout; `--mode eager` times one Python call per step instead).  Timing: barrier +
synchronize on both sides, max over ranks.  Rank 0 prints ONE JSON line.

roofline: the step kernel is HBM-bound integer/fp64 work.  achieved = ALGORITHMIC bytes per launch (SURVEY.md
section 8d: 260 B per agent-step at 1v1, 289.3 B at 4v4, x E*A agent-steps per launch) / the kernel's average launch
duration, measured here with HIP events on the launch stream over the timed region.  peak = 8 TB/s.
cpu_baseline: the CPU oracle (oracle/battlespace_ref.py, the scalar Python restatement of the reference's step()) on
configs[0] -- 1 game of 1v1, the same uniform random actions, reset on done -- timed on one host core of this box.
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_ALG = {1: 260.0, 2: 279.25, 3: 285.67, 4: 289.3}   # algorithmic bytes per agent-step (SURVEY.md section 8d formula)
HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: HBM3E 8 TB/s


def b_alg(n):
    """[13 + 146 + 13/A] + [13 + 50 + 5/A] + [4 + 4(3n+2) + 4 + 1] bytes per agent-step."""
    A = 2 * n
    return (13 + 146 + 13 / A) + (13 + 50 + 5 / A) + (4 + 4 * (3 * n + 2) + 4 + 1)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()


def cpu_baseline(seconds_target=12.0):
    """configs[0]: the oracle's step() on 1 game of 1v1, uniform random actions (seed 1234), reset on done; 1 core."""
    import random
    import numpy as np
    from oracle import battlespace_ref as ref
    random.seed(1234)
    env = ref.RefEnv(n_agents=1)
    ids = env.possible_agents
    acts = np.random.default_rng(1234).integers(0, 4, size=(200_000, 2)).tolist()
    env.reset()
    for k in range(2000):                                   # warm-up
        if env.env_done:
            env.reset()
        env.step({ids[0]: acts[k][0], ids[1]: acts[k][1]})
    calls = 0
    t0 = time.perf_counter()
    while True:
        for k in range(10_000):
            if env.env_done:
                env.reset()
            a = acts[(calls + k) % 200_000]
            env.step({ids[0]: a[0], ids[1]: a[1]})
        calls += 10_000
        dt = time.perf_counter() - t0
        if dt >= seconds_target:
            break
    extra = {}
    try:    # context only: the C restatement of the same path (oracle/battlespace_ref.c, OpenMP over games), all host cores
        from oracle import cref
        Ec, Tc = 16384, 60
        c = cref.CRefBatch(Ec, n_agents=1, seed=1234, auto_reset=True)
        c.reset()
        a = np.random.default_rng(1).integers(0, 4, size=(Tc, Ec, 2)).astype(np.int32)
        for t in range(5):
            c.step(a[t])
        t1 = time.perf_counter()
        for t in range(Tc):
            c.step(a[t])
        dtc = time.perf_counter() - t1
        extra = {"c_port_all_cores": {"value": round(Ec * 2 * Tc / dtc, 1), "unit": "agent-steps/s", "cores": os.cpu_count(),
                                      "sample": f"{Tc} steps of {Ec} games x 1v1, oracle/battlespace_ref.c, OpenMP"}}
    except Exception as exc:      # the C oracle is optional context; the Python port above is the reported baseline
        extra = {"c_port_all_cores": {"error": str(exc)[:120]}}
    return {**extra, "value": round(calls * 2 / dt, 1), "unit": "agent-steps/s", "cores": 1, "kind": "port",
            "sample": f"{calls} step() calls of 1 game x 1v1 (configs[0]), uniform random actions seed 1234, "
                      f"reset on done, {dt:.1f} s on 1 of {os.cpu_count()} host cores ({_cpu_model()}, {platform.machine()}, "
                      f"CPython {platform.python_version()}); oracle/battlespace_ref.py"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--n-agents", type=int, default=1, help="planes per team (1 = configs[1], 4 = configs[2])")
    ap.add_argument("--action-mix", choices=("uniform", "forward", "shoot"), default="uniform",
                    help="uniform = i.i.d. over {0,1,2,3} (the headline); forward = nobody ever shoots (traffic calibration: every "
                         "byte moved is known); shoot = everybody shoots every tick (stress: ~11 live bullets per agent)")
    ap.add_argument("--continuous", action="store_true", help="continuous [speed, turn, shoot] actions (battle_env.py:418-424) instead of discrete")
    ap.add_argument("--mode", choices=("graph", "eager", "many"), default="graph")
    ap.add_argument("--graph-len", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the two extra (non-headline) measurements")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend for the barrier / timing reduction (nccl = RCCL; gloo only to rehearse N>1 on a 1-GPU box)")
    ap.add_argument("--rehearse-on-device0", action="store_true", help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from deep_rl_battlespace_amd import sharding

    rank, world, local_rank = sharding.rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev_index = 0 if args.rehearse_on_device0 else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")
    if world > 1:                                           # used for the barrier / max-time reduction only
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
            except TypeError:                                   # older torch: no device_id argument
                dist.init_process_group("nccl")
        else:
            dist.init_process_group("gloo")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def measure(n, E, K, W, mode, graph_len):
        """K timed step() calls of E games x n-v-n on this rank; returns (env, seconds, kernel ms per launch, graph len)."""
        A = 2 * n
        env = sharding.make_shard(E * world, rank, world, n_agents=n, device=dev, seed=1234, auto_reset=True,
                                  continuous_actions=args.continuous)
        env.reset()
        G = max(1, min(graph_len, K))                       # steps per graph replay; a remainder runs as plain calls
        gen = torch.Generator(device=dev)
        gen.manual_seed(1234 + rank)
        if args.continuous:
            actions = torch.rand((G, E, A, 3), generator=gen, device=dev, dtype=torch.float32) * 2 - 1
        elif args.action_mix == "uniform":
            actions = torch.randint(0, 4, (G, E, A), generator=gen, device=dev, dtype=torch.int32)
        else:
            actions = torch.full((G, E, A), 0 if args.action_mix == "forward" else 1, device=dev, dtype=torch.int32)
        if mode == "graph":
            graph, _ = env.capture_steps(actions)

            def run(steps):
                for _ in range(steps // G):
                    graph.replay()
                for t in range(steps % G):
                    env.step_batch(actions[t])
        elif mode == "many":
            # K ticks as K/G multi-tick launches (bsx_step_many_*): every tick's obs / rew / done go to their own slice of
            # [G, E, A, ...] buffers, nothing is skipped or overwritten within a launch
            D = env.obs_size
            outs = (torch.empty((G, E, A, D), dtype=torch.float32, device=dev), torch.empty((G, E, A), dtype=torch.float32, device=dev),
                    torch.empty((G, E, A), dtype=torch.uint8, device=dev))

            def run(steps):
                for _ in range(steps // G):
                    env.step_many(actions, store=True, out=outs)
                r = steps % G
                if r:
                    env.step_many(actions[:r], store=True, out=tuple(o[:r] for o in outs))
        else:
            def run(steps):
                for t in range(steps):
                    env.step_batch(actions[t % G])
        run(W)
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        run(K)
        ev1.record()
        barrier()
        dt = time.perf_counter() - t0
        kernel_ms = ev0.elapsed_time(ev1) / K              # average launch-to-launch duration on the launch stream
        if world > 1:
            t = torch.tensor([dt, kernel_ms], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, kernel_ms = float(t[0]), float(t[1])
        return env, dt, kernel_ms, G

    n, E = args.n_agents, args.envs_per_gpu
    A = 2 * n
    K, W = args.steps, args.warmup
    env, dt, kernel_ms, G = measure(n, E, K, W, args.mode, args.graph_len)

    games = sharding.reduce_counters(sharding.local_counter_sums(env).to(red_dev))   # logging only, after the timed region

    # not the headline: the same kernel on BASELINE.json configs[2] and in the streaming regime (working set > Infinity
    # Cache), a few hundred steps each, so one run shows how the roofline fraction moves with the batch
    others = {}
    multi = None
    if world == 1 and not args.no_other_workloads and args.mode != "many":
        # the same K ticks as K/100 multi-tick launches (bsx_step_many_*: every tick's outputs stored, state carried in
        # registers / L2 between ticks): what an open-loop caller (random or scripted play) gets without kernel boundaries
        del env
        torch.cuda.empty_cache()
        env, dtm, kmm, Gm = measure(n, E, K, W, "many", 100)
        achm = b_alg(n) * E * A / (kmm * 1e-3) / 1e9
        multi = {"agent_steps_per_s": round(E * A * K / dtm, 1), "us_per_tick": round(kmm * 1e3, 3), "ticks_per_launch": Gm,
                 "roofline_frac": round(achm / HBM_PEAK_GBS, 4), "kernel": f"bsx_step_kernel<{n if n <= 4 else 0},{'true' if args.continuous else 'false'},true,false>",
                 "note": "not the headline: north_star asks for one kernel per step"}
        try:
            multi["traffic_per_tick"] = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[f"E{E}_n{n}_many"]["hbm_bytes_per_tick"]
        except Exception:
            multi["traffic_per_tick"] = None
    loop_sampling = None
    if world == 1 and not args.no_other_workloads and not args.continuous:
        # the same loop with the actions SAMPLED inside it (one torch.randint + one step() call per
        # tick from Python, no graph): what a caller that draws random actions tick by tick sees
        del env
        torch.cuda.empty_cache()
        env = sharding.make_shard(E, 0, 1, n_agents=n, device=dev, seed=1234, auto_reset=True)
        env.reset()
        gen2 = torch.Generator(device=dev); gen2.manual_seed(1234)
        Ks = min(K, 500)
        for _ in range(50):
            env.step_batch(torch.randint(0, 4, (E, A), generator=gen2, device=dev, dtype=torch.int32))
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(Ks):
            env.step_batch(torch.randint(0, 4, (E, A), generator=gen2, device=dev, dtype=torch.int32))
        torch.cuda.synchronize(dev); dts = time.perf_counter() - t0
        loop_sampling = {"agent_steps_per_s": round(E * A * Ks / dts, 1), "us_per_step": round(dts / Ks * 1e6, 2), "steps": Ks,
                         "note": "eager Python loop: torch.randint + step_batch per tick"}
    if world == 1 and not args.no_other_workloads and (n, E) == (1, 65536):
        for tag, (n2, E2, K2) in {"configs[2] 65536 x 4v4": (4, 65536, 400), "1048576 x 1v1 (streaming)": (1, 1048576, 200),
                                  "65536 x 1v1, all-shoot stress (action = 1 every tick, ~11 live bullets per agent)": (1, 65536, 400)}.items():
            del env
            torch.cuda.empty_cache()
            saved_mix = args.action_mix
            if "all-shoot" in tag:
                args.action_mix = "shoot"
            env, dt2, km2, _ = measure(n2, E2, K2, 50, args.mode, 100)
            args.action_mix = saved_mix
            ach = b_alg(n2) * E2 * 2 * n2 / (km2 * 1e-3) / 1e9
            others[tag] = {"agent_steps_per_s": round(E2 * 2 * n2 * K2 / dt2, 1), "avg_launch_us": round(km2 * 1e3, 2),
                           "roofline_frac": round(ach / HBM_PEAK_GBS, 4)}
        del env
        torch.cuda.empty_cache()
    if rank == 0:
        agent_steps = E * world * A * K
        bytes_per_launch = b_alg(n) * E * A
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"E{E}_n{n}" + ("_many" if args.mode == "many" else "")
                if key in tj:
                    traffic = tj[key]["hbm_bytes_per_tick" if args.mode == "many" else "hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        out = {
            "metric": "agent-steps/sec", "value": round(agent_steps / dt, 1), "unit": "agent-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64/i16", "data": "synthetic",
            "config": {"workload": f"{E} games x {n}v{n} per GPU, {'uniform random' if args.action_mix == 'uniform' else 'all-' + args.action_mix} {'continuous' if args.continuous else 'discrete'} actions, fused HIP step(), auto-reset "
                                   f"(BASELINE.json configs[{1 if n == 1 else 2}]{' x ' + str(world) + ' shards' if world > 1 else ''})",
                       "envs_per_gpu": E, "n_agents_per_team": n, "agents_per_env": A, "launch": args.mode,
                       "graph_len": G if args.mode == "graph" else None, "parallelism": f"{world} independent env shards, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": f"bsx_step_kernel<{n if n <= 4 else 0},{'true' if args.continuous else 'false'},{'true' if args.mode == 'many' else 'false'},false>", "avg_launch_us": round(kernel_ms * 1e3, 3),
                         "algorithmic_bytes_per_launch": round(bytes_per_launch), "bytes_per_agent_step": round(b_alg(n), 2),
                         # SURVEY.md section 8d asks for these beside it: the API-only lower bound (action in; obs, reward, done out)
                         # and the fraction on the bytes that actually reached HBM (PMC)
                         "io_only_bytes_per_agent_step": 4 + 4 * (3 * n + 2) + 4 + 1,
                         "io_only_frac": round((4 + 4 * (3 * n + 2) + 4 + 1) * E * A / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "traffic_frac": round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
                         "regime": "latency-bound at this size: 2 wavefronts per SIMD, working set 48 MB inside the 256 MB Infinity Cache "
                                   "(DESIGN.md section 6)" if (n, E) == (1, 65536) and args.mode != "many" else None},
            "games_finished": int(games[0]), "ties": int(games[1]), "red_wins": int(games[2]), "blue_wins": int(games[3]),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        out["other_workloads"] = others
        out["multi_tick_launch"] = multi
        out["loop_incl_action_sampling"] = loop_sampling
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
