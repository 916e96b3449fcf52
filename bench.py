#!/usr/bin/env python3
"""bench.py -- agent-steps/sec of the fused Battlespace step() on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Launched plainly with --gpus N > 1 this process touches no GPU: it starts N children (one per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), waits for them and exits with their code; rank 0's child prints the
JSON line.  Under torch.distributed.run the environment is already there and each rank measures directly.

A "step" is one pass of the hot path over one batch: ONE step() of every game a rank owns (65 536 games of 1v1 per GPU =
BASELINE.json configs[1]; N GPUs = N independent shards of 65 536, configs[3] at N=8; no collective on the step path).
Actions are i.i.d. uniform over {0,1,2,3}, a counter hash of (seed 1234, tick, GLOBAL game index, agent) so that the job's
action table does not depend on the sharding, generated before the timed region and resident in HBM; bullet jitter and auto-reset spawns are drawn in-kernel (Philox4x32-10).  Finished games are re-spawned
by the next step() call (auto_reset), so resets are inside the measurement; the games' clocks are staggered first (game e
starts e mod tie_tick ticks late) so that the time-limit ties -- 98 % of all game ends under random play -- are spread
evenly over any window instead of arriving in lock-step every 121 calls.

Timing (median of repeats).  After the stagger, the W warm-up steps and an untimed device ramp, the
block of EXACTLY K steps is timed R times -- first the R wall-clock blocks (A), one after the other, then the R event brackets (B):
  (A) wall clock of each rank's K steps between barrier + torch.cuda.synchronize pairs, MAX over ranks -> ms_per_step, value
  (B) HIP events on the launch stream around the same K steps, recorded while an untimed K-step block queued just
      before is still running, so the interval holds device time only (no host launch latency of the first graph);
      when K < --graph-len the bracket holds the whole action table (graph-len launches) as ONE graph, so that the
      ~11 us a graph replay costs on top of its kernels is not booked on the kernel     -> roofline.avg_launch_us
and the MEDIAN over the R repeats is reported (all samples are in the line).  The K steps are launched as replays of
HIP graphs holding G = min(K, --graph-len) consecutive step() launches (the launch-bound inner loop of a rollout; `--mode
eager` times one Python call per step instead); the action table always spans --graph-len ticks and a short block walks
through it slice by slice, so the workload does not depend on K.  Rank 0 prints ONE JSON line.

roofline: the step kernel is HBM-bound integer/fp64 work by the contract's accounting.  achieved = ALGORITHMIC bytes per launch
(260 B per agent-step at 1v1, 289.3 B at 4v4, x E*A agent-steps per launch) / the kernel's average launch
duration (B); peak = 8 TB/s; frac = achieved / peak is the contract figure (printed as null when it exceeds 1: the 12-slot count is
then more than the launch moves).  frac_on_traffic is the same duration against the bytes that actually reached HBM: PMC
FETCH_SIZE x 2 + WRITE_SIZE, collected by two child runs of this script under `rocprofv3 --kernel-trace --pmc` (one counter per
pass, as MI355X_MICROARCH.md prescribes) right after the timed region -- the counters cannot be read from inside this process;
`traffic_source` says so, or names the profiles/traffic.json series that was used instead (`--no-live-traffic`, no rocprofv3, a
profiler already attached, N > 1; always for `other_workloads`).  frac_claimed = min(frac, frac_on_traffic) is the figure to quote;
live_aware_bytes_per_launch is what this build's layout must move with the measured bullet load (b_live below).
chained_graphs (not the headline; `--chains P` runs any workload that way): the same K steps with every graph's launches as P
independent chains over game ranges (parallel_env.capture_steps(chains=P), bsx_step_*_range): per step the whole batch still
advances one tick -- as P launches that wait only for their own range's previous launch.  Same games bit for bit.
cpu_baseline: the CPU oracle (oracle/battlespace_ref.py, the scalar Python restatement of the reference's step()) on
configs[0] -- 1 game of 1v1, the same uniform random actions, reset on done -- one process PINNED to one host core of this box
(BASELINE.md section 3); beside it the same port at 4v4 (`port_4v4`, the same-run CPU figure for configs[2]) and, as context, the C
port on all cores and the reference's own survey-time figures.
roofline.bound: "hbm" only where the measured traffic runs at half of the peak or more, else "issue/latency" (65 536 x 1v1 sits at
the floor of its form: kernel boundary + first loads + instruction issue at two waves per SIMD).
baseline_configs: the line's LAST key, numbers only (< 1 800 characters: the driver keeps the last 2 000): agent-steps/s, us and frac_claimed of every BASELINE.json
config this run measured -- it survives a record that keeps only the tail of the line.
N > 1: every rank pins itself to its own block of host cores (its card's NUMA node when sysfs names it) before its first GPU call;
gloo control group first, the RCCL probe beside it, and the ranks agree on the timing backend before anyone proceeds
(sharding.init_timing_group).  Before the barriered blocks every rank times the same block R times ALONE (the ranks take turns, the
other cards idle): rank 0's is the same-run single-shard reference.  The line then carries, beside `value` (all ranks' agent-steps over
the MAX over ranks of the wall-clock block): `value_device` (sum over ranks of E*A / the rank's median kernel time: immune to host
jitter), `scaling_efficiency` = value / (N x the single-shard reference) and `scaling_efficiency_device`, `per_rank` (every rank's
medians, min / median / max), `process_group`, `rccl_version`.  R: --repeats, else 5, else 25 when a block is shorter than 2 ms.
"""
import argparse
import json
import os
import platform
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: HBM3E 8 TB/s


def b_alg(n, continuous=False):
    """[13 + 146 + 13/A] + [13 + 50 + 5/A] + [4 + 4(3n+2) + 4 + 1] bytes per agent-step
    (continuous actions: +8 B, three float32 instead of one int32)."""
    A = 2 * n
    return (13 + 146 + 13 / A) + (13 + 50 + 5 / A) + (4 + 4 * (3 * n + 2) + 4 + 1) + (8 if continuous else 0)


def b_io(n, continuous=False):
    return 4 + 4 * (3 * n + 2) + 4 + 1 + (8 if continuous else 0)


def b_live(n, live, shots, continuous=False):
    """What THIS build's layout (v2, ABI 14) has to move per agent-step when an agent holds `live` bullets and fires `shots` per call (the
    contract formula b_alg charges all 12 slots of a canonical layout, however few are in flight): read plane record 8 (+ 8: the float64
    heading, continuous) + the game's two records 16/A + action 4 + 8 per live bullet (its pool entry: position, age, owner, integer
    step code); written plane 8 (+ 8) + the game's dynamic record 8/A + 8 per surviving bullet (entries are rewritten whole, compacted)
    + 8 per shot (its heading in the export ring) + the outputs the API mandates (observation row, reward, done).  Not counted: the
    first 64 pool entries of a wave are read whether or not they exist (512 bytes per wave = 8 per agent at most)."""
    A = 2 * n
    c = 8 if continuous else 0
    return (8 + c + 16 / A + 4 + 8 * live) + (8 + c + 8 / A + 8 * live + 8 * shots) + (4 * (3 * n + 2) + 4 + 1) + c


def claim(frac, frac_on_traffic):
    """The roofline fraction this line CLAIMS: the smaller of the contract figure (algorithmic bytes / time / peak) and the same time
    against the bytes that actually reached HBM ("if actual HBM bytes < B_alg, quote the smaller").  A
    contract figure above 1 says that the 12-slot algorithmic count is not what the kernel moves, nothing else: it is printed as null."""
    vals = [v for v in (frac, frac_on_traffic) if v is not None]
    return (round(min(vals), 5) if vals else None), (None if frac is not None and frac > 1.0 else frac)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()


def _pin_one_core():
    """"1 process pinned to 1 core".  Pins this process to ONE of the cores it may run on (the highest-numbered
    one: core 0 takes the box's interrupts) and returns (restore(), core or None when the platform has no affinity call)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        core = allowed[-1]
        os.sched_setaffinity(0, {core})
        return (lambda: os.sched_setaffinity(0, set(allowed))), core
    except (AttributeError, OSError):
        return (lambda: None), None


def _time_port(n_agents, seconds_target, warm=2000):
    """The oracle's step() on ONE game of n_agents-v-n_agents, uniform random actions (seed 1234), reset on done -> (calls, seconds)."""
    import random
    import numpy as np
    from oracle import battlespace_ref as ref
    random.seed(1234)
    env = ref.RefEnv(n_agents=n_agents)
    ids = env.possible_agents
    A = len(ids)
    acts = np.random.default_rng(1234).integers(0, 4, size=(200_000, A)).tolist()
    env.reset()
    for k in range(warm):
        if env.env_done:
            env.reset()
        env.step(dict(zip(ids, acts[k])))
    calls = 0
    t0 = time.perf_counter()
    while True:
        for k in range(2_000):
            if env.env_done:
                env.reset()
            env.step(dict(zip(ids, acts[(calls + k) % 200_000])))
        calls += 2_000
        dt = time.perf_counter() - t0
        if dt >= seconds_target:
            return calls, dt


def cpu_baseline(seconds_target=12.0):
    """configs[0]: the oracle's step() (the Python port of the reference's, oracle/battlespace_ref.py) on 1 game of 1v1, uniform
    random actions (seed 1234), reset on done; one process pinned to one core, as BASELINE.md section 3 words it.  Beside it the
    same port on 1 game of 4v4 (`port_4v4`: the same-run CPU figure for configs[2]) and, as context, the C port on all cores."""
    import numpy as np
    restore, core = _pin_one_core()
    try:
        calls, dt = _time_port(1, seconds_target)
        calls4, dt4 = _time_port(4, seconds_target / 2, warm=500)
    finally:
        restore()
    pin = f"pinned to core {core}" if core is not None else "not pinned (no sched_setaffinity)"
    extra = {"port_4v4": {"value": round(calls4 * 8 / dt4, 1), "unit": "agent-steps/s", "cores": 1, "kind": "port",
                          "sample": f"{calls4} step() calls of 1 game x 4v4 in {dt4:.1f} s, {pin}"},
             "reference_published": {"1v1": 25500, "2v2": 23600, "4v4": 18600, "unit": "agent-steps/s",
                                     "note": "BASELINE.md section 2: the reference itself (get_pos blits included), other hardware"}}
    try:    # context only: the C restatement of the same path (oracle/battlespace_ref.c, OpenMP over games), all host cores
        from oracle import cref
        Ec, Tc = 16384, 100
        c = cref.CRefBatch(Ec, n_agents=1, seed=1234, auto_reset=True)
        c.reset()
        a = np.random.default_rng(1).integers(0, 4, size=(Tc, Ec, 2)).astype(np.int32)
        for t in range(20):                                 # thread pool up, pages touched
            c.step(a[t])
        done_steps, t1 = 0, time.perf_counter()
        while True:                                         # at least 2 s of steady stepping: short samples swing by 4x with the host's state
            for t in range(Tc):
                c.step(a[t])
            done_steps += Tc
            dtc = time.perf_counter() - t1
            if dtc >= 2.0:
                break
        extra["c_port_all_cores"] = {"value": round(Ec * 2 * done_steps / dtc, 1), "unit": "agent-steps/s", "cores": os.cpu_count(),
                                     "sample": f"{done_steps} steps of {Ec} games x 1v1 in {dtc:.1f} s, battlespace_ref.c, OpenMP"}
    except Exception as exc:      # the C oracle is optional context; the Python port above is the reported baseline
        extra["c_port_all_cores"] = {"error": str(exc)[:120]}
    return {"value": round(calls * 2 / dt, 1), "unit": "agent-steps/s", "cores": 1, "kind": "port",
            "sample": f"{calls} step() calls of 1 game x 1v1 (configs[0]) in {dt:.1f} s, {pin} of {os.cpu_count()}",
            "host": f"{_cpu_model()}, CPython {platform.python_version()}", **extra}


def hashed_bits(T, lo, hi, A, k, seed, device):
    """int64 [T, hi-lo, A, k] of 31 well-mixed bits each, a pure function of (seed, tick, GLOBAL game index, agent, component):
    the job's action table does not depend on how its games are sharded over ranks (splitmix64 finaliser on a counter)."""
    import torch
    M = (1 << 64)

    def c(v):                                               # two's-complement int64 constant
        v %= M
        return v - M if v >= (1 << 63) else v
    t = torch.arange(T, device=device, dtype=torch.int64).view(T, 1, 1, 1)
    e = torch.arange(lo, hi, device=device, dtype=torch.int64).view(1, hi - lo, 1, 1)
    a = torch.arange(A, device=device, dtype=torch.int64).view(1, 1, A, 1)
    j = torch.arange(k, device=device, dtype=torch.int64).view(1, 1, 1, k)
    x = (e * A + a) * c(0x9E3779B97F4A7C15) + (t * 4 + j) * c(0xD1B54A32D192ED03) + c(seed * 0x2545F4914F6CDD1D + 0x632BE59BD9B4E019)
    for sh, mul in ((30, 0xBF58476D1CE4E5B9), (27, 0x94D049BB133111EB)):
        x = (x ^ ((x >> sh) & ((1 << (64 - sh)) - 1))) * c(mul)
    x = x ^ ((x >> 31) & ((1 << 33) - 1))
    return (x >> 20) & 0x7FFFFFFF


def live_traffic(args, kernel, grid_threads, dense_file=None):
    """HBM bytes per launch of the headline kernel, measured NOW: two child runs of this script under `rocprofv3 --kernel-trace
    --pmc <counter>` -- FETCH_SIZE and WRITE_SIZE in SEPARATE passes, nothing else traced, as MI355X_MICROARCH.md's HBM section
    prescribes -- on the same workload (eager launches, so every launch is its own dispatch record), per-launch mean over the
    second half of the run.  The profiled child issues the step launches and next to nothing else: counters are collected for kernels
    matching `bsx_step` only, and the bullet-heavy workload replays the trajectory THIS (unprofiled) process recorded (`dense_file`) instead
    of preparing it under the profiler (~21 000 small dispatches and a 300-node graph replay under per-dispatch counter collection: the one
    kind of pass that ever hung, profiles/r06_traced_runs.json).  Units are KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read stream, so the read side is
    doubled (calibrated on this kernel in round 1, profiles/r01_traffic_calibration.json).  Returns (bytes, detail) or (None, why)."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    attached = [k for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_OUTPUT_PATH") if os.environ.get(k)] + \
        [v for v in os.environ.get("LD_PRELOAD", "").split(":") if "rocprof" in v]
    if attached:                                            # this process already runs under a profiler: no nested counter passes
        return None, f"a profiler is attached to this run ({attached[0]}): live counter passes skipped"
    many = args.mode == "many"
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bsx_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--kernel-include-regex", "bsx_step", "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.abspath(__file__),
               "--steps", "100" if many else "200", "--warmup", "20", "--repeats", "1", "--ramp-ms", "0", "--mode", "many" if many else "eager",
               "--envs-per-gpu", str(args.envs_per_gpu), "--n-agents", str(args.n_agents),
               "--action-mix", args.action_mix, "--no-cpu-baseline", "--no-other-workloads", "--no-live-traffic"]
        if args.continuous:
            cmd.append("--continuous")
        if args.action_mix == "dense":
            if not dense_file:
                return None, "bullet-heavy workload: no recorded trajectory to replay under the profiler (see --dense-record)"
            cmd += ["--dense-replay", dense_file]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=120)
            files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} pass failed (rc {r.returncode}): {(r.stderr or '')[-160:]}"
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                    if row["Counter_Name"] == ctr and kernel.replace(",", ", ") in row["Kernel_Name"] and int(row["Grid_Size"]) == grid_threads]
            vals = vals[len(vals) // 2:]
            if not vals:
                return None, f"no {kernel} launches in the {ctr} pass"
            got[ctr] = (sum(vals) / len(vals), len(vals))
        except Exception as exc:
            return None, f"{type(exc).__name__}: {str(exc)[:160]}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    f_kib, w_kib = got["FETCH_SIZE"][0], got["WRITE_SIZE"][0]
    ticks = 100 if many else 1
    return int((2 * f_kib + w_kib) * 1024 / ticks), {"fetch_size_kib_raw": round(f_kib, 1), "write_size_kib_raw": round(w_kib, 1),
                                                      "launches_averaged": got["FETCH_SIZE"][1], "ticks_per_launch": ticks}


def dropin_one_game(dev, calls=3000):
    """BASELINE.json configs[0] through the DROP-IN surface: one game of 1v1 on the MI355X with the reference's return types (dicts
    of numpy rows, Python numbers), stdlib `random` draws in the reference's order, uniform random actions, reset on done -- what a
    maintainer who only swaps the import gets.  One upload, one launch, one download, one synchronisation per step()."""
    import random
    import numpy as np
    import deep_rl_battlespace_amd as bsx
    random.seed(1234)
    env = bsx.parallel_env(n_agents=1, device=dev)
    ids = env.possible_agents
    acts = np.random.default_rng(1234).integers(0, 4, size=(calls + 200, 2)).tolist()
    env.reset()
    for k in range(200):
        if env.env_done:
            env.reset()
        env.step({ids[0]: acts[k][0], ids[1]: acts[k][1]})
    t0 = time.perf_counter()
    for k in range(200, 200 + calls):
        if env.env_done:
            env.reset()
        env.step({ids[0]: acts[k][0], ids[1]: acts[k][1]})
    dt = time.perf_counter() - t0
    return {"agent_steps_per_s": round(calls * 2 / dt, 1), "step_calls_per_s": round(calls / dt, 1), "us_per_call": round(dt / calls * 1e6, 1),
            "calls": calls, "note": "1 game x 1v1 behind the reference's surface: one host<->device round trip per call"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--repeats", type=int, default=None,
                    help="the K-step block is timed this many times; medians are reported.  Default: 5, and 25 when one block is shorter "
                         "than 2 ms (the driver's --steps 20 is ~0.14 ms: more samples instead of a median of five)")
    ap.add_argument("--ramp-ms", type=float, default=150.0, help="untimed device ramp before the first timed block (clock ramp-up)")
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--n-agents", type=int, default=1, help="planes per team (1 = configs[1], 4 = configs[2])")
    ap.add_argument("--action-mix", choices=("uniform", "forward", "shoot", "dense"), default="uniform",
                    help="uniform = i.i.d. over {0,1,2,3} (the headline); forward = nobody ever shoots (traffic calibration: every "
                         "byte moved is known); shoot = action 1 every tick (planes run into a wall: ~1.6 live bullets per agent); "
                         "dense = recorded closed-loop keep-shooting play (~7 live bullets per agent, the bullet-heavy regime)")
    ap.add_argument("--continuous", action="store_true", help="continuous [speed, turn, shoot] actions (battle_env.py:418-424) instead of discrete")
    ap.add_argument("--mode", choices=("graph", "eager", "many"), default="graph")
    ap.add_argument("--graph-len", type=int, default=100)
    ap.add_argument("--chains", type=int, default=1,
                    help="graph mode: the batch as this many game ranges, each its own chain of launches on a branch of the graph "
                         "(capture_steps(chains=)); 1 = one launch per step over the whole batch (the headline)")
    ap.add_argument("--dense-record", default=None, metavar="FILE",
                    help="--action-mix dense: write the recorded trajectory (state snapshot + the K calls' actions) to FILE after the closed-loop "
                         "preparation, for a later --dense-replay")
    ap.add_argument("--dense-replay", default=None, metavar="FILE",
                    help="--action-mix dense: load the trajectory a --dense-record run wrote instead of preparing it -- the form every run under "
                         "rocprofv3 --pmc takes (live_traffic, tools/profile_round.sh): the profiled process then issues the step launches and "
                         "nothing else (profiles/r06_traced_runs.json: why)")
    ap.add_argument("--one-wave", action="store_true", help="A/B: keep the one-wave 1v1 kernels (BSX_F_ONE_WAVE) where the library would take a two-wave form; same results")
    ap.add_argument("--no-stagger", action="store_true", help="leave all games on the same clock (time-limit ties in lock-step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the extra (non-headline) measurements")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) that measure the headline's HBM bytes; "
                         "roofline.traffic then comes from profiles/traffic.json")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default=None,
                    help="process-group backend for the barrier / timing reduction (nccl = RCCL; gloo only to rehearse N>1 on a 1-GPU box)")
    ap.add_argument("--rehearse-on-device0", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (implies --backend gloo; at most 6 ranks may share the card)")
    ap.add_argument("--digest-dir", default=None, help="every rank writes a digest of its shard's final state here (cross-process shard check)")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` started plainly: N children, one per GPU.  The parent initialises no GPU (counting devices
    does not), so nothing is exec'd or forked from a process that holds the card."""
    import torch
    N = args.gpus
    have = torch.cuda.device_count()
    if have < N and not args.rehearse_on_device0:
        raise SystemExit(f"--gpus {N} but only {have} GPU(s) visible; to rehearse the {N}-rank path on one card: "
                         f"python bench.py --gpus {N} --rehearse-on-device0")
    if args.rehearse_on_device0 and N > 6:
        raise SystemExit("at most 6 processes may share one card on this pool")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(N):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(N), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), BSX_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    try:
        deadline = time.time() + 1500
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is not None:
                    pending.remove(p)
                    if r != 0 and rc == 0:
                        rc = r
            if rc != 0 or time.time() > deadline:
                break
            time.sleep(0.05)
        if pending:                                         # a rank failed (or the deadline passed): stop the others, by PID
            rc = rc or 124
            for p in pending:
                p.terminate()
            for p in pending:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    raise SystemExit(rc)


def main():
    args = parse_args()
    if args.backend is None:
        args.backend = "gloo" if args.rehearse_on_device0 else "nccl"
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist
    from deep_rl_battlespace_amd import sharding

    rank, world, local_rank = sharding.rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # N > 1: every rank on its own block of host cores (its card's NUMA node when sysfs says which), BEFORE the first GPU call, so that
    # the runtime's threads inherit it: eight ranks' launch / sync loops must not migrate over each other (a 142 us timed block is
    # booked MAX over ranks: one descheduled rank is a scaling loss).  Nothing is re-executed; a rank that cannot pin says so.
    pinned = sharding.pin_rank_to_its_cores(local_rank, world, same_card_for_all=args.rehearse_on_device0)
    dev_index = 0 if args.rehearse_on_device0 else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend_note, tgroup, clean_exit = None, None, True
    if world > 1:                                           # used for the barrier / max-time reduction only
        # RCCL carries nothing but the barrier and a few tiny reductions here (the step path has no collective), so a node on which
        # it cannot come up must not cost the measurement -- and the decision must be the SAME on every rank: gloo comes up first,
        # the RCCL probe runs beside it, and the ranks agree over gloo before anyone proceeds (sharding.init_timing_group)
        tgroup, args.backend, backend_note, clean_exit = sharding.init_timing_group(args.backend, dev)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier(group=tgroup)
        torch.cuda.synchronize(dev)

    per_rank = {}                                           # name -> [world][R] samples of every rank (N > 1 only)

    def max_over_ranks(values, name=None):
        if world == 1:
            return list(values)
        t = torch.tensor(values, device=red_dev, dtype=torch.float64)
        if name is not None:                                # every rank's own samples, for the per-rank arrays of the line
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t, group=tgroup)
            per_rank[name] = [q.tolist() for q in parts]
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=tgroup)
        return t.tolist()

    def timed_blocks(run, K, R, restore=None, run_b=None, Kb=None, tag=None):
        """-> (wall seconds per repeat, kernel ms per launch per repeat); see the module docstring (A), (B).
        restore: a recorded trajectory is rewound before every block (its device copy is what keeps the queue busy in (B)).
        run_b / Kb: (B) brackets Kb >= 100 launches -- the K-step block repeated inside ONE graph -- when K itself is shorter than
        that, so that the per-replay cost of a short graph (~11 us) is not booked on the kernel."""
        walls, kms = [], []
        if run_b is None:
            run_b, Kb = run, K
        # (A) first, R blocks one after the other, nothing between a block's closing synchronisation and the next block's opening one; then
        # (B), R event brackets.  Until round 6 the two alternated per repeat; a 20-step block is ~108 us of kernels + 12 ... 20 us of
        # graph launch and wake-up, and that remainder moves by a few us with what the process did just before the block
        # (tools/micro/block_cycle.py: 120 ... 124 us in a tight loop of blocks, 124 ... 129 behind an event bracket over a 100-node
        # replay): the blocks are measured as a tight loop of blocks, which is what the contract's wording describes.
        for _ in range(R):
            if restore:
                restore()
            barrier()
            t0 = time.perf_counter()
            run(K)
            torch.cuda.synchronize(dev)
            walls.append(time.perf_counter() - t0)          # this rank's K steps, synchronised; the MAX over ranks is taken below
            barrier()
        for _ in range(R):
            if restore:
                restore()
            else:
                run_b(Kb)                                   # untimed: keeps the device busy while the events and the next block are queued
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            run_b(Kb)
            ev1.record()
            torch.cuda.synchronize(dev)
            kms.append(ev0.elapsed_time(ev1) / Kb)
        return max_over_ranks(walls, tag and tag + "walls"), max_over_ranks(kms, tag and tag + "kms")

    def solo_blocks(run, K, R, run_b, Kb):
        """The same K-step block, R times, WITHOUT the group: the ranks take turns (the others wait in a gloo barrier, their cards idle),
        so each rank's figure is what its shard does when it has the node's host side to itself -- the same-run N = 1 reference that
        `scaling_efficiency` divides by.  Every rank plays the same number of steps here, so the shards stay the games of the unsplit
        job.  -> (wall seconds per repeat, kernel ms per launch per repeat) of THIS rank."""
        walls, kms = [], []
        for turn in range(world):
            if world > 1:
                torch.cuda.synchronize(dev)
                dist.barrier()                              # the gloo control group: no kernel on anybody's card while a rank measures
            if turn != rank:
                continue
            for _ in range(R):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                run(K)
                torch.cuda.synchronize(dev)
                walls.append(time.perf_counter() - t0)
            for _ in range(R):                              # (wall-clock blocks first, event brackets after: as timed_blocks)
                run_b(Kb)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                run_b(Kb)
                ev1.record()
                torch.cuda.synchronize(dev)
                kms.append(ev0.elapsed_time(ev1) / Kb)
        if world > 1:
            torch.cuda.synchronize(dev)
            dist.barrier()
        return walls, kms

    def ramp(run, ms, K):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            run(K)
            torch.cuda.synchronize(dev)

    def stagger(env, step_fn):
        """Spread the games' clocks: game e is re-spawned e mod tie_tick calls into the pre-roll, so afterwards the time-limit
        ties (and with them the auto-resets) fall evenly on every call instead of all on call 121, 242, ..."""
        P = env.tie_tick
        idx = (torch.arange(env.n_envs, device=dev) + env.env_offset) % P      # global game index: the same games whatever the sharding
        for k in range(P):
            step_fn(k)
            env.reset(mask=(idx == k))

    def dense_policy(env):
        """Closed-loop 'keep shooting' play for the bullet-heavy regime: fire unless the point 450 px ahead is off the field or
        the enemy base sits on the line of fire (a base hit ends games and clears every bullet); then turn toward the middle."""
        st = env.export_state(("px", "py", "pdir", "base_xy"))
        px, py = st["px"].double(), st["py"].double()
        d = torch.deg2rad(st["pdir"])
        hx, hy = torch.cos(d), -torch.sin(d)
        ax, ay = px + 450.0 * hx, py + 450.0 * hy
        out = (ax < 0) | (ax > 1200) | (ay < 0) | (ay > 800)
        b = st["base_xy"].double()
        n = env.n_agents
        ebx = torch.cat([b[:, 2:3].expand(-1, n), b[:, 0:1].expand(-1, n)], 1)
        eby = torch.cat([b[:, 3:4].expand(-1, n), b[:, 1:2].expand(-1, n)], 1)
        rx, ry = ebx - px, eby - py
        along, perp = rx * hx + ry * hy, (rx * hy - ry * hx).abs()
        near_base = (along > -40) & (along < 560) & (perp < 60)
        turn = torch.where(hy * (600 - px) - hx * (400 - py) > 0, 2, 3)
        return torch.where(out | near_base, turn, torch.ones_like(turn)).to(torch.int32)

    def measure(n, E, K, W, mode, graph_len, mix="uniform", continuous=False, R=None, do_stagger=True, chains=1, tag=None):
        """K timed step() calls of E games x n-v-n on this rank, R times -> dict(env, walls, kms, G, live)."""
        A = 2 * n
        R = R or args.repeats
        env = sharding.make_shard(E * world, rank, world, n_agents=n, device=dev, seed=1234, auto_reset=True,
                                  continuous_actions=continuous, one_wave=args.one_wave)
        env.reset()
        G = max(1, min(graph_len, K))                       # steps per graph replay; a remainder runs as plain calls
        lo = env.env_offset
        restore, live = None, None
        pos = [0]                                           # which G-tick slice of the table comes next
        if mix == "dense":
            if continuous or mode == "many":
                raise SystemExit("--action-mix dense is a recorded discrete trajectory replayed call by call (as one HIP graph, or eagerly)")
            G = K
            if args.dense_replay:
                # the trajectory an earlier, UNPROFILED run recorded (--dense-record): nothing of the closed-loop preparation -- ~480 ticks of
                # dense_policy, ~21 000 small dispatches -- is issued by this process
                rec = torch.load(args.dense_replay, map_location=dev)
                if rec["actions"].shape[0] < K or tuple(rec["actions"].shape[1:]) != (E, A) or rec["meta"] != [world, rank, n]:
                    raise SystemExit(f"--dense-replay: {args.dense_replay} holds {tuple(rec['actions'].shape)} for (world, rank, n) = {rec['meta']}, "
                                     f"this run needs {(K, E, A)} for {[world, rank, n]}")
                snap, actions, live = rec["snap"], rec["actions"][:K].contiguous(), rec["live"]   # (a shorter run replays the recording's first K calls)
            else:
                # pre-roll to the steady state of the closed-loop play, snapshot, record the next K calls' actions, rewind
                stagger(env, lambda k: env.step_batch(dense_policy(env)))
                for _ in range(60):
                    env.step_batch(dense_policy(env))
                snap = env.state_dict()
                actions = torch.empty((K, E, A), dtype=torch.int32, device=dev)
                lv = []
                for t in range(K):
                    actions[t] = dense_policy(env)
                    env.step_batch(actions[t])
                    if t % 10 == 0:
                        lv.append(float(env.export_state(("bl_live",))["bl_live"].float().sum()) / (E * A))
                live = round(sum(lv) / len(lv), 3)
                if args.dense_record:
                    torch.save({"snap": snap, "actions": actions, "live": live, "meta": [world, rank, n]}, args.dense_record)

            def restore():
                env.load_state_dict(snap)
                pos[0] = 0
            restore()
        # The action table always spans TT >= graph_len ticks (a multiple of G), whatever K is: a short block (the driver's
        # --steps 20) walks through it slice by slice, so the games see the same 100-tick action cycle as in a long run
        NG = max(1, graph_len // G) if mix != "dense" else 1
        TT = NG * G
        if mix == "dense":
            pass
        elif continuous:
            actions = (hashed_bits(TT, lo, lo + E, A, 3, 1234, dev).to(torch.float32) * (2.0 / 2147483648.0) - 1.0).contiguous()
        elif mix == "uniform":
            actions = (hashed_bits(TT, lo, lo + E, A, 1, 1234, dev)[..., 0] >> 29).to(torch.int32).contiguous()
        else:
            actions = torch.full((TT, E, A), 0 if mix == "forward" else 1, device=dev, dtype=torch.int32)
        if mix != "dense" and do_stagger and not args.no_stagger:
            stagger(env, lambda k: env.step_batch(actions[k % TT]))
        run_b, Kb = None, None
        if mode == "graph":
            graphs = [env.capture_steps(actions[i * G:(i + 1) * G], chains=chains)[0] for i in range(NG)]
            graph = graphs[0]

            def run(steps):
                for _ in range(steps // G):
                    graphs[pos[0] % NG].replay()
                    pos[0] += 1
                for t in range(steps % G):
                    env.step_batch(actions[(pos[0] % NG) * G + t])
            if NG > 1:                                      # short blocks: the event bracket holds the whole table as ONE graph (timed_blocks)
                graph_b, _ = env.capture_steps(actions, chains=chains)
                Kb = TT

                def run_b(steps):
                    graph_b.replay()
        elif mode == "many":
            # K ticks as K/G multi-tick launches (bsx_step_many_*): every tick's obs / rew / done go to their own slice of
            # [G, E, A, ...] buffers, nothing is skipped or overwritten within a launch
            D = env.obs_size
            outs = (torch.empty((G, E, A, D), dtype=torch.float32, device=dev), torch.empty((G, E, A), dtype=torch.float32, device=dev),
                    torch.empty((G, E, A), dtype=torch.uint8, device=dev))

            def run(steps):
                for _ in range(steps // G):
                    env.step_many(actions[(pos[0] % NG) * G:(pos[0] % NG + 1) * G], store=True, out=outs)
                    pos[0] += 1
                r = steps % G
                if r:
                    env.step_many(actions[:r], store=True, out=tuple(o[:r] for o in outs))
        else:
            def run(steps):
                for t in range(steps):
                    env.step_batch(actions[pos[0] % TT])
                    pos[0] += 1
        if not restore:
            run(W)
            ramp(run, args.ramp_ms, K)
        else:
            ramp(lambda k: (restore(), run(k)), args.ramp_ms, K)
        if R is None:                                       # no --repeats: 5 blocks, 25 when a block is shorter than 2 ms (decided by all ranks together)
            R = 5
            if not restore:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                run(K)
                torch.cuda.synchronize(dev)
                if max_over_ranks([time.perf_counter() - t0])[0] < 2e-3:
                    R = 25
        solo = None
        if tag and not restore:                             # the headline: each rank's own un-barriered blocks first (the same-run N = 1 reference)
            solo = solo_blocks(run, K, R, run_b or run, Kb or K)
            if world > 1:
                max_over_ranks(solo[0], tag + "solo_walls")
                max_over_ranks(solo[1], tag + "solo_kms")
        walls, kms = timed_blocks(run, K, R, restore, run_b, Kb, tag)
        if live is None and E * A <= (1 << 22):
            live = round(float(env.export_state(("bl_live",))["bl_live"].float().sum()) / (E * A), 3)
        return dict(env=env, walls=walls, kms=kms, G=G, live=live, Kb=Kb or K, solo=solo)

    def two_wave(n, continuous, many, E):
        """csrc split_applies(): 1v1 launches run as a two-wave kernel (bsx_step_split.h) -- discrete multi-tick launches of up to 65 536
        games, discrete per-call launches of up to 114 688, continuous per-call launches of up to 81 920."""
        if n != 1 or args.one_wave or (continuous and many):
            return False
        return E <= (65536 if many else (81920 if continuous else 114688))

    def kernel_name(n, continuous, many, E):
        narrow = E * 2 * n * 200 <= 0xFFFFFFFF            # csrc narrow_offsets_ok(): 32-bit offsets while every array stays below 4 GB
        if two_wave(n, continuous, many, E):
            form = (2 if E > 32768 else 1) if many else 0     # csrc launch_for_n(): multi-tick launches of more than 32 768 games take form 2
            draw = not many and (continuous or E <= 98304)    # ... per-call launches of up to 98 304 games the kernel whose geometry wave draws
            return (f"bsx_step_split_kernel<false,{'true' if narrow else 'false'},{form},"
                    f"{'true' if continuous else 'false'},{'true' if draw else 'false'}>")      # <LG, OFF32, MANY (0 per call, 1 / 2 multi-tick forms), CONT, DRAW>
        return (f"bsx_step_kernel<{n if n <= 4 else 0},{'true' if continuous else 'false'},{'true' if many else 'false'},false,false,"
                f"{'true' if narrow else 'false'}>")       # <N, CONT, MULTI, ACTOR, LG, OFF32>

    def form_floor(kernel, km):
        try:
            ff = json.load(open(os.path.join(ROOT, "profiles", "form_floor.json")))
        except Exception:
            return None
        w = ff["workload"]
        here = {"n_agents_per_team": n, "envs_per_gpu": E, "mode": args.mode, "action_mix": args.action_mix, "continuous": bool(args.continuous)}
        if ff["kernel"] != kernel or w != here or args.chains != 1:
            return None                                     # another kernel or workload: nothing re-derives the floor for it
        return {"us": ff["us"], "terms_us": ff["terms_us"], "frac": round(ff["us"] / (km * 1e3), 4), "derived_for_kernel": ff["kernel"],
                "series": ff["series"], "source": ff["source"], "note": ff["note"]}

    def traffic_entry(key):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
        except Exception:
            return None

    def summary(m, n, E, K, continuous=False, many=False, key=None):
        """One non-headline workload as a dict (medians over the repeats)."""
        A = 2 * n
        km, wall = statistics.median(m["kms"]), statistics.median(m["walls"])
        ach = b_alg(n, continuous) * E * A / (km * 1e-3) / 1e9
        frac = round(ach / HBM_PEAK_GBS, 4)
        out = {"agent_steps_per_s": round(E * A * K / wall, 1), "avg_launch_us": round(km * 1e3, 3),
               "kernel": kernel_name(n, continuous, many, E), "steps": K, "repeats": len(m["kms"]),
               "live_bullets_per_agent": m["live"]}
        te = traffic_entry(key) if key else None
        fot = None
        if te:
            tb = te["hbm_bytes_per_tick" if many else "hbm_bytes_per_launch"]
            fot = round(tb / (km * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            out.update(traffic=tb, frac_on_traffic=fot, traffic_source=f"profiles/traffic.json[{key}] (series {te.get('series')})")
        out["frac_claimed"], out["roofline_frac"] = claim(frac, fot)
        if frac > 1.0:
            out["roofline_frac_note"] = f"contract formula gives {frac} > 1: its 12-slot count exceeds what moves"
        return out

    n, E = args.n_agents, args.envs_per_gpu
    A = 2 * n
    K, W = args.steps, args.warmup
    if args.chains != 1 and args.mode != "graph":
        raise SystemExit("--chains is a property of the captured graph (--mode graph)")
    dense_tmp = None
    if (args.action_mix == "dense" and world == 1 and not args.no_live_traffic and args.chains == 1 and not args.dense_replay
            and not args.dense_record and K >= 200):
        # the counter passes below replay this run's trajectory (live_traffic): recorded here, where no profiler is attached
        import tempfile
        fd, dense_tmp = tempfile.mkstemp(prefix="bsx_dense_", suffix=".pt", dir="/tmp")
        os.close(fd)
        args.dense_record = dense_tmp
    head = measure(n, E, K, W, args.mode, args.graph_len, mix=args.action_mix, continuous=args.continuous, chains=args.chains, tag="head_")
    env = head["env"]
    games = sharding.reduce_counters(sharding.local_counter_sums(env).to(red_dev), group=tgroup)   # logging only, after the timed region
    if args.digest_dir:
        # cross-process shard check (tests/test_hip_sharding.py): what this rank's games look like after the run
        import hashlib
        st = env.export_state()
        os.makedirs(args.digest_dir, exist_ok=True)
        h = hashlib.sha256()
        for k in sorted(st):
            h.update(st[k].cpu().numpy().tobytes())
        with open(os.path.join(args.digest_dir, f"rank{rank}.json"), "w") as f:
            json.dump({"rank": rank, "world": world, "env_offset": env.env_offset, "n_envs": env.n_envs, "sha256": h.hexdigest()}, f)
        torch.save({k: v.cpu() for k, v in st.items()}, os.path.join(args.digest_dir, f"rank{rank}.pt"))
    head_env_tie_tick = env.tie_tick
    del env
    head["env"] = None
    torch.cuda.empty_cache()

    # Not the headline: the same kernel on BASELINE.json configs[2], in the streaming regime (working set > Infinity Cache),
    # with many live bullets, with continuous actions, as a multi-tick launch, and with the policy in the loop (configs[4]).
    others, multi, loop_sampling, rollouts, dropin, chained = {}, None, None, None, None, None
    extra = world == 1 and not args.no_other_workloads and (n, E) == (1, 65536) and args.mode == "graph" \
        and args.action_mix == "uniform" and not args.continuous and args.chains == 1
    if extra:
        Ko = min(K, 400)
        m = measure(n, E, min(K, 1000), W, "many", 100)
        multi = summary(m, n, E, min(K, 1000), many=True, key=f"E{E}_n{n}_many")
        multi.update(ticks_per_launch=m["G"], us_per_tick=multi.pop("avg_launch_us"),
                     note="not the headline: north_star asks for one kernel per step")
        del m
        torch.cuda.empty_cache()
        # the same loop with the actions SAMPLED inside it (one torch.randint + one step() call per
        # tick from Python, no graph): what a caller that draws random actions tick by tick sees
        envs = sharding.make_shard(E, 0, 1, n_agents=n, device=dev, seed=1234, auto_reset=True)
        envs.reset()
        gen2 = torch.Generator(device=dev); gen2.manual_seed(1234)
        Ks = min(K, 500)
        for _ in range(50):
            envs.step_batch(torch.randint(0, 4, (E, A), generator=gen2, device=dev, dtype=torch.int32))
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(Ks):
            envs.step_batch(torch.randint(0, 4, (E, A), generator=gen2, device=dev, dtype=torch.int32))
        torch.cuda.synchronize(dev); dts = time.perf_counter() - t0
        loop_sampling = {"agent_steps_per_s": round(E * A * Ks / dts, 1), "us_per_step": round(dts / Ks * 1e6, 2), "steps": Ks,
                         "note": "eager Python loop: torch.randint + step_batch per tick"}
        del envs
        torch.cuda.empty_cache()
        for tag, (n2, E2, K2, mix2, cont2, key2) in {
                "configs[2] 65536 x 4v4": (4, 65536, Ko, "uniform", False, "E65536_n4"),
                "1048576 x 1v1 (streaming)": (1, 1048576, min(K, 200), "uniform", False, "E1048576_n1"),
                "65536 x 1v1, bullet-heavy (recorded closed-loop keep-shooting play)": (1, 65536, min(K, 300), "dense", False, "E65536_n1_dense"),
                "65536 x 1v1, action 1 every tick (planes end up on a wall)": (1, 65536, Ko, "shoot", False, None),
                "65536 x 1v1, continuous actions": (1, 65536, Ko, "uniform", True, "E65536_n1_cont"),
                "65536 x 4v4, continuous actions": (4, 65536, Ko, "uniform", True, "E65536_n4_cont")}.items():
            m = measure(n2, E2, K2, 50, "graph", 100, mix=mix2, continuous=cont2, R=3)
            others[tag] = summary(m, n2, E2, K2, continuous=cont2, key=key2)
            del m
            torch.cuda.empty_cache()
        # The same step() workloads with the graph's launches as independent chains over game ranges (capture_steps(chains=)): per step
        # the whole batch still advances one tick, as P launches that wait only for their own range's previous launch
        chained = {"note": "one HIP graph, P chains of per-step launches over game ranges; same games bit for bit"}
        for tag, (n2, E2, K2, cont2, P2) in {
                "65536 x 1v1, 2 chains": (1, 65536, Ko, False, 2),
                "65536 x 4v4, 2 chains": (4, 65536, Ko, False, 2),
                "65536 x 4v4, 3 chains": (4, 65536, Ko, False, 3),
                "65536 x 4v4, continuous actions, 2 chains": (4, 65536, Ko, True, 2),
                "1048576 x 1v1, 2 chains": (1, 1048576, min(K, 200), False, 2)}.items():
            m = measure(n2, E2, K2, 50, "graph", 100, continuous=cont2, R=3, chains=P2)
            km2, wall2 = statistics.median(m["kms"]), statistics.median(m["walls"])
            chained[tag] = {"chains": P2, "us_per_step": round(km2 * 1e3, 3), "agent_steps_per_s": round(E2 * 2 * n2 * K2 / wall2, 1),
                            "steps": K2, "repeats": len(m["kms"]),
                            "is_capture_steps_default": len(sharding.chain_ranges(E2, n2, "auto")) == P2}
            te2 = traffic_entry(f"E{E2}_n{n2}" + ("_cont" if cont2 else ""))
            if te2:                                         # the chains run the same kernels over the same games: a step moves the bytes of the one-launch form
                chained[tag]["frac_claimed"] = round(te2["hbm_bytes_per_launch"] / (km2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                chained[tag]["frac_note"] = f"HBM bytes per step of the one-launch form (profiles/traffic.json[E{E2}_n{n2}{'_cont' if cont2 else ''}]) / this form's time per step"
            del m
            torch.cuda.empty_cache()
        rollouts = rollout_lines(dev, E, min(K, 320))
        try:
            dropin = dropin_one_game(dev)
        except Exception as exc:
            dropin = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
    if rank == 0:
        wall, km = statistics.median(head["walls"]), statistics.median(head["kms"])
        agent_steps = E * world * A * K
        bytes_per_launch = b_alg(n, args.continuous) * E * A
        achieved = bytes_per_launch / (km * 1e-3) / 1e9
        many = args.mode == "many"
        key = f"E{E}_n{n}" + ("_cont" if args.continuous else "") + ("_dense" if args.action_mix == "dense" else "") + ("_many" if many else "")
        te = traffic_entry(key) if args.action_mix in ("uniform", "dense") else None
        traffic = te["hbm_bytes_per_tick" if many else "hbm_bytes_per_launch"] if te else None
        tsrc = f"profiles/traffic.json[{key}] series {te.get('series')} (a constant, not this run)" if te else None
        tdetail = None
        cpu_base = cpu_baseline() if (world == 1 and not args.no_cpu_baseline) else None      # before the counter passes: a quiet host
        if world == 1 and not args.no_live_traffic and args.chains == 1:   # (a chained graph's launches cover a range each: the per-launch counters do not describe a step)
            Gw = 2
            while Gw < A:
                Gw *= 2
            grid_threads = ((E + 64 // Gw - 1) // (64 // Gw)) * 64 * (2 if two_wave(n, args.continuous, many, E) else 1)
            live_b, info = live_traffic(args, kernel_name(n, args.continuous, many, E), grid_threads,
                                        dense_file=args.dense_replay or args.dense_record)
            if dense_tmp:
                os.remove(dense_tmp)
            if live_b is not None:
                traffic, tdetail = live_b, info
                tsrc = "this run: 2 rocprofv3 --pmc child passes, 2 x FETCH_SIZE + WRITE_SIZE (KiB)"
            else:
                tsrc = (tsrc or "none") + f" [live passes unavailable: {str(info)[:60]}]"
        mixname = {"uniform": "uniform random", "forward": "all-forward", "shoot": "all-shoot", "dense": "recorded keep-shooting"}[args.action_mix]
        fot = round(traffic / (km * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None
        frac_raw = round(achieved / HBM_PEAK_GBS, 5)
        frac_claimed, frac_contract = claim(frac_raw, fot)
        frac_note = None if frac_contract is not None else f"contract formula gives {frac_raw} > 1: its 12-slot count exceeds what moves"
        # shots per agent-step: the share of `shoot` in the action table (an upper bound: dead planes and finished games do not fire)
        shots = {"uniform": 0.25, "forward": 0.0, "shoot": 1.0, "dense": 0.78}[args.action_mix] if not args.continuous else 0.5
        # what bounds the launch: HBM only where the measured traffic runs at half of the peak or more; below that the step is bound by
        # instruction issue and the kernel boundary (two waves per SIMD at C2)
        bound = "hbm" if (fot is not None and fot >= 0.5) else "issue/latency"
        cfg_no = {1: 1, 4: 2}.get(n)
        out = {
            "metric": "agent-steps/sec", "value": round(agent_steps / wall, 1), "unit": "agent-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(wall / K * 1e3, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64/i16", "data": "synthetic",
            "config": {"workload": f"{E} games x {n}v{n} per GPU, {mixname} {'continuous' if args.continuous else 'discrete'} actions, fused HIP step(), auto-reset"
                                   + (f" (BASELINE configs[{3 if (world == 8 and n == 1 and E == 65536) else cfg_no}])" if cfg_no else ""),
                       "envs_per_gpu": E, "n_agents_per_team": n, "agents_per_env": A, "launch": args.mode + (f", {args.chains} chains" if args.chains > 1 else ""),
                       "graph_len": head["G"] if args.mode == "graph" else None, "parallelism": f"{world} independent env shards, no collective",
                       "process_group": (args.backend if world > 1 else None), "process_group_note": backend_note,
                       "rehearsal_all_ranks_on_device0": bool(args.rehearse_on_device0) or None},
            "timing": {"repeats": len(head["walls"]), "statistic": "median", "ramp_ms": args.ramp_ms,
                       "ms_per_step_samples": [round(w / K * 1e3, 6) for w in head["walls"]],
                       "avg_launch_us_samples": [round(k * 1e3, 3) for k in head["kms"]], "launches_per_event_bracket": head["Kb"],
                       "note": "ms_per_step: wall clock, barrier+sync pairs; avg_launch_us: HIP events, device time only"},
            "roofline": {"bound": bound, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": frac_contract, "traffic": traffic, "frac_on_traffic": fot, "frac_claimed": frac_claimed,
                         "claim": "frac_claimed = min(frac [contract bytes], frac_on_traffic [PMC bytes])",
                         "frac_note": frac_note,
                         "live_aware_bytes_per_launch": round(b_live(n, head["live"] or 0.0, shots, args.continuous) * E * A),
                         "traffic_source": tsrc, "traffic_detail": tdetail,
                         "kernel": kernel_name(n, args.continuous, many, E), "avg_launch_us": round(km * 1e3, 3),
                         "algorithmic_bytes_per_launch": round(bytes_per_launch), "bytes_per_agent_step": round(b_alg(n, args.continuous), 2),
                         # SURVEY.md section 8d asks for these beside it: the API-only lower bound (action in; obs, reward, done out)
                         "io_only_bytes_per_agent_step": b_io(n, args.continuous),
                         "io_only_frac": round(b_io(n, args.continuous) * E * A / (km * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "live_bullets_per_agent": head["live"], "shots_per_agent_step": round(shots, 3),
                         "regime": ("4 waves per SIMD (two workgroups of the two-wave kernel: a chain wave at priority 1 + a geometry wave each), "
                                    "48 MB working set inside the 256 MB Infinity Cache") if (n, E) == (1, 65536) and not many and two_wave(n, args.continuous, many, E)
                         else ("2 waves per SIMD, 48 MB working set inside the 256 MB Infinity Cache" if (n, E) == (1, 65536) and not many else None),
                         # what bounds THIS launch form at this size: not bytes -- a kernel boundary, the first loads behind it,
                         # the vector-port time of a SIMD's waves, the store drain.  Constants of round 5 (profiles/form_floor.json), printed only
                         # while the kernel that runs is the one they were derived for; floor / measured says how close the kernel stands to it.
                         "form_floor": form_floor(kernel_name(n, args.continuous, many, E), km)},
            "games_finished": int(games[0]), "ties": int(games[1]), "red_wins": int(games[2]), "blue_wins": int(games[3]),
            "tie_tick": head_env_tie_tick,
        }
        if head.get("solo"):
            out["timing"]["unbarriered_ms_per_step"] = round(statistics.median(head["solo"][0]) / K * 1e3, 6)
            out["timing"]["unbarriered_avg_launch_us"] = round(statistics.median(head["solo"][1]) * 1e3, 3)
        out["host"] = {"cores_of_rank0": _ranges(pinned.get("cores")), "numa_node_of_rank0": pinned.get("numa_node"), "pinning": pinned.get("how")}
        if world > 1:
            out.update(multi_rank_fields(per_rank, world, E, A, K, out["value"], args.backend))
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        elif world > 1:
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = ("measured on rank 0 at N = 1 only (bench contract): the N = 1 line of the same tree carries it; "
                                        "at N > 1 every rank is pinned to its card's cores and none is free to time the CPU port")
            out["roofline"]["traffic_source"] = (tsrc or "none") + " [N > 1: no counter passes -- a rank's launches are the N = 1 shard's, kernel and games; the N = 1 line measures them live]"
        out["other_workloads"] = others
        out["chained_graphs"] = chained
        out["multi_tick_launch"] = multi
        out["loop_incl_action_sampling"] = loop_sampling
        out["policy_rollouts"] = rollouts
        out["drop_in_one_game"] = dropin
        # LAST key, numbers only: every BASELINE.json config in a form that survives a truncated record of this line
        try:
            out["baseline_configs"] = baseline_summary(out, n, E, world, km, frac_claimed)
        except Exception as exc:                            # noqa: BLE001 -- the summary is a convenience: it must never cost the line
            out["baseline_configs"] = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                                      # the gloo control group: every rank is done
        if clean_exit:
            dist.destroy_process_group()
        else:                                               # a probe thread is still blocked inside RCCL: no destructors, no joins
            sys.stdout.flush()
            os._exit(0)


def _ranges(cores):
    """[0,1,2,3,8,9] -> '0-3,8-9' (a rank's core block in the line, short)."""
    if not cores:
        return None
    cores, parts, a = sorted(cores), [], None
    for i, c in enumerate(cores):
        if a is None:
            a = c
        if i + 1 == len(cores) or cores[i + 1] != c + 1:
            parts.append(str(a) if a == c else f"{a}-{c}")
            a = None
    return ",".join(parts)


def _mmm(vals):
    return {"min": round(min(vals), 6), "median": round(statistics.median(vals), 6), "max": round(max(vals), 6)} if vals else None


def multi_rank_fields(per_rank, world, E, A, K, value, backend):
    """What the N > 1 line says beyond `value` (= all ranks' agent-steps over the MAX over ranks of the wall-clock block, as the bench
    contract words it), from every rank's own samples (per_rank[name] = [world][R]):
      value_device       = sum over ranks of E*A / that rank's median kernel time per launch (HIP events on the launch stream: device
                           time only -- host jitter on one of N ranks, which the MAX books as a scaling loss, does not enter);
      scaling_efficiency = value / (N x rank 0's own un-barriered figure, measured in this run while the other ranks' cards were idle);
                           `_device` the same on kernel time; the denominator is printed (`single_shard_reference`);
      per_rank           = every rank's medians (barriered block, kernel time, solo block) + min / median / max over ranks: placement
                           (which card, which NUMA node) is the one thing that could bend a curve of independent shards."""
    med = lambda name: [statistics.median(v) for v in per_rank.get(name, [])]     # noqa: E731
    walls, kms, swalls, skms = med("head_walls"), med("head_kms"), med("head_solo_walls"), med("head_solo_kms")
    out = {"per_rank": {"ms_per_step": [round(w / K * 1e3, 6) for w in walls], "avg_launch_us": [round(k * 1e3, 3) for k in kms],
                        "solo_ms_per_step": [round(w / K * 1e3, 6) for w in swalls], "solo_avg_launch_us": [round(k * 1e3, 3) for k in skms],
                        "ms_per_step_min_median_max": _mmm([w / K * 1e3 for w in walls]),
                        "avg_launch_us_min_median_max": _mmm([k * 1e3 for k in kms])}}
    if kms and all(k > 0 for k in kms):
        out["value_device"] = round(sum(E * A / (k * 1e-3) for k in kms), 1)
    if swalls and swalls[0] > 0:
        ref = E * A * K / swalls[0]
        out["single_shard_reference"] = {"agent_steps_per_s": round(ref, 1), "ms_per_step": round(swalls[0] / K * 1e3, 6),
                                         "avg_launch_us": round(skms[0] * 1e3, 3) if skms else None,
                                         "what": "rank 0's shard alone, un-barriered, the other ranks' cards idle, same run"}
        out["scaling_efficiency"] = round(value / (world * ref), 4)
        if skms and skms[0] > 0 and "value_device" in out:
            out["scaling_efficiency_device"] = round(out["value_device"] / (world * E * A / (skms[0] * 1e-3)), 4)
    try:
        import torch
        v = torch.cuda.nccl.version()
        out["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as exc:                                # noqa: BLE001
        out["rccl_version"] = f"unavailable ({type(exc).__name__})"
    out["process_group"] = backend
    return out


def baseline_summary(out, n, E, world, km, frac_claimed):
    """BASELINE.json's five configs (+ the streaming and evaluation workloads) as one compact dict of numbers: agent-steps/s,
    us per step (HIP events; per tick and wall clock for the rollouts) and the claimed roofline fraction.  A config this run did not
    measure is null."""
    def pick(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d

    def line(d, us="avg_launch_us"):
        if not isinstance(d, dict) or "agent_steps_per_s" not in d:
            return None
        r = {"agent_steps_per_s": round(d["agent_steps_per_s"]), "us": d.get(us)}
        if d.get("frac_claimed") is not None:
            r["frac_claimed"] = round(d["frac_claimed"], 3)
        rf = d.get("roofline")
        if isinstance(rf, dict) and rf.get("frac_mfma") is not None:      # the rollouts: matrix-core fraction (and the HBM one when PMC bytes exist)
            r["frac"] = round(rf["frac_mfma"], 3)
            if rf.get("frac_hbm") is not None:
                r["frac_hbm"] = round(rf["frac_hbm"], 3)
        return r
    def eval_line(ev):                                      # (a variant that failed is recorded as {"error": ...}: then null, not a crash)
        l = line(ev, "us_per_tick")
        if l is None:
            return None
        for k_out, k_in in (("red_win_rate", "win_rate_red"), ("red_win_rate_stale_first_obs", "win_rate_red_stale_first_obs")):
            if isinstance(ev.get(k_in), (int, float)):
                l[k_out] = round(ev[k_in], 4)
        return l
    def c3_default():
        l = line(by(ch, "4v4, 3 chains"), "us_per_step")
        if l is not None:
            l["form"] = "3 chains of per-step launches in one graph (capture_steps default, chains='auto'); C3_one_launch: one launch per step"
        return l
    head = {"agent_steps_per_s": round(out["value"]), "us": round(km * 1e3, 3), "frac_claimed": frac_claimed and round(frac_claimed, 3)}
    ow, ro, ch = out.get("other_workloads") or {}, pick(out, "policy_rollouts", "variants") or {}, out.get("chained_graphs") or {}
    cb = out.get("cpu_baseline") or {}
    by = lambda d, word: next((v for k, v in d.items() if word in k), None)     # noqa: E731
    ev = by(ro, "reference evaluation workload (")
    s = {"C1_cpu_port_1core": cb.get("value"), "C1_cpu_port_4v4_1core": pick(cb, "port_4v4", "value"),
         "C1_gpu_dropin_1game": pick(out, "drop_in_one_game", "agent_steps_per_s"),
         "C2": head if (n, E, world) == (1, 65536, 1) else None,
         # configs[2] in the form capture_steps() takes by default there (chains="auto": 3 chains at 4v4) and as ONE launch per step
         "C3": c3_default(),
         "C3_one_launch": head if (n, E, world) == (4, 65536, 1) else line(by(ow, "configs[2]")),
         "C4": head if (n, E, world) == (1, 65536, 8) else None,
         "C5_graph": line(by(ro, "graph of 2 kernels per tick, both"), "us_per_tick"),
         "C5_one_launch": line(by(ro, "one launch for all ticks, both"), "us_per_tick"),
         "C5_one_launch_bf16x6": line(by(ro, "six bf16"), "us_per_tick"),
         "C5_ppo_one_launch": line(by(ro, "PPO-shaped rollout, one launch: "), "us_per_tick"),
         "eval_2v2": eval_line(ev),
         "1M_1v1": line(by(ow, "1048576")),
         "multi_tick_C2": line(out.get("multi_tick_launch"), "us_per_tick")}
    if world > 1 and s["C4"] is None:
        s[f"N{world}_x_{E}_{n}v{n}"] = head
    return s


MFMA_PEAK_F32_TFLOPS = 157.3                          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, f32 in / f32 accumulate (= the vector peak)
MFMA_PEAK_BF16_TFLOPS = 2500.0                        # MI355X_MICROARCH.md: dense bf16 matrix peak


def rollout_roofline(us_per_tick, rows, obs, n_out, precision="f32", heads=1, traffic_key=None):
    """Roofline of one rollout tick (actor -> step): the actor is the one GEMM-shaped op on this path (maddpg/networks.py:54-85:
    obs -> 64 -> LayerNorm -> 64 -> LayerNorm -> n_out per plane), so it has a flops figure; the step has the bytes.
      flops_per_tick   algorithmic: rows x heads x 2 x (obs*64 + 64*64 + 64*n_out), rows = planes acting through a network
      frac_mfma        the time the matrix cores need for it at their dense peaks / the tick: f32 layers against 157.3 TFLOP/s; in the
                       split-bf16 forms the 64 x 64 layer runs as 3 / 6 bf16 products, priced as executed against 2 500 TFLOP/s
      bytes_per_tick   HBM bytes per tick by PMC (profiles/traffic.json[rollout_*], tools/rollout_traffic.py), frac_hbm against 8 TB/s
      bound            "mfma" / "hbm" where that fraction is at least one half, else "issue/latency"; frac = the larger fraction."""
    edge, mid = 2.0 * rows * heads * (obs * 64 + 64 * n_out), 2.0 * rows * heads * 64 * 64
    k = {"f32": 0, "bf16x3": 3, "bf16x6": 6}[precision]
    t_mfma = ((edge + mid) if k == 0 else edge) / (MFMA_PEAK_F32_TFLOPS * 1e12) + (k * mid / (MFMA_PEAK_BF16_TFLOPS * 1e12) if k else 0.0)
    t = us_per_tick * 1e-6
    r = {"flops_per_tick": round(edge + mid), "mfma_peak_tflops": MFMA_PEAK_F32_TFLOPS if k == 0 else {"f32_layers": MFMA_PEAK_F32_TFLOPS, "bf16_64x64": MFMA_PEAK_BF16_TFLOPS},
         "achieved_tflops": round((edge + mid) / t / 1e12, 2), "frac_mfma": round(t_mfma / t, 4), "bytes_per_tick": None, "frac_hbm": None}
    try:
        te = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(traffic_key) if traffic_key else None
    except Exception:                                       # noqa: BLE001
        te = None
    if te and te.get("hbm_bytes_per_tick"):
        r.update(bytes_per_tick=te["hbm_bytes_per_tick"], frac_hbm=round(te["hbm_bytes_per_tick"] / t / 1e9 / HBM_PEAK_GBS, 4),
                 traffic_source=f"profiles/traffic.json[{traffic_key}] (series {te.get('series')})")
    fh = r["frac_hbm"] or 0.0
    r["frac"] = max(r["frac_mfma"], fh)
    r["bound"] = "mfma" if (r["frac_mfma"] >= 0.5 and r["frac_mfma"] >= fh) else ("hbm" if fh >= 0.5 else "issue/latency")
    return r


def rollout_lines(dev, E, K):
    """BASELINE.json configs[4]: 65 536 games x 1v1 with the policy in the loop, end-to-end agent-steps/s.  The caller's loop
    `actions = actor(obs) + noise; obs, rew, done = env.step(actions)` (reference main.py:177-181, maddpg/agent.py:25-33) on
    device: as a HIP graph of two kernels per tick (bsx_actor_forward -> bsx_step_*), and as ONE launch for all T ticks
    (bsx_rollout_*); actors are random-init networks of the reference's shape, Gaussian exploration noise 0.1."""
    import torch
    import deep_rl_battlespace_amd as bsx
    from deep_rl_battlespace_amd import instinct
    from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
    T = 32
    out = {}
    tkeys = {0: "rollout_graph", 1: "rollout_one_launch", 2: "rollout_one_launch_bf16x6", 3: "rollout_one_launch_bf16x3", 4: "rollout_scripted_blue"}
    variants = [("graph of 2 kernels per tick, both teams on actors", dict(), False, False),
                ("one launch for all ticks, both teams on actors", dict(one_launch=True), False, False),
                ("one launch, 64x64 layer as six bf16 matrix products of three-term splits (float32-class accuracy)", dict(one_launch=True, precision="bf16x6"), False, False),
                ("one launch, 64x64 layer as three bf16 matrix products of two-term splits (~1e-5 on a score)", dict(one_launch=True, precision="bf16x3"), False, False),
                ("one launch, red = actor vs blue = scripted instinct opponent (main.py:119-122)", dict(one_launch=True), True, False),
                ("PPO-shaped rollout, one launch: categorical draw from softmax(scores) + log-prob + value head per plane", dict(one_launch=True, sample="categorical", value=True), False, False),
                ("PPO-shaped rollout, graph of 2 kernels per tick: categorical draw + log-prob + value head", dict(sample="categorical", value=True), False, False),
                ("PPO-shaped rollout, one launch, both MLPs' 64x64 layers as three bf16 matrix products of two-term splits", dict(one_launch=True, sample="categorical", value=True, precision="bf16x3"), False, False),
                ("continuous actions: graph of 2 kernels per tick", dict(), False, True),
                ("continuous actions: one launch for all ticks", dict(one_launch=True), False, True)]
    for vi, (tag, kw, scripted, cont) in enumerate(variants):
        try:
            env = bsx.parallel_env(n_agents=1, n_envs=E, auto_reset=True, seed=1234, device=dev, continuous_actions=cont)
            env.reset()
            torch.manual_seed(0)
            actor = StackedActor(2, 5, 3 if cont else 4, device=dev)
            with torch.no_grad():
                actor.w3.mul_(100.0)                        # random init leaves the head near 0: spread the scores so that play is varied
            opp = instinct.Team(env.possible_blue, env.possible_red, env) if scripted else None
            kw = dict(kw)
            heads = 2 if kw.get("value") else 1
            if kw.pop("value", False):                      # a value head: a second MLP of the actor's shape with one output per plane
                critic = StackedActor(2, 5, 1, device=dev)
                with torch.no_grad():
                    critic.w3.mul_(100.0)
                kw["value_actor"] = critic
            ro = PolicyRollout(env, actor, T, noise_std=0.0 if "sample" in kw else 0.1, opponent=opp, **kw)
            ro.start(); ro.capture()
            reps = max(1, K // T)
            for _ in range(3 + 150 // T):                   # past the first time-limit ties
                ro.run()
            samples = []
            for _ in range(3):
                torch.cuda.synchronize(dev); t0 = time.perf_counter()
                for _ in range(reps):
                    ro.run()
                torch.cuda.synchronize(dev)
                samples.append((time.perf_counter() - t0) / (reps * T))
            dt = statistics.median(samples)
            c = env.counters().sum(0)
            out[tag] = {"agent_steps_per_s": round(E * 2 / dt, 1), "us_per_tick": round(dt * 1e6, 3), "ticks_per_launch_or_graph": T,
                        "ticks_timed": reps * T, "repeats": 3, "games_finished": int(c[0]), "ties": int(c[1]), "red_wins": int(c[2]), "blue_wins": int(c[3]),
                        # (a value head is a second MLP with ONE output: priced with the actor's n_out, < 4 % more flops than it has)
                        "roofline": rollout_roofline(dt * 1e6, E * (1 if scripted else 2), 5, 3 if cont else 4, kw.get("precision", "f32"), heads, tkeys.get(vi))}
            del ro, env, actor
        except Exception as exc:                            # a variant this build does not offer is reported, not hidden
            out[tag] = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
        torch.cuda.empty_cache()
    for tag, prec in (("reference evaluation workload (evaluate.py:32-76): 2v2, shipped checkpoints vs scripted instinct team", "f32"),
                      ("reference evaluation workload, the checkpoints' 64x64 layers as six bf16 matrix products of three-term splits (float32-class accuracy)", "bf16x6")):
        try:
            out[tag] = evaluation_line(dev, E, prec)
        except Exception as exc:
            out[tag] = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
    return {"workload": f"BASELINE.json configs[4]: {E} games x 1v1 + on-device actor per plane (obs 5 -> 64 -> LayerNorm -> 64 -> LayerNorm -> 4, "
                        "maddpg/networks.py:54-85), end to end", "variants": out}


def evaluation_line(dev, E, precision="f32"):
    """The reference's own evaluation workload on this path: 2v2, reward config models/completed_model/cf.json, red = the shipped
    checkpoints actor_plane0 / actor_plane1 (weights recorded in tests/golden/g12_evaluation.npz: data, not code) with
    Ornstein-Uhlenbeck noise 0.1 that is never restarted, blue = the scripted instinct team in-kernel, one launch per 32 ticks;
    the red win rate beside the tally the unmodified evaluate.main() produced in the build container and README.md:30's figure."""
    import numpy as np
    import torch
    import deep_rl_battlespace_amd as bsx
    from deep_rl_battlespace_amd.rollout import play_reference_evaluation, reference_checkpoint_actor
    g12 = np.load(os.path.join(ROOT, "tests", "golden", "g12_evaluation.npz"))
    cf = dict(zip(("hit_base_reward", "hit_plane_reward", "miss_punishment", "die_punishment", "lose_punishment"), (float(v) for v in g12["cf"])))
    n = int(g12["n_agents"])
    actor = reference_checkpoint_actor(g12, n, device=dev)
    env = bsx.parallel_env(n_agents=n, n_envs=E, auto_reset=True, seed=1234, device=dev, **cf)
    # the tally: exactly the first 3 games of every slot (whole games, as the script plays them one after another; stopping all slots
    # at one moment would over-count short games), looked at every 4 ticks; the timing below: 32 ticks per launch
    res = play_reference_evaluation(env, actor, games=0, T=4, one_launch=True, seed=12, precision=precision, games_per_slot=3)
    res.pop("rollout")
    res["tally"] = "the first 3 games of each of the %d slots" % E
    ro = play_reference_evaluation(env, actor, games=1, T=32, one_launch=True, seed=13, precision=precision).pop("rollout")
    torch.cuda.synchronize(dev)
    samples = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(8):
            ro.run()
        torch.cuda.synchronize(dev)
        samples.append((time.perf_counter() - t0) / (8 * 32))
    dt = statistics.median(samples)
    ref = {k: int(g12[k]) for k in ("games", "ties", "red_wins", "blue_wins")}
    # the same workload WITH the script's quirk (evaluate.py:53-66: a game's first tick sees the observations of a discarded reset),
    # tick by tick; and how far each tally is from the reference's own, in standard deviations of ITS sample (3 018 games)
    p_ref = ref["red_wins"] / ref["games"]
    sigma = (p_ref * (1 - p_ref) / ref["games"]) ** 0.5
    stale = None
    if precision == "f32":
        del ro
        env2 = bsx.parallel_env(n_agents=n, n_envs=E, auto_reset=True, seed=1234, device=dev, **cf)
        stale = play_reference_evaluation(env2, actor, games=0, T=4, seed=12, precision=precision, first_tick_stale_obs=True, games_per_slot=3)
        stale.pop("rollout")
        stale["sigmas_from_reference_tally"] = round((stale["win_rate_red"] - p_ref) / sigma, 2)
        res["win_rate_red_stale_first_obs"] = round(stale["win_rate_red"], 5)
    res["sigmas_from_reference_tally"] = round((res["win_rate_red"] - p_ref) / sigma, 2)
    res["with_the_scripts_stale_first_observation"] = stale
    return {**res, "agent_steps_per_s": round(E * 2 * n / dt, 1), "us_per_tick": round(dt * 1e6, 3), "games_per_s": round(res["games"] / (res["ticks"] * dt), 1),
            "roofline": rollout_roofline(dt * 1e6, E * n, 3 * n + 2, 4, precision),
            "envs": E, "n_agents_per_team": n, "ticks_per_launch": 32,
            "reference_tally_evaluate_py": {**ref, "win_rate_red": round(ref["red_wins"] / ref["games"], 4),
                                            "source": "tests/golden/g12_evaluation.npz: evaluate.main() unmodified, run in the build container"},
            "published_win_rate": "~80% (README.md:30)"}


if __name__ == "__main__":
    main()
