"""Multi-GPU layout of the step() path: independent games shard trivially -- one contiguous env range per rank, one
process per GPU, NO collective on the step path.  In-kernel randomness is keyed by the GLOBAL env
index (`env_offset + e`), so a job's games are the same games however many GPUs it is spread over.

The only communication offered is an optional all-reduce of a handful of int64 game counters for logging
(torch.distributed: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" on CPU in the tests); the payload is
~32 bytes, pure latency, and it is never issued by step()."""
import os

import torch


def rank_world():
    """(rank, world_size, local_rank) from the torch.distributed.run environment; (0, 1, 0) when launched plainly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def env_range(total_envs, rank, world_size):
    """Contiguous range [lo, hi) of global env indices owned by `rank`; sizes differ by at most one."""
    if total_envs < 0 or world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad shard request")
    q, r = divmod(total_envs, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def chain_ranges(n_envs, n_agents_per_team, chains):
    """The batch as contiguous game ranges [(first, count), ...] in whole 256-game blocks -- what bsx_step_*_range takes and
    parallel_env.capture_steps(chains=) runs as independent chains of launches; fewer ranges than asked when there are not that many
    blocks.  chains="auto" (capture_steps' default) takes chains only where they do not lose in EITHER use of the graph -- replays queued
    back to back (what the chains were built for: profiles/r03_4v4_issue_bound.json, r06_chains_1v1.json), or every replay synchronised.
    A multi-branch graph is launched node by node at ~6 us each, which a synchronised replay pays for in full: chained steps run at
    (chains x ~6 us) per step at best.  So: 3 chains at 4v4 (back to back 20.2 -> 16.5 us per step, synchronised 21.0 -> 20.9), 2 for
    3v3 and for 5v5 ... 16v16 (steps of 17 ... 97 us), from ~260 k agents per step; 1v1 from 1 M games (-4 ... -5 % in both uses);
    NOT 2v2 (back to back 9.7 -> 8.6, synchronised 9.5 -> 10.8) and not 1v1 below that (196 608 games: 10.3 -> 8.9 against 10.7 -> 12.0):
    ask for those (chains=2) when replays are queued."""
    blocks = -(-int(n_envs) // 256)
    if chains == "auto":
        n, agents = int(n_agents_per_team), int(n_envs) * 2 * int(n_agents_per_team)
        if n == 1:
            chains = 2 if agents >= (1 << 21) else 1
        elif n == 2:
            chains = 1
        else:
            chains = {4: 3}.get(n, 2) if agents >= (1 << 18) else 1
    chains = max(1, min(int(chains), blocks))
    cuts = [(blocks * r // chains) * 256 for r in range(chains)] + [int(n_envs)]
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(chains)]


def make_shard(total_envs, rank=None, world_size=None, **env_kwargs):
    """Construct this rank's `parallel_env` over its env range (env_offset = first global index)."""
    from .envs.battle_env import parallel_env
    if rank is None or world_size is None:
        rank, world_size, _ = rank_world()
    lo, hi = env_range(total_envs, rank, world_size)
    return parallel_env(n_envs=hi - lo, env_offset=lo, **env_kwargs)


def reduce_counters(counters, group=None):
    """Sum per-rank game counters (games, ties, red wins, blue wins -- any int64 vector) over all ranks.
    `counters`: 1-D int64 tensor on the backend's device (cuda for nccl/RCCL, cpu for gloo).  Off the step stream,
    at logging cadence only.  Without an initialised process group this is the identity."""
    import torch.distributed as dist
    t = counters.clone()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def local_counter_sums(env):
    """int64 [4] on env.device: this shard's (games, ties, red wins, blue wins)."""
    c = env.export_state(("counters",))["counters"]
    return c.to(torch.int64).sum(0)


# ---------------------------------------------------------------------------------------------- host cores of a multi-rank job
def _parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist) -> sorted list of ints."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(out))


def card_numa_nodes(sysfs="/sys"):
    """NUMA node of every AMD card, in PCI-address order (the order the runtime numbers its devices in when no *_VISIBLE_DEVICES
    variable re-orders them), read from /sys/class/drm/card*/device/numa_node WITHOUT touching the GPU; [] when not readable.
    A card whose node reads -1 (no affinity reported) is listed as None."""
    import glob
    cards = {}
    for d in glob.glob(os.path.join(sysfs, "class", "drm", "card[0-9]*", "device")):
        try:
            if open(os.path.join(d, "vendor")).read().strip() != "0x1002":
                continue
            node = int(open(os.path.join(d, "numa_node")).read().strip())
            cards[os.path.basename(os.path.realpath(d))] = node if node >= 0 else None
        except (OSError, ValueError):
            continue
    return [cards[k] for k in sorted(cards)]


def affinity_blocks(world_size, allowed, card_nodes=None, node_cpus=None):
    """Disjoint blocks of host cores, one per local rank -> ([sorted core list] * world_size, "numa" | "plain").

    With the cards' NUMA nodes known (card_nodes[r] = node of rank r's card, node_cpus[node] = that node's cores) the ranks whose cards
    share a node split THAT node's allowed cores evenly, in rank order; otherwise -- nothing readable, a card without a node, a node
    with fewer allowed cores than ranks on it -- the allowed cores are cut into world_size consecutive blocks of
    len(allowed) // world_size.  A pure function of its arguments: every rank computes the same table, so the blocks are disjoint
    without any communication.  Fewer allowed cores than ranks: every rank keeps all of them (nothing to separate)."""
    allowed = sorted(set(allowed))
    W = int(world_size)
    if W < 1:
        raise ValueError("world_size < 1")
    if len(allowed) < W:
        return [list(allowed) for _ in range(W)], "plain"
    plain = [allowed[r * (len(allowed) // W):(r + 1) * (len(allowed) // W)] for r in range(W)], "plain"
    if not card_nodes or not node_cpus or len(card_nodes) < W or any(card_nodes[r] is None for r in range(W)):
        return plain
    out = [None] * W
    for node in sorted(set(card_nodes[:W])):
        ranks = [r for r in range(W) if card_nodes[r] == node]
        cores = [c for c in node_cpus.get(node, []) if c in set(allowed)]
        per = len(cores) // len(ranks)
        if per < 1:
            return plain
        for i, r in enumerate(ranks):
            out[r] = cores[i * per:(i + 1) * per]
    return out, "numa"


def pin_rank_to_its_cores(local_rank, world_size, same_card_for_all=False, sysfs="/sys"):
    """Pin THIS process (and every thread it starts afterwards: the HIP runtime's, torch's) to its block of host cores.  To be called
    before the first GPU call of a rank.  -> dict(cores=[...], numa_node=int|None, how="numa"|"plain"|"unpinned: <why>").
    world_size 1: the process keeps what it had (cores = all allowed)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError) as exc:
        return {"cores": None, "numa_node": None, "how": f"unpinned: {type(exc).__name__}"}
    if world_size <= 1:
        return {"cores": allowed, "numa_node": None, "how": "unpinned: one rank keeps the cores it was given"}
    nodes = card_numa_nodes(sysfs)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):     # a visible-devices list re-numbers the cards
        v = os.environ.get(var)
        if v:
            try:
                nodes = [nodes[int(t)] for t in v.split(",")]
            except (ValueError, IndexError):
                nodes = []
            break
    if same_card_for_all and nodes:
        nodes = [nodes[0]] * world_size
    node_cpus = {}
    for nd in set(x for x in nodes if x is not None):
        try:
            node_cpus[nd] = _parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{nd}", "cpulist")).read())
        except (OSError, ValueError):
            nodes = []
            break
    blocks, how = affinity_blocks(world_size, allowed, nodes, node_cpus)
    mine = blocks[local_rank]
    try:
        os.sched_setaffinity(0, set(mine))
    except OSError as exc:
        return {"cores": allowed, "numa_node": None, "how": f"unpinned: {exc}"}
    return {"cores": mine, "numa_node": nodes[local_rank] if how == "numa" else None, "how": how}


# ---------------------------------------------------------------------------------------------- the timing group of a multi-rank job
def _rccl_probe(device, world_size, timeout_s):
    """Create the RCCL (backend "nccl") group next to the gloo control group and prove it with one all-reduce.  Blocking and
    collective: it returns only when EVERY rank has joined, so it is run on a worker thread (init_timing_group)."""
    import datetime
    import torch.distributed as dist
    g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s))
    probe = torch.ones(1, device=device)
    dist.all_reduce(probe, group=g)
    torch.cuda.synchronize(device)
    if int(probe.item()) != world_size:
        raise RuntimeError(f"all_reduce probe returned {probe.item()} for world size {world_size}")
    return g


def _rccl_preflight(device):
    """What a rank can check alone, before anything collective: the backend is compiled in, its card is there and takes work."""
    import torch.distributed as dist
    if not dist.is_nccl_available():
        raise RuntimeError("torch.distributed was built without the nccl (RCCL) backend")
    if not torch.cuda.is_available() or device.index >= torch.cuda.device_count():
        raise RuntimeError(f"{device} is not visible to this rank")
    torch.ones(1, device=device).add_(1)
    torch.cuda.synchronize(device)


def init_timing_group(backend, device, timeout_s=180.0, probe_wait_s=90.0, poll_s=0.05):
    """Process groups of a multi-rank bench job -> (group for the barrier / timing reductions, backend in use, note, clean).

    The step path has no collective; the groups carry a barrier and a few tiny reductions.  gloo always comes up first, as the
    default (control) group.  With backend "nccl" every rank then (1) checks alone what it can check alone, (2) runs the collective
    RCCL probe on a worker thread while its main thread watches the control store, and the ranks AGREE on the outcome before
    anyone proceeds: a rank that fails says so through the store at once, every rank that sees a failure (its own, a peer's, or
    no result within probe_wait_s) stops waiting, and a MIN all-reduce over gloo makes the decision the same everywhere -- all
    ranks time over RCCL, or all fall back to gloo together within seconds; never a mixture (the ranks that succeeded would sit
    in their next RCCL collective until its timeout).  clean = False says that something of RCCL was left behind in this process -- a
    probe thread blocked inside it, or a communicator this rank created that the job does not adopt: the caller must leave through
    os._exit() once its line is printed (no destructors, no joins)."""
    import datetime
    import threading
    import time
    import torch.distributed as dist
    rank, world, _ = rank_world()
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))
    if backend != "nccl":
        return None, "gloo", None, True
    try:
        from torch.distributed.distributed_c10d import _get_default_store
        store = dist.PrefixStore("bsx_rccl_probe", _get_default_store())
    except Exception:                                       # noqa: BLE001 -- no store to publish through: the timeout alone bounds the wait
        store = None
    box = {}

    def attempt():
        try:
            if device.type == "cuda":
                torch.cuda.set_device(device)               # (the current device is per thread)
            _rccl_preflight(device)
            # (the probe group's own timeout is no longer than the wait below: a probe that is abandoned resolves -- RCCL's watchdog
            # gives up on it -- before a measurement of the fallback path could be cut short by that watchdog's process abort)
            box["group"] = _rccl_probe(device, world, min(timeout_s, probe_wait_s))
        except BaseException as exc:                        # noqa: BLE001 -- whatever RCCL raised, the fallback is the same
            box["error"] = f"{type(exc).__name__}: {str(exc)[:160]}"
            if store is not None:
                try:
                    store.add("failed", 1)
                except Exception:                           # noqa: BLE001
                    pass
    th = threading.Thread(target=attempt, name="bsx-rccl-probe", daemon=True)
    th.start()
    t0 = time.time()
    seen_peer_failure = False
    while th.is_alive() and time.time() - t0 < probe_wait_s:
        th.join(poll_s)
        if store is not None and th.is_alive():
            try:
                seen_peer_failure = store.add("failed", 0) > 0
            except Exception:                               # noqa: BLE001
                store = None
            if seen_peer_failure:
                break
    mine_ok = (not th.is_alive()) and "group" in box
    agreed = torch.tensor([1 if mine_ok else 0], dtype=torch.int32)
    dist.all_reduce(agreed, op=dist.ReduceOp.MIN)           # over gloo: every rank leaves with the same decision
    if int(agreed.item()) == 1:
        return box["group"], "nccl", None, True
    why = box.get("error") or ("a peer rank reported a failure" if seen_peer_failure else
                               ("this rank's probe passed, another rank's did not" if mine_ok else f"no result within {probe_wait_s:.0f} s"))
    note = f"nccl (RCCL) did not come up on every rank ({why}); all ranks use gloo"
    # clean only if nothing of RCCL is left in this process: no probe thread blocked inside it, and no communicator that was created
    # here but is not going to be used (its peers failed: tearing it down at exit could wait for them)
    return None, "gloo", note[:200], (not th.is_alive()) and "group" not in box
