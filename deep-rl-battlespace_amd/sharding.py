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
    blocks.  chains="auto": by what was measured (profiles/r03_4v4_issue_bound.json) -- nothing to gain at 1v1 or below ~260 k agents
    per step (short launches: a multi-branch graph's bookkeeping, ~1 us per step, costs more than it hides), else 3 chains at 4v4, 2
    for every other team size."""
    blocks = -(-int(n_envs) // 256)
    if chains == "auto":
        chains = {1: 1, 4: 3}.get(int(n_agents_per_team), 2) if int(n_envs) * 2 * int(n_agents_per_team) >= (1 << 18) else 1
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
