"""Query-block sharding across the GPUs of one node (one process per GPU, torch.distributed; backend 'nccl' = RCCL over
xGMI on the GPU box, 'gloo' in the CPU tests).

Queries are independent given the replicated per-shape state (cloud + per-point table, ~104 MB), so the data path needs a
single exchange per growth round: a variable-length all-gather of 4 bytes per query."""
import torch


_QUERY_SHARDING = False


def set_query_sharding(on: bool):
    """Query-block sharding inside one shape is opt-in: with shape-level sharding the ranks work on different shapes and
    must not meet in a collective."""
    global _QUERY_SHARDING
    _QUERY_SHARDING = bool(on)


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, world_size: int):
    """Contiguous, balanced [lo, hi) slice of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_map(fn, items: torch.Tensor) -> torch.Tensor:
    """Every rank evaluates fn on its contiguous slice of `items` [n, ...] (fn returns one float32 per item) and all
    ranks receive the concatenated result [n] in the original order."""
    import torch.distributed as dist
    rank, ws = world()
    n = items.shape[0]
    if ws == 1 or not _QUERY_SHARDING:
        return fn(items)
    lo, hi = shard_range(n, rank, ws)
    local = fn(items[lo:hi]).to(torch.float32).contiguous()
    width = -(-n // ws)
    pad = torch.zeros((width,), dtype=torch.float32, device=local.device)
    pad[:hi - lo] = local
    parts = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad)
    out = [parts[r][:shard_range(n, r, ws)[1] - shard_range(n, r, ws)[0]] for r in range(ws)]
    return torch.cat(out)


def max_over_ranks(seconds: float, device) -> float:
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_latents(latent_sum: torch.Tensor, counts: torch.Tensor):
    """Sum the per-rank partial latent sums / counts of a latent loop whose encoder passes were dealt round-robin."""
    import torch.distributed as dist
    _, ws = world()
    if ws > 1:
        dist.all_reduce(latent_sum, op=dist.ReduceOp.SUM)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return latent_sum, counts
