"""Multi-GPU layout of a many-chain job: independent chains are cut into contiguous blocks, one
per rank (= one process per GPU); nothing on the data path crosses ranks.  The reference has no
multi-device logic at all (its chains never interact: src/mcmc.jl:258-286); the only
collectives here gather results and timing over torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests)."""
import numpy as np


def shard_chains(total_chains, world_size, rank):
    """Contiguous block [offset, offset + count) of rank `rank`; the first `total % world` ranks get one more.
    The offset is the RNG `chain_offset`, so results do not depend on the partition."""
    base, rem = divmod(int(total_chains), int(world_size))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def gather_chain_major(local, dist, total_chains, world_size):
    """All-gather a chain-major tensor [C_local, ...] into [total_chains, ...] on every rank
    (ragged shards are padded to the largest block)."""
    import torch
    counts = [shard_chains(total_chains, world_size, r)[1] for r in range(world_size)]
    cmax = max(counts)
    pad = torch.zeros((cmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def job_throughput(local_units, local_seconds, dist=None, device="cpu"):
    """Whole-job rate: Σ units over ranks ÷ max seconds over ranks."""
    if dist is None:
        return local_units / local_seconds
    import torch
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u[0]) / float(t[0])
