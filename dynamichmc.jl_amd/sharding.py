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


class TorchAllReduce:
    """In-place SUM of a tensor over the ranks of a torch.distributed group — what DeviceContext.set_metric_allreduce wants.
    backend "nccl" (= RCCL over xGMI) reduces the CUDA tensor where it lies; with "gloo" (the CPU tests, and several ranks
    sharing one GPU) the tensor makes the round trip through host memory."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group

    def __call__(self, t):
        backend = self.dist.get_backend(self.group)
        if t.is_cuda and backend != "nccl":
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t


def pooled_covariance(draws, allreduce=None):
    """The job-wide estimate of the shared dense metric, spelled out on the host (numpy): what dhmc_update_metric_dense computes
    on the device once an all-reduce is installed (include/dhmc.h) — column sums and row counts added over the ranks, the scatter
    about the job's mean added over the ranks, Σ = S / (J_total - 1).  `draws`: this rank's [C_local][N][D] (or [J][D]);
    `allreduce(array)` adds a float64 numpy array over the ranks in place (None: one rank).  The protocol's reference for the
    tests; the device path uses the same two collectives."""
    x = np.asarray(draws, np.float64).reshape(-1, np.shape(draws)[-1])
    buf = np.concatenate([x.sum(0), [float(x.shape[0]), 0.0]])      # column sums, row count, error slot (include/dhmc.h)
    if allreduce is not None:
        allreduce(buf)
    if buf[-1] != 0.0:
        raise RuntimeError("a rank of the job failed before the estimate")
    J = buf[-2]
    mean = buf[:-2] / J
    d = x - mean
    S = d.T @ d
    if allreduce is not None:
        allreduce(S)
    return S / (J - 1), mean, int(J)
