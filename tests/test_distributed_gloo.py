"""world_size-2 run of the multi-GPU layout on CPU (gloo): chains are cut into contiguous blocks
with the block offset as RNG chain_offset, each rank runs its block (the CPU oracle stands in
for the per-rank engine here — there is no GPU in this container), results are gathered, and
the gathered job equals the single-process job chain for chain (partition independence)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, total, D, N, outdir):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import oracle_lib as ol
    from __graft_entry__ import load_package
    pkg = load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = pkg.sharding.shard_chains(total, world, rank)
    eng = ol.Oracle(D, cnt, seed=77, chain_offset=off)
    eng.init(); eng.find_initial_stepsize()
    r = eng.run(N, da={})
    draws = pkg.sharding.gather_chain_major(torch.from_numpy(r["draws"]), dist, total, world)
    steps = pkg.sharding.gather_chain_major(torch.from_numpy(r["steps"]), dist, total, world)
    rate = pkg.sharding.job_throughput(int(r["steps"].sum()), 2.0 + rank, dist)
    if rank == 0:
        np.savez(os.path.join(outdir, "gathered.npz"), draws=draws.numpy(), steps=steps.numpy(), rate=rate)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_equals_single_process(tmp_path):
    sys.path.insert(0, HERE)
    import oracle_lib as ol
    total, D, N, world = 5, 24, 12, 2      # ragged: blocks of 3 and 2 chains
    mp.spawn(_worker, args=(world, _free_port(), total, D, N, str(tmp_path)), nprocs=world, join=True)
    g = np.load(tmp_path / "gathered.npz")
    single = ol.Oracle(D, total, seed=77)
    single.init(); single.find_initial_stepsize()
    r = single.run(N, da={})
    assert np.array_equal(g["draws"], r["draws"])
    assert np.array_equal(g["steps"], r["steps"])
    assert np.isclose(float(g["rate"]), r["steps"].sum() / 3.0)   # Σ units / max seconds over ranks


def _pool_worker(rank, world, port, total, N, D, outdir):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    from __graft_entry__ import load_package
    pkg = load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = np.random.default_rng(3).normal(size=(total, N, D)) * np.linspace(0.5, 3, D) + np.linspace(-2, 2, D)
    off, cnt = pkg.sharding.shard_chains(total, world, rank)
    red = pkg.sharding.TorchAllReduce(dist)

    def allreduce(a):                         # numpy array, in place, through the same wrapper the device path gets
        t = torch.from_numpy(a)
        red(t)
    S, mean, J = pkg.sharding.pooled_covariance(x[off:off + cnt], allreduce)
    np.savez(os.path.join(outdir, f"pool{rank}.npz"), S=S, mean=mean, J=J)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pooled_metric_estimate_is_the_single_process_estimate(tmp_path):
    """SURVEY §8e / include/dhmc.h dhmc_set_metric_allreduce: the shared dense metric of a sharded job is estimated from the draws
    of ALL ranks by two collectives (column sums + row count, then the scatter about the job's mean).  The protocol, on the host
    (sharding.pooled_covariance, the device path's reference) with world size 2 over gloo and ragged blocks: both ranks get the
    same matrix, equal to numpy's covariance of all draws to rounding."""
    sys.path.insert(0, ROOT)
    total, N, D, world = 7, 40, 9, 2
    mp.spawn(_pool_worker, args=(world, _free_port(), total, N, D, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "pool0.npz"), np.load(tmp_path / "pool1.npz")
    assert np.array_equal(a["S"], b["S"]) and np.array_equal(a["mean"], b["mean"]) and int(a["J"]) == total * N
    x = np.random.default_rng(3).normal(size=(total, N, D)) * np.linspace(0.5, 3, D) + np.linspace(-2, 2, D)
    X = x.reshape(-1, D)
    assert np.allclose(a["mean"], X.mean(0), rtol=1e-13, atol=1e-13)
    assert np.allclose(a["S"], np.cov(X.T), rtol=1e-12, atol=1e-13)
