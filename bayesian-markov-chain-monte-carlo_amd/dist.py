"""
Multi-GPU sharding of independent chains.

Chains never exchange data while sampling, so the path shards with NO data-path collective:
rank r owns the contiguous global chain ids [r*C/G, (r+1)*C/G) and runs them on its own GPU.
The Philox counter is keyed by the GLOBAL chain id, hence the pooled result is bit-identical
for any G.  The only collective is one all-gather of the post-burn-in sample block at the end
(`torch.distributed.all_gather_into_tensor`: RCCL over xGMI with backend "nccl", gloo on CPU in
the test-suite) — the "posterior pool".  One process per GPU; launch with torch.distributed.run.
"""
import os

import numpy as np


def shard_bounds(n_chains, world_size, rank):
    """Contiguous block of global chain ids owned by `rank` (n_chains must divide evenly)."""
    if n_chains % world_size:
        raise ValueError(f"n_chains={n_chains} is not divisible by world_size={world_size}")
    per = n_chains // world_size
    return rank * per, per


def init_process_group(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them)."""
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(1, torch.cuda.device_count()))
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def allgather_pool(local, group=None):
    """local: tensor (n_keep, C_local, d) on this rank → (G, n_keep, C_local, d) on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    local = local.contiguous()
    # concatenated-along-dim-0 output: the one layout both RCCL and gloo accept
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out.view((world,) + tuple(local.shape))


def comm_init_from_process_group(engine, group=None, device=None):
    """Create `engine`'s own RCCL communicator (C ABI: rsf_comm_unique_id / rsf_comm_init) for the ranks of a
    torch.distributed group.  torch.distributed is only the side channel that carries rank 0's 128-byte id; the
    collectives themselves (Engine.pool_allgather / pool_allreduce_sum) then run inside the library on the ctx
    stream, so a caller without PyTorch can do the same with any other way of moving 128 bytes."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1:
        engine.comm_init(1, 0, engine.comm_unique_id())  # a real one-rank communicator, same code path as world > 1
        return
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    uid = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        uid = torch.tensor(list(engine.comm_unique_id()), dtype=torch.uint8, device=device)
    dist.broadcast(uid, src=0, group=group)
    engine.comm_init(world, rank, bytes(uid.cpu().tolist()))


def allreduce_summary(local, group=None, device=None):
    """Summary path: combine per-rank `Engine.pool_summary` dicts {n, mean, var, min, max} into the
    global moments with three tiny all-reduces (SUM of n / sum / centred sum of squares, MIN, MAX) instead of
    materialising the pooled samples on every rank.  Returns the same dict for the union of all ranks' draws."""
    import torch
    import torch.distributed as dist

    n, mean, var = float(local["n"]), float(local["mean"]), float(local["var"])
    # sums about zero are combined in float64; the centred second moment is rebuilt with the parallel formula
    t = torch.tensor([n, n * mean], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    N, gmean = float(t[0]), float(t[1] / t[0])
    m2 = torch.tensor([var * (n - 1.0) + n * (mean - gmean) ** 2], dtype=torch.float64, device=device)
    dist.all_reduce(m2, op=dist.ReduceOp.SUM, group=group)
    lo = torch.tensor([float(local["min"])], dtype=torch.float64, device=device)
    hi = torch.tensor([float(local["max"])], dtype=torch.float64, device=device)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return dict(n=N, mean=gmean, var=float(m2[0]) / (N - 1.0) if N > 1 else 0.0, min=float(lo[0]), max=float(hi[0]))


def allreduce_histogram(counts, group=None, device=None):
    """Summary path: per-rank `Engine.pool_histogram` counts (exact integers in float64) → the histogram of the union of all
    ranks' draws, one all-reduce of nbins + 2 doubles (RCCL with backend "nccl", gloo in the CPU test-suite)."""
    import torch
    import torch.distributed as dist

    t = counts.clone() if isinstance(counts, torch.Tensor) else torch.as_tensor(np.array(counts, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def pool_to_chain_major(pool):
    """(G, n_keep, C_local, d) → (n_keep, G*C_local, d): global chain id = g*C_local + c."""
    G, n, C, d = pool.shape
    if isinstance(pool, np.ndarray):
        return np.ascontiguousarray(pool.transpose(1, 0, 2, 3)).reshape(n, G * C, d)
    return pool.permute(1, 0, 2, 3).reshape(n, G * C, d)


def run_single_process(engines, model, substeps, data, q0_global, lo, hi, n_iters, nburn, seed=0, mcmc_kwargs=None, thin=1):
    """SURVEY §8e's process model without a launcher: ONE host thread drives `engines` (one per GPU, typically
    `[Engine(mem="device", device=g) for g in range(n)]`).  Engine r samples the contiguous block r of global chain ids
    — device-memory engines only enqueue, so all GPUs compute at once — then the kept draws are pooled with one grouped
    RCCL all-gather through the C ABI (rsf_comm_init_all + rsf_pool_allgather_all).  The reference runs the same chains
    one after another (RSF.py:1042-1044).  Returns (pools, stats): pools[r] is the (n_keep, C, d) pool as engine r
    holds it (identical on every rank), stats[r] engine r's counters."""
    if __package__:
        from .engine import Engine
    else:
        from engine import Engine
    G = len(engines)
    q0_global = np.asarray(q0_global, dtype=np.float64)
    q0_global = q0_global.reshape(q0_global.shape[0], -1)
    kept = []
    for r, eng in enumerate(engines):
        off, per = shard_bounds(q0_global.shape[0], G, r)
        eng.set_model(model, substeps)
        eng.mcmc_init(q0_global[off:off + per], data, lo, hi, seed=seed, chain_offset=off, **(mcmc_kwargs or {}))
    for eng in engines:  # (a second loop: every init has been synchronous, the runs overlap across GPUs)
        tq, _, _ = eng.mcmc_run(n_iters, traces=("q",))
        kept.append(tq[max(nburn - 1, 0):][::thin])
    for eng in engines:
        eng.sync()
    made = not engines[0].world
    if made:
        Engine.comm_init_all(engines)
    try:
        pools = [pool_to_chain_major(p) for p in Engine.pool_allgather_all(engines, kept)]
    finally:
        if made:
            for eng in engines:
                eng.comm_destroy()
    return pools, [eng.stats() for eng in engines]


def run_sharded(engine_factory, model, substeps, data, q0_global, lo, hi, n_iters, nburn, seed=0, mcmc_kwargs=None,
                device=None, thin=1):
    """Each rank samples its shard, then the kept draws are all-gathered.

    engine_factory() → Engine for this rank (the product passes `lambda: Engine(mem="device")`;
    the CPU test-suite passes an oracle-backed factory).  q0_global: (C, d) start points for ALL
    chains, identical on every rank.  Returns (pool (n_keep, C, d) torch tensor, local stats)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    q0_global = np.asarray(q0_global, dtype=np.float64)
    q0_global = q0_global.reshape(q0_global.shape[0], -1)
    off, per = shard_bounds(q0_global.shape[0], world, rank)
    eng = engine_factory()
    try:
        eng.set_model(model, substeps)
        eng.mcmc_init(q0_global[off:off + per], data, lo, hi, seed=seed, chain_offset=off, **(mcmc_kwargs or {}))
        tq, _, _ = eng.mcmc_run(n_iters, traces=("q",))
        eng.sync()
        stats = eng.stats()
        kept = tq[max(nburn - 1, 0):][::thin]  # thin=k: every k-th kept draw goes into the pool
        if not isinstance(kept, torch.Tensor):
            kept = torch.from_numpy(np.ascontiguousarray(kept))
        if device is not None:
            kept = kept.to(device)
        pool = pool_to_chain_major(allgather_pool(kept))
    finally:
        eng.close()
    return pool, stats
