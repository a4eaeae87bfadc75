"""
CPU test of the multi-GPU path with world_size 2 over gloo: chains are sharded by contiguous global
id, sampled with no data-path collective, and the kept draws are pooled with one all-gather.  Because
the RNG is keyed by the global chain id the pool must be bit-identical to a single-process run.
(The CPU oracle's engine stands in for the HIP engine; on GPUs the same code runs over RCCL.)
"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

WORLD = 2
C, N_ITERS, NBURN = 24, 12, 6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(pkg, lib, oracle_mod):
    model = oracle_mod.ModelSpec(500)
    with pkg.Engine(lib=lib) as e:
        e.set_model(model, 1)
        _, acc = e.forward([1000.0])
    acc = acc[:, 0]
    data = acc + np.abs(acc) * np.random.default_rng(2025).standard_normal(acc.shape[0])
    q0 = np.random.default_rng(1).uniform(800.0, 1200.0, (C, 1))
    return model, data, q0


def _worker(rank, port, out_dir):
    import ctypes

    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(WORLD), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import rsf_oracle
    import torch.distributed as dist

    import bayesian_markov_chain_monte_carlo_amd as pkg
    from bayesian_markov_chain_monte_carlo_amd import dist as rdist

    lib = pkg._abi.bind(ctypes.CDLL(rsf_oracle.lib_path()))
    assert rdist.init_process_group("gloo") == (rank, WORLD)
    model, data, q0 = _problem(pkg, lib, rsf_oracle)
    pool, stats = rdist.run_sharded(lambda: pkg.Engine(lib=lib, cpu_threads=2), model, 1, data, q0, [0.0], [1e4], N_ITERS, NBURN,
                                    seed=77, mcmc_kwargs=dict(prior_len=3))
    np.save(os.path.join(out_dir, f"pool_{rank}.npy"), pool.numpy())
    # summary path: per-rank moments of the local shard, combined with all-reduces
    per = C // WORLD
    with pkg.Engine(lib=lib) as e:
        local = e.pool_summary(pool[:, rank * per:(rank + 1) * per].contiguous().numpy())
    glob = rdist.allreduce_summary(local)
    np.save(os.path.join(out_dir, f"summary_{rank}.npy"), np.array([glob[k] for k in ("n", "mean", "var", "min", "max")]))
    with pkg.Engine(lib=lib) as e:  # ... and a fixed-bin histogram of the local shard, summed over ranks with one all-reduce
        hist = rdist.allreduce_histogram(e.pool_histogram(pool[:, rank * per:(rank + 1) * per].contiguous().numpy(), 16, 900.0, 1100.0))
    np.save(os.path.join(out_dir, f"hist_{rank}.npy"), hist.numpy())
    assert stats["iters_done"] == N_ITERS
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds(pkg):
    from bayesian_markov_chain_monte_carlo_amd import dist as rdist

    assert [rdist.shard_bounds(64, 4, r) for r in range(4)] == [(0, 16), (16, 16), (32, 16), (48, 16)]
    with pytest.raises(ValueError):
        rdist.shard_bounds(10, 4, 0)


@pytest.mark.timeout(300)
def test_two_rank_pool_equals_single_process(pkg, oracle_lib, oracle_mod, tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    pools = [np.load(tmp_path / f"pool_{r}.npy") for r in range(WORLD)]
    np.testing.assert_array_equal(pools[0], pools[1])  # every rank holds the whole pool
    model, data, q0 = _problem(pkg, oracle_lib, oracle_mod)
    with pkg.Engine(lib=oracle_lib) as e:
        e.set_model(model, 1)
        e.mcmc_init(q0, data, [0.0], [1e4], seed=77, chain_offset=0, prior_len=3)
        tq, _, _ = e.mcmc_run(N_ITERS, traces=("q",))
    x = pools[0].ravel()
    for r in range(WORLD):
        np.testing.assert_allclose(np.load(tmp_path / f"summary_{r}.npy"), [x.size, x.mean(), x.var(ddof=1), x.min(), x.max()], rtol=1e-12)
    ref, _ = np.histogram(x, 16, (900.0, 1100.0))
    for r in range(WORLD):
        h = np.load(tmp_path / f"hist_{r}.npy")
        np.testing.assert_array_equal(h, np.concatenate([[(x < 900.0).sum()], ref, [(x > 1100.0).sum()]]))
    assert pools[0].shape == (N_ITERS - NBURN + 1, C, 1)
    np.testing.assert_array_equal(pools[0], tq[NBURN - 1:])
