"""
GPU tests of the drop-in surface: the reference's own CPU-runnable case (BASELINE configs[0]: 1 chain,
1000 proposals, nsteps 500) through MCMC.sample() on the GPU, and the main.py / RSF flow end to end.
"""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mc_standard_error(x, batch=50):
    """Standard error of the mean of one correlated chain, by batch means."""
    n = (x.size // batch) * batch
    return float(x[:n].reshape(-1, batch).mean(axis=1).std(ddof=1) / np.sqrt(n // batch))


def test_config1_chain_matches_the_reference_run(pkg, golden):
    """np.random.seed(2025) + MCMC.sample(False) with the forward model integrated by fixed-step RK4, eight steps per output
    interval (1.7e-8 from dop853 on SSq, SURVEY §8c ladder), against the reference's 1000-proposal chain
    (tests/golden/config1.npz) AND against this package's own run of the reference scheme (integrator = "dop853", which
    reproduces that golden chain to 1e-9: next test).  Same seed => same variates => the two chains are the same chain until an
    accept decision lands inside what the integrators disagree by.  So: (i) up to their first differing iteration the chains are
    identical (1e-6: the integrators' difference carried by q); (ii) if they fork, the fork is PROVEN a near-tie — the log
    acceptance ratios the two integrators assign to that very proposal differ by no more than their SSq difference allows
    (< 5e-5), yet decide differently, so the decision margin was smaller than that; (iii) the posterior summaries agree within
    three Monte-Carlo standard errors of the chains themselves."""
    g, meta = golden.npz("config1"), golden.json("config1")
    runs = {}
    for tag, setup in (("rk4_s8", dict(substeps=8)), ("dop853", dict(integrator="dop853"))):
        model = pkg.RateStateModel(number_time_steps=500)
        for k, v in setup.items():
            setattr(model, k, v)
        np.random.seed(2025)
        mc = pkg.MCMC(model, g["data"], 1000.0, ["Uniform", 0.0, 10000.0], 1000.0, nsamples=1000, lstm_model=None, verbose=False)
        mc.nburn = 0  # keep the whole chain: the fork, if any, may sit in the burn-in
        runs[tag] = (mc.sample(False)[0], np.asarray(mc.std2), mc, model)
    qa, sa, mca, model_a = runs["rk4_s8"]
    qb, sb, mcb, model_b = runs["dop853"]
    assert qa.shape == qb.shape == (1001,)
    np.testing.assert_allclose(qb[500:], g["qparams_kept"][0], rtol=1e-9)  # the reference's chain itself
    differ = ~np.isclose(qa, qb, rtol=1e-6)
    fork = int(np.argmax(differ)) if differ.any() else None
    print("config-1 chain, RK4 (8 steps per interval) vs the reference scheme: " + (f"first differing sample {fork} of 1000" if fork else "no fork"))
    if fork is not None:
        assert fork >= 1 and not differ[:fork].any()
        # iteration `fork` (1-based column) proposed the same q_new from the same point on both sides; exactly one side moved
        q_prev = qa[fork - 1]
        moved_a, moved_b = qa[fork] != q_prev, qb[fork] != qb[fork - 1]
        assert moved_a != moved_b, "the chains differ without one of them having accepted what the other rejected"
        q_new = qa[fork] if moved_a else qb[fork]
        std2 = sb[fork - 1]
        ratio = {}
        for tag, (mc, model) in (("a", (mca, model_a)), ("b", (mcb, model_b))):
            ssq_prev = float(mc.SSqcalc(np.array([[q_prev]]))[0, 0])
            ssq_new = float(mc.SSqcalc(np.array([[q_new]]))[0, 0])
            ratio[tag] = 0.5 * (ssq_prev - ssq_new) / std2
        gap = abs(ratio["a"] - ratio["b"])
        print(f"  log acceptance ratio of that proposal: RK4 {ratio['a']:.9f}, reference scheme {ratio['b']:.9f} (|difference| {gap:.2e})")
        assert gap < 5e-5, f"the integrators disagree on the log acceptance ratio by {gap:.2e}: not the 1e-8-level SSq difference"
        assert fork >= 100, f"a near-tie already at sample {fork}: possible, but worth a look"
    se = (_mc_standard_error(qa[500:]) ** 2 + _mc_standard_error(qb[500:]) ** 2) ** 0.5
    assert abs(qa[500:].mean() - qb[500:].mean()) < 3 * se, (qa[500:].mean(), qb[500:].mean(), se)
    assert abs(qa[500:].mean() - meta["mean"]) < 3 * se and abs(qa[500:].std() / meta["std"] - 1) < 0.35
    assert 0.5 < mca.acceptance_ratio < 0.9
    assert sa.shape == (1001,)


def test_config1_chain_is_identical_with_the_reference_integrator(pkg, golden):
    """integrator = "dop853": seeded MCMC.sample(False) gives the reference's 1000-proposal chain, all 501 kept samples."""
    g = golden.npz("config1")
    model = pkg.RateStateModel(number_time_steps=500)
    model.integrator = "dop853"
    np.random.seed(2025)
    mc = pkg.MCMC(model, g["data"], 1000.0, ["Uniform", 0.0, 10000.0], 1000.0, nsamples=1000, lstm_model=None, verbose=False)
    q = mc.sample(False)
    np.testing.assert_allclose(q, g["qparams_kept"], rtol=1e-9)
    np.testing.assert_allclose(mc.std2, g["std2_kept"], rtol=1e-7)


def test_batched_posterior_agrees_with_reference_posterior(pkg, golden):
    """Tier 3: pooled GPU posterior (Philox path, RK4 S = 1) vs the reference's long chain."""
    g, meta = golden.npz("config1"), golden.json("config1")
    model = pkg.RateStateModel(number_time_steps=500)
    mc = pkg.MCMC(model, g["data"], 1000.0, ["Uniform", 0.0, 10000.0], 1000.0, nsamples=200, lstm_model=None)
    pool = mc.sample_batched(2048, seed=11, mem="device", iters_per_launch=50)
    x = pool.pooled()[0]
    assert x.size == 2048 * 101
    # the reference's 501 correlated samples pin the mean only to a few units: compare within 3 of its standard errors
    ess = 501 / 8.0
    assert abs(x.mean() - meta["mean"]) < 3 * meta["std"] / np.sqrt(ess) + 1.0
    assert 0.7 < x.std() / meta["std"] < 1.3
    assert 0.5 < pool.accept_rate < 0.9 and pool.stats["nonfinite"] == 0


def test_main_entry_runs_on_the_gpu(pkg, tmp_path, monkeypatch):
    """main.py's own flow (main.py:50-56, 156-164, 300-301: five true Dc between 100 and 5000, every chain started at 1000, the
    LIST prior — no adaptation ever, MCMC.py:524-527), which the reference cannot run as shipped, so its
    chains are held to what they must be rather than to recorded ones (RSF.inference against the reference's recorded
    chains: test_inference_slices_and_chains_equal_the_reference): one kept block per Dc in dc_list order, inside the prior
    box; for the four well-posed cases the proposal Vstart gives is accepted at the usual rate and the chain walks TOWARDS
    the truth; for Dc = 100 — a proposal ~600 wide around a posterior ~5 wide — almost nothing is accepted (the wide-proposal
    case the kernels' counters and run-ahead exist for), and the sample stays inside the box all the same."""
    from bayesian_markov_chain_monte_carlo_amd import main as entry

    monkeypatch.chdir(tmp_path)
    np.random.seed(1)
    problem = entry.setup_problem()
    assert problem.data.shape == (5 * 500,) and np.isfinite(problem.data).all()
    np.testing.assert_allclose(problem.dc_list, np.linspace(100.0, 5000.0, 5))
    problem.make_animations, problem.verbose = False, False
    nsamples = 120
    with redirect_stdout(io.StringIO()) as out:
        seconds = entry.perform_inference(problem, "json", nsamples)
    assert seconds > 0 and os.path.exists(tmp_path / "data.json")
    assert out.getvalue().count("--- Dc is") == 5
    from bayesian_markov_chain_monte_carlo_amd.json_save_load import load_object

    np.testing.assert_array_equal(np.asarray(load_object(str(tmp_path / "data.json"))), problem.data)  # the slices' source
    assert [float(k) for k in problem.posteriors] == [float(dc) for dc in problem.dc_list]
    for dc in problem.dc_list:
        q = np.asarray(problem.posteriors[float(dc)])
        assert q.shape == (1, nsamples + 1 - nsamples // 2) and np.isfinite(q).all() and (q > 0).all() and (q < 1e4).all()
        moves = int((np.diff(q[0]) != 0).sum())
        if dc > 1000:
            assert moves > 0.3 * (q.shape[1] - 1), (dc, moves)       # Vstart's proposal is accepted at the usual rate ...
            assert q[0, -10:].mean() > 1000.0 + 20.0, (dc, q[0, -10:])  # ... and the chain walks up from 1000 towards the truth
        else:
            assert moves <= 0.2 * (q.shape[1] - 1), (dc, moves)       # Dc = 100: hardly anything is accepted


def test_dc_list_sweep_in_one_launch(pkg):
    """RSF.inference_batched: every true Dc of dc_list is an observation group of independent chains."""
    np.random.seed(4)
    problem = pkg.RSF(number_slip_values=3, lowest_slip_value=500.0, largest_slip_value=2500.0, qstart=1000.0,
                      qpriors=["Uniform", 0.0, 10000.0])
    problem.model = pkg.RateStateModel(number_time_steps=500)
    problem.data = problem.generate_time_series()
    pools = problem.inference_batched(nsamples=120, chains_per_dc=256, seed=1)
    assert sorted(pools) == [500.0, 1500.0, 2500.0]
    for dc, pool in pools.items():
        assert pool.samples.shape == (120 + 1 - 60, 256, 1)
        x = pool.pooled()[0]
        assert abs(x.mean() - dc) < 0.25 * dc, (dc, x.mean())
        assert 0.2 < pool.accept_rate < 0.98


def test_integration_md_binding_stub_works(pkg, golden):
    """The ctypes stub INTEGRATION.md shows a maintainer of the reference (section 2) is executed verbatim."""
    import re
    import types

    from conftest import ROOT

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "rsf_binding.py" in b][0]
    block = block.replace("/path/to/bayesian-markov-chain-monte-carlo_amd/csrc/librsf_hip.so", pkg._abi.LIB_PATH)
    mod = types.ModuleType("rsf_binding")
    exec(compile(block, "INTEGRATION.md", "exec"), mod.__dict__)
    model = pkg.RateStateModel(number_time_steps=500)  # any object with the reference's attribute names will do
    g = golden.npz("ssq")
    ctx = mod.make_ctx(model)
    got = mod.ssq(ctx, g["qgrid"], g["data"])
    with pkg.Engine(mem="host") as e:
        e.set_model(model, 1)
        want, _ = e.forward(g["qgrid"], data=g["data"], want_ssq=True, want_acc=False)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_allclose(got[4:], g["ssq"][4:], rtol=1e-3)  # and it is the reference's SSq (Tier 2, S = 1)


def test_plain_c_caller(pkg, tmp_path):
    """a C program linked against librsf_hip.so and the SYSTEM HIP runtime (the library
    carries no DT_NEEDED on a runtime) gets the same numbers as the ctypes path running on torch's bundled runtime."""
    import subprocess

    from conftest import ROOT

    exe = tmp_path / "abi_smoke"
    csrc = os.path.dirname(pkg._abi.LIB_PATH)
    subprocess.check_call(["gcc", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + csrc, "-lrsf_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    assert out[0] == "nout 500" and out[3].startswith("backend hip-gfx950 version 1 devices")
    assert "pool ok" in out  # (RCCL prints a version banner first) RCCL bound at run time from a process without PyTorch (the system librccl.so.1)
    c_ssq = np.array(out[1].split()[1:], dtype=np.float64)
    c_mcmc = out[2].split()[1:]
    model = pkg.RateStateModel(500)
    with pkg.Engine(mem="host") as e:
        e.set_model(model, 1)
        dc = np.array([500.0, 1000.0, 2000.0])
        _, acc = e.forward(dc)
        data = acc[:, 1] * (1.0 + 0.3 * np.sin(0.7 * np.arange(500)))
        ssq, _ = e.forward(dc, data=data, want_ssq=True, want_acc=False)
        np.testing.assert_allclose(c_ssq, ssq, rtol=1e-13)  # sin() of libm vs NumPy may differ in the last bit
        e.mcmc_init(np.full((64, 1), 1000.0), data, [0.0], [1e4], seed=2025, prior_len=3)
        tq, ts, _ = e.mcmc_run(10)
        st = e.stats()
    np.testing.assert_allclose(float(c_mcmc[0]), tq[-1, :, 0].mean(), rtol=1e-9)
    np.testing.assert_allclose(float(c_mcmc[1]), ts[-1, 0], rtol=1e-9)
    assert [int(v) for v in c_mcmc[2:]] == [st["accepted"], st["evaluated"], st["nonfinite"], st["iters_done"]]


def test_bench_c_abi_pool_exchange_helper(pkg):
    """bench.py's N > 1 leg pools the samples a second time through the C ABI (rsf_comm_init + rsf_pool_allgather,
    the library's own RCCL communicator) under a watchdog thread.  One rank is all this box allows: a 1-rank NCCL
    process group + a real 1-rank RCCL communicator inside the library."""
    import socket
    import sys

    import torch
    import torch.distributed as dist

    from conftest import ROOT
    from bayesian_markov_chain_monte_carlo_amd import dist as rdist

    sys.path.insert(0, ROOT)
    import bench

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        local = torch.arange(6 * 128, dtype=torch.float64, device="cuda").reshape(6, 128, 1)
        pool = rdist.pool_to_chain_major(rdist.allgather_pool(local))
        with pkg.Engine(mem="device") as e:
            res = bench.abi_pool_allgather(e, local, pool, rdist, timeout_s=60.0)
        assert res["status"] == "ok" and res["equals_torch_pool"] is True, res
    finally:
        dist.destroy_process_group()


def test_bench_self_launches_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO external launcher: the parent (which touches no GPU API) spawns two fresh ranks
    under torch.distributed.run, relays rank 0's single JSON line and exits 0.  gloo backend: both ranks share the one
    card of this box, a rehearsal of the N > 1 code path (sharding by global chain id, barrier, max-over-ranks timing,
    pool all-gather), not a scaling number."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--chains", "2048",
                        "--nsteps", "200", "--steps", "2", "--warmup", "1", "--iters-per-step", "5", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["chains_per_gpu"] == 2048
    assert out["value"] > 0 and out["roofline"]["bound"] == "valu_fp64" and "pool_allgather_ms" in out


def _rsf_problem(pkg, meta):
    problem = pkg.RSF(**meta["rsf_kwargs"])
    problem.model = pkg.RateStateModel(number_time_steps=meta["number_time_steps"])
    problem.model.integrator = "dop853"      # the reference's own scheme: its numbers, not a convergence argument
    problem.make_animations, problem.verbose = False, False
    return problem


def test_generate_time_series_equals_the_reference_vector(pkg, golden):
    """A13 on the GPU: RSF.generate_time_series() under np.random.seed gives the reference's concatenated (num_dc*N,)
    vector (tests/golden/rsf_driver.npz, written by the reference's RSF.generate_time_series, RSF.py:355-371)."""
    g, meta = golden.npz("rsf_driver"), golden.json("rsf_driver")
    problem = _rsf_problem(pkg, meta)
    np.random.seed(meta["seed_data"])
    data = problem.generate_time_series()
    assert data.shape == g["data"].shape == (3 * 500,)
    np.testing.assert_allclose(data, g["data"], rtol=1e-9, atol=1e-9 * np.abs(g["data"]).max())


def test_inference_slices_and_chains_equal_the_reference(pkg, golden, oracle_lib, tmp_path, monkeypatch):
    """A14 on the GPU: RSF.inference() — JSON round trip, data[i*N:(i+1)*N] per Dc, one MCMC per Dc in dc_list order
    from one RNG stream — gives the chains the reference's MCMC gives on those slices (RSF.py:874-894, 1040-1046)."""
    g, meta = golden.npz("rsf_driver"), golden.json("rsf_driver")
    monkeypatch.chdir(tmp_path)
    problem = _rsf_problem(pkg, meta)
    problem.data, problem.format = g["data"], "json"
    np.random.seed(meta["seed_chains"])
    with redirect_stdout(io.StringIO()):
        seconds = problem.inference(meta["nsamples"])
    assert seconds > 0
    # (a) the chains as the product runs them: every kept sample of every chain.  The proposal std sqrt(Vstart) comes from a
    # forward difference with relative step 1e-6 (MCMC.py:251) that amplifies the ~1e-12 GPU-vs-libm rounding of the
    # trajectories a million-fold, and q = q_cur + sqrt(Vstart) z carries half of that times sqrt(V) z / q.  That is a property
    # of the reference's Vstart formula, not of the chain logic, so it is held to its own honest bound here ...
    for i, dc in enumerate(g["dc_list"]):
        np.testing.assert_allclose(problem.posteriors[float(dc)], g[f"qparams_{i}"], rtol=2e-8, err_msg=f"dc {dc}")
    # (b) ... and separated out: with Vstart taken from the checker (whose CPU twin of this test, tests/test_host_logic.py,
    # reproduces the reference's chains to 1e-9) the GPU chains — slices, JSON round trip, RNG order, every accept
    # decision, sigma^2 — equal the reference's to 1e-9, and Vstart itself is asserted with the bound the formula allows.
    from bayesian_markov_chain_monte_carlo_amd.engine import Engine

    checker = {}

    def checker_vstart(mc):
        with Engine(lib=oracle_lib) as e:
            e.set_model(mc.model, 1)
            e.mcmc_init([[float(mc.qstart)]], np.asarray(mc.data, dtype=np.float64), [0.0], [1e4], prior_len=len(mc.qpriors))
            checker[float(mc.dc_true)] = float(e.get_state()[3][0, 0, 0])
        return checker[float(mc.dc_true)]

    problem2 = _rsf_problem(pkg, meta)
    problem2.data, problem2.format = g["data"], "json"
    device_init = pkg.MCMC.compute_initial_covariance

    def init_with_checker_vstart(mc):  # the product's own init, then the proposal covariance replaced by the checker's
        device_init(mc)
        v = checker_vstart(mc)
        mc._engine().set_state(V=np.reshape(v, (1, 1, 1)))
        mc.Vstart = np.array([[v]])

    monkeypatch.setattr(pkg.MCMC, "compute_initial_covariance", init_with_checker_vstart)
    np.random.seed(meta["seed_chains"])
    with redirect_stdout(io.StringIO()):
        problem2.inference(meta["nsamples"])
    monkeypatch.setattr(pkg.MCMC, "compute_initial_covariance", device_init)
    for i, dc in enumerate(g["dc_list"]):
        np.testing.assert_allclose(problem2.posteriors[float(dc)], g[f"qparams_{i}"], rtol=1e-9, err_msg=f"dc {dc} (checker Vstart)")
    N = meta["number_time_steps"]
    for i, dc in enumerate(g["dc_list"]):
        mc = pkg.MCMC(problem.model, g["data"][i * N:(i + 1) * N], float(dc), problem.qpriors, problem.qstart, nsamples=4, lstm_model=None)
        mc.compute_initial_covariance()
        rel = abs(mc.Vstart[0, 0] / checker[float(dc)] - 1)
        print(f"Vstart GPU vs checker, dc {dc}: rel {rel:.2e}")
        assert rel < 2e-6, (dc, rel)
    assert not np.allclose(g["qparams_0"], g["qparams_1"], rtol=1e-2) and not np.allclose(g["qparams_1"], g["qparams_2"], rtol=1e-2)


def test_config3_shard_at_full_size_with_rccl_pool(pkg):
    """BASELINE configs[3] as far as one GPU goes: ONE rank's shard in its stated size — 262 144 chains (2 097 152 / 8),
    1000 proposals each, nsteps 500, post-burn-in pool of 501 draws per chain = 1.05 GB — through `dist.run_sharded`
    (the multi-GPU driver: shard by global chain id, sample, one all-gather of the kept block) on a real NCCL/RCCL
    process group of world size 1, then the same pool once more through the C ABI's own RCCL communicator.  Checked:
    the gathered pool IS the shard's kept block (bit for bit, both routes); shifting the shard to the chain-id range rank
    3 of 8 would own changes the draws (the RNG is keyed by the global id); posterior moments are sane.  The 8-rank run
    itself needs the driver's node."""
    import socket

    import torch
    import torch.distributed as dist

    from bayesian_markov_chain_monte_carlo_amd import dist as rdist

    C, n_iters, nburn = 262144, 1000, 500
    model = pkg.RateStateModel(number_time_steps=500)
    with pkg.Engine(mem="host") as e:
        e.set_model(model, 1)
        from conftest import synthetic_data

        data = synthetic_data(e)
    q0 = np.full((C, 1), 1000.0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        pool, stats = rdist.run_sharded(lambda: pkg.Engine(mem="device"), model, 1, data, q0, [0.0], [1.0e4], n_iters, nburn,
                                        seed=2025, mcmc_kwargs=dict(prior_len=3))
        torch.cuda.synchronize()
        assert pool.shape == (n_iters + 1 - nburn, C, 1) and pool.is_cuda       # 501 kept rows x 262 144 chains: 1.05 GB
        assert stats["iters_done"] == n_iters and stats["evaluated"] == stats["iters_done"] * C and stats["nonfinite"] == 0
        mean, std = float(pool.mean()), float(pool.std())
        # one noise realisation (|acc| N(0,1), seed 2025) puts the posterior at 1040 +- 51 for nsteps 500; the reference's own
        # single-chain run on ITS realisation: 990 +- 43 (tests/golden/config1.json)
        assert abs(mean - 1000.0) < 100.0 and 20.0 < std < 80.0, (mean, std)
        assert 0.5 < stats["accepted"] / (n_iters * C) < 0.9
        # the same shard again by hand: the pool is exactly the kept block of the shard's trace
        with pkg.Engine(mem="device") as e:
            e.set_model(model, 1)
            e.mcmc_init(q0, data, [0.0], [1.0e4], seed=2025, chain_offset=0, prior_len=3)
            tq, _, _ = e.mcmc_run(n_iters, traces=("q",))
            e.sync()
            assert torch.equal(pool, tq[nburn - 1:])
            # ... and through the C ABI (rsf_comm_init + rsf_pool_allgather: the library's own RCCL communicator)
            rdist.comm_init_from_process_group(e)
            out = e.pool_allgather(tq[nburn - 1:].contiguous())
            e.sync()
            assert out.shape == (1, n_iters + 1 - nburn, C, 1) and torch.equal(out[0], pool)
            e.comm_destroy()
            del out
            # rank 3 of 8 would own global chains [3C, 4C): different draws from the same start
            e.mcmc_init(q0[:4096], data, [0.0], [1.0e4], seed=2025, chain_offset=3 * C, prior_len=3)
            t3, _, _ = e.mcmc_run(50, traces=("q",))
            e.sync()
            assert not torch.equal(t3, tq[:50, :4096])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("prior", [["Uniform", 0.0, 1e4], {1: 0.0, 2: 1e4}])
def test_public_submethods_compose_into_the_sample_loop_on_the_gpu(pkg, golden, prior):
    """MCMC.py:494-527 composed by the CALLER from the public sub-methods — each of which runs its step on the device
    (acceptreject / update_standard_deviation: one replayed kernel iteration; update_covariance_matrix: rsf_mcmc_adapt) —
    walks the chain the fused sample() walks.  CPU twin with the checker's engine: tests/test_host_logic.py."""
    from test_host_logic import compose_like_the_reference_loop

    g = golden.npz("ssq")
    n = 40
    model = pkg.RateStateModel(number_time_steps=500)
    np.random.seed(17)
    fused = pkg.MCMC(model, g["data"], 1000.0, prior, 1000.0, nsamples=n, lstm_model=None, verbose=False)
    q_fused = fused.sample(False)
    assert model.engine().lib.rsf_backend() == b"hip-gfx950"
    np.random.seed(17)
    mc = pkg.MCMC(model, g["data"], 1000.0, prior, 1000.0, nsamples=n, lstm_model=None, verbose=False)
    qparams, std2 = compose_like_the_reference_loop(mc, n)
    assert len(np.unique(qparams)) > 5
    np.testing.assert_allclose(qparams[:, mc.nburn:], q_fused, rtol=1e-9)
    np.testing.assert_allclose(std2[mc.nburn:], fused.std2, rtol=1e-9)


def test_drop_in_sample_with_the_float32_solve(pkg, golden):
    """`RateStateModel.precision = "float32"` under the drop-in single-chain `MCMC.sample()`: one chain is one half-filled
    lane of the two-chains-per-lane float32 sampler, driven through the one-proposal replay graph.  Same seed, same
    variates: the float32 chain follows the float64 one until a decision falls inside the ~1e-4 SSq difference."""
    g = golden.npz("ssq")
    out = {}
    for precision in ("float64", "float32"):
        model = pkg.RateStateModel(number_time_steps=500)
        model.precision = precision
        np.random.seed(23)
        mc = pkg.MCMC(model, g["data"], 1000.0, ["Uniform", 0.0, 1e4], 1000.0, nsamples=60, lstm_model=None, verbose=False)
        out[precision] = (mc.sample(False), mc.std2, mc.acceptance_ratio)
    q64, q32 = out["float64"][0], out["float32"][0]
    assert q32.shape == q64.shape == (1, 31) and np.isfinite(q32).all()
    assert abs(out["float32"][2] - out["float64"][2]) <= 0.1
    same = np.isclose(q32, q64, rtol=1e-6)
    assert same[0, :5].all() or same.mean() > 0.5, (q32[0, :8], q64[0, :8])
    np.testing.assert_allclose(out["float32"][1][:3], out["float64"][1][:3], rtol=1e-3)
