"""
CPU tests of the host-side mirror (RateStateModel / MCMC / RSF / main / json_save_load).  The classes
normally drive the HIP library; here the CPU oracle's engine is injected in its place so the host
logic (signatures, shapes, RNG draw order, quirk modes, persistence) is exercised without a GPU.
"""
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest


@pytest.fixture()
def model(pkg, oracle_lib):
    m = pkg.RateStateModel(number_time_steps=500)
    m._engine = pkg.Engine(lib=oracle_lib)  # test-only injection of the checker
    yield m
    m._engine.close()


def test_rate_state_model_surface(model, golden):
    for attr, val in dict(a=0.011, b=0.014, mu_ref=0.6, V_ref=1.0, k1=1e-7, t_start=0.0, t_final=50.0, num_tsteps=500,
                          delta_t=0.1, mu_t_zero=0.6, RadiationDamping=True, Dc=None).items():
        assert getattr(model, attr) == val
    with pytest.raises(ValueError):
        model.evaluate()
    model.Dc = np.array([1000.0])  # MCMC.py:381 assigns a 1-element array
    np.random.seed(0)
    t, acc, acc_noise = model.evaluate()
    assert t.shape == acc.shape == acc_noise.shape == (500,) and acc[0] == 0.0
    assert t[-1] == pytest.approx(49.9) and t[-1] == golden.json("forward")["cases"][3]["t_last"]  # accumulated like r.t
    np.random.seed(0)
    np.testing.assert_array_equal(acc_noise, acc + np.abs(acc) * np.random.randn(500))  # same draw as the reference
    ref = golden.npz("forward")["n500_dc1000"]
    assert np.abs(acc - ref).max() < 5e-4 * np.abs(ref).max()  # S = 1 vs dop853 (Tier 2)
    model.substeps = 8                                          # attribute change re-arms the engine
    model.Dc = 1000.0
    assert np.abs(model.evaluate()[1] - ref).max() < 2e-7 * np.abs(ref).max()
    batch = model.evaluate_batch([100.0, 1000.0, 5000.0])
    assert batch.shape == (3, 500)
    np.testing.assert_array_equal(batch[1], model.evaluate()[1])


@pytest.mark.parametrize("tag", ["list", "dict", "dict3"])
def test_mcmc_sample_reproduces_the_seeded_reference_chain(pkg, model, golden, tag):
    """np.random.seed selects the same chain as in the reference: same draw order from the global RNG
    (MCMC.py:497/331/160 + the unused randn(N) of every forward solve)."""
    g, meta = golden.npz("replay_" + tag), golden.json("replay_" + tag)
    prior = meta["prior"] if isinstance(meta["prior"], list) else {int(k): v for k, v in meta["prior"].items()}
    model.substeps = 8
    np.random.seed(meta["data_seed"])
    np.random.randn(meta["nsteps"])  # the reference generated its data with this seed: one evaluate() = randn(N)
    mc = pkg.MCMC(model, g["data"], meta["dc_true"], prior, meta["qstart"], nsamples=meta["nsamples"], lstm_model=None,
                  verbose=False)
    q = mc.sample(False)
    assert q.shape == (1, meta["nsamples"] + 1 - meta["nburn"]) == g["qparams_kept"].shape
    np.testing.assert_allclose(q, g["qparams_kept"], rtol=1e-6)
    np.testing.assert_allclose(mc.std2, g["std2_kept"], rtol=1e-5)
    assert mc.std2.shape == (meta["nsamples"] + 1 - meta["nburn"],)
    np.testing.assert_allclose(mc.Vstart[0, 0], meta["vstart"], rtol=5e-3)
    assert mc.nburn == meta["nburn"] and mc.n0 == 0.01
    np.testing.assert_array_equal(mc.qstart_limits, [[0.0, 1e4]])


def test_mcmc_public_submethods(pkg, model, golden):
    g = golden.npz("ssq")
    mc = pkg.MCMC(model, g["data"], 1000.0, ["Uniform", 0.0, 1e4], 1000.0, nsamples=10, lstm_model=None, verbose=False)
    s = mc.SSqcalc(np.array([[1000.0]]))
    assert s.shape == (1, 1) and s[0, 0] == pytest.approx(g["ssq"][4], rel=2e-4)
    np.random.seed(3)
    acc, s_new = mc.acceptreject(np.array([[20000.0]]), s, 1e-5)  # out of bounds: no solve, no uniform drawn
    assert not acc and s_new is s
    assert np.random.rand() == np.random.RandomState(3).rand()
    acc, s_new = mc.acceptreject(np.array([[1000.0]]), s * 2, 1e-5)  # much better fit: always accepted
    assert acc and s_new[0, 0] == pytest.approx(s[0, 0])
    mc.compute_initial_covariance()
    assert mc.std2[0] == pytest.approx(golden.json("init")["cases"]["list_q1000"]["std2_0"], rel=2e-4)
    assert model.Dc == pytest.approx(1000.0 * (1 + 1e-6))  # left perturbed like the reference
    mc.update_standard_deviation(s)
    assert len(mc.std2) == 2 and mc.std2[-1] > 0
    with pytest.raises(AttributeError):  # list prior: the reference's adaptation raises (and swallows) this
        mc.update_covariance_matrix(np.arange(20.0).reshape(1, -1))
    mcd = pkg.MCMC(model, g["data"], 1000.0, {1: 0.0, 2: 1e4}, 1000.0, nsamples=10, lstm_model=None)
    L = mcd.update_covariance_matrix(np.arange(20.0).reshape(1, -1))
    assert L.shape == (1, 1) and L[0, 0] == pytest.approx(np.sqrt(2.38 ** 2 / 2 * np.var(np.arange(10.0, 20.0), ddof=1)))
    with pytest.raises(NotImplementedError):
        pkg.MCMC(model, g["data"], 1000.0, ["Uniform", 0.0, 1e4], 1000.0, lstm_model={"x": 1}).sample(False)

    with pytest.raises(NotImplementedError):  # the quirk mode is the reference's: one parameter (MCMC.py:98, 381)
        mcd.update_covariance_matrix(np.arange(40.0).reshape(2, -1))
    with pytest.raises(np.linalg.LinAlgError):  # a window that never moved: np.linalg.cholesky's failure, and only that one
        mcd.update_covariance_matrix(np.full((1, 20), 3.0))


def test_any_model_with_dc_and_evaluate_is_accepted(pkg, oracle_lib):
    """The reference's model contract (MCMC.py:65-66, 127, 381-384): a settable .Dc and .evaluate()[1].  Such a model is
    evaluated on the host, its chain steps run on the engine (here the checker, injected); the batched device path, which
    integrates the model itself, says what it needs instead.  Without a GPU the product's own engine fails loudly."""
    from duck_model import observation

    model, data = observation()
    mc = pkg.MCMC(model, data, 4.0, ["Uniform", 0.5, 40.0], 6.0, nsamples=12, lstm_model=None, verbose=False)
    if pkg._abi.load().rsf_device_count() <= 0:
        with pytest.raises(pkg._abi.RsfError):
            mc.sample(False)
    mc._host_engine = pkg.Engine(lib=oracle_lib)
    calls = model.calls
    s = mc.SSqcalc(np.array([[4.0]]))
    assert s.shape == (1, 1) and model.calls == calls + 1 and np.asarray(model.Dc).shape == (1,)  # a 1-element array, MCMC.py:381
    np.random.seed(5)
    out = mc.sample(False)
    assert out.shape == (1, 12 + 1 - 6) and mc.std2.shape == (7,) and 0.0 <= mc.acceptance_ratio <= 1.0
    # the sub-methods work on such a model too (host evaluation, device chain step)
    acc, s_new = mc.acceptreject(np.array([[100.0]]), s, 1e-3)  # out of bounds: the model is not called
    assert not acc and s_new is s
    calls = model.calls
    acc, s_new = mc.acceptreject(np.array([[4.0]]), s * 3, 1e-3)  # much better fit: accepted, one model call
    assert acc and model.calls == calls + 1 and s_new[0, 0] == pytest.approx(s[0, 0])
    mc.std2 = [1e-3]
    mc.update_standard_deviation(s)
    assert len(mc.std2) == 2 and mc.std2[-1] > 0
    with pytest.raises(TypeError):
        mc.sample_batched(64)
    mc._host_engine.close()


def compose_like_the_reference_loop(mc, nsamples):
    """The reference's own sample() body (MCMC.py:464-468, 494-527) written out with the public sub-methods."""
    N = len(mc.data)
    mc.compute_initial_covariance()
    np.random.randn(N), np.random.randn(N), np.random.randn(N)   # its three forward solves each waste N normals
    qparams = np.copy(np.array([[float(mc.qstart)]]))
    Vold = np.copy(mc.Vstart)
    SSqprev = mc.SSqcalc(qparams)
    for isample in range(nsamples):
        q_new = np.reshape(np.random.multivariate_normal(qparams[:, -1], Vold), (-1, 1))
        accept, SSqnew = mc.acceptreject(q_new, SSqprev, mc.std2[-1])
        if accept:
            qparams = np.concatenate((qparams, q_new), axis=1)
            SSqprev = SSqnew
        else:
            qparams = np.concatenate((qparams, np.reshape(qparams[:, -1], (-1, 1))), axis=1)
        mc.update_standard_deviation(SSqprev)
        if (isample + 1) % mc.adapt_interval == 0:
            try:
                Vold = mc.update_covariance_matrix(qparams)
            except Exception:  # noqa: BLE001 — the reference's bare `except: pass` (MCMC.py:524-527)
                pass
    return qparams, np.asarray(mc.std2)


@pytest.mark.parametrize("prior", [["Uniform", 0.0, 1e4], {1: 0.0, 2: 1e4}])
def test_public_submethods_compose_into_the_sample_loop(pkg, model, golden, prior):
    """A caller that drives compute_initial_covariance / SSqcalc / acceptreject / update_standard_deviation /
    update_covariance_matrix itself, the way the reference's loop does (MCMC.py:494-527), walks the chain sample() walks
    — every sub-method runs its step through the engine (here the checker's; tests/test_gpu_dropin.py: the HIP library's)."""
    g = golden.npz("ssq")
    n = 40
    np.random.seed(17)
    fused = pkg.MCMC(model, g["data"], 1000.0, prior, 1000.0, nsamples=n, lstm_model=None, verbose=False)
    q_fused = fused.sample(False)
    np.random.seed(17)
    mc = pkg.MCMC(model, g["data"], 1000.0, prior, 1000.0, nsamples=n, lstm_model=None, verbose=False)
    qparams, std2 = compose_like_the_reference_loop(mc, n)
    assert qparams.shape == (1, n + 1) and len(np.unique(qparams)) > 5
    np.testing.assert_allclose(qparams[:, mc.nburn:], q_fused, rtol=1e-9)
    np.testing.assert_allclose(std2[mc.nburn:], fused.std2, rtol=1e-9)


def test_json_round_trip(pkg, tmp_path):
    from bayesian_markov_chain_monte_carlo_amd import json_save_load as j

    x = np.random.default_rng(0).standard_normal(37)
    f = tmp_path / "data.json"
    j.save_object(x, str(f))
    wire = json.load(open(f))
    assert wire["__ndarray__"] is True and wire["shape"] == [37] and len(wire["data"]) == 37
    np.testing.assert_array_equal(j.load_object(str(f)), x)
    j.save_object({"a": x.reshape(1, 37)}, str(f))
    assert j.load_object(str(f))["a"].shape == (1, 37)
    with pytest.raises(TypeError):
        j.save_object({"a": {1, 2}}, str(f))


def test_rsf_driver_and_main_entry(pkg, oracle_lib, tmp_path, monkeypatch):
    from bayesian_markov_chain_monte_carlo_amd import main as entry

    assert (entry.NUMBER_SLIP_VALUES, entry.LOWEST_SLIP_VALUE, entry.LARGEST_SLIP_VALUE, entry.QSTART, entry.QPRIORS,
            entry.NUMBER_TIME_STEPS, entry.NSAMPLES) == (5, 100.0, 5000.0, 1000.0, ["Uniform", 0.0, 10000.0], 500, 500)
    monkeypatch.chdir(tmp_path)
    problem = pkg.RSF(number_slip_values=2, lowest_slip_value=500.0, largest_slip_value=1500.0, qstart=1000.0,
                      qpriors=["Uniform", 0.0, 10000.0])
    np.testing.assert_array_equal(problem.dc_list, [500.0, 1500.0])
    assert problem.num_dc == 2 and problem.num_features == 2 and not problem.plotfigs and not problem.reduction
    problem.model = pkg.RateStateModel(number_time_steps=500)
    problem.model._engine = pkg.Engine(lib=oracle_lib)
    np.random.seed(5)
    problem.data = problem.generate_time_series()
    assert problem.data.shape == (1000,)
    np.random.seed(5)  # same noise stream as the reference: one randn(N) per Dc, in dc_list order
    clean = problem.model.evaluate_batch(problem.dc_list)
    np.testing.assert_array_equal(problem.data[:500], clean[0] + np.abs(clean[0]) * np.random.randn(500))
    problem.make_animations, problem.verbose = False, False
    with redirect_stdout(io.StringIO()) as out:
        seconds = entry.perform_inference(problem, "json", 12)
    assert isinstance(seconds, float) and seconds > 0            # the decorator returns seconds, like the reference
    assert "--- Dc is 500.0 ---" in out.getvalue()
    assert os.path.exists(tmp_path / "data.json")
    assert set(problem.posteriors) == {500.0, 1500.0} and problem.posteriors[500.0].shape == (1, 12 + 1 - 6)
    problem.format = "mysql"
    with pytest.raises(RuntimeError):
        problem.prepare_data(problem.data)
    problem.model._engine.close()


def test_sample_batched_host_logic(pkg, model, golden, monkeypatch):
    """sample_batched slices the post-burn block correctly across several launches (oracle engine injected)."""
    import sys

    mcmc_mod = sys.modules[pkg.MCMC.__module__]  # the module, not the class the package re-exports under the same name
    g = golden.npz("ssq")
    lib = model._engine.lib
    monkeypatch.setattr(mcmc_mod, "Engine", lambda mem="host", device=-1, **k: pkg.Engine(lib=lib))
    mc = pkg.MCMC(model, g["data"], 1000.0, ["Uniform", 0.0, 1e4], 1000.0, nsamples=20, lstm_model=None)
    one = mc.sample_batched(8, seed=4, mem="host")
    many = mc.sample_batched(8, seed=4, mem="host", iters_per_launch=7)
    assert one.samples.shape == (20 + 1 - 10, 8, 1) and one.std2.shape == (11, 8)
    np.testing.assert_array_equal(one.samples, many.samples)
    assert one.pooled().shape == (1, 88) and 0 < one.accept_rate <= 1
    jit = mc.sample_batched(8, seed=4, mem="host", jitter=(500.0, 1500.0))
    assert not np.array_equal(jit.samples, one.samples)
    thinned = mc.sample_batched(8, seed=4, mem="host", iters_per_launch=7, thin=3)
    np.testing.assert_array_equal(thinned.samples, one.samples[::3])
    np.testing.assert_array_equal(thinned.std2, one.std2[::3])
    with pytest.raises(ValueError):
        mc.sample_batched(8, seed=4, mem="host", thin=0)


def test_flat_module_imports_like_the_reference(tmp_path):
    """The reference imports its modules flat (main.py:44-46, RSF.py:1-4, MCMC.py:1).  With the package directory first on
    sys.path the very same statements must work (fresh interpreter: nothing of the package pre-imported), and the objects
    must be the drop-in classes bound to the HIP library (which refuses to run here: no GPU, no CPU fallback)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = """
import sys
sys.path.insert(0, sys.argv[1])
from imports import *
from RSF import RSF
from RateStateModel import RateStateModel
from MCMC import MCMC
from json_save_load import save_object, load_object
from RSF import measure_execution_time
import main
assert main.QPRIORS == ["Uniform", 0.0, 10000.0] and main.NSAMPLES == 500 and callable(main.setup_problem)
assert "bayesian_markov_chain_monte_carlo_amd" not in sys.modules            # no package detour
problem = RSF(number_slip_values=5, lowest_slip_value=100., largest_slip_value=5000., qstart=1000., qpriors=main.QPRIORS)
problem.model = RateStateModel(number_time_steps=500)
assert problem.dc_list.tolist() == [100.0, 1325.0, 2550.0, 3775.0, 5000.0] and problem.model.delta_t == 0.1
mc = MCMC(problem.model, np.zeros(500), 100.0, main.QPRIORS, 1000.0, nsamples=10, lstm_model=None)
assert mc.nburn == 5 and mc.qstart_limits.tolist() == [[0.0, 10000.0]]
import _abi
print("LIB", _abi.LIB_PATH)
try:
    problem.model.Dc = 100.0
    problem.model.evaluate()
except _abi.RsfError as e:
    print("REFUSED", e)
"""
    r = subprocess.run([sys.executable, "-c", script, os.path.join(root, "bayesian-markov-chain-monte-carlo_amd")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(tmp_path),
                       env=dict(os.environ, MPLBACKEND="Agg"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "csrc/librsf_hip.so" in r.stdout
    import bayesian_markov_chain_monte_carlo_amd as pkg
    if pkg._abi.load().rsf_device_count() == 0:
        assert "REFUSED" in r.stdout and "no CPU fallback" in r.stdout


def test_bench_self_launch_reports_child_failure(tmp_path):
    """`python bench.py --gpus 2` starts its own ranks (no external launcher).  Without a GPU the ranks fail — the parent must
    then exit non-zero and print no result line (with GPUs the same path is covered by tests/test_gpu_dropin.py)."""
    import subprocess
    import sys

    import bayesian_markov_chain_monte_carlo_amd as pkg
    if pkg._abi.load().rsf_device_count() > 0:
        pytest.skip("GPU present: the successful self-launch is tested in test_gpu_dropin.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--chains", "256",
                        "--nsteps", "100", "--steps", "1", "--warmup", "0", "--iters-per-step", "2", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=str(tmp_path), timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert "child ranks exited with status" in r.stderr


def _rsf_problem(pkg, meta, engine=None):
    problem = pkg.RSF(**meta["rsf_kwargs"])
    problem.model = pkg.RateStateModel(number_time_steps=meta["number_time_steps"])
    problem.model.integrator = "dop853"      # the reference's own scheme: its numbers, not a convergence argument
    if engine is not None:
        problem.model._engine = engine       # test-only injection of the checker
    problem.make_animations, problem.verbose = False, False
    return problem


def test_generate_time_series_equals_the_reference_vector(pkg, oracle_lib, golden):
    """A13: RSF.generate_time_series() under np.random.seed gives the reference's concatenated (num_dc*N,) vector
    (tests/golden/rsf_driver.npz, written by the reference's RSF.generate_time_series, RSF.py:355-371)."""
    g, meta = golden.npz("rsf_driver"), golden.json("rsf_driver")
    with pkg.Engine(lib=oracle_lib) as e:
        problem = _rsf_problem(pkg, meta, e)
        np.testing.assert_array_equal(problem.dc_list, g["dc_list"])
        np.random.seed(meta["seed_data"])
        data = problem.generate_time_series()
    assert data.shape == g["data"].shape == (3 * 500,)
    np.testing.assert_allclose(data, g["data"], rtol=1e-9, atol=1e-9 * np.abs(g["data"]).max())


def test_inference_slices_and_chains_equal_the_reference(pkg, oracle_lib, golden, tmp_path, monkeypatch):
    """A14: RSF.inference() — JSON round trip, data[i*N:(i+1)*N] per Dc, one MCMC per Dc in dc_list order from one RNG
    stream — gives the chains the reference's MCMC gives on those slices (RSF.py:874-894, 1040-1046)."""
    g, meta = golden.npz("rsf_driver"), golden.json("rsf_driver")
    monkeypatch.chdir(tmp_path)
    with pkg.Engine(lib=oracle_lib) as e:
        problem = _rsf_problem(pkg, meta, e)
        problem.data, problem.format = g["data"], "json"
        np.random.seed(meta["seed_chains"])
        with redirect_stdout(io.StringIO()):
            problem.inference(meta["nsamples"])
    for i, dc in enumerate(g["dc_list"]):
        np.testing.assert_allclose(problem.posteriors[float(dc)], g[f"qparams_{i}"], rtol=1e-9, err_msg=f"dc {dc}")
    # a wrong slice cannot pass: the three chains differ from one another by far more than the tolerance
    assert not np.allclose(g["qparams_0"], g["qparams_1"], rtol=1e-3)


def test_reference_written_json_files(pkg, golden, tmp_path):
    """Files written by the reference's own json_save_load.save_object (tests/golden/ref_written_*.json) load here, and
    this module's save_object writes the same bytes."""
    from bayesian_markov_chain_monte_carlo_amd import json_save_load as J

    for name in ("ref_written_array", "ref_written_dict"):
        src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".json")
        obj = J.load_object(src)
        J.save_object(obj, tmp_path / "again.json")
        assert open(src, "rb").read() == open(tmp_path / "again.json", "rb").read(), name
    arr = J.load_object(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_written_array.json"))
    assert isinstance(arr, np.ndarray) and arr.shape == (12,) and arr.dtype == np.float64
    np.testing.assert_array_equal(arr, np.random.default_rng(3).standard_normal(12) * 1e-3)
    d = J.load_object(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_written_dict.json"))
    assert d["grid"].shape == (2, 3) and d["ints"].tolist() == [0, 1, 2, 3] and d["meta"] == {"dc": 1000.0, "n": 12, "tags": ["a", "b"]}


def test_posterior_figure_uses_the_engine_kde(pkg, model, monkeypatch):
    """RSF.plot_dist (RSF.py:717-746): trace panel + density panel; the density drawn is the engine's KDE of the kept
    samples (here the oracle engine's; on a GPU box the device KDE) and equals scipy.stats.gaussian_kde on the same grid."""
    pytest.importorskip("matplotlib")
    monkeypatch.setenv("MPLBACKEND", "Agg")
    import matplotlib

    matplotlib.use("Agg", force=True)
    from scipy.stats import gaussian_kde

    problem = pkg.RSF(number_slip_values=1, lowest_slip_value=1000.0, largest_slip_value=1000.0, qstart=1000.0)
    problem.model, problem.format = model, "json"
    q = np.random.default_rng(3).normal(1000.0, 40.0, (1, 300))
    fig = problem.plot_dist(q, 1000.0)
    assert fig is not None and len(fig.axes) == 2
    pdf, grid = fig.axes[1].lines[0].get_data()
    assert len(grid) == 1000
    np.testing.assert_allclose(pdf, gaussian_kde(q[0]).pdf(grid), rtol=1e-9, atol=1e-300)
    np.testing.assert_array_equal(fig.axes[0].lines[0].get_ydata(), q[0])
    with pytest.warns(UserWarning, match="posterior figure skipped"):   # a chain that never moved: no density, no crash
        assert problem.plot_dist(np.full((1, 50), 1000.0), 1000.0) is None
    import matplotlib.pyplot as plt

    plt.close("all")


def test_unknown_dc_is_reported_not_sampled(pkg, model, capsys):
    """RSF.perform_sampling_and_plotting (RSF.py:874-878): a dc that is not in dc_list prints the reference's message and
    returns without sampling."""
    problem = pkg.RSF(number_slip_values=2, lowest_slip_value=500.0, largest_slip_value=1500.0, qstart=1000.0)
    problem.model, problem.format = model, "json"
    problem.perform_sampling_and_plotting(np.zeros(1000), 777.0, 10, None)
    assert "Error: dc value 777.0 not found in dc_list." in capsys.readouterr().out
    assert problem.posteriors == {}
