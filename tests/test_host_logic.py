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


@pytest.mark.parametrize("tag", ["list", "dict"])
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

    class Duck:
        Dc = None

        def evaluate(self):
            return None, np.zeros(500), None

    with pytest.raises(TypeError):
        pkg.MCMC(Duck(), g["data"], 1000.0, ["Uniform", 0.0, 1e4], 1000.0, lstm_model=None).sample(False)


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
    import bayesian_markov_chain_monte_carlo_amd.engine as eng_mod

    g = golden.npz("ssq")
    lib = model._engine.lib
    monkeypatch.setattr(eng_mod, "Engine", lambda mem="host", device=-1, **k: pkg.Engine(lib=lib))
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
