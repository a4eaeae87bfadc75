"""
Checks of the sampler as an operator over a caller-evaluated likelihood (rsf_mcmc_init_state / rsf_mcmc_propose /
rsf_mcmc_replay_ssq), shared by the CPU suite (the oracle library) and the GPU suite (the HIP kernels).

SURVEY §8(c) G4/G5: the reference's instrumented runs recorded, per iteration, the normal behind the proposal, the uniform,
the gamma variate AND the proposal's sum of squares.  Fed back, the chain logic alone — no integrator anywhere — must
reproduce the reference's chain to rounding: accept rule, sigma^2 update, box test, adaptation quirks (MCMC.py:494-527).
"""
import numpy as np


def _config(meta):
    is_list = isinstance(meta["prior"], list)
    lo, hi = (meta["prior"][1], meta["prior"][2]) if is_list else (meta["prior"]["1"], meta["prior"]["2"])
    return dict(lo=[lo], hi=[hi], n0=meta["n0"], prior_len=len(meta["prior"]), adapt_mode="none" if is_list else "reference_dict",
                adapt_interval=meta["adapt_interval"])


def _init(engine, meta):
    cfg = _config(meta)
    engine.mcmc_init_state([[meta["qstart"]]], [meta["ssq0"]], [meta["std2_0"]], [[[meta["vstart"]]]], cfg.pop("lo"), cfg.pop("hi"), **cfg)


def _q_after(g):
    """The chain's point after each recorded iteration (the fixture's q_cur[i] is the point iteration i STARTS from)."""
    return np.concatenate((g["q_cur"][1:], g["qparams_kept"][0][-1:]))


def replay_in_one_call(engine, g, meta):
    """All recorded iterations in ONE call: q, sigma^2 and the accept flags of every iteration against the reference's."""
    _init(engine, meta)
    n = len(g["z"])
    u = np.where(np.isnan(g["u"]), 1.0, g["u"])          # the reference draws no uniform for an out-of-bounds proposal
    sn = np.where(np.isnan(g["ssq_new"]), 0.0, g["ssq_new"])
    tq, ts, ta = engine.mcmc_replay_ssq(g["z"].reshape(n, 1, 1), u.reshape(n, 1), g["g"].reshape(n, 1), sn.reshape(n, 1))
    tq, ts, ta = np.asarray(tq)[:, 0, 0], np.asarray(ts)[:, 0], np.asarray(ta)[:, 0]
    np.testing.assert_allclose(tq, _q_after(g), rtol=1e-14, atol=0)
    np.testing.assert_allclose(ts, g["std2_after"], rtol=1e-13, atol=0)
    assert np.array_equal(ta.astype(bool), g["q_cur"] != _q_after(g))  # accepted <=> the chain moved
    nb = meta["nburn"]
    np.testing.assert_allclose(tq[nb - 1:], g["qparams_kept"][0], rtol=1e-14)
    np.testing.assert_allclose(ts[nb - 1:], g["std2_kept"], rtol=1e-13)
    c = engine.counters()
    inb = g["inb"].astype(bool)
    assert c["evaluated"] == int(inb.sum()) and c["out_of_bounds"] == int((~inb).sum()) and c["accepted"] == int(ta.sum())
    return tq, ts, ta


def replay_step_by_step(engine, g, meta):
    """One iteration per call, the way a host that evaluates the model itself drives the chain: rsf_mcmc_propose must
    announce the reference's proposal and box decision, the proposal covariance before every iteration must be the
    reference's Vold (the dict prior's Cholesky-factor-as-covariance quirk included), then the step."""
    _init(engine, meta)
    q_after = _q_after(g)
    for i in range(len(g["z"])):
        V = float(np.asarray(engine.get_state()[3])[0, 0, 0])
        assert abs(V - g["vold"][i]) <= 1e-13 * abs(g["vold"][i]), (i, V, g["vold"][i])
        q_new, inb = engine.mcmc_propose([[g["z"][i]]])
        assert bool(np.asarray(inb)[0]) == bool(g["inb"][i]), i
        assert abs(float(np.asarray(q_new)[0, 0]) - g["q_prop"][i]) <= 1e-14 * abs(g["q_prop"][i]), i
        u = 1.0 if np.isnan(g["u"][i]) else g["u"][i]
        sn = 0.0 if np.isnan(g["ssq_new"][i]) else g["ssq_new"][i]
        tq, ts, _ = engine.mcmc_replay_ssq([[[g["z"][i]]]], [[u]], [[g["g"][i]]], [[sn]])
        assert abs(float(np.asarray(tq)[0, 0, 0]) - q_after[i]) <= 1e-14 * abs(q_after[i]), i
        ssq = float(np.asarray(engine.get_state()[1])[0])
        assert abs(ssq - g["ssq_after"][i]) <= 1e-14 * abs(g["ssq_after"][i]), i
        assert abs(float(np.asarray(ts)[0, 0]) - g["std2_after"][i]) <= 1e-13 * g["std2_after"][i], i


def duck_model_chain(pkg, golden, case, engine=None):
    """This package's MCMC.sample() on tests/duck_model.DecayModel (any object with .Dc and .evaluate()) under the seed of
    the reference's recorded run → (chain, std2, Vstart, model calls) and the golden arrays."""
    from duck_model import observation

    g = golden.npz("duck_model")
    tag = case["tag"]
    prior = case["prior"] if isinstance(case["prior"], list) else {int(k): v for k, v in case["prior"].items()}
    model, data = observation(case["dc_true"], case["seed_data"])
    np.testing.assert_array_equal(data, g[f"{tag}_data"])
    mc = pkg.MCMC(model, data, case["dc_true"], prior, case["qstart"], nsamples=case["nsamples"], lstm_model=None, adapt_interval=10,
                  verbose=False)
    if engine is not None:
        mc._host_engine = engine  # the CPU suite drives the checker library through the same class
    np.random.seed(case["seed_chain"])
    qp = mc.sample(False)
    return qp, np.asarray(mc.std2), np.asarray(mc.Vstart), model.calls, g


def adapt_matches_numpy_on_degenerate_windows(pkg, engine, trials=600):
    """MCMC.update_covariance_matrix's arithmetic (MCMC.py:200-203) through rsf_mcmc_adapt, against NumPy itself: for a window
    in which the chain never moved, np.cov is exactly 0 for some sample values (np.linalg.cholesky raises: the reference
    keeps its proposal, MCMC.py:524-527) and ~1e-31 for others (the Cholesky "succeeds" and the reference's proposal
    collapses) — which of the two is a matter of the low bits of the sample through NumPy's pairwise summation.  The library
    must raise exactly where NumPy raises and return NumPy's factor elsewhere; also for two-valued and ordinary windows, and
    for every window length up to the limit."""
    rng = np.random.default_rng(12)
    raised = collapsed = 0
    for t in range(trials):
        n = int(rng.integers(2, 129)) if t % 3 == 0 else 10
        kind = t % 4
        if kind == 0:
            w = np.full(n, rng.uniform(1.0, 5000.0))                      # the chain never moved
        elif kind == 1:
            a, b = rng.uniform(1.0, 5000.0, 2)
            w = np.where(np.arange(n) < rng.integers(1, n), a, b)         # it moved once
        else:
            w = rng.uniform(900.0, 1100.0, n)
        scale = 2.38 ** 2 / 2
        try:
            ref = np.linalg.cholesky(np.atleast_2d(scale * np.cov(w.reshape(1, n))))[0, 0]
        except np.linalg.LinAlgError:
            ref = None
        try:
            got = engine.mcmc_adapt(w.reshape(n, 1), "reference_dict", prior_len=2)[0, 0]
        except pkg._abi.RsfError as ex:
            assert ex.code == pkg._abi.ERR_NOT_POSDEF
            got = None
        assert (ref is None) == (got is None), (t, n, kind, ref, got, w[:3])
        if ref is None:
            raised += 1
        else:
            assert abs(got - ref) <= 4e-15 * ref, (t, n, kind, ref, got)
            collapsed += kind == 0
    assert raised > 20 and collapsed > 20, (raised, collapsed)  # both outcomes of the never-moved window really occur
