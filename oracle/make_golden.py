#!/usr/bin/env python3
"""
make_golden.py — capture golden vectors from the LIVE reference (build container only).

The reference tree (/root/reference, read-only) ships no tests, fixtures or golden files, so
every parity vector under tests/golden/ is produced here by importing the reference unmodified
and recording what it computes.  Only data is written: inputs, outputs and the random variates
the reference consumed.  No reference source text is copied anywhere.

Import recipe: the reference's `imports.py` hard-imports `mysql.connector`,
which is not installed and is not on the hot path; an empty module object is registered under
that name so the import line passes.  Nothing from it is ever called.  MPLBACKEND=Agg.

Usage:  python oracle/make_golden.py [--long]      (--long adds the 1000-proposal config-1 run,
                                                     about 3 minutes of single-core CPU)
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
_m, _c = types.ModuleType("mysql"), types.ModuleType("mysql.connector")
_m.connector = _c
sys.modules["mysql"], sys.modules["mysql.connector"] = _m, _c
REF = os.environ.get("RSF_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import MCMC as ref_mcmc_module  # noqa: E402
from MCMC import MCMC  # noqa: E402
from RateStateModel import RateStateModel  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


# ---------------------------------------------------------------------------------------------
def forward_vectors():
    """G1: RateStateModel.evaluate()[1] (clean acceleration) on a grid of inputs."""
    cases, series = [], {}

    def run(tag, n, dc, damping=True, a=None, b=None):
        m = RateStateModel(number_time_steps=n)
        m.RadiationDamping = damping
        if a is not None:
            m.a, m.b = a, b
        m.Dc = dc
        np.random.seed(0)
        t, acc, _ = m.evaluate()
        series[tag] = acc
        cases.append(dict(tag=tag, nsteps=n, dc=dc, damping=damping, a=m.a, b=m.b, t_last=float(t[-1]), nout=len(acc)))

    for dc in (1.0, 10.0, 100.0, 1000.0, 1325.0, 5000.0, 9999.0):
        run(f"n500_dc{dc:g}", 500, dc)
    for dc in (100.0, 1000.0, 5000.0):
        run(f"n500_dc{dc:g}_nodamp", 500, dc, damping=False)
        run(f"n2000_dc{dc:g}", 2000, dc)
    for a, b in ((0.015, 0.020), (0.012, 0.010)):
        run(f"n500_dc1000_a{a:g}_b{b:g}", 500, 1000.0, a=a, b=b)
    np.savez_compressed(os.path.join(OUT, "forward.npz"), **series)
    with open(os.path.join(OUT, "forward.json"), "w") as f:
        json.dump(dict(source="RateStateModel.evaluate()[1], RateStateModel.py:188-395", cases=cases), f, indent=1)


def nondefault_vectors():
    """G1b: the same, with every model attribute moved off its default (V_ref != 1, mu_t_zero != mu_ref, t_start != 0,
    other k1 / a / b) — pins how the restatements treat those attributes against the reference itself."""
    attrs = dict(V_ref=1.7, mu_ref=0.55, mu_t_zero=0.58, k1=3.0e-7, a=0.012, b=0.0155)
    cases, series = [], {}
    for damping in (True, False):
        for dc in (300.0, 1000.0, 6000.0):
            m = RateStateModel(number_time_steps=400, start_time=1.5, end_time=37.0)
            for k, v in attrs.items():
                setattr(m, k, v)
            m.RadiationDamping = damping
            m.Dc = dc
            np.random.seed(0)
            t, acc, _ = m.evaluate()
            tag = f"dc{dc:g}" + ("" if damping else "_nodamp")
            series[tag] = acc
            cases.append(dict(tag=tag, dc=dc, damping=damping, nout=len(acc), t_first=float(t[0]), t_last=float(t[-1])))
    np.savez_compressed(os.path.join(OUT, "forward_nondefault.npz"), **series)
    with open(os.path.join(OUT, "forward_nondefault.json"), "w") as f:
        json.dump(dict(source="RateStateModel.evaluate()[1] with non-default attributes, RateStateModel.py:167-395",
                       number_time_steps=400, start_time=1.5, end_time=37.0, attrs=attrs, cases=cases), f, indent=1)


# ---------------------------------------------------------------------------------------------
def make_data(n, dc_true, seed):
    m = RateStateModel(number_time_steps=n)
    m.Dc = dc_true
    np.random.seed(seed)
    _, acc, acc_noise = m.evaluate()
    return m, acc, acc_noise


def ssq_and_init_vectors():
    """G2: MCMC.SSqcalc on a q grid.  G3: compute_initial_covariance for list and dict priors."""
    model, _, data = make_data(500, 1000.0, 0)
    out = dict(data=data)
    qgrid = np.array([50.0, 200.0, 700.0, 950.0, 1000.0, 1000.001, 1050.0, 1500.0, 4000.0, 9000.0])
    mc = MCMC(model, data, 1000.0, ["Uniform", 0.0, 10000.0], 1000.0, nsamples=10, lstm_model=None)
    ssq = []
    for q in qgrid:
        np.random.seed(1)
        ssq.append(float(mc.SSqcalc(np.array([[q]]))[0, 0]))
    out["qgrid"], out["ssq"] = qgrid, np.array(ssq)
    init = {}
    for name, prior in (("list", ["Uniform", 0.0, 10000.0]), ("dict", {1: 0.0, 2: 10000.0})):
        for qstart in (1000.0, 400.0):
            mc = MCMC(model, data, 1000.0, prior, qstart, nsamples=10, lstm_model=None)
            np.random.seed(2)
            mc.compute_initial_covariance()
            init[f"{name}_q{qstart:g}"] = dict(prior_len=len(prior), qstart=qstart, std2_0=float(mc.std2[0]),
                                              vstart=float(mc.Vstart[0, 0]))
    np.savez_compressed(os.path.join(OUT, "ssq.npz"), **out)
    with open(os.path.join(OUT, "init.json"), "w") as f:
        json.dump(dict(source="MCMC.compute_initial_covariance, MCMC.py:206-266; data = ssq.npz['data']", cases=init), f, indent=1)


# ---------------------------------------------------------------------------------------------
class Recorder:
    """Wraps the RNG entry points the reference's sample() uses and records what it drew."""

    def __init__(self, mc):
        self.mc = mc
        self.rows = []
        self.cur = None

    def __enter__(self):
        self._mvn, self._rand = np.random.multivariate_normal, np.random.rand
        self._gamma = ref_mcmc_module.gamma
        self._ssqcalc = self.mc.SSqcalc
        rec = self

        def mvn(mean, cov, *a, **k):
            st = np.random.get_state()
            q_new = rec._mvn(mean, cov, *a, **k)
            after = np.random.get_state()
            np.random.set_state(st)
            z = np.random.standard_normal(1)[0]          # the one normal multivariate_normal consumed
            chk = np.random.get_state()
            assert chk[2] == after[2] and np.array_equal(chk[1], after[1]) and chk[3:] == after[3:]
            rec.cur = dict(z=float(z), q_prop=float(np.ravel(q_new)[0]), q_cur=float(np.ravel(mean)[0]),
                           vold=float(np.ravel(cov)[0]), inb=0, ssq_new=np.nan, u=np.nan)
            return q_new

        def rand(*a, **k):
            u = rec._rand(*a, **k)
            rec.cur["u"] = float(np.ravel(u)[0])
            return u

        class GammaProxy:
            @staticmethod
            def rvs(aval, scale=1.0, size=None):
                st = np.random.get_state()
                x = rec._gamma.rvs(aval, scale=scale, size=size)
                after = np.random.get_state()
                np.random.set_state(st)
                g = np.random.standard_gamma(aval)
                chk = np.random.get_state()
                assert chk[2] == after[2] and np.array_equal(chk[1], after[1]) and chk[3:] == after[3:]
                assert abs(g * float(np.ravel(scale)[0]) - float(np.ravel(x)[0])) <= 1e-15 * abs(float(np.ravel(x)[0]))
                rec.cur["g"] = float(g)
                return x

        def ssqcalc(q_new):
            s = rec._ssqcalc(q_new)
            if rec.cur is not None:
                rec.cur["inb"], rec.cur["ssq_new"] = 1, float(np.ravel(s)[0])
            return s

        np.random.multivariate_normal, np.random.rand = mvn, rand
        ref_mcmc_module.gamma = GammaProxy
        self.mc.SSqcalc = ssqcalc
        return self

    def __exit__(self, *exc):
        np.random.multivariate_normal, np.random.rand = self._mvn, self._rand
        ref_mcmc_module.gamma = self._gamma
        del self.mc.SSqcalc


def replay_vectors(tag, prior, qstart, nsamples, dc_true=1000.0, n=500, seed=2025):
    """G4-G6: an instrumented MCMC.sample(False): variates consumed + per-iteration results."""
    model, _, data = make_data(n, dc_true, seed)
    mc = MCMC(model, data, dc_true, prior, qstart, nsamples=nsamples, lstm_model=None)
    rec = Recorder(mc)
    rows = rec.rows
    # sample() gives no per-iteration hook; record a row each time update_standard_deviation runs
    orig_usd = mc.update_standard_deviation

    def usd(ssqprev):
        orig_usd(ssqprev)
        row = rec.cur
        row["ssq_after"], row["std2_after"] = float(np.ravel(ssqprev)[0]), float(np.ravel(mc.std2[-1])[0])
        rows.append(row)
        rec.cur = None

    mc.update_standard_deviation = usd
    with rec, quiet():
        qparams = mc.sample(False)
    std2_0, vstart = None, float(mc.Vstart[0, 0])
    # std2 was trimmed to [nburn:] by sample(); recompute std2[0] from the recorded first row
    mc2 = MCMC(model, data, dc_true, prior, qstart, nsamples=nsamples, lstm_model=None)
    np.random.seed(0)
    mc2.compute_initial_covariance()
    std2_0 = float(mc2.std2[0])
    ssq0 = float(np.ravel(mc2.SSqcalc(np.array([[qstart]])))[0])
    cols = {k: np.array([r[k] for r in rows]) for k in ("z", "u", "g", "inb", "ssq_new", "q_prop", "q_cur", "vold", "ssq_after", "std2_after")}
    np.savez_compressed(os.path.join(OUT, f"replay_{tag}.npz"), data=data, qparams_kept=qparams, std2_kept=np.asarray(mc.std2, dtype=np.float64).ravel(), **cols)
    meta = dict(source="MCMC.sample(False), MCMC.py:391-544", prior=(prior if isinstance(prior, list) else {str(k): v for k, v in prior.items()}),
                prior_is_dict=isinstance(prior, dict), qstart=qstart, nsamples=nsamples, nburn=mc.nburn, dc_true=dc_true, nsteps=n,
                data_seed=seed, std2_0=std2_0, vstart=vstart, ssq0=ssq0, n0=mc.n0, adapt_interval=mc.adapt_interval)
    with open(os.path.join(OUT, f"replay_{tag}.json"), "w") as f:
        json.dump(meta, f, indent=1)


def config1_run():
    """BASELINE config 1: 1 chain, 1000 proposals, nsteps=500, Dc_true=1000, list prior."""
    model, _, data = make_data(500, 1000.0, 2025)
    np.random.seed(2025)
    mc = MCMC(model, data, 1000.0, ["Uniform", 0.0, 10000.0], 1000.0, nsamples=1000, lstm_model=None)
    t0 = time.time()
    with quiet():
        q = mc.sample(False)
    wall = time.time() - t0
    np.savez_compressed(os.path.join(OUT, "config1.npz"), data=data, qparams_kept=q, std2_kept=np.asarray(mc.std2, dtype=np.float64).ravel())
    with open(os.path.join(OUT, "config1.json"), "w") as f:
        json.dump(dict(source="MCMC.sample(False) config 1", wall_s=wall, proposals_per_s=1000 / wall, mean=float(q.mean()), std=float(q.std()),
                       kept=int(q.shape[1]), cpu="build container, 1 core"), f, indent=1)


def rsf_driver_vectors(nsamples=30, seed_data=7, seed_chains=11):
    """A13 / A14: the reference's RSF.generate_time_series() (RSF.py:355-371) for a 3-value dc_list under
    np.random.seed, then — from ONE further seed, sequentially in dc_list order like RSF.inference's loop
    (RSF.py:1042-1044) — MCMC(model, data[i*N:(i+1)*N], dc, qpriors, qstart, nsamples=...).sample(False) per Dc
    (the slice and constructor arguments of RSF.perform_sampling_and_plotting, RSF.py:874-894).  MCMC is driven
    directly because RSF.inference cannot run as shipped (its JSON helpers are shadowed by the MySQL ones and
    sample(True) needs ffmpeg)."""
    from RSF import RSF

    kw = dict(number_slip_values=3, lowest_slip_value=500.0, largest_slip_value=2500.0, qstart=1000.0,
              qpriors=["Uniform", 0.0, 10000.0])
    problem = RSF(**kw)
    problem.model = RateStateModel(number_time_steps=500)
    np.random.seed(seed_data)
    data = problem.generate_time_series()
    n = problem.model.num_tsteps
    out = dict(data=data, dc_list=np.asarray(problem.dc_list, dtype=np.float64))
    np.random.seed(seed_chains)
    for i, dc in enumerate(problem.dc_list):
        mc = MCMC(problem.model, data[i * n:(i + 1) * n], dc, problem.qpriors, problem.qstart, lstm_model=None, nsamples=nsamples)
        with quiet():
            q = mc.sample(False)
        out[f"qparams_{i}"], out[f"std2_{i}"] = q, np.asarray(mc.std2, dtype=np.float64).ravel()
    np.savez_compressed(os.path.join(OUT, "rsf_driver.npz"), **out)
    with open(os.path.join(OUT, "rsf_driver.json"), "w") as f:
        json.dump(dict(source="RSF.generate_time_series (RSF.py:355-371) + per-Dc MCMC.sample(False) on data[i*N:(i+1)*N] (RSF.py:874-894)",
                       rsf_kwargs=kw, number_time_steps=500, nsamples=nsamples, seed_data=seed_data, seed_chains=seed_chains,
                       note="np.random.seed(seed_data) before generate_time_series(); np.random.seed(seed_chains) once before the "
                            "sequential per-Dc chains"), f, indent=1)


def duck_model_vectors():
    """Round 4: the reference's sampler on a model that is not a rate-and-state model (tests/duck_model.py: any object with
    .Dc and .evaluate(), MCMC.py:65-66, 127) — list prior (never adapts), dict prior (the adaptation quirk) and a box tight
    enough for out-of-bounds proposals.  Pins the drop-in's duck-typed path: same seed => same chain."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from duck_model import observation

    out, meta = {}, dict(source="MCMC(model=tests/duck_model.DecayModel).sample(False), MCMC.py:391-544", cases=[])
    for tag, prior, qstart, nsamples in (("list", ["Uniform", 0.5, 40.0], 6.0, 150), ("dict", {1: 0.5, 2: 40.0}, 6.0, 150),
                                         ("tightbox", ["Uniform", 3.93, 4.02], 3.98, 100)):
        model, data = observation()
        mc = MCMC(model, data, 4.0, prior, qstart, nsamples=nsamples, lstm_model=None, adapt_interval=10)
        np.random.seed(99)
        with quiet():
            qp = mc.sample(False)
        out[f"{tag}_data"], out[f"{tag}_qparams"], out[f"{tag}_std2"] = data, qp, np.asarray(mc.std2)
        out[f"{tag}_vstart"] = np.asarray(mc.Vstart)
        meta["cases"].append(dict(tag=tag, prior=prior if isinstance(prior, list) else {str(k): v for k, v in prior.items()},
                                  prior_is_dict=not isinstance(prior, list), qstart=qstart, nsamples=nsamples, dc_true=4.0,
                                  seed_data=314, seed_chain=99, model_calls=model.calls))
    np.savez_compressed(os.path.join(OUT, "duck_model.npz"), **out)
    with open(os.path.join(OUT, "duck_model.json"), "w") as f:
        json.dump(meta, f, indent=1)


def json_fixtures():
    """Files written by the REFERENCE's json_save_load.save_object (json_save_load.py:37-39, 128-130): an ndarray and a
    dict of ndarrays (nested, 2-D and scalar members) — fixtures for the repo's load_object / byte-equal save_object."""
    import json_save_load as ref_json

    rng = np.random.default_rng(3)
    vec = rng.standard_normal(12) * 1e-3
    obj = {"acc": vec, "grid": np.arange(6, dtype=np.float64).reshape(2, 3), "meta": {"dc": 1000.0, "n": 12, "tags": ["a", "b"]},
           "ints": np.arange(4)}
    ref_json.save_object(vec, os.path.join(OUT, "ref_written_array.json"))
    ref_json.save_object(obj, os.path.join(OUT, "ref_written_dict.json"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--long", action="store_true")
    ap.add_argument("--only-nondefault", action="store_true", help="write only forward_nondefault.* (added later)")
    ap.add_argument("--only-round2", action="store_true", help="write only the vectors added in round 2 (rsf_driver.*, "
                    "ref_written_*.json, replay_dict3.*)")
    ap.add_argument("--only-round4", action="store_true", help="write only the vectors added in round 4 (duck_model.*)")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    duck_model_vectors()
    if args.only_round4:
        sys.exit(0)
    rsf_driver_vectors()
    json_fixtures()
    replay_vectors("dict3", {0: "Uniform", 1: 0.0, 2: 10000.0}, 1000.0, 120)
    if args.only_round2:
        sys.exit(0)
    nondefault_vectors()
    if args.only_nondefault:
        sys.exit(0)
    forward_vectors()
    ssq_and_init_vectors()
    replay_vectors("list", ["Uniform", 0.0, 10000.0], 1000.0, 200)
    replay_vectors("dict", {1: 0.0, 2: 10000.0}, 1000.0, 200)
    replay_vectors("tightbox", ["Uniform", 950.0, 1050.0], 1000.0, 120)
    if args.long:
        config1_run()
    print("golden vectors written to", os.path.abspath(OUT))
