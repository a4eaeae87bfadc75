#!/usr/bin/env python3
"""Wall time of the drop-in single-chain MCMC.sample() (reference semantics, NumPy-seeded variates, one fused
kernel launch per proposal) for both integrators.   python tools/sample_latency.py"""
import time, numpy as np, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bayesian_markov_chain_monte_carlo_amd as rsf
for integ in ("rk4", "dop853"):
    model = rsf.RateStateModel(number_time_steps=500); model.integrator = integ
    model.Dc = 1000.0
    np.random.seed(2025)
    t, acc, acc_noise = model.evaluate()
    mc = rsf.MCMC(model, acc_noise, 1000.0, ["Uniform", 0.0, 10000.0], 1000.0, nsamples=1000, lstm_model=None, verbose=False)
    mc.sample(False)
    t0 = time.perf_counter(); q = mc.sample(False); dt = time.perf_counter() - t0
    print(integ, "MCMC.sample 1000 proposals: %.3f s  (%.3f ms per proposal)" % (dt, dt), q.shape, q.mean())
