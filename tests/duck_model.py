"""
A model that is NOT a rate-and-state model: the smallest object that satisfies the reference sampler's model contract
(MCMC.py:65-66, 127, 381-384) — a settable `.Dc` and an `.evaluate()` whose second element is the clean series.

Shared by oracle/make_golden.py (which runs the REFERENCE's MCMC on it and records the chain) and the tests (which run this
package's MCMC on it): a damped oscillation whose decay time is the parameter.  `evaluate()` also draws from NumPy's
global stream, like the reference's own model does (RateStateModel.py:392), so the fixture pins the order in which the
sampler and the model consume that stream.
"""
import numpy as np


class DecayModel:
    def __init__(self, n=160, t_end=12.0):
        self.t = np.linspace(0.0, t_end, n)
        self.Dc = None
        self.calls = 0

    def evaluate(self):
        dc = float(np.asarray(self.Dc, dtype=np.float64).reshape(-1)[0])  # the sampler assigns a 1-element array (MCMC.py:381)
        self.calls += 1
        clean = np.exp(-self.t / dc) * np.sin(3.0 * self.t) + 0.1 * np.log1p(dc)
        noisy = clean + 0.02 * np.random.randn(self.t.shape[0])
        return self.t, clean, noisy


def observation(dc_true=4.0, seed=314):
    """(model, noisy data) in the way the reference's driver makes its data: the model's own noisy series."""
    m = DecayModel()
    m.Dc = dc_true
    np.random.seed(seed)
    return m, m.evaluate()[2]
