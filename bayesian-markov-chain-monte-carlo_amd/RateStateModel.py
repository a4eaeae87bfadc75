"""
RateStateModel — drop-in for the reference class of the same name (RateStateModel.py:15-395).

Same constructor, same public attributes, same `evaluate() -> (t, acc, acc_noise)`; the forward
solve runs on the GPU through the C ABI (rsf_forward_batch) as a fixed-step RK4 integration
(`substeps` steps per output interval) instead of the reference's SciPy dop853 call.  DESIGN.md
states the resulting tolerance ladder against the reference trajectory.

Additive surface: `substeps`, `precision`, `integrator`, `evaluate_batch(dc, a=None, b=None)`.
"""
import numpy as np

if __package__:
    from .engine import Engine
else:  # flat layout: this directory on sys.path, the reference's own import style (main.py:44-46)
    from engine import Engine

# RateStateModel.py:5-11
A = 0.011
B = 0.014
MU_REF = 0.6
V_REF = 1.0
K1 = 1.0e-7
START_TIME = 0.0
END_TIME = 50.0


def _scalar(x):
    """`model.Dc` may be a Python float or a 1-element array (MCMC.py:381 assigns q_new[0,])."""
    return float(np.asarray(x, dtype=np.float64).reshape(-1)[0])


class RateStateModel:
    def __init__(self, number_time_steps=500, start_time=START_TIME, end_time=END_TIME):
        self.a = A
        self.b = B
        self.mu_ref = MU_REF
        self.V_ref = V_REF
        self.k1 = K1
        self.t_start = start_time
        self.t_final = end_time
        self.num_tsteps = number_time_steps
        self.delta_t = (end_time - start_time) / number_time_steps
        self.mu_t_zero = MU_REF
        self.RadiationDamping = True
        self.Dc = None
        self.substeps = 1  # RK4 steps per delta_t (additive knob; 1 = BASELINE's "fixed-step RK4 nsteps")
        self.precision = "float64"  # additive: "float32" integrates the ODE in single precision (tolerance sweeps)
        self.integrator = "rk4"     # additive: "dop853" = the reference's own adaptive scheme (scipy ode, rtol 1e-6, atol 1e-10)
        self._engine = None
        self._engine_key = None

    # ---- engine plumbing ----------------------------------------------------------------
    def _model_key(self):
        return (self.a, self.b, self.mu_ref, self.V_ref, self.k1, self.t_start, self.t_final, self.num_tsteps, self.delta_t,
                self.mu_t_zero, bool(self.RadiationDamping), int(self.substeps), self.precision, self.integrator)

    def engine(self):
        """Host-memory Engine bound to the HIP library, re-armed when an attribute changed."""
        if self._engine is None:
            self._engine = Engine(mem="host")
        key = self._model_key()
        if key != self._engine_key:
            self._engine.set_model(self, self.substeps)
            self._engine_key = key
        return self._engine

    def num_outputs(self):
        return int(np.floor((self.t_final - self.t_start) / self.delta_t))  # RateStateModel.py:358

    def time_axis(self):
        """t[k] accumulated the way the reference stores r.t (RateStateModel.py:382-384)."""
        n = self.num_outputs()
        t = np.empty(n)
        t[0] = self.t_start
        for k in range(1, n):
            t[k] = t[k - 1] + self.delta_t
        return t

    # ---- reference surface --------------------------------------------------------------
    def evaluate(self):
        """→ (t, acc, acc_noise); acc_noise = acc + |acc|·N(0,1) from the global NumPy RNG
        (RateStateModel.py:392), so a seeded caller sees the reference's draw order."""
        if self.Dc is None:
            raise ValueError("RateStateModel.Dc must be set before evaluate()")
        _, acc = self.engine().forward([_scalar(self.Dc)], want_acc=True)
        acc = np.ascontiguousarray(acc[:, 0])
        acc_noise = acc + 1.0 * np.abs(acc) * np.random.randn(acc.shape[0])
        return self.time_axis(), acc, acc_noise

    # ---- batched surface (additive) -----------------------------------------------------
    def evaluate_batch(self, dc, a=None, b=None):
        """Clean acceleration for C parameter sets in one launch → ndarray (C, nout)."""
        _, acc = self.engine().forward(np.asarray(dc, dtype=np.float64).reshape(-1), a=a, b=b, want_acc=True)
        return np.ascontiguousarray(acc.T)
