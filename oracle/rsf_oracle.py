"""
rsf_oracle.py — NumPy/SciPy twin of the CPU restatement (TEST INFRASTRUCTURE ONLY).

Nothing in the product path may import this module; only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg do.  It exists to tie the C restatement (rsf_oracle.c) and the
HIP kernels back to the reference's own behaviour:

  forward_dop853   RateStateModel.evaluate with the reference's actual integrator
                   (scipy dop853, rtol 1e-6, atol 1e-10, restarted per output interval;
                   RateStateModel.py:358-395).  Checked against tests/golden/forward_*.npz.
  forward_rk4      the fixed-step RK4 restatement, vectorised over parameter sets; same
                   arithmetic as rsf_oracle.c `solve`.
  ReferenceSampler MCMC.sample's logic (MCMC.py:391-544) as a replay machine that consumes
                   recorded variates, used to pin accept/reject, sigma^2 and adaptation logic
                   against tests/golden/replay_*.npz.

Parity pinning: the reference has no tests or golden vectors of its own; every comparison
here is against vectors captured from the live reference import by oracle/make_golden.py.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# RateStateModel.py:5-11
A, B, MU_REF, V_REF, K1, START_TIME, END_TIME = 0.011, 0.014, 0.6, 1.0, 1.0e-7, 0.0, 50.0


class ModelSpec:
    """Attributes of RateStateModel (RateStateModel.py:167-184) + the RK4 substep knob."""

    def __init__(self, number_time_steps=500, start_time=START_TIME, end_time=END_TIME, substeps=1):
        self.a, self.b, self.mu_ref, self.V_ref, self.k1 = A, B, MU_REF, V_REF, K1
        self.t_start, self.t_final = start_time, end_time
        self.num_tsteps = number_time_steps
        self.delta_t = (end_time - start_time) / number_time_steps
        self.mu_t_zero = MU_REF
        self.RadiationDamping = True
        self.substeps = substeps
        self.precision = "float64"
        self.integrator = "rk4"

    @property
    def nout(self):
        return int(np.floor((self.t_final - self.t_start) / self.delta_t))  # RateStateModel.py:358


def _friction(m, t, y, dc, a, b):
    """RateStateModel.py:318-355 (literal; works on scalars or arrays of lanes)."""
    kprime = 1e-2 * 10 / dc
    V_l = m.V_ref * (1 + np.exp(-t / 20) * np.sin(10 * t))
    temp = 1 / a * (y[0] - m.mu_ref - b * np.log(m.V_ref * y[1] / dc))
    v = m.V_ref * np.exp(temp)
    d1 = 1.0 - v * y[1] / dc
    d0 = kprime * V_l - kprime * v
    d2 = v / a * (d0 - b / y[1] * d1)
    if m.RadiationDamping:
        d0 = d0 - m.k1 * d2
        d2 = v / a * (d0 - b / y[1] * d1)
    return [d0, d1, d2]


def forward_dop853(m, dc, a=None, b=None):
    """The reference's own integration scheme (RateStateModel.py:374-389) → clean acc[nout]."""
    from scipy import integrate

    a = m.a if a is None else a
    b = m.b if b is None else b
    n = m.nout
    vel = np.zeros(n)
    acc = np.zeros(n)
    vel[0] = m.V_ref

    def rhs(t, y):
        return np.array(_friction(m, t, y, dc, a, b)).reshape(3, 1)

    r = integrate.ode(rhs).set_integrator("dop853", rtol=1e-6, atol=1e-10)
    r.set_initial_value([m.mu_t_zero, dc / m.V_ref, m.V_ref], m.t_start)
    k = 1
    while r.successful() and k < n:
        r.integrate(r.t + m.delta_t)
        vel[k] = r.y[2]
        acc[k] = (vel[k] - vel[k - 1]) / m.delta_t
        k += 1
    return acc


def forward_rk4(m, dc, a=None, b=None):
    """Fixed-step RK4 restatement, vectorised over lanes → acc[nout, C] (time-major)."""
    dc = np.atleast_1d(np.asarray(dc, dtype=np.float64))
    a = np.full_like(dc, m.a) if a is None else np.broadcast_to(np.asarray(a, dtype=np.float64), dc.shape)
    b = np.full_like(dc, m.b) if b is None else np.broadcast_to(np.asarray(b, dtype=np.float64), dc.shape)
    n, S = m.nout, m.substeps
    h = m.delta_t / S
    hh = 0.5 * h
    y = [np.full_like(dc, m.mu_t_zero), dc / m.V_ref, np.full_like(dc, m.V_ref)]
    acc = np.zeros((n, dc.size))
    vprev = y[2].copy()
    j = 0
    with np.errstate(all="ignore"):
        for k in range(1, n):
            for _ in range(S):
                t0, tm, t1 = m.t_start + j * hh, m.t_start + (j + 1) * hh, m.t_start + (j + 2) * hh
                k1 = _friction(m, t0, y, dc, a, b)
                k2 = _friction(m, tm, [y[i] + hh * k1[i] for i in range(3)], dc, a, b)
                k3 = _friction(m, tm, [y[i] + hh * k2[i] for i in range(3)], dc, a, b)
                k4 = _friction(m, t1, [y[i] + h * k3[i] for i in range(3)], dc, a, b)
                y = [y[i] + h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]) for i in range(3)]
                j += 2
            acc[k] = (y[2] - vprev) / m.delta_t
            vprev = y[2].copy()
    return acc


def ssq_rk4(m, dc, data, a=None, b=None):
    """MCMC.SSqcalc (MCMC.py:381-387) over lanes, sequential sum like rsf_oracle.c."""
    acc = forward_rk4(m, dc, a, b)
    s = np.zeros(acc.shape[1])
    for k in range(acc.shape[0]):
        s += (acc[k] - data[k]) ** 2
    return s


class ReferenceSampler:
    """MCMC.sample (MCMC.py:391-544) as a replay machine for ONE chain, one parameter.

    `ssq_fn(q) -> float` is the likelihood's sum of squares; the caller supplies either a
    forward model or the SSq values recorded from the reference.  `prior` is the reference's
    qpriors object (list ["Uniform", lo, hi] or dict {1: lo, 2: hi}); its type selects the
    adaptation quirk exactly as in the reference.
    """

    def __init__(self, ssq_fn, nout, prior, qstart, n0=0.01, adapt_interval=10):
        self.ssq_fn, self.nout, self.prior = ssq_fn, nout, prior
        self.lo, self.hi = prior[1], prior[2]  # MCMC.py:98
        self.n0, self.adapt_interval = n0, adapt_interval
        self.q = float(qstart)

    def set_initial(self, std2_0, vstart, ssq0):
        self.std2, self.V, self.ssq = [float(std2_0)], float(vstart), float(ssq0)
        self.qparams = [self.q]

    def step(self, isample, z, u, g, ssq_new=None):
        q_new = self.q + np.sqrt(self.V) * z  # MCMC.py:497 (d = 1)
        accept = bool(q_new > self.lo and q_new < self.hi)  # MCMC.py:318-320
        inb = accept
        if accept:
            s_new = self.ssq_fn(q_new) if ssq_new is None else ssq_new
            with np.errstate(all="ignore"):
                logalpha = min(0.5 * (self.ssq - s_new) / self.std2[-1], 0.0)  # MCMC.py:327
                accept = bool(logalpha > np.log(u))  # MCMC.py:331
        if accept:
            self.q, self.ssq = float(q_new), float(s_new)
        self.qparams.append(self.q)
        aval = 0.5 * (self.n0 + self.nout)  # MCMC.py:158
        bval = 0.5 * (self.n0 * self.std2[-1] + self.ssq)
        self.std2.append(1.0 / (g * (1.0 / bval)))  # gamma.rvs(aval, scale=1/bval) == g/bval
        if (isample + 1) % self.adapt_interval == 0:  # MCMC.py:523-527
            try:
                window = np.asarray(self.qparams[-self.adapt_interval:]).reshape(1, -1)
                vnew = 2.38 ** 2 / len(self.prior.keys()) * np.cov(window)  # list prior: AttributeError
                vnew = np.linalg.cholesky(np.reshape(vnew, (-1, 1)))
                self.V = float(vnew[0, 0])  # the factor is used as the covariance from here on
            except Exception:
                pass
        return inb, accept, float(q_new)


# --------------------------------------------------------------------------------------------
# ctypes binding of the C restatement (same ABI as the product library; include/rsf_abi.h)
# --------------------------------------------------------------------------------------------
def lib_path():
    return os.path.join(HERE, "librsf_oracle.so")


def build(force=False):
    """Compile rsf_oracle.c → librsf_oracle.so (gcc, no FMA contraction, OpenMP over chains)."""
    import subprocess

    src, out = os.path.join(HERE, "rsf_oracle.c"), lib_path()
    hdr = os.path.join(HERE, "..", "include", "rsf_abi.h")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return out
    subprocess.check_call(["make", "-s", "-C", HERE, "librsf_oracle.so"])
    return out


def load():
    """ctypes handle on the oracle library with argtypes set (tests/abi.py does the typing)."""
    path = lib_path()
    if not os.path.exists(path):
        build()
    return ctypes.CDLL(path)
