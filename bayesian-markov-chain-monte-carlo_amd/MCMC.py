"""
MCMC — drop-in for the reference sampler class (MCMC.py:4-544) with the hot loop on the GPU.

`MCMC(model, data, dc_true, qpriors, qstart, nsamples, lstm_model, adapt_interval, verbose)` and
`.sample(MAKE_ANIMATIONS)` keep the reference's signatures, return shapes and quirk modes
(SURVEY Appendix A).  Each iteration — propose, box test, forward solve, sum of squares,
accept/reject, inverse-gamma σ² update, adaptation — executes in the fused HIP kernel
(rsf_mcmc_replay / rsf_mcmc_run); the host only feeds random variates.

`sample()` draws those variates from the *global NumPy RNG in the reference's order*
(MCMC.py:497, 331, 160 and the N unused normals of every RateStateModel.evaluate call,
RateStateModel.py:392), so `np.random.seed(s)` selects the same chain the reference would run.
`sample_batched()` (additive) runs many independent chains per launch with the on-device
Philox stream and is the throughput path.

The model contract is the reference's (MCMC.py:65-66, 127, 381-384): ANY object with a settable `.Dc`
and `.evaluate()` whose second element is the clean series.  This package's RateStateModel is integrated
on the device inside the fused kernel; for any other model the host calls `model.evaluate()` where the
reference would and the device runs the chain step on the sum of squares it is handed
(rsf_mcmc_init_state / rsf_mcmc_propose / rsf_mcmc_replay_ssq).
"""
import numpy as np

if __package__:
    from . import _figures
    from ._abi import ADAPT_MODES, ERR_NOT_POSDEF, RsfError
    from .engine import Engine
else:  # flat layout: this directory on sys.path, the reference's own import style (main.py:44-46)
    import _figures
    from _abi import ADAPT_MODES, ERR_NOT_POSDEF, RsfError
    from engine import Engine


class PosteriorPool:
    """Result of sample_batched: kept draws of every chain, iteration-major."""

    def __init__(self, samples, std2, accept_rate, stats, nburn):
        self.samples = samples          # (n_keep, C, d)
        self.std2 = std2                # (n_keep, C)
        self.accept_rate = accept_rate  # accepted / proposals over all chains
        self.stats = stats
        self.nburn = nburn

    def pooled(self):
        """(d, n_keep*C): every kept draw of every chain, the reference's (d, n) layout."""
        n, C, d = self.samples.shape
        return np.ascontiguousarray(self.samples.reshape(n * C, d).T)


class MCMC:
    def __init__(self, model, data, dc_true, qpriors, qstart, nsamples=100, lstm_model={}, adapt_interval=10,
                 verbose=True):
        self.model = model
        self.qstart = qstart
        self.qpriors = qpriors
        self.nsamples = nsamples
        self.nburn = int(nsamples / 2)
        self.verbose = verbose
        self.adapt_interval = adapt_interval
        self.data = data
        self.lstm_model = lstm_model
        self.n0 = 0.01
        if self._multi_parameter():
            # additive (BASELINE config 5): joint (Dc, a, b) — one ["Uniform", lo, hi] spec per parameter, qstart a 3-vector.  The
            # reference's sampler has one parameter (MCMC.py:98, 381); the three-parameter chains run through sample_batched only.
            if len(self.qpriors) != 3 or np.size(self.qstart) != 3:
                raise ValueError("joint inference takes three prior specs [[name, lo, hi]] x 3 for (Dc, a, b) and a 3-vector qstart")
            self.qstart_limits = np.array([[p[1], p[2]] for p in self.qpriors], dtype=np.float64)
        else:
            self.qstart_limits = np.array([[self.qpriors[1], self.qpriors[2]]])
        self.dc_true = dc_true
        # additive: consume the N normals each reference forward solve wastes (RateStateModel.py:392)
        self.replay_reference_rng = True

    # ---- helpers ------------------------------------------------------------------------
    def _device_model(self):
        """True: the model is this package's RateStateModel, integrated on the device inside the sampler kernel.
        False: any other object with `.Dc` and `.evaluate()` — the host evaluates it, the device runs the chain step."""
        return hasattr(self.model, "engine")

    def _engine(self):
        if self.lstm_model:
            raise NotImplementedError("the reduced-order-model hook (MCMC.py:124-125) has no implementation "
                                      "in the reference either; pass a falsy lstm_model")
        if self._device_model():
            return self.model.engine()
        if getattr(self, "_host_engine", None) is None:
            self._host_engine = Engine(mem="host")  # no set_model: the chain logic alone (rsf_mcmc_init_state)
        return self._host_engine

    def _prior_is_dict(self):
        return hasattr(self.qpriors, "keys")

    def _multi_parameter(self):
        """Three prior specs, one per parameter of (Dc, a, b), instead of the reference's single ["Uniform", lo, hi]."""
        try:
            first = self.qpriors[0]
        except (KeyError, IndexError, TypeError):
            return False
        return not isinstance(first, (str, bytes)) and np.ndim(first) == 1 and len(first) == 3

    @property
    def n_params(self):
        return 3 if self._multi_parameter() else 1

    def _adapt_mode(self):
        # list prior: update_covariance_matrix raises AttributeError, swallowed at MCMC.py:524-527 => never adapts.
        # Three parameters (this build's extension): corrected adaptive Metropolis — the initial proposal follows the (Dc, a)
        # ridge only locally, and what the chains learn about it is what makes them mix
        if self._multi_parameter():
            return "am"
        return "reference_dict" if self._prior_is_dict() else "none"

    def _init_chains(self, eng, q0, seed=0, chain_offset=0, adapt_mode=None):
        data = np.ascontiguousarray(self.data, dtype=np.float64).reshape(-1)
        lo, hi = self.qstart_limits[:, 0], self.qstart_limits[:, 1]
        # forward-difference step of the initial sensitivities: the reference's 1e-6 for its one parameter (MCMC.py:251); 1e-4 for
        # the three-parameter extension, whose regularised covariance does not need the small step and is cleaner without it
        eng.mcmc_init(q0, data, lo, hi, seed=seed, chain_offset=chain_offset, n0=self.n0,
                      prior_len=len(self.qpriors), adapt_mode=adapt_mode or self._adapt_mode(),
                      adapt_interval=self.adapt_interval, fd_rel_step=1e-4 if self._multi_parameter() else 1e-6)

    # ---- reference sub-methods (public names kept; each one runs its step on the device) -----------------------
    # A caller that composes them the way the reference's own loop does (MCMC.py:494-527) gets the arithmetic of the fused
    # kernel: acceptreject and update_standard_deviation are one replayed kernel iteration on the engine's chain,
    # update_covariance_matrix the kernel's adaptation step on the given window.  The variates come from NumPy's global
    # stream in the reference's order, exactly as in sample().
    def evaluate_model(self):
        if self.lstm_model:
            raise NotImplementedError("reduced-order-model hook: no such class exists in the reference")
        return self.model.evaluate()[1]

    def SSqcalc(self, q_new):
        """Sum of squares at q_new (d, 1) → (1, 1) (MCMC.py:381-387); one GPU forward solve — or, for a model that is not
        this package's RateStateModel, one `model.evaluate()` on the host."""
        self.model.Dc = q_new[0, ]
        if not self._device_model():
            acc = np.asarray(self.evaluate_model(), dtype=np.float64)
            return np.sum((acc.reshape(1, -1) - np.asarray(self.data, dtype=np.float64).reshape(1, -1)) ** 2, axis=1, keepdims=True)
        eng = self._engine()
        dc = float(np.asarray(self.model.Dc, dtype=np.float64).reshape(-1)[0])
        ssq, _ = eng.forward([dc], data=np.asarray(self.data, dtype=np.float64).reshape(-1), want_ssq=True, want_acc=False)
        return ssq.reshape(1, 1)

    def _chain_key(self):
        # the chain on the engine belongs to THIS sampler and THIS observation: a different `data` array means new chains
        return (id(self), id(self.data), len(self.data))

    def _scratch_chain(self, eng):
        """The engine's one-chain sampler state the sub-methods work on (created by compute_initial_covariance, or here)."""
        if eng.n_chains != 1 or eng.n_params != 1 or getattr(eng, "_chain_owner", None) != self._chain_key():
            if self._device_model():
                self._init_chains(eng, np.array([[float(self.qstart)]]))
            else:  # any state will do: every sub-method sets the state it works from
                eng.mcmc_init_state([[float(self.qstart)]], [0.0], [1.0], [[[0.0]]], self.qstart_limits[:, 0], self.qstart_limits[:, 1],
                                    n0=self.n0, prior_len=len(self.qpriors), adapt_mode=self._adapt_mode(),
                                    adapt_interval=self.adapt_interval)
            eng._chain_owner = self._chain_key()
        return eng

    def _one_iteration(self, eng, q, ssq, std2, V, z, u, g, ssq_new=None):
        """One kernel iteration from an explicit chain state → (q, SSq, sigma^2, accepted) after it.  rsf_mcmc_replay (the
        kernel solves the proposal itself), or rsf_mcmc_replay_ssq when the host has evaluated the model (ssq_new)."""
        eng.set_state(q=[[q]], ssq=[ssq], std2=[std2], V=[[[V]]])
        if ssq_new is None:
            tq, ts, ta = eng.mcmc_replay(np.array([[[z]]]), np.array([[u]]), np.array([[g]]))
        else:
            tq, ts, ta = eng.mcmc_replay_ssq(np.array([[[z]]]), np.array([[u]]), np.array([[g]]), np.array([[ssq_new]]))
        return float(tq[0, 0, 0]), float(eng.get_state()[1][0]), float(ts[0, 0]), bool(ta[0, 0])

    def acceptreject(self, q_new, SSqprev, std2):
        """(accept, SSq) for the proposal q_new (MCMC.py:268-333): the kernel's box test, forward solve and accept test, with
        the uniform drawn from NumPy only when the proposal is inside the box (the reference's draw order)."""
        eng = self._scratch_chain(self._engine())
        q = float(np.asarray(q_new, dtype=np.float64).reshape(-1)[0])
        lo, hi = float(self.qstart_limits[0, 0]), float(self.qstart_limits[0, 1])
        u, ssq_new = 1.0, None
        if q > lo and q < hi:
            if not self._device_model():
                ssq_new = float(self.SSqcalc(np.asarray(q_new, dtype=np.float64).reshape(-1, 1))[0, 0])  # the model draws what it draws
            elif self.replay_reference_rng:
                np.random.randn(len(self.data))  # the N normals the reference's forward solve wastes (RateStateModel.py:392)
            u = np.random.rand()
        elif not self._device_model():
            ssq_new = 0.0  # out of bounds: never read
        # proposal = q + chol(V) z with V = 0: the kernel proposes exactly q_new from the state (q_new, SSqprev, std2)
        _, ssq, _, accept = self._one_iteration(eng, q, float(np.asarray(SSqprev).reshape(-1)[0]), float(std2), 0.0, 0.0, u, 1.0, ssq_new)
        return accept, (np.array([[ssq]]) if accept else SSqprev)

    def update_standard_deviation(self, SSqprev):
        """Appends sigma^2 ~ InvGamma(0.5 (n0 + N), 0.5 (n0 sigma^2 + SSq)) (MCMC.py:129-160) — the kernel's Gibbs step: one
        iteration whose proposal is out of bounds (z = +inf), so that nothing else of the chain moves."""
        eng = self._scratch_chain(self._engine())
        g = np.random.standard_gamma(0.5 * (self.n0 + len(self.data)))  # == gamma.rvs(aval, scale=1/bval) * bval
        _, _, s2, _ = self._one_iteration(eng, float(self.qstart), float(np.asarray(SSqprev).reshape(-1)[0]), float(self.std2[-1]),
                                          1.0, np.inf, 1.0, g, None if self._device_model() else 0.0)
        self.std2.append(s2)

    def update_covariance_matrix(self, qparams):
        """chol(2.38^2 / len(qpriors.keys()) * cov(last adapt_interval samples)) (MCMC.py:162-204), on the device.  A list
        prior has no .keys(): AttributeError, as in the reference (whose loop swallows it: the chain never adapts)."""
        n_keys = len(self.qpriors.keys())
        window = np.asarray(qparams, dtype=np.float64)[:, -self.adapt_interval:].T
        if window.shape[1] != 1:
            # the reference's np.cov / cholesky take any number of rows, but its sampler has ONE parameter (MCMC.py:98, 381) and
            # so has this quirk mode on the device; say so instead of reporting a covariance failure the caller would swallow
            raise NotImplementedError(f"update_covariance_matrix: qparams has {window.shape[1]} rows; the reference sampler and its "
                                      "dict-prior adaptation are one-parameter (corrected adaptation for 3 parameters: "
                                      "sample_batched(adapt_mode='am'))")
        try:
            return self._engine().mcmc_adapt(window, "reference_dict", prior_len=n_keys)
        except RsfError as ex:
            if ex.code == ERR_NOT_POSDEF:  # only this status is np.linalg.cholesky's failure (MCMC.py:203, 524-527)
                raise np.linalg.LinAlgError("Matrix is not positive definite") from ex
            raise

    def compute_initial_covariance(self):
        """std2[0] and Vstart (MCMC.py:244-266): from the device init kernel, or — for a model that is not this package's
        RateStateModel — from two `model.evaluate()` calls on the host, the reference's own steps."""
        eng = self._engine()
        if self._device_model():
            self._init_chains(eng, np.array([[float(self.qstart)]]))
            eng._chain_owner = self._chain_key()
            _, _, std2, V = eng.get_state()
            self.std2 = [float(std2[0])]
            self.Vstart = V.reshape(1, 1).copy()
            self.model.Dc = self.qstart * (1 + 1e-6)  # the reference leaves the model perturbed (MCMC.py:251)
            return
        data = np.asarray(self.data, dtype=np.float64).reshape(1, -1)
        self.model.Dc = self.qstart
        acc = np.asarray(self.evaluate_model(), dtype=np.float64).reshape(1, -1)           # MCMC.py:245-248
        self.model.Dc = self.model.Dc * (1 + 1e-6)                                         # :251
        acc_dq = np.asarray(self.evaluate_model(), dtype=np.float64).reshape(1, -1)        # :254
        self.std2 = [np.sum((acc - data) ** 2, axis=1).item() / (acc.shape[1] - len(self.qpriors))]   # :261
        X = ((acc_dq - acc) / (self.model.Dc * 1e-6)).T                                    # :264, perturbed Dc in the denominator
        self.Vstart = self.std2[-1] * np.linalg.inv(X.T @ X)                               # :265-266

    # ---- the hot loop -------------------------------------------------------------------
    def sample(self, MAKE_ANIMATIONS=False):
        """One chain, nsamples proposals → ndarray (1, nsamples + 1 - nburn)  (MCMC.py:391-544)."""
        if self._multi_parameter():
            raise NotImplementedError("sample() is the reference's one-parameter loop (MCMC.py:98, 381); joint (Dc, a, b) chains run "
                                      "through sample_batched")
        eng = self._engine()
        if not self._device_model():
            return self._sample_host_model(eng, MAKE_ANIMATIONS)
        N = len(self.data)
        burn = self.replay_reference_rng
        self.compute_initial_covariance()
        if burn:  # two solves in compute_initial_covariance + the initial SSqcalc
            np.random.randn(N), np.random.randn(N), np.random.randn(N)
        lo, hi = float(self.qstart_limits[0, 0]), float(self.qstart_limits[0, 1])
        aval = 0.5 * (self.n0 + N)
        qparams = np.empty((1, self.nsamples + 1))
        qparams[0, 0] = self.qstart
        std2 = list(self.std2)
        iaccept = 0
        # the host needs the current point and proposal variance only to know whether the reference would draw u
        # (in-bounds test): q follows from the trace row, V changes only where the chain adapts
        q_state, _, _, V_state = eng.get_state()
        q_cur, v_cur = float(q_state[0, 0]), float(V_state[0, 0, 0])
        adapts = self._adapt_mode() != "none"
        for isample in range(self.nsamples):
            z = np.random.standard_normal()  # the single normal multivariate_normal consumes (MCMC.py:497)
            with np.errstate(invalid="ignore"):
                q_new = q_cur + np.sqrt(v_cur) * z
            u = 1.0
            if q_new > lo and q_new < hi:  # the reference draws u only for in-bounds proposals
                if burn:
                    np.random.randn(N)
                u = np.random.rand()
            g = np.random.standard_gamma(aval)
            tq, ts, ta = eng.mcmc_replay(np.array([[[z]]]), np.array([[u]]), np.array([[g]]))
            accept = bool(ta[0, 0])
            iaccept += accept
            q_cur = float(tq[0, 0, 0])
            qparams[0, isample + 1] = q_cur
            std2.append(float(ts[0, 0]))
            if adapts and (isample + 1) % self.adapt_interval == 0:
                v_cur = float(eng.get_state()[3][0, 0, 0])
            if self.verbose:
                print(isample, accept)
                print("Generated Sample ---- ", q_new)
        if self.verbose:
            print("acceptance ratio:", iaccept / self.nsamples)
        self.std2 = np.asarray(std2)[self.nburn:]
        self.acceptance_ratio = iaccept / self.nsamples
        if MAKE_ANIMATIONS:
            self._animate(qparams)
        return qparams[:, self.nburn:]

    def _sample_host_model(self, eng, MAKE_ANIMATIONS):
        """sample() for ANY model object with `.Dc` and `.evaluate()` (the reference's contract, MCMC.py:65-66, 127): the
        host calls the model exactly where the reference does — so whatever the model draws from NumPy's global stream
        is drawn in the reference's order — and every chain step (proposal, box test, accept test, sigma^2, adaptation)
        runs on the device on the sum of squares it is handed."""
        N = len(self.data)
        self.compute_initial_covariance()                                   # MCMC.py:464
        ssq0 = float(self.SSqcalc(np.array([[float(self.qstart)]]))[0, 0])  # :468
        lo, hi = self.qstart_limits[:, 0], self.qstart_limits[:, 1]
        eng.mcmc_init_state([[float(self.qstart)]], [ssq0], [float(self.std2[-1])], np.reshape(self.Vstart, (1, 1, 1)), lo, hi,
                            n0=self.n0, prior_len=len(self.qpriors), adapt_mode=self._adapt_mode(), adapt_interval=self.adapt_interval)
        eng._chain_owner = self._chain_key()
        aval = 0.5 * (self.n0 + N)
        qparams = np.empty((1, self.nsamples + 1))
        qparams[0, 0] = self.qstart
        std2 = list(self.std2)
        iaccept = 0
        for isample in range(self.nsamples):
            z = np.array([[np.random.standard_normal()]])  # the single normal multivariate_normal consumes (MCMC.py:497)
            q_new, inb = eng.mcmc_propose(z)
            u, ssq_new = 1.0, 0.0
            if inb[0]:  # MCMC.py:322-331: the model is evaluated, then the uniform is drawn
                ssq_new = float(self.SSqcalc(q_new.reshape(-1, 1))[0, 0])
                u = np.random.rand()
            g = np.random.standard_gamma(aval)
            tq, ts, ta = eng.mcmc_replay_ssq(z.reshape(1, 1, 1), np.array([[u]]), np.array([[g]]), np.array([[ssq_new]]))
            accept = bool(ta[0, 0])
            iaccept += accept
            qparams[0, isample + 1] = float(tq[0, 0, 0])
            std2.append(float(ts[0, 0]))
            if self.verbose:
                print(isample, accept)
                print("Generated Sample ---- ", q_new.reshape(-1, 1))
        if self.verbose:
            print("acceptance ratio:", iaccept / self.nsamples)
        self.std2 = np.asarray(std2)[self.nburn:]
        self.acceptance_ratio = iaccept / self.nsamples
        if MAKE_ANIMATIONS:
            self._animate(qparams)
        return qparams[:, self.nburn:]

    def sample_batched(self, n_chains, seed=0, q0=None, jitter=None, n_iters=None, iters_per_launch=None,
                       adapt_mode=None, mem="device", device=-1, chain_offset=0, keep="post_burn", thin=1):
        """Throughput path (additive): n_chains independent chains, Philox variates on device.

        q0: (C,) / (C, d) start points; default qstart for every chain, optionally jittered
        uniformly in `jitter=(lo, hi)` with a NumPy generator seeded by `seed` and keyed by
        global chain id.  Returns a PosteriorPool of the post-burn-in draws; `thin=k` keeps every k-th kept draw
        (the pool of a long multi-GPU run need not hold, or all-gather, every iteration: SURVEY §8e)."""
        if int(thin) < 1:
            raise ValueError("thin must be >= 1")
        thin = int(thin)
        n_iters = self.nsamples if n_iters is None else n_iters
        nburn = int(n_iters / 2) if keep == "post_burn" else 0
        gids = chain_offset + np.arange(n_chains)
        d = self.n_params
        if q0 is None:
            q0 = np.tile(np.asarray(self.qstart, dtype=np.float64).reshape(1, d), (n_chains, 1))
            if jitter is not None:  # (lo, hi) for Dc, or one (lo, hi) per parameter: start points spread over that box
                jit = np.broadcast_to(np.asarray(jitter, dtype=np.float64).reshape(-1, 2), (d, 2)) if np.ndim(jitter) > 1 else None
                for r, g in enumerate(gids):
                    rng = np.random.default_rng([seed, int(g)])
                    if jit is None:
                        q0[r, 0] = rng.uniform(*jitter)
                    else:
                        q0[r] = rng.uniform(jit[:, 0], jit[:, 1])
        q0 = np.asarray(q0, dtype=np.float64).reshape(n_chains, -1)
        if q0.shape[1] != d:
            raise ValueError(f"q0 has {q0.shape[1]} columns, the sampler {d} parameter(s)")
        if not self._device_model():
            raise TypeError("sample_batched integrates the model on the device: `model` must be this package's RateStateModel "
                            "(sample() takes any model object with .Dc and .evaluate())")
        eng = Engine(mem=mem, device=device)
        try:
            eng.set_model(self.model, getattr(self.model, "substeps", 1))
            self._init_chains(eng, q0, seed=seed, chain_offset=chain_offset, adapt_mode=adapt_mode)
            step = iters_per_launch or n_iters
            kept_q, kept_s, done = [], [], 0
            while done < n_iters:
                n = min(step, n_iters - done)
                tq, ts, _ = eng.mcmc_run(n, traces=("q", "std2"))
                first = max(nburn - 1 - done, 0)  # trace row r is qparams column done + r + 1
                if first < n:
                    kept_q.append(tq[first:])
                    kept_s.append(ts[first:])
                done += n
            eng.sync()
            stats = eng.stats()
            cat = (lambda xs: np.concatenate([np.asarray(x.cpu() if hasattr(x, "cpu") else x) for x in xs], axis=0))
            samples, std2 = cat(kept_q)[::thin], cat(kept_s)[::thin]
        finally:
            eng.close()
        rate = stats["accepted"] / max(1, n_iters * n_chains)
        return PosteriorPool(samples, std2, rate, stats, nburn)

    # ---- visualisation (off the hot path; degrades gracefully) --------------------------
    def _animate(self, qparams):
        _figures.chain_movie(qparams[0], f"MCMC Sampling Evolution for dc = {self.dc_true:.2f} as True value",
                             f"mcmc_animation_dc_{self.dc_true:.2f}.mp4")


assert set(ADAPT_MODES) == {"none", "reference_dict", "am"}
