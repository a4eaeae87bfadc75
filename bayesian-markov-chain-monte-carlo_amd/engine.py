"""
Engine — thin object wrapper over the C ABI (include/rsf_abi.h).

One Engine owns one rsf_ctx.  With mem="host" every array argument is a NumPy array and calls
are synchronous; with mem="device" arguments are torch CUDA tensors (PyTorch is used only for
device memory and streams) and the work is ordered on the current torch stream.

The class is library-agnostic (`lib` is any handle typed by _abi.bind) so the test-suite can
drive the CPU oracle through the very same code; the product always passes _abi.load().
"""
import ctypes
import os

import numpy as np

if __package__:
    from . import _abi
else:  # flat layout: this directory on sys.path, the reference's own import style (main.py:44-46)
    import _abi


def _model_struct(model, substeps):
    m = _abi.Model()
    m.size = ctypes.sizeof(_abi.Model)
    m.flags = _abi.FLAG_RADIATION_DAMPING if getattr(model, "RadiationDamping", True) else 0
    precision = getattr(model, "precision", "float64")
    if precision not in ("float64", "float32"):
        raise ValueError(f"precision must be 'float64' or 'float32', not {precision!r}")
    if precision == "float32":
        m.flags |= _abi.FLAG_FP32_SOLVE
    integrator = getattr(model, "integrator", "rk4")
    if integrator not in ("rk4", "dop853"):
        raise ValueError(f"integrator must be 'rk4' or 'dop853', not {integrator!r}")
    if integrator == "dop853":
        if precision != "float64":
            raise ValueError("the dop853 integrator is float64 only")
        m.flags |= _abi.FLAG_DOP853
    m.nsteps = int(model.num_tsteps)
    # the reference steps by its `delta_t` attribute and counts floor((t_final - t_start)/delta_t) samples
    # (RateStateModel.py:176,358); the C ABI derives delta_t from (t_start, t_final, num_tsteps), so a model whose
    # delta_t was edited out of step with them would silently integrate something else here: refuse it
    dt_attr = getattr(model, "delta_t", None)
    if dt_attr is not None:
        dt = (float(model.t_final) - float(model.t_start)) / m.nsteps
        if abs(float(dt_attr) - dt) > 1e-12 * abs(dt):
            raise ValueError(f"model.delta_t = {dt_attr!r} is not (t_final - t_start)/num_tsteps = {dt!r}; "
                             "set t_start, t_final, num_tsteps and delta_t consistently")
    m.substeps = int(substeps)
    m.t_start, m.t_final = float(model.t_start), float(model.t_final)
    m.mu_ref, m.V_ref, m.k1 = float(model.mu_ref), float(model.V_ref), float(model.k1)
    m.mu_t_zero = float(model.mu_t_zero)
    m.a, m.b = float(model.a), float(model.b)
    return m


class Engine:
    def __init__(self, lib=None, mem="host", device=-1, block_threads=0, cpu_threads=0, stream=None, checker=False):
        if lib is None:
            lib = _abi.load()
            _abi.require_device(lib)
        elif lib.rsf_backend() != b"hip-gfx950" and not checker and os.environ.get("RSF_ALLOW_CHECKER_ENGINE") != "1":
            # `lib` exists so that the test-suite can drive the CPU oracle through this very class; nothing in the product may
            # end up on it by accident: a non-HIP library is refused unless THIS CALL declares itself a checker
            # (checker=True: __graft_entry__.smoke(), bench.py's cpu_baseline leg, tools/) — the environment variable is
            # the test-suite's process-wide form of the same declaration (tests/conftest.py) and is set nowhere else
            raise _abi.RsfError(-2, f"Engine(lib=...) was handed the {lib.rsf_backend().decode()!r} library: the product runs on "
                                    "csrc/librsf_hip.so only (no CPU fallback); a checker passes checker=True")
        self.lib = lib
        self.mem = mem
        self.device = device
        self._torch = None
        cfg = _abi.Config()
        cfg.size, cfg.version = ctypes.sizeof(_abi.Config), _abi.ABI_VERSION
        cfg.device = device
        cfg.mem_space = _abi.MEM_DEVICE if mem == "device" else _abi.MEM_HOST
        cfg.block_threads, cfg.cpu_threads = block_threads, cpu_threads
        if mem == "device":
            import torch

            self._torch = torch
            if device < 0:
                device = torch.cuda.current_device()
            self.device = cfg.device = device  # buffers and launches of this engine stay on this GPU whatever is current
            if stream is None:
                stream = torch.cuda.current_stream(device).cuda_stream
        cfg.stream = stream
        self._ctx = ctypes.c_void_p()
        _abi.check(lib, lib.rsf_create(ctypes.byref(cfg), ctypes.byref(self._ctx)))
        self.nout = None
        self.n_chains = self.n_params = None
        self.world = self.rank = 0  # set by comm_init

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.rsf_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        _abi.check(self.lib, self.lib.rsf_sync(self._ctx))

    # -- buffers --------------------------------------------------------------------------
    def _empty(self, shape, dtype=np.float64):
        if self.mem == "device":
            t = self._torch
            return t.empty(shape, dtype=t.uint8 if dtype == np.uint8 else t.float64, device=f"cuda:{self.device}")
        return np.empty(shape, dtype=dtype)

    def _in(self, x, dtype=np.float64):
        """Contiguous array in this engine's memory space (None passes through)."""
        if x is None:
            return None
        if self.mem == "device":
            t = self._torch
            if not isinstance(x, t.Tensor):
                x = t.as_tensor(np.ascontiguousarray(x, dtype=dtype))
            return x.to(device=f"cuda:{self.device}", dtype=t.float64).contiguous()
        return np.ascontiguousarray(x, dtype=dtype)

    @staticmethod
    def _ptr(x):
        if x is None:
            return None
        if isinstance(x, np.ndarray):
            return x.ctypes.data
        return x.data_ptr()

    # -- forward model --------------------------------------------------------------------
    def set_model(self, model, substeps=1):
        m = _model_struct(model, substeps)
        _abi.check(self.lib, self.lib.rsf_set_model(self._ctx, ctypes.byref(m)))
        n = ctypes.c_int32()
        _abi.check(self.lib, self.lib.rsf_model_nout(self._ctx, ctypes.byref(n)))
        self.nout = n.value
        self.model_args = (model, substeps)
        self.n_chains = self.n_params = None  # a new model invalidates the chains (rsf_abi.h: rsf_set_model)
        return self.nout

    def _need_model(self):
        if self.nout is None:
            raise _abi.RsfError(-3, "call set_model first")

    def _need_chains(self):
        if self.n_chains is None:
            raise _abi.RsfError(-3, "call mcmc_init first")

    def forward(self, dc, a=None, b=None, data=None, want_ssq=False, want_acc=True):
        """→ (ssq[C] | None, acc[nout, C] | None) for C parameter sets."""
        self._need_model()
        dc = self._in(np.atleast_1d(dc) if not hasattr(dc, "data_ptr") else dc)
        C = int(dc.shape[0])
        a, b, data = self._in(a), self._in(b), self._in(data)
        if want_ssq and data is None:
            raise ValueError("want_ssq needs data")
        if data is not None and int(data.shape[0]) != self.nout:
            raise ValueError(f"data has {int(data.shape[0])} entries, the model produces {self.nout}")
        ssq = self._empty((C,)) if want_ssq else None
        acc = self._empty((self.nout, C)) if want_acc else None
        _abi.check(self.lib, self.lib.rsf_forward_batch(self._ctx, C, self._ptr(dc), self._ptr(a), self._ptr(b),
                                                        self._ptr(data), self._ptr(ssq), self._ptr(acc)))
        return ssq, acc

    # -- sampler --------------------------------------------------------------------------
    def mcmc_init(self, q0, data, lo, hi, seed=0, chain_offset=0, n0=0.01, prior_len=0, adapt_mode="none",
                  adapt_interval=10, fd_rel_step=1e-6):
        self._need_model()
        q0 = self._in(q0)
        if q0.ndim == 1:
            q0 = q0.reshape(-1, 1)
        C, d = int(q0.shape[0]), int(q0.shape[1])
        data = self._in(data)
        n_groups = int(data.shape[0]) if data.ndim == 2 else 1  # (G, nout): one observation series per chain group
        if int(data.shape[-1]) != self.nout:
            raise ValueError(f"data has {int(data.shape[-1])} entries per series, the model produces {self.nout}")
        if C % n_groups:
            raise ValueError(f"{C} chains cannot be split evenly over {n_groups} observation groups")
        cfg = self._mcmc_config(C, d, lo, hi, seed, chain_offset, n0, prior_len, adapt_mode, adapt_interval, fd_rel_step, n_groups)
        _abi.check(self.lib, self.lib.rsf_mcmc_init(self._ctx, ctypes.byref(cfg), self._ptr(q0), self._ptr(data)))
        self.n_chains, self.n_params = C, d

    def _mcmc_config(self, C, d, lo, hi, seed=0, chain_offset=0, n0=0.01, prior_len=0, adapt_mode="none", adapt_interval=10,
                     fd_rel_step=1e-6, n_groups=1):
        lo, hi = np.broadcast_to(np.asarray(lo, dtype=np.float64), (d,)), np.broadcast_to(np.asarray(hi, dtype=np.float64), (d,))
        cfg = _abi.McmcConfig()
        cfg.size = ctypes.sizeof(_abi.McmcConfig)
        cfg.n_params, cfg.n_chains, cfg.chain_offset = d, C, int(chain_offset)
        cfg.seed, cfg.n0, cfg.prior_len = int(seed), float(n0), int(prior_len)
        cfg.adapt_mode = _abi.ADAPT_MODES[adapt_mode] if isinstance(adapt_mode, str) else int(adapt_mode)
        cfg.adapt_interval, cfg.fd_rel_step, cfg.n_groups = int(adapt_interval), float(fd_rel_step), n_groups
        for p in range(d):
            cfg.lo[p], cfg.hi[p] = float(lo[p]), float(hi[p])
        return cfg

    def mcmc_init_state(self, q, ssq, std2, V, lo, hi, **kw):
        """Chains from an explicit state (rsf_mcmc_init_state): no model, no observation — the sampler as an operator over
        a likelihood the caller evaluates; advanced by mcmc_replay_ssq only.  q (C, d), ssq (C,), std2 (C,), V (C, d, d)."""
        q = self._in(q)
        if q.ndim == 1:
            q = q.reshape(-1, 1)
        C, d = int(q.shape[0]), int(q.shape[1])
        ssq, std2, V = self._in(ssq), self._in(std2), self._in(V)
        if int(np.prod(ssq.shape)) != C or int(np.prod(std2.shape)) != C or int(np.prod(V.shape)) != C * d * d:
            raise ValueError("ssq and std2 hold one value per chain, V one (d, d) matrix per chain")
        cfg = self._mcmc_config(C, d, lo, hi, **kw)
        _abi.check(self.lib, self.lib.rsf_mcmc_init_state(self._ctx, ctypes.byref(cfg), self._ptr(q), self._ptr(ssq), self._ptr(std2),
                                                          self._ptr(V)))
        self.n_chains, self.n_params = C, d

    def mcmc_propose(self, z):
        """The proposals the next iteration will make from the normals z (C, d) → (q_new (C, d), in_bounds (C,) uint8)."""
        self._need_chains()
        C, d = self.n_chains, self.n_params
        z = self._in(z)
        if int(np.prod(z.shape)) != C * d:
            raise ValueError(f"z must hold {d} normals for each of the {C} chains")
        qn, inb = self._empty((C, d)), self._empty((C,), np.uint8)
        _abi.check(self.lib, self.lib.rsf_mcmc_propose(self._ctx, self._ptr(z), self._ptr(qn), self._ptr(inb)))
        return qn, inb

    def mcmc_replay_ssq(self, z, u, g, ssq_new, traces=True):
        """mcmc_replay with the proposals' sums of squares supplied by the caller (n, C): the chain logic alone."""
        z, u, g, ssq_new = self._in(z), self._in(u), self._in(g), self._in(ssq_new)
        n_iters = int(u.shape[0])
        tq, ts, ta = self._traces(n_iters, traces)
        _abi.check(self.lib, self.lib.rsf_mcmc_replay_ssq(self._ctx, n_iters, self._ptr(z), self._ptr(u), self._ptr(g), self._ptr(ssq_new),
                                                          self._ptr(tq), self._ptr(ts), self._ptr(ta)))
        return tq, ts, ta

    def get_state(self):
        self._need_chains()
        C, d = self.n_chains, self.n_params
        q, ssq, std2, V = self._empty((C, d)), self._empty((C,)), self._empty((C,)), self._empty((C, d, d))
        _abi.check(self.lib, self.lib.rsf_mcmc_get_state(self._ctx, self._ptr(q), self._ptr(ssq), self._ptr(std2), self._ptr(V)))
        return q, ssq, std2, V

    def set_state(self, q=None, ssq=None, std2=None, V=None):
        q, ssq, std2, V = self._in(q), self._in(ssq), self._in(std2), self._in(V)
        _abi.check(self.lib, self.lib.rsf_mcmc_set_state(self._ctx, self._ptr(q), self._ptr(ssq), self._ptr(std2), self._ptr(V)))

    def _traces(self, n_iters, want):
        self._need_chains()
        C, d = self.n_chains, self.n_params
        if want is True:
            want = ("q", "std2", "accept")
        want = want or ()
        tq = self._empty((n_iters, C, d)) if "q" in want else None
        ts = self._empty((n_iters, C)) if "std2" in want else None
        ta = self._empty((n_iters, C), np.uint8) if "accept" in want else None
        return tq, ts, ta

    def mcmc_run(self, n_iters, traces=True, out=None):
        """n_iters fused iterations for every chain → (trace_q[n,C,d], trace_std2[n,C], accept[n,C])."""
        tq, ts, ta = out if out is not None else self._traces(n_iters, traces)
        _abi.check(self.lib, self.lib.rsf_mcmc_run(self._ctx, int(n_iters), self._ptr(tq), self._ptr(ts), self._ptr(ta)))
        return tq, ts, ta

    def mcmc_replay(self, z, u, g, traces=True):
        z, u, g = self._in(z), self._in(u), self._in(g)
        n_iters = int(u.shape[0])
        tq, ts, ta = self._traces(n_iters, traces)
        _abi.check(self.lib, self.lib.rsf_mcmc_replay(self._ctx, n_iters, self._ptr(z), self._ptr(u), self._ptr(g),
                                                      self._ptr(tq), self._ptr(ts), self._ptr(ta)))
        return tq, ts, ta

    def stats(self):
        v = [ctypes.c_int64() for _ in range(4)]
        _abi.check(self.lib, self.lib.rsf_mcmc_stats(self._ctx, *[ctypes.byref(x) for x in v]))
        return dict(zip(("accepted", "evaluated", "nonfinite", "iters_done"), (x.value for x in v)))

    def counters(self):
        """rsf_mcmc_counters → dict: the chain totals plus, on the HIP library, how the float64 RK4 kernels spent their
        wave-steps (tier by tier, redone trips, lane utilisation); `lane_utilisation` is derived."""
        v = (ctypes.c_int64 * len(_abi.COUNTERS))()
        _abi.check(self.lib, self.lib.rsf_mcmc_counters(self._ctx, v, len(_abi.COUNTERS)))
        c = dict(zip(_abi.COUNTERS, (int(x) for x in v)))
        steps = c["steps_tight"] + c["steps_narrow"] + c["steps_wide"] + c["steps_full"]
        c["lane_utilisation"] = c["lane_steps"] / (64.0 * steps) if steps else None
        return c

    # -- posterior post-processing (RSF.plot_dist, RSF.py:717-746) -------------------------
    def _column(self, samples, param):
        """samples: (n,) or trace block (..., d) in this engine's memory space → (array, n, stride)."""
        x = self._in(samples)
        d = int(x.shape[-1]) if x.ndim > 1 else 1
        n = int(np.prod(x.shape)) // d
        return x, n, d, int(param)

    def pool_summary(self, samples, param=0):
        """→ dict(n, mean, var (ddof=1), min, max) of parameter `param` over all pooled draws."""
        x, n, d, p = self._column(samples, param)
        out = (ctypes.c_double * 5)()
        _abi.check(self.lib, self.lib.rsf_pool_summary(self._ctx, n, self._ptr(x) + 8 * p, d, out))
        return dict(zip(("n", "mean", "var", "min", "max"), list(out)))

    def pool_kde(self, samples, grid, param=0, bw_factor=0.0):
        """scipy.stats.gaussian_kde(samples).pdf(grid) (Scott bandwidth unless bw_factor > 0) → density[m]."""
        x, n, d, p = self._column(samples, param)
        grid = self._in(grid)
        m = int(grid.shape[0])
        dens = self._empty((m,))
        _abi.check(self.lib, self.lib.rsf_pool_kde(self._ctx, n, self._ptr(x) + 8 * p, d, m, self._ptr(grid), float(bw_factor),
                                                   self._ptr(dens)))
        return dens

    def pool_histogram(self, samples, nbins, lo, hi, param=0):
        """numpy.histogram(samples, nbins, (lo, hi)) of parameter `param` over all pooled draws, plus the out-of-range counts:
        → counts[nbins + 2] (float64 holding exact integers): [below lo, bin 0 .. bin nbins-1, above hi or NaN].  The summary path
        of SURVEY §8e: a few KB per rank, summed across ranks with pool_allreduce_sum / dist.allreduce_histogram."""
        x, n, d, p = self._column(samples, param)
        counts = self._empty((int(nbins) + 2,))
        _abi.check(self.lib, self.lib.rsf_pool_histogram(self._ctx, n, self._ptr(x) + 8 * p, d, int(nbins), float(lo), float(hi),
                                                         self._ptr(counts)))
        return counts

    # -- multi-GPU posterior pool through the C ABI (RCCL bound inside the library) -----
    def comm_unique_id(self):
        """Rank 0: the 128-byte id every rank passes to comm_init (send it over any channel)."""
        buf = (ctypes.c_uint8 * 128)()
        _abi.check(self.lib, self.lib.rsf_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, world, rank, unique_id=None):
        """Collective: create this ctx's communicator (world = 1 needs no id)."""
        buf = (ctypes.c_uint8 * 128)(*unique_id) if unique_id is not None else None
        _abi.check(self.lib, self.lib.rsf_comm_init(self._ctx, int(world), int(rank), buf))
        self.world, self.rank = int(world), int(rank)

    def comm_destroy(self):
        _abi.check(self.lib, self.lib.rsf_comm_destroy(self._ctx))
        self.world = 0

    def pool_allgather(self, local):
        """local: array/tensor of any shape in this engine's memory space → (world,) + shape on every rank."""
        if not self.world:
            raise _abi.RsfError(-3, "pool_allgather: call comm_init first")
        x = self._in(local)
        out = self._empty((self.world,) + tuple(x.shape))
        n = int(np.prod(x.shape))
        _abi.check(self.lib, self.lib.rsf_pool_allgather(self._ctx, self._ptr(x), n, self._ptr(out)))
        return out

    def pool_allreduce_sum(self, buf):
        """In-place element-wise sum over ranks of a float64 array/tensor in this engine's memory space."""
        if not self.world:
            raise _abi.RsfError(-3, "pool_allreduce_sum: call comm_init first")
        x = self._in(buf)
        _abi.check(self.lib, self.lib.rsf_pool_allreduce_sum(self._ctx, self._ptr(x), int(np.prod(x.shape))))
        return x

    # -- the same exchange driven by ONE host thread that owns several engines (one per GPU): SURVEY §8e's process model --
    @staticmethod
    def _ctx_array(engines):
        if not engines:
            raise ValueError("need at least one engine")
        lib = engines[0].lib
        if any(e.lib is not lib for e in engines):
            raise ValueError("all engines of a group must come from the same library")
        return lib, (ctypes.c_void_p * len(engines))(*[e._ctx for e in engines])

    @staticmethod
    def comm_init_all(engines):
        """rsf_comm_init_all: engines[i] becomes rank i of a len(engines)-rank group (ncclCommInitAll; no id, no launcher)."""
        lib, ctxs = Engine._ctx_array(engines)
        _abi.check(lib, lib.rsf_comm_init_all(ctxs, len(engines)))
        for r, e in enumerate(engines):
            e.world, e.rank = len(engines), r

    @staticmethod
    def pool_allgather_all(engines, locals_):
        """One grouped all-gather: locals_[r] (same shape on every rank, in engines[r]'s memory space) → list of
        (world,) + shape arrays, one per engine, each holding every rank's block."""
        lib, ctxs = Engine._ctx_array(engines)
        n = len(engines)
        xs = [e._in(x) for e, x in zip(engines, locals_)]
        if len(xs) != n or any(tuple(x.shape) != tuple(xs[0].shape) for x in xs):
            raise ValueError("one block of the same shape per engine")
        outs = [e._empty((n,) + tuple(xs[0].shape)) for e in engines]
        send = (ctypes.c_void_p * n)(*[Engine._ptr(x) for x in xs])
        recv = (ctypes.c_void_p * n)(*[Engine._ptr(o) for o in outs])
        _abi.check(lib, lib.rsf_pool_allgather_all(ctxs, n, send, int(np.prod(xs[0].shape)), recv))
        return outs

    @staticmethod
    def pool_allreduce_sum_all(engines, bufs):
        """One grouped in-place sum over ranks: bufs[r] lives in engines[r]'s memory space → the summed buffers."""
        lib, ctxs = Engine._ctx_array(engines)
        n = len(engines)
        xs = [e._in(x) for e, x in zip(engines, bufs)]
        if len(xs) != n or any(tuple(x.shape) != tuple(xs[0].shape) for x in xs):
            raise ValueError("one buffer of the same shape per engine")
        ptrs = (ctypes.c_void_p * n)(*[Engine._ptr(x) for x in xs])
        _abi.check(lib, lib.rsf_pool_allreduce_sum_all(ctxs, n, ptrs, int(np.prod(xs[0].shape))))
        return xs

    def mcmc_adapt(self, window, adapt_mode, prior_len=0):
        """MCMC.update_covariance_matrix for one window of samples (n, d) on the device (rsf_mcmc_adapt) → the (d, d) matrix
        the reference's loop would assign to Vold; raises RsfError(-1) where np.linalg.cholesky would raise."""
        w = np.ascontiguousarray(window, dtype=np.float64)
        w = w.reshape(w.shape[0], -1)
        n, d = w.shape
        out = np.empty((d, d))
        mode = _abi.ADAPT_MODES[adapt_mode] if isinstance(adapt_mode, str) else int(adapt_mode)
        _abi.check(self.lib, self.lib.rsf_mcmc_adapt(d, n, w.ctypes.data, mode, int(prior_len), out.ctypes.data))
        return out

    # -- RNG helpers (tests) --------------------------------------------------------------
    def philox(self, ctr, key):
        c, k, o = (ctypes.c_uint32 * 4)(*ctr), (ctypes.c_uint32 * 2)(*key), (ctypes.c_uint32 * 4)()
        _abi.check(self.lib, self.lib.rsf_philox4x32_10(c, k, o))
        return list(o)

    def draws(self, seed, chain, iteration, n_params, shape):
        z, u, g = (ctypes.c_double * 3)(), ctypes.c_double(), ctypes.c_double()
        _abi.check(self.lib, self.lib.rsf_mcmc_draws(seed, chain, iteration, n_params, shape, z, ctypes.byref(u), ctypes.byref(g)))
        return list(z)[:n_params], u.value, g.value
