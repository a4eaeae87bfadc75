"""
RSF — drop-in for the reference driver class (RSF.py:59-1046): sweeps `dc_list`, makes the
synthetic observation, round-trips it through the chosen persistence format and runs one
MCMC per Dc.  The compute (forward solves, sampling) goes to the GPU; plotting stays optional
host work and never blocks the result.

Deviations from the reference, all needed for `main.py` to run at all:
the JSON helpers are imported under their own names (the reference shadows them with the MySQL
ones), the MySQL helpers are imported lazily, and a missing ffmpeg/display only skips figures.
"""
import time

import numpy as np

if __package__:
    from . import _figures, json_save_load
    from .engine import Engine
    from .MCMC import MCMC, PosteriorPool
else:  # flat layout: this directory on sys.path, the reference's own import style (main.py:44-46)
    import _figures
    import json_save_load
    from engine import Engine
    from MCMC import MCMC, PosteriorPool


def measure_execution_time(func):
    """Wall-clock decorator; like the reference (RSF.py:49-54) it RETURNS the elapsed seconds."""

    def wrapper(*args, **kwargs):
        start_time = time.time()
        func(*args, **kwargs)
        return time.time() - start_time

    return wrapper


class RSF:
    def __init__(self, number_slip_values=1, lowest_slip_value=1.0, largest_slip_value=1000.0, qstart=10.0,
                 qpriors=["Uniform", 0.0, 10000.0], reduction=False, plotfigs=False):
        self.num_dc = number_slip_values
        self.dc_list = np.linspace(lowest_slip_value, largest_slip_value, self.num_dc)
        self.num_features = 2
        self.plotfigs = plotfigs
        self.qstart = qstart
        self.qpriors = qpriors
        self.reduction = reduction
        self.make_animations = True   # the reference hard-codes sample(True), RSF.py:897
        self.verbose = True
        self.posteriors = {}          # additive: dc -> kept samples of the last inference

    def generate_time_series(self):
        """Noisy acceleration for every Dc of dc_list, concatenated (RSF.py:355-371).  The
        forward solves run as ONE batched launch; the noise is drawn per Dc in the reference's
        order from the global NumPy RNG."""
        n = self.model.num_tsteps
        acc = self.model.evaluate_batch(self.dc_list)  # (num_dc, nout)
        if acc.shape[1] != n:
            raise ValueError(f"model produces {acc.shape[1]} samples per series, num_tsteps is {n}")
        acc_appended_noise = np.zeros(len(self.dc_list) * n)
        t = self.model.time_axis()
        for index, dc_value in enumerate(self.dc_list):
            self.model.Dc = dc_value
            acc_noise = acc[index] + 1.0 * np.abs(acc[index]) * np.random.randn(n)  # RateStateModel.py:392
            self.plot_time_series(t, acc[index])
            acc_appended_noise[index * n:(index + 1) * n] = acc_noise
        return acc_appended_noise

    def plot_time_series(self, time, acceleration):
        if self.plotfigs:
            _figures.series_figure(time, acceleration, self.model.Dc, self.model.t_start, self.model.t_final)

    def prepare_data(self, data):
        if self.format == "json":
            self.lstm_file = "model_lstm.json"
            self.data_file = "data.json"
            json_save_load.save_object(data, self.data_file)
            data = json_save_load.load_object(self.data_file)
        elif self.format == "mysql":
            raise RuntimeError("the MySQL format needs a server and mysql.connector (RSF.py:597-602); "
                               "neither is part of this build — use 'json'")
        return data

    def plot_dist(self, qparams, dc):
        """Kept samples beside their kernel density (RSF.py:717-746); the density is the device KDE of the pooled draws."""
        def kde():  # resolved inside the figure's guard: an engine failure only skips the figure
            return self.model.engine().pool_kde

        kde._deferred = True
        return _figures.trace_with_density(qparams[0, :], f"$d_c={dc:.2f}\\,\\mu m$ with {self.format} formatting", kde)

    def perform_sampling_and_plotting(self, data, dc, nsamples, model_lstm):
        hits = np.flatnonzero(np.asarray(self.dc_list) == dc)  # the series of true value dc_list[i] is data[i*N:(i+1)*N], RSF.py:874-882
        if hits.size == 0:
            print(f"Error: dc value {dc} not found in dc_list.")
            return
        n = self.model.num_tsteps
        noisy_data = data[int(hits[0]) * n:(int(hits[0]) + 1) * n]
        print(f"--- Dc is {dc} ---")
        mc = MCMC(self.model, noisy_data, dc, self.qpriors, self.qstart, lstm_model=model_lstm, nsamples=nsamples,
                  verbose=self.verbose)
        qparams = mc.sample(self.make_animations)
        self.posteriors[float(dc)] = qparams
        self.plot_dist(qparams, dc)

    def inference_batched(self, nsamples, chains_per_dc=256, seed=0, mem="device", device=-1, adapt_mode=None):
        """Additive throughput path: the whole dc_list sweep as ONE launch per block of iterations — every
        true Dc is an observation group with `chains_per_dc` independent chains (Philox variates on device).
        Returns {dc: PosteriorPool} of the post-burn-in draws (nburn = int(nsamples/2) like MCMC)."""
        n, G = self.model.num_tsteps, len(self.dc_list)
        data = np.ascontiguousarray(np.asarray(self.data, dtype=np.float64).reshape(G, n))
        probe = MCMC(self.model, data[0], self.dc_list[0], self.qpriors, self.qstart, nsamples=nsamples)
        nburn = probe.nburn
        C = G * int(chains_per_dc)
        eng = Engine(mem=mem, device=device)
        try:
            eng.set_model(self.model, getattr(self.model, "substeps", 1))
            eng.mcmc_init(np.full((C, 1), float(self.qstart)), data, probe.qstart_limits[:, 0], probe.qstart_limits[:, 1],
                          seed=seed, n0=probe.n0, prior_len=len(self.qpriors), adapt_mode=adapt_mode or probe._adapt_mode(),
                          adapt_interval=probe.adapt_interval)
            tq, ts, ta = eng.mcmc_run(nsamples)
            eng.sync()
            tq, ts, ta = (np.asarray(x.cpu() if hasattr(x, "cpu") else x) for x in (tq, ts, ta))
        finally:
            eng.close()
        out, first = {}, max(nburn - 1, 0)
        for g, dc in enumerate(self.dc_list):
            sl = slice(g * chains_per_dc, (g + 1) * chains_per_dc)
            acc = int(ta[:, sl].sum())
            out[float(dc)] = PosteriorPool(tq[first:, sl], ts[first:, sl], acc / (nsamples * chains_per_dc),
                                           dict(accepted=acc), nburn)
        self.posteriors = {dc: pool.pooled() for dc, pool in out.items()}
        return out

    @measure_execution_time
    def inference(self, nsamples):
        data = self.prepare_data(self.data)
        for dc in self.dc_list:
            self.perform_sampling_and_plotting(data, dc, nsamples, None)
        return
