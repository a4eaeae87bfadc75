"""
Optional figures of the drop-in classes (off the hot path; SURVEY §2 marks the reference's plotting out of scope).
What the reference shows is kept — the chain's evolution as an animation (MCMC.py:471-492, 538) and, per true Dc, the
kept samples beside their kernel density (RSF.py:717-746) — but the code is this package's own: the figures are built
from plain arrays by two free functions, the density comes from the device KDE (`Engine.pool_kde`, the product's
posterior post-processing) rather than from SciPy on the host, and a missing matplotlib / ffmpeg / display only skips
the figure with a warning; results never depend on it.
"""
import warnings

import numpy as np


def _pyplot():
    import matplotlib.pyplot as plt

    return plt


def chain_movie(chain, title, path, fps=30):
    """Grow the trace of one chain frame by frame and write it as a movie; → True if written."""
    chain = np.asarray(chain, dtype=np.float64).ravel()
    fig = None
    try:
        from matplotlib.animation import FuncAnimation

        plt = _pyplot()
        fig, ax = plt.subplots()
        # frames and x-range as the reference draws them: nsamples frames over (0, nsamples), MCMC.py:471-492
        ax.set(title=title, xlabel="Sample Index", ylabel="Sample Value", xlim=(0, chain.size - 1),
               ylim=(chain.min() - 1.0, chain.max() + 1.0))
        (trace,) = ax.plot([], [], lw=2)
        steps = np.arange(chain.size)

        def draw(n):
            trace.set_data(steps[:n], chain[:n])
            return (trace,)

        FuncAnimation(fig, draw, frames=chain.size - 1, blit=True).save(path, fps=fps, writer="ffmpeg")
        return True
    except Exception as ex:  # no ffmpeg / no display / no matplotlib
        warnings.warn(f"MCMC animation skipped: {ex}")
        return False
    finally:
        if fig is not None:
            _pyplot().close(fig)


def trace_with_density(samples, title, density, points=1000):
    """Left: the kept samples in order.  Right: their density on a grid spanning the left panel's y-range, drawn sideways.
    `density(samples, grid) -> pdf` supplies the estimate (the device KDE).  → the figure, or None if skipped."""
    samples = np.asarray(samples, dtype=np.float64).ravel()
    fig = None
    try:
        plt = _pyplot()
        if callable(density) and getattr(density, "_deferred", False):
            density = density()  # the engine is resolved inside the guard: a failure there only skips the figure
        fig, (left, right) = plt.subplots(1, 2, gridspec_kw=dict(width_ratios=(0.7, 0.15), wspace=0.15))
        fig.suptitle(title, fontsize=10)
        left.plot(samples, color="b", linewidth=1.0)
        left.set(xlabel="Sample number", xlim=(0, samples.size))
        left.set_ylabel("$d_c$", fontsize=10)
        grid = np.linspace(*left.get_ylim(), points)
        pdf = np.asarray(density(samples, grid), dtype=np.float64)
        if not np.isfinite(pdf).all():
            raise ValueError("the density estimate is not finite (a chain that never moved has zero variance)")
        right.plot(pdf, grid, color="b", linewidth=1.0)
        right.fill_betweenx(grid, 0.0, pdf, alpha=0.3)
        right.set(xlabel="Prob. density", xlim=(0, None), xticks=[])
        right.yaxis.set_visible(False)
        return fig
    except Exception as ex:
        if fig is not None:
            _pyplot().close(fig)  # a skipped figure must not stay registered with pyplot (one per Dc and inference)
        warnings.warn(f"posterior figure skipped: {ex}")
        return None


def series_figure(t, acc, dc, t_start, t_final):
    """One clean acceleration series (RSF.plot_time_series)."""
    fig = None
    try:
        plt = _pyplot()
        fig, ax = plt.subplots()
        ax.plot(t, acc, linewidth=1.0, label="True")
        ax.set(title=f"$d_c$={dc} $\\mu m$ RSF solution", xlabel="Time (sec)", ylabel="Acceleration $(\\mu m/s^2)$",
               xlim=(t_start - 2.0, t_final))
        ax.grid(True)
        ax.legend()
        return fig
    except Exception as ex:
        if fig is not None:
            _pyplot().close(fig)
        warnings.warn(f"time-series figure skipped: {ex}")
        return None
