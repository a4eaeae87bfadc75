"""
MI355X-native MCMC-over-ODE hot path (rate-and-state friction), drop-in for the reference's
RateStateModel / MCMC / RSF classes and main.py entry.  See DESIGN.md.

The directory name is fixed by the build contract and is not a Python identifier; import it as
`bayesian_markov_chain_monte_carlo_amd` (alias module at the repo root), or put this directory
on sys.path and use the reference's flat module names (`from MCMC import MCMC`).
"""
from . import _abi  # noqa: F401
from ._abi import RsfError  # noqa: F401
from .engine import Engine  # noqa: F401
from .RateStateModel import RateStateModel  # noqa: F401
from .MCMC import MCMC  # noqa: F401
from .RSF import RSF, measure_execution_time  # noqa: F401

__all__ = ["Engine", "RsfError", "RateStateModel", "MCMC", "RSF", "measure_execution_time"]
