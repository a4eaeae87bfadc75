"""
Star-import prelude, kept because the reference's modules and any caller written against them begin with
`from imports import *` (imports.py:1-12, main.py:44).  Same names as the reference's prelude; the one difference
is that `mysql.connector` — which the hot path never calls — is optional here instead of a hard import-time
dependency, and so is matplotlib (figures are skipped without it).
"""
import json  # noqa: F401
import time  # noqa: F401

import numpy as np  # noqa: F401
import scipy  # noqa: F401
from numpy import exp, log, sin  # noqa: F401
from scipy import integrate  # noqa: F401
from scipy.stats import gamma, gaussian_kde  # noqa: F401

try:  # figures only (RSF.plot_*, MCMC animation); absent or unusable => those are skipped with a warning
    import matplotlib.pyplot as plt  # noqa: F401
    from matplotlib.animation import FuncAnimation  # noqa: F401
except Exception:
    pass
try:  # MySQL persistence only (out of scope here, RSF.prepare_data raises for it); the reference hard-imports this
    import mysql.connector  # noqa: F401
except Exception:
    pass
