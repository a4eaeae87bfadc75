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

for _optional in ("import matplotlib.pyplot as plt", "from matplotlib.animation import FuncAnimation", "import mysql.connector"):
    try:
        exec(_optional)
    except Exception:  # absent or unusable on this host: only plotting / MySQL persistence need them
        pass
del _optional
