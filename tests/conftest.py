import ctypes
import json
import os
import sys

import numpy as np
import pytest

os.environ["RSF_ALLOW_CHECKER_ENGINE"] = "1"  # the test-suite is the checker: it may bind Engine to the CPU oracle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import bayesian_markov_chain_monte_carlo_amd as pkg

        return pkg._abi.load().rsf_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def pkg():
    import bayesian_markov_chain_monte_carlo_amd as p

    return p


@pytest.fixture(scope="session")
def oracle_mod():
    import rsf_oracle

    rsf_oracle.build()
    return rsf_oracle


@pytest.fixture(scope="session")
def oracle_lib(pkg, oracle_mod):
    return pkg._abi.bind(ctypes.CDLL(oracle_mod.lib_path()))


@pytest.fixture()
def cpu_engine(pkg, oracle_lib):
    """Engine on the CPU oracle (the checker)."""
    e = pkg.Engine(lib=oracle_lib)
    yield e
    e.close()


@pytest.fixture()
def gpu_engine(pkg):
    """Engine on the product library; host buffers through the C ABI."""
    e = pkg.Engine(mem="host")
    assert e.lib.rsf_backend() == b"hip-gfx950"
    yield e
    e.close()


class Golden:
    def npz(self, name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    def json(self, name):
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return Golden()


def synthetic_data(engine, nout_expected=None, dc_true=1000.0, seed=2025):
    """Observation in the bench's recipe: own forward solve + |acc| N(0,1)."""
    _, acc = engine.forward([dc_true])
    acc = np.asarray(acc)[:, 0]
    return acc + np.abs(acc) * np.random.default_rng(seed).standard_normal(acc.shape[0])
