"""The kernel's own fp64 log / exp / reciprocal (csrc/rsf_math.h) against NumPy, on the GPU.
Tolerance: a few ulp (the RHS needs ~1e-15; Tier-1 parity is checked end-to-end elsewhere)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    out = tmp_path_factory.mktemp("probe") / "probe_math"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-o", str(out),
                           os.path.join(HERE, "hip", "probe_math.hip")])

    def run(kind, x):
        d = out.parent
        np.asarray(x, dtype=np.float64).tofile(d / "in.f64")
        subprocess.check_call([str(out), kind, str(d / "in.f64"), str(d / "out.f64")])
        return np.fromfile(d / "out.f64", dtype=np.float64)

    return run


def _ulp_err(got, want):
    return np.abs(got - want) / np.spacing(np.abs(want))


def test_log(probe):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(0.05, 20.0, 200000), np.exp(rng.uniform(-700, 700, 50000)),
                        1.0 + rng.uniform(-1e-3, 1e-3, 50000), [1.0, 0.5, 2.0, np.sqrt(0.5), np.sqrt(2.0), 5e-324, 1e-310]])
    got, want = probe("log", x), np.log(x)
    ok = want != 0
    assert _ulp_err(got[ok], want[ok]).max() <= 4.0
    assert (got[~ok] == 0).all()
    bad = probe("log", [0.0, -1.0, np.nan, -np.inf, np.inf])
    assert np.isnan(bad[:4]).all() and not np.isfinite(bad[4])


def test_exp(probe):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-20.0, 20.0, 200000), rng.uniform(-700.0, 700.0, 50000), rng.uniform(-1e-3, 1e-3, 50000), [0.0]])
    got, want = probe("exp", x), np.exp(x)
    assert _ulp_err(got, want).max() <= 4.0
    edge = probe("exp", [800.0, -800.0, np.nan, np.inf])
    assert edge[0] == np.inf and edge[1] == 0.0 and np.isnan(edge[2]) and not np.isfinite(edge[3])


def test_rcp(probe):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(1e-3, 1e5, 200000), np.exp(rng.uniform(-600, 600, 50000))])
    got = probe("rcp", x)
    assert _ulp_err(got, 1.0 / x).max() <= 2.0
    seed = probe("rcp_seed", x)  # documents what v_rcp_f64 alone delivers (why Newton steps are needed)
    rel = np.abs(seed * x - 1.0).max()
    print(f"v_rcp_f64 seed max relative error: {rel:.3e}")
    assert rel < 1e-6


def test_sincos2pi(probe):
    """Box-Muller angle: sin/cos(2 pi u) for u = k * 2^-53 in (0, 1], absolute error at rounding level (the
    reference value is formed in extended precision from the exactly reduced argument)."""
    rng = np.random.default_rng(3)
    u = (rng.integers(1, 2 ** 53, 300000, dtype=np.int64).astype(np.float64)) * 2.0 ** -53
    u = np.concatenate([u, [2.0 ** -53, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 1.0, 1.0 - 2.0 ** -53, 0.25 + 2.0 ** -53]])
    k = np.rint(4.0 * u)
    r = (u - 0.25 * k).astype(np.longdouble)  # exact
    th = r * (2 * np.longdouble(np.pi) + np.longdouble(1.2246467991473532e-16) * 2)
    s0, c0 = np.sin(th), np.cos(th)
    q = k.astype(np.int64) & 3
    want_s = np.where(q == 0, s0, np.where(q == 1, c0, np.where(q == 2, -s0, -c0))).astype(np.float64)
    want_c = np.where(q == 0, c0, np.where(q == 1, -s0, np.where(q == 2, -c0, s0))).astype(np.float64)
    got_s, got_c = probe("sin2pi", u), probe("cos2pi", u)
    assert np.abs(got_s - want_s).max() < 3e-16 and np.abs(got_c - want_c).max() < 3e-16
    # relative accuracy also where the value is tiny (near the axes the reduced angle is exact)
    big = np.abs(want_s) > 1e-300
    assert (np.abs(got_s - want_s)[big] / np.abs(want_s)[big]).max() < 1e-15
    assert np.abs(got_s ** 2 + got_c ** 2 - 1.0).max() < 1e-15
