"""
CPU unit tests of tests/chain_parity.py — the assertion every GPU chain-parity test relies on.  Both "sides" are the CPU
oracle here; forks are fabricated: one that is NOT a near-tie (must be refused) and one that IS (u placed on the accept
threshold by bisection, the two sides a hair either way of it: must be accepted).
"""
import numpy as np
import pytest

from chain_parity import Rerun, assert_chains_match
from conftest import synthetic_data


def _setup(cpu_engine, oracle_mod, C=6, n=12, adapt="none"):
    cpu_engine.set_model(oracle_mod.ModelSpec(300, 0.0, 30.0, 1), 1)
    data = synthetic_data(cpu_engine)
    q0 = np.full((C, 1), 1000.0)
    kw = dict(seed=77, chain_offset=1000, prior_len=3 if adapt == "none" else 2, adapt_mode=adapt, adapt_interval=5)
    cpu_engine.mcmc_init(q0, data, [0.0], [1e4], **kw)
    state0 = cpu_engine.get_state()
    rerun = Rerun(type(cpu_engine), cpu_engine, q0, data, [0.0], [1e4], state0, kw)
    return cpu_engine.mcmc_run(n), rerun


@pytest.mark.parametrize("adapt", ["none", "reference_dict"])
def test_identical_runs_pass_and_unexplained_fork_is_refused(cpu_engine, oracle_mod, adapt):
    tc, rerun = _setup(cpu_engine, oracle_mod, adapt=adapt)
    assert assert_chains_match(tc, tc, rerun).all()
    # the Philox variates the helper regenerates really are the run's: a one-chain replay walks the same chain
    z, u, g = rerun.chain_variates(2, 12)
    back = rerun.replay(2, z, u, g)
    np.testing.assert_array_equal(back[2][:, 0], tc[2][:, 2])
    np.testing.assert_allclose(back[0][:, 0], tc[0][:, 2], rtol=1e-14)
    np.testing.assert_allclose(back[1][:, 0], tc[1][:, 2], rtol=1e-14)
    # a fabricated fork: flip one decision of chain 3 on the "GPU" side — this is no near-tie and must fail loudly
    k = 4
    fake = [x.copy() for x in tc]
    fake[2][k, 3] ^= 1
    with pytest.raises(AssertionError, match="NOT a near-tie"):
        assert_chains_match(tuple(fake), tc, rerun)
    # ... and so must a chain that already differs BEFORE its first differing decision
    fake[0][k - 2, 3] *= 1.0 + 1e-6
    with pytest.raises(AssertionError, match="differs before its fork"):
        assert_chains_match(tuple(fake), tc, rerun)
    # equal decisions but different samples: plain Tier-1 failure
    bad = [x.copy() for x in tc]
    bad[0][5, 1] *= 1.0 + 1e-7
    with pytest.raises(AssertionError):
        assert_chains_match(tuple(bad), tc, rerun)


def test_a_genuine_near_tie_fork_is_accepted(cpu_engine, oracle_mod):
    tc, rerun = _setup(cpu_engine, oracle_mod)
    n = 12

    def threshold(c, k, z, u, g):
        lo, hi = -60.0, 0.0                   # bisection on log u for the accept threshold log alpha of iteration k
        for _ in range(80):
            mid = 0.5 * (lo + hi)
            u2 = u.copy()
            u2[k, 0] = np.exp(mid)
            lo, hi = (mid, hi) if rerun.replay(c, z, u2, g)[2][k, 0] else (lo, mid)
        return lo, hi

    for c, k in ((c, k) for c in range(6) for k in range(2, n)):
        z, u, g = rerun.chain_variates(c, n)
        lo, hi = threshold(c, k, z, u, g)
        if -50.0 < lo < -1e-3:                # a downhill in-bounds proposal: log alpha strictly inside (-inf, 0)
            break
    else:
        pytest.fail("no downhill proposal found in 6 chains x 10 iterations")
    ua, ub = u.copy(), u.copy()
    ua[k, 0], ub[k, 0] = np.exp(lo - 2e-12 * abs(lo)), np.exp(hi + 2e-12 * abs(hi))  # accept / reject, 4e-12 apart
    # the run as a replay test sees it: "GPU" walked the chain with ua, the oracle with ub — same variates to 1e-12
    def all_chains(uk):
        rows = [rerun.replay(cc, *((z, uk, g) if cc == c else rerun.chain_variates(cc, n))) for cc in range(6)]
        return tuple(np.concatenate([r[j] for r in rows], axis=1) for j in range(3))

    tg, to = all_chains(ua), all_chains(ub)
    assert tg[2][k, c] == 1 and to[2][k, c] == 0
    zs, us, gs = (np.concatenate([rerun.chain_variates(cc, n)[j] if cc != c else (z, ub, g)[j] for cc in range(6)], axis=1) for j in range(3))
    rr = Rerun(rerun.Engine, rerun.cpu, rerun.q0, rerun.data, rerun.lo, rerun.hi, rerun.state0, rerun.kw, variates=(zs, us, gs))
    same = assert_chains_match(tg, to, rr)
    assert not same[c] and same.sum() == 5
