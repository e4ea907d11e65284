"""The first GPU tests of a run (the file name sorts first): the call every user of the library makes first --
`new MembershipView(K, nodeIds, endpoints)` (R/MembershipView.java:74-89) -- at the sizes GPUTEST_r05 died at and below, built
again and again on fresh engines and on the same engine, against the CPU oracle.  If the device or the runtime on this box
cannot do THIS, the failure is reported here, by name, before the parity suite starts (and tests/conftest.py keeps the
rest of the run alive)."""
import numpy as np
import pytest

from rapid_amd import scenarios as S
from tests.helpers import oracle_view

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from rapid_amd import engine
    if engine.device_count() < 1:
        pytest.fail("no gfx950 device visible: the product has no CPU fallback")
    return engine


def test_canary_engine_self_test(E):
    """rapid_engine_self_test: a 5-node view and one decided round with known answers, checked inside the library."""
    eng = E.Engine(n_max=64, K=3, H=3, L=1)
    eng.self_test()
    eng.self_test()


@pytest.mark.parametrize("n,K", [(1, 3), (2, 3), (3, 10), (50, 3), (400, 10)])
def test_canary_view_builds(E, n, K):
    pop = S.Population.make(n)
    reg, oview = oracle_view(pop, K)
    want_cfg = oview.getCurrentConfigurationId()
    want_rings = [oview.getRing(k) for k in range(K)]
    oobs, osubj, omember = oview.tables(n)

    def check(view):
        assert view.getCurrentConfigurationId() == want_cfg
        for k in range(K):
            assert np.array_equal(view.getRing(k), want_rings[k])
        obs, subj, member = view.tables()
        assert np.array_equal(obs, oobs) and np.array_equal(subj, osubj) and np.array_equal(member, omember)

    for rep in range(10):  # a fresh engine every time (create, build, destroy)
        eng = E.Engine(n_max=n, K=K, H=K, L=1)
        check(E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo))
        del eng
    eng = E.Engine(n_max=n + 7, K=K, H=K, L=1)  # the same engine, built over and over
    view = E.MembershipView(eng)
    for rep in range(10):
        check(view.build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo))
