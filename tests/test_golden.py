"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle): the oracle still reproduces
them, the fast CPU formulation and the SIMT-emulated HIP kernel reproduce them (CPU), and the engine reproduces them on the
device (-m gpu), including the decided cut and the next configuration id."""
import glob
import os

import numpy as np
import pytest

from oracle import pyoracle as O
from rapid_amd import scenarios as S
from tests.helpers import oracle_view

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
assert FIXTURES, "no golden fixtures"


def load(path):
    g = np.load(path)
    return {k: g[k] for k in g.files}


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_and_emulated_kernel_reproduce_golden(path):
    from tests.emu import pyemu
    g = load(path)
    n, K, H, L = int(g["n"]), int(g["K"]), int(g["H"]), int(g["L"])
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    assert view.getCurrentConfigurationId() == int(g["config_id"])
    obs, subj, member = view.tables(n)
    assert np.array_equal(obs, g["obs"]) and np.array_equal(subj, g["subj"])
    for k in range(K):
        assert np.array_equal(view.getRing(k), g["rings"][k])
    emit, nprop, poff, props = O.sim_run(view, K, H, L, pop.id_hi, pop.id_lo, g["records"], g["rec_off"])
    assert np.array_equal(emit, g["emit_batch"]) and np.array_equal(nprop, g["num_proposals"])
    assert np.array_equal(poff, g["prop_off"]) and np.array_equal(props, g["props"])
    fe, fn, fo, fp = O.fast_sim_run(n, K, H, L, int(g["config_id"]), obs, subj, member, g["records"], g["rec_off"])
    assert np.array_equal(fe, emit) and np.array_equal(fp, props)
    rx = np.arange(0, len(emit), max(1, len(emit) // 24))  # a sample keeps the emulation quick
    off = np.zeros(len(rx) + 1, dtype=np.int64)
    parts = []
    for i, r in enumerate(rx):
        parts.append(g["records"][g["rec_off"][r]:g["rec_off"][r + 1]])
        off[i + 1] = off[i] + len(parts[-1])
    e2, n2, c2, f2, p2, st = pyemu.tally(np.concatenate(parts), off, n, K, H, L, int(g["config_id"]), obs, subj, member,
                                        trusted=True)
    assert np.array_equal(e2, emit[rx]) and np.array_equal(n2, nprop[rx])
    for i, r in enumerate(rx):
        assert p2[i, :c2[i]].tolist() == props[poff[r]:poff[r + 1]].tolist()
    svc = O.AlertBatchService(view, K, H, L, pop.id_hi, pop.id_lo)
    svc.decideViewChange(g["cut"].tolist())
    assert view.getCurrentConfigurationId() == int(g["next_config_id"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_engine_reproduces_golden(path):
    from rapid_amd import engine as E
    g = load(path)
    n, K, H, L = int(g["n"]), int(g["K"]), int(g["H"]), int(g["L"])
    pop = S.Population.make(n)
    eng = E.Engine(n_max=n, K=K, H=H, L=L)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    assert view.getCurrentConfigurationId() == int(g["config_id"])
    obs, subj, member = view.tables()
    assert np.array_equal(obs, g["obs"]) and np.array_equal(subj, g["subj"])
    for k in range(K):
        assert np.array_equal(view.getRing(k), g["rings"][k])
    sim = E.ClusterSimulation(eng)
    sim.load_streams(g["records"], g["rec_off"])
    sim.set_alert_set(g["alert_set"], trust_copies=True)
    sim.tally()
    emit, nprop, pcount, fp = sim.results()
    assert np.array_equal(emit, g["emit_batch"]) and np.array_equal(nprop, g["num_proposals"])
    assert np.array_equal(pcount, np.diff(g["prop_off"]))
    for r in range(0, len(emit), 3):
        assert sorted(sim.proposal(r)) == g["props"][g["prop_off"][r]:g["prop_off"][r + 1]].tolist()
    rr, new_cfg = sim.round(apply=True)
    assert rr.decided == 1 and rr.votes_winner == int(g["votes_winner"])
    assert sim.decided_cut() == g["cut"].tolist()
    assert new_cfg == int(g["next_config_id"])
