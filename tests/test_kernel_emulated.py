"""The HIP tally kernels (rapid_amd/csrc/tally_kernel.h), compiled UNMODIFIED with g++ against the SIMT
emulator in tests/emu/, checked against the CPU oracle.  Lanes are scheduled in random order between wave
collectives, so order-dependence of the LDS atomics would show up here.  This validates the kernel LOGIC on
CPU; the same comparisons run on the real device in tests/test_gpu_parity.py (-m gpu)."""
import numpy as np
import pytest

from oracle import pyoracle as O
from rapid_amd import scenarios as S
from tests.emu import pyemu
from tests.helpers import oracle_view, proposal_fingerprints, random_stream


def _check(records, rec_off, n_nodes, K, H, L, cfg, obs, subj, member, **kw):
    fe, fn, fo, fp = O.fast_sim_run(n_nodes, K, H, L, cfg, obs, subj, member, records, rec_off)
    out = {}
    valid, _ = pyemu.validate_alerts(records, n_nodes, K, cfg, member)
    for force in (0, 1, 2) if valid else (0, 1):  # 2: the kTrusted instantiation (only for streams that qualify)
        emit, nprop, pcount, fpr, props, stats = pyemu.tally(records, rec_off, n_nodes, K, H, L, cfg, obs, subj, member,
                                                            force_exact=force & 1, trusted=(force == 2), **kw)
        bad = np.flatnonzero(emit != fe)
        assert len(bad) == 0, (force, bad[:5], emit[bad[:5]], fe[bad[:5]])
        assert np.array_equal(nprop, fn), force
        assert np.array_equal(pcount, np.diff(fo)), force
        for r in range(len(fe)):
            assert props[r, : pcount[r]].tolist() == fp[fo[r]: fo[r + 1]].tolist(), (force, r)
        assert np.array_equal(fpr, proposal_fingerprints(fo, fp, fe >= 0)), force  # the fingerprint is a function of the proposal alone
        out[force] = (fpr, stats)
    assert np.array_equal(out[0][0], out[1][0])  # fingerprints identical on both paths
    return out[0][1]


@pytest.mark.parametrize("seed", range(8))
def test_random_valid_streams_trusted_variant(seed):
    """Streams in which every alert passes the filter (what the generators produce): exercises kTrusted."""
    rng = np.random.default_rng(7000 + seed)
    n_nodes = int(rng.integers(8, 48))
    K = int(rng.integers(3, 11))
    H = int(rng.integers(1, K + 1))
    L = int(rng.integers(1, H + 1))
    pop = S.Population.make(n_nodes)
    n_members = int(rng.integers(max(2, n_nodes // 2), n_nodes + 1))
    members = sorted(rng.permutation(n_nodes)[:n_members].tolist())
    reg, view = oracle_view(pop, K, members)
    obs, subj, member = view.tables(n_nodes)
    cfg = view.getCurrentConfigurationId()
    recs, off = [], [0]
    for r in range(16):
        hot = rng.permutation(n_nodes)[: int(rng.integers(1, min(n_nodes, 12) + 1))]
        n_rec = int(rng.integers(0, 500))
        recs.append(random_stream(rng, n_nodes, K, member, cfg, n_rec, hot, p_bad_cfg=0.0, p_bad_status=0.0,
                                  p_eob=float(rng.choice([0.05, 0.3, 1.0]))))
        off.append(off[-1] + n_rec)
    records = np.concatenate(recs)
    assert pyemu.validate_alerts(records, n_nodes, K, cfg, member)[0]
    before = pyemu.CURRENT_RUNS
    _check(records, np.array(off), n_nodes, K, H, L, cfg, obs, subj, member, seed=seed)
    # every record carries the view's configuration id: the trusted run was taken twice -- comparing the ids per delivery, and by the
    # instantiation that never loads them (kCurrent, what the product launches for such a round) -- with equal results
    assert pyemu.CURRENT_RUNS == before + 1


@pytest.mark.parametrize("seed", range(10))
def test_random_adversarial_streams(seed):
    rng = np.random.default_rng(1000 + seed)
    n_nodes = int(rng.integers(8, 48))
    K = int(rng.integers(3, 11))
    H = int(rng.integers(1, K + 1))
    L = int(rng.integers(1, H + 1))
    pop = S.Population.make(n_nodes)
    n_members = int(rng.integers(max(2, n_nodes // 2), n_nodes + 1))
    members = sorted(rng.permutation(n_nodes)[:n_members].tolist())
    reg, view = oracle_view(pop, K, members)
    obs, subj, member = view.tables(n_nodes)
    cfg = view.getCurrentConfigurationId()
    recs, off = [], [0]
    for r in range(24):
        hot = rng.permutation(n_nodes)[: int(rng.integers(1, min(n_nodes, 12) + 1))]
        n_rec = int(rng.integers(0, 400))
        recs.append(random_stream(rng, n_nodes, K, member, cfg, n_rec, hot, p_eob=float(rng.choice([0.02, 0.3, 1.0]))))
        off.append(off[-1] + n_rec)
    _check(np.concatenate(recs), np.array(off), n_nodes, K, H, L, cfg, obs, subj, member, seed=seed)
    _check(np.concatenate(recs), np.array(off), n_nodes, K, H, L, cfg, obs, subj, member, seed=seed, tables_in_lds=0 if seed % 2 else 3, packed=True)


@pytest.mark.parametrize("name,n,f,K,H,L", [("C1", 50, 1, 3, 3, 1), ("C2", 300, 12, 10, 9, 4),
                                            ("C3a", 400, 20, 10, 9, 4), ("C3b", 400, 20, 10, 9, 4),
                                            ("C2", 200, 20, 10, 8, 2)])
def test_scenarios(name, n, f, K, H, L):
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    rx = None if n <= 60 else np.arange(0, n, 7)  # a sample of receivers keeps the emulation fast
    sc = S.build_scenario(name, subj, cfg, n=n, f=f, H=H, L=L, receivers=rx)
    is_f = np.zeros(n, dtype=bool)
    is_f[sc.faulty] = True
    keep = ~is_f[sc.receivers]
    stats = _check(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member)
    if n > 60 and 3 * pyemu.window_records() <= np.diff(sc.rec_off).max():
        assert stats[1] > 0  # the cold / fast windows were exercised (streams of several windows)


@pytest.mark.parametrize("n,n_out,n_crash,n_join,K,H,L", [(300, 30, 10, 12, 10, 9, 4), (150, 25, 4, 20, 10, 8, 3)])
def test_churn_scenario(n, n_out, n_crash, n_join, K, H, L):
    """Joins (UP alerts about non-members, expected observers) and crashes in one configuration, SURVEY 8f rank 1."""
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K, list(range(0, n - n_out)))
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs, member, cfg, n_crash, n_join, H, L)
    sc = S.build_churn_scenario(obs, member, cfg, n_crash, n_join, H, L, receivers=sc.receivers[::11])
    _check(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member)


@pytest.mark.parametrize("mode", [0, 2, 3])
def test_other_dictionary_placements(mode):
    """The tally kernel with the node -> slot dictionary in memory (0), as compressed tables -- bitmap + rank -- in LDS (2),
    and with no dictionary at all: the records carry their subjects' entries (3, kDictResolved: what the engine runs), on
    scenario and adversarial streams (the other tests run the direct tables)."""
    n, K, H, L = 300, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K, list(range(0, n - 20)))
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs, member, cfg, 10, 8, H, L)
    sc = S.build_churn_scenario(obs, member, cfg, 10, 8, H, L, receivers=sc.receivers[::13])
    _check(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member, tables_in_lds=mode)
    rng = np.random.default_rng(99 + mode)
    recs, off = [], [0]
    for r in range(12):
        hot = rng.permutation(n)[:10]
        n_rec = int(rng.integers(0, 600))
        recs.append(random_stream(rng, n, K, member, cfg, n_rec, hot, p_eob=float(rng.choice([0.05, 0.5]))))
        off.append(off[-1] + n_rec)
    _check(np.concatenate(recs), np.array(off), n, K, H, L, cfg, obs, subj, member, tables_in_lds=mode)
    _check(np.concatenate(recs), np.array(off), n, K, H, L, cfg, obs, subj, member, tables_in_lds=mode, pool=True)
    if mode == 0:  # ... the dictionary as hashed buckets of one-byte remainders in LDS (kDictHashed: packed rounds whose every named subject
        # is hot), slots renumbered in hash order by index_hash_kernel; stale and uncovered reports among the deliveries
        _check(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member, tables_in_lds=4, packed=True)
        _check(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member, tables_in_lds=4, packed=True, pool=True, waves=2, grid=3)
        late = sc.records.copy()
        late["cfg_id"][::7] = cfg + 3  # late deliveries of another configuration: dropped per delivery, their subjects may be anybody
        late["dst"][::21] = np.arange(len(late["dst"][::21])) % n
        fe, fn, fo, fp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, late, sc.rec_off)
        for trusted in (False, True):
            emit, nprop, pcount, fpr, props, stats, covered = pyemu.tally(late, sc.rec_off, n, K, H, L, cfg, obs, subj, member, tables_in_lds=4, packed=True,
                                                                          declared=sc.batches.recs, trusted=trusted)
            assert covered and np.array_equal(emit, fe) and np.array_equal(nprop, fn) and np.array_equal(pcount, np.diff(fo))
            assert np.array_equal(fpr, proposal_fingerprints(fo, fp, fe >= 0))
        stray = sc.records.copy()  # a report of the CURRENT configuration about a node the alert set does not name: not covered
        stray["dst"][5] = int(np.setdiff1d(np.arange(n), sc.batches.recs["dst"])[0])
        stray["status"][5] = 1 if member[stray["dst"][5]] else 0
        *_, covered = pyemu.tally(stray, sc.rec_off, n, K, H, L, cfg, obs, subj, member, tables_in_lds=4, packed=True, declared=sc.batches.recs)
        assert not covered
    if mode in (0, 3):  # ... and with two slots per LDS word (PackedSlotDetector: the state of rounds with thousands of hot subjects)
        _check(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member, tables_in_lds=mode, packed=True)
        _check(np.concatenate(recs), np.array(off), n, K, H, L, cfg, obs, subj, member, tables_in_lds=mode, packed=True)
        _check(np.concatenate(recs), np.array(off), n, K, H, L, cfg, obs, subj, member, tables_in_lds=mode, packed=True, pool=True, waves=2, grid=3)


def test_receivers_claimed_from_the_common_pool():
    """Beyond the first deal the receivers come from the pool every workgroup may claim from (on the device: the last
    eighth of a large population, so that fast workgroups take work off slow ones): same results, and the last workgroup
    leaves the pool words zeroed (checked by the emulator's entry point)."""
    n, K, H, L = 300, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario("C3b", subj, cfg, n=n, f=15, H=H, L=L, receivers=np.arange(0, n, 9))
    assert len(sc.receivers) > 4 * 6  # several claims per wave at the emulator's 2 workgroups x 3 waves
    _check(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member, pool=True)
    _check(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member, pool=True, waves=2, grid=3, tables_in_lds=2)


def test_unaligned_starts_and_empty_receivers():
    n, K, H, L = 24, 5, 4, 2
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    rng = np.random.default_rng(3)
    lens = [0, 1, 3, 0, 205, 7, 64, 65, 63, 128, 0, 409, 2]
    parts = [random_stream(rng, n, K, member, cfg, m, [1, 2, 3, 4, 5]) for m in lens]
    off = np.cumsum([0] + lens)
    _check(np.concatenate(parts), off, n, K, H, L, cfg, obs, subj, member)


def test_cd_instance_kernel_kats():
    """CutDetectionTest.java:42-252 (KAT-1..6) through cd_instance_kernel."""
    K, H, L = 10, 8, 2
    n = 16
    obs = np.full((n, K), -1, dtype=np.int32)
    subj = np.full((n, K), -1, dtype=np.int32)
    member = np.zeros(n, dtype=np.uint8)

    def alert(src, dst, ring, status=S.UP):
        a = np.zeros(1, dtype=S.ALERT_DTYPE)
        a["src"], a["dst"], a["ring_mask"], a["status"], a["cfg_id"] = src, dst, 1 << ring, status, -1
        return a

    # KAT-1
    cd = pyemu.CdInstance(n, K, H, L, obs, subj, member)
    for i in range(H - 1):
        ret, _ = cd.aggregate(alert(i + 1, 12, i))
        assert ret == [] and cd.num_proposals() == 0
    ret, _ = cd.aggregate(alert(H, 12, H - 1))
    assert ret == [12] and cd.num_proposals() == 1
    # KAT-4 (three blockers, duplicate reports past H)
    cd = pyemu.CdInstance(n, K, H, L, obs, subj, member)
    for d in (12, 13, 14):
        for i in range(H - 1):
            assert cd.aggregate(alert(i + 1, d, i))[0] == []
    for d in (12, 14):
        cd.aggregate(alert(H, d, H - 1))
        assert cd.aggregate(alert(H + 1, d, H - 1))[0] == [] and cd.num_proposals() == 0
    ret, _ = cd.aggregate(alert(H, 13, H - 1))
    assert sorted(ret) == [12, 13, 14] and cd.num_proposals() == 1
    # KAT-5 (below L is noise)
    cd = pyemu.CdInstance(n, K, H, L, obs, subj, member)
    for i in range(H - 1):
        cd.aggregate(alert(i + 1, 12, i))
    cd.aggregate(alert(1, 13, 0))
    for i in range(H - 1):
        cd.aggregate(alert(i + 1, 14, i))
    assert cd.aggregate(alert(H, 12, H - 1))[0] == []
    assert sorted(cd.aggregate(alert(H, 14, H - 1))[0]) == [12, 14]
    # KAT-6 (batch of 3 x 10 rings, one multi-alert call): three separate 1-node emissions
    cd = pyemu.CdInstance(n, K, H, L, obs, subj, member)
    batch = np.concatenate([alert(1, d, r) for d in (12, 13, 14) for r in range(K)])
    ret, counts = cd.aggregate(batch)
    assert ret == [12, 13, 14] and sum(counts) == 3 and cd.num_proposals() == 3


def test_cd_instance_kernel_kat7_link_invalidation():
    """CutDetectionTest.java:254-301 on the real ring topology of 127.0.0.2:2..31."""
    K, H, L = 10, 8, 2
    reg = O.Registry()
    view = O.MembershipView(reg, K)
    nodes = [reg.intern("127.0.0.2", 2 + i) for i in range(30)]
    for i, nd in enumerate(nodes):
        view.ringAdd(nd, (i + 1, i + 1))
    obs, subj, member = view.tables(30)
    cd = pyemu.CdInstance(30, K, H, L, obs, subj, member)

    def alert(src, dst, ring):
        a = np.zeros(1, dtype=S.ALERT_DTYPE)
        a["src"], a["dst"], a["ring_mask"], a["status"], a["cfg_id"] = src, dst, 1 << ring, S.DOWN, -1
        return a

    dst = nodes[0]
    observers = view.getObserversOf(dst)
    for i in range(H - 1):
        assert cd.aggregate(alert(observers[i], dst, i))[0] == []
    failed = set()
    for i in range(H - 1, K):
        oo = view.getObserversOf(observers[i])
        failed.add(observers[i])
        for j in range(K):
            assert cd.aggregate(alert(oo[j], observers[i], j))[0] == []
            assert cd.num_proposals() == 0
    ret = cd.invalidate()
    assert len(ret) == 4 and cd.num_proposals() == 1
    assert set(ret) == failed | {dst}


def test_declared_alert_set_must_cover_the_delivered_streams():
    """rapid_sim_set_alert_set semantics in the kernel: the index is built from the declared set; a delivered report about a
    subject / ring the set does not contain (which could make a subject hot that has no slot) is detected and reported,
    not silently under-counted."""
    n, K, H, L = 300, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario("C2", subj, cfg, n=n, f=12, H=H, L=L, receivers=np.arange(0, n, 23))
    fe, fn, fo, fp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off)
    for trusted in (False, True):
        *res, covered = pyemu.tally(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member, trusted=trusted,
                                    declared=sc.batches.recs)
        assert covered and np.array_equal(res[0], fe) and np.array_equal(res[2], np.diff(fo))
        # the set without every alert about one faulty subject: that subject is not hot in the index, but it is delivered
        short = sc.batches.recs[sc.batches.recs["dst"] != sc.faulty[0]]
        *_, covered = pyemu.tally(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member, trusted=trusted, declared=short)
        assert not covered
        # ... and without ONE ring of a subject that stays hot: harmless, the slot exists and holds all K rings
        one = np.flatnonzero((sc.batches.recs["dst"] == sc.faulty[1]))[0]
        *res, covered = pyemu.tally(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member, trusted=trusted,
                                    declared=np.delete(sc.batches.recs, one))
        assert covered and np.array_equal(res[0], fe) and np.array_equal(res[2], np.diff(fo))


@pytest.mark.parametrize("n,n_out,n_crash,n_join,chunked", [(300, 30, 10, 12, False), (300, 30, 10, 12, True), (9000, 60, 120, 40, True),
                                                            (17000, 40, 200, 30, True), (300, 30, 10, 12, "fused"), (301, 0, 25, 0, "fused"),
                                                            (9000, 60, 120, 40, "fused"), (17000, 40, 200, 30, "fused")])
def test_round_index_kernels_match_their_host_statement(n, n_out, n_crash, n_join, chunked):
    """The kernels that build the per-round index (rapid_amd/csrc/index_kernels.h), emulated: the one-launch form of a declared
    alert set (index_fused_kernel: everything in one workgroup's LDS), the two-kernel form (touch, then one workgroup) and the
    form of large populations (nodes walked by workgroups of 8,192: count, assign, then one workgroup for the adjacency)
    must produce, entry for entry, what pyemu.build_round_index states in numpy -- the statement every emulated tally test
    already runs on: slots ascending by node, dictionary and declared masks, the hot adjacency triples and per-slot masks,
    the compressed tables, the counts and validation flags the host reads, and a work area left zeroed."""
    K, H, L = 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K, list(range(0, n - n_out)))
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs, member, cfg, n_crash, n_join, H, L, receivers=[0])
    alerts = sc.batches.recs
    want = pyemu.build_round_index(alerts, n, K, L, np.asarray(obs), member)
    fused = chunked == "fused"
    chunked = chunked is True
    got = pyemu.index_run(alerts, n, K, L, cfg, obs, member, chunked=chunked, fused=fused)
    nh, na, nt = want["n_hot"], want["n_adj"], want["n_touched"]
    assert (int(got["info"][0]), int(got["info"][1]), int(got["info"][3]), int(got["info"][5])) == (nh, nh, na, nt)
    assert got["info"][2] == 0 and got["info"][6] == 1 and got["info"][7] == 0
    ok, all_down = pyemu.validate_alerts(alerts, n, K, cfg, member)
    assert int(got["info"][4]) == (0 if ok else 1) | (0 if all_down else 2)
    assert np.array_equal(got["dict"][:n], want["dict"][:n]) and np.array_equal(got["decl"][:n], want["decl"][:n])
    assert np.array_equal(got["node_of_slot"][:nh], want["node_of_slot"][:nh])
    assert np.array_equal(got["smask"][:nh], want["adj_off"][:nh]) and np.array_equal(got["pairs"][:na], want["adj"][:na])
    nw = (n + 31) // 32
    assert np.array_equal(got["tbits"][:nw], want["tbits"]) and np.array_equal(got["trank"][:nw], want["trank"])
    assert np.array_equal(got["tent"][:nt], want["tent"][:nt])
    if fused:  # ... and the 32-bit entries the tally stages, which the other forms leave to dict_entries_kernel
        assert np.array_equal(got["entries"], pyemu.dict_entries(want, n))
    if not chunked:  # with room for the direct tables the one-workgroup form skips the compressed ones and says so
        lean = pyemu.index_run(alerts, n, K, L, cfg, obs, member, direct_budget=160 * 1024, fused=fused)
        assert lean["info"][7] == 1 and lean["info"][6] == 0 and np.array_equal(lean["dict"][:n], want["dict"][:n])
        assert np.array_equal(lean["pairs"][:na], want["adj"][:na]) and (lean["tbits"][:nw] == 0xEEEEEEEE).all()


def _population_of_voters(rng, R, n_props, prop_cap, p_vote=0.8):
    """R receivers, each silent or voting for one of n_props random node lists (skewed towards the first)."""
    cuts = [sorted(rng.choice(5000, size=int(rng.integers(1, prop_cap + 1)), replace=False).tolist()) for _ in range(n_props)]
    w = np.array([8.0] + [1.0] * (n_props - 1))
    which = rng.choice(n_props, size=R, p=w / w.sum())
    votes = rng.random(R) < p_vote
    props = np.zeros((R, prop_cap), dtype=np.int32)
    pcount = np.zeros(R, dtype=np.int32)
    poff = [0]
    flat = []
    for r in range(R):
        if votes[r]:
            c = cuts[which[r]]
            props[r, : len(c)] = c
            pcount[r] = len(c)
            flat += c
        poff.append(len(flat))
    fp = proposal_fingerprints(np.array(poff), np.array(flat, dtype=np.int64), votes)
    return cuts, which, votes, props, pcount, fp


@pytest.mark.parametrize("seed", range(6))
def test_vote_kernels_settle_a_round(seed):
    """The fast-round kernels (rapid_amd/csrc/vote_kernels.h), emulated, against a direct count (R/FastPaxos.java:125-156:
    votes per identical proposal): the counting pass + verification find the plurality proposal, its votes and its list;
    the verification alone, on the statistics the tally kernel gathers, settles the lowest voter's proposal; a receiver whose
    list differs under an equal fingerprint is counted as an offender, never as a vote."""
    rng = np.random.default_rng(4200 + seed)
    R, prop_cap = int(rng.integers(1, 700)), 12
    n_props = int(rng.integers(1, 5))
    cuts, which, votes, props, pcount, fp = _population_of_voters(rng, R, n_props, prop_cap, p_vote=float(rng.choice([0.0, 0.5, 0.95])) if seed else 0.9)
    voters = np.flatnonzero(votes)
    counts = np.bincount(which[voters], minlength=n_props) if len(voters) else np.zeros(n_props, dtype=int)
    res, ref = pyemu.vote_settle(fp, pcount, props, mode=0)
    assert int(res[2]) == len(voters)
    if len(voters) == 0:
        assert int(res[1]) == 0
    else:
        best = int(counts.max())
        assert int(res[1]) == best == int(res[7]) and int(res[6]) == 0
        winners = [i for i in range(n_props) if counts[i] == best]
        assert ref in [cuts[i] for i in winners]
        assert int(res[4]) == int(fp[voters[which[voters] == cuts.index(ref)][0]]) and int(res[5]) == (~int(res[4])) & ((1 << 64) - 1)
        assert int(res[3]) == int((counts > 0).sum())
        # the candidate form: the lowest voter's proposal, whatever the others hold
        res1, ref1 = pyemu.vote_settle(fp, pcount, props, mode=1)
        lead = int(which[voters[0]])
        assert ref1 == cuts[lead] and int(res1[1]) == int(counts[lead]) == int(res1[7]) and int(res1[2]) == len(voters)
        assert int(res1[3]) == (1 if counts[lead] == len(voters) else 2) and int(res1[4]) == int(fp[voters[0]]) and int(res1[6]) == 0
        # an equal fingerprint over a different list: an offender
        if len(voters) >= 3:
            forged = int(voters[-1])
            if which[forged] == lead:
                props2 = props.copy()
                props2[forged, 0] += 1
                res2, _ = pyemu.vote_settle(fp, pcount, props2, mode=1)
                assert int(res2[6]) == 1 and int(res2[7]) == int(counts[lead])
        # the same settled on the voters' bitmaps (what the tally kernel writes beside the lists: one bit per hot slot, the
        # slot numbering = the rank of a node among the round's proposed nodes), 9 and 1 words per receiver
        nodes = sorted({x for c in cuts for x in c})
        slot = {x: i for i, x in enumerate(nodes)}
        for words in (1, 9, 300):
            bits = np.zeros((R, words), dtype=np.uint64)
            spread = (words * 64) // max(len(nodes), 1)
            for r in voters:
                for x in cuts[int(which[r])]:
                    b = slot[x] * max(spread, 1)
                    bits[r, b >> 6] |= np.uint64(1 << (b & 63))
            for mode in (2, 3):  # a thread per receiver / a wave per receiver
                res3, ref3 = pyemu.vote_settle(fp, pcount, props, mode=mode, bits=bits)
                assert ref3 == ref1 and res3.tolist() == res1.tolist()
                if len(voters) >= 3 and which[int(voters[-1])] == lead:
                    bits2 = bits.copy()
                    bits2[int(voters[-1]), words - 1] ^= np.uint64(1 << 63)
                    res4, _ = pyemu.vote_settle(fp, pcount, props, mode=mode, bits=bits2)
                    assert int(res4[6]) == 1 and int(res4[7]) == int(counts[lead])


@pytest.mark.parametrize("seed", range(8))
def test_vote_merge_kernel_equals_the_host_statement(seed):
    """vote_merge_kernel (what every rank runs over the all-gathered answers) against rapid_amd.parallel.merge_candidates on
    random ranks: a common candidate with and without a quorum, dissenters, silent ranks, ranks whose candidates differ,
    lists that differ under an equal fingerprint."""
    from rapid_amd import parallel as P
    rng = np.random.default_rng(9100 + seed)
    prop_cap, membership = 16, int(rng.integers(40, 400))
    quorum = P.fast_quorum(membership)
    cut_a = sorted(rng.choice(1000, size=int(rng.integers(1, 12)), replace=False).tolist())
    cut_b = sorted(rng.choice(1000, size=len(cut_a), replace=False).tolist())
    fa, fb = 0x1111222233334444 + seed, 0x9999AAAABBBBCCCC + seed
    for trial in range(12):
        n_ranks = int(rng.integers(1, 6))
        blocks, host = [], []
        for k in range(n_ranks):
            kind = rng.choice(["silent", "a", "a_dissent", "b", "a_otherlist"], p=[0.15, 0.45, 0.2, 0.1, 0.1])
            voters = 0 if kind == "silent" else int(rng.integers(1, membership // max(1, n_ranks) + 2))
            votes = voters if kind != "a_dissent" else max(1, voters - int(rng.integers(1, 4)))
            fp, cut = (fb, cut_b) if kind == "b" else (fa, cut_a)
            if kind == "a_otherlist":
                cut = cut_b
            if voters == 0:
                fp, cut, votes = 0, None, 0
            blocks.append(pyemu.vote_block(votes, voters, fp, cut, prop_cap))
            host.append(dict(voters=voters, votes=votes, fp=fp, cut=cut))
        settled, hvotes, hvoters, hcut = P.merge_candidates(host, membership)
        status, votes, voters, cut, res = pyemu.vote_merge(blocks, prop_cap, quorum, seed=trial + 1)
        assert (status == 1) == settled, (host, status)
        assert voters == hvoters
        if settled:
            assert votes == hvotes and int(res[7]) == votes and int(res[6]) == 0
            if hvoters:
                assert cut == host[[h["voters"] > 0 for h in host].index(True)]["cut"]
                assert (hcut is not None) == (votes >= quorum)


@pytest.mark.parametrize("n,K,n_members", [(1, 10, 1), (2, 10, 2), (3, 10, 3), (50, 3, 50), (64, 10, 40), (300, 10, 270), (1100, 7, 1100)])
def test_view_kernels_match_the_oracle(n, K, n_members):
    """The view kernels (rapid_amd/csrc/view_kernels.h), emulated, against the restated MembershipView: XXH64 ring keys
    (R/MembershipView.java:562-587; hostnames of every tail length, negative ports), the K rings, observers / subjects of
    members and expected observers of non-members (:210-322), the configuration id (:544-556, the ordered (v, 37^len)
    reduction), and the rings of a removal-only view change obtained by compaction instead of a new sort (:167-201)."""
    rng = np.random.default_rng(77 + n)
    pop = S.Population.make(n)
    if n >= 50:  # every XXH64 path: short tails, the 32-byte stripe loop, ports with the sign bit set
        pop.hostnames = [bytes(rng.integers(33, 127, size=int(l)).astype(np.uint8)) + b"-%d" % i for i, l in enumerate(rng.integers(0, 90, size=n))]
        pop.ports = rng.integers(-2**31, 2**31 - 1, size=n).astype(np.int32)
    members = sorted(rng.permutation(n)[:n_members].tolist())
    reg, oview = oracle_view(pop, K, members)
    keep = np.zeros(n, dtype=bool)
    keep[members] = True
    gone = members[:: max(2, n_members // 7)] if n_members > 3 else []
    keep[gone] = False
    outsiders = [x for x in range(n) if x not in set(members)]
    come = outsiders[:: max(1, len(outsiders) // 9)]  # joiners admitted by the same cut (none where everybody is a member)
    keep[come] = True
    got = pyemu.view_build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo, K, members, keep=keep)
    for k in range(K):
        for node in range(0, n, max(1, n // 40)):
            assert int(got["keys"][k, node]) == oview.ringKey(k, node), (k, node)
        assert np.array_equal(got["ring"][k], oview.getRing(k)), k
    oobs, osubj, omember = oview.tables(n)
    assert np.array_equal(got["obs"], oobs) and np.array_equal(got["subj"], osubj)
    assert int(got["cfg"][0]) == oview.getCurrentConfigurationId()
    for node in gone:
        oview.ringDelete(node)
    for j, node in enumerate(come):
        oview.ringAdd(node, (int(pop.id_hi[node]), int(pop.id_lo[node])))
    assert got["ring2"].shape[1] == n_members - len(gone) + len(come)
    for k in range(K):
        assert np.array_equal(got["ring2"][k], oview.getRing(k)), ("view change", k)
    if n_members >= 2 and got["ring2"].shape[1] >= 2:  # the tables of the new view, patched from the old ones where the cut touched them
        # (against a view built afresh over the new membership: the changed one's tables() go through its observer memo, which may
        # hold stale rows by now -- quirk Q4, tested where it belongs)
        _, fresh = oracle_view(pop, K, sorted(np.flatnonzero(keep).tolist()))
        oobs2, osubj2, _ = fresh.tables(n)
        assert np.array_equal(got["obs2"], oobs2) and np.array_equal(got["subj2"], osubj2)


def test_identifiers_merged_on_the_device():
    """identifiersSeen stays sorted on the device (signed high, then signed low: R/MembershipView.java:474-500); the NodeIds a
    cut admits are merged in by ids_merge_kernel."""
    rng = np.random.default_rng(5)
    for n_old, n_new in ((0, 3), (5, 0), (1, 1), (700, 40), (300, 300)):
        ids = {(int(a), int(b)) for a, b in rng.integers(-2**62, 2**62, size=(n_old + n_new, 2))}
        ids |= {(7, -3), (7, 5), (-7, 0)} if n_old + n_new > 3 else set()
        ids = list(ids)
        rng.shuffle(ids)
        old, new = sorted(ids[:n_old]), sorted(ids[n_old:])
        assert pyemu.ids_merge(old, new) == sorted(ids)


def _core_words(recs):
    return (recs["ring_mask"].astype(np.uint32) & 0x3FFF) | np.where(recs["status"] != 0, 1 << 14, 1 << 15).astype(np.uint32) | \
        ((recs["flags"].astype(np.uint32) & 1) << 16)


def test_streams_generated_on_the_device_equal_the_host_statement():
    """rapid_sim_generate (gen_resolve_alerts / gen_streams kernels, emulated) against scenarios.deliver_hashed: the same records
    in the same order for every receiver -- as 20-byte boundary records (byte for byte) and as resolved 8-byte records (the
    subject's dictionary entry of the round's index, ring mask + status + the batch end on the last alert of every batch);
    an alert of another configuration in the set resolves to the poison entry; with per-batch delivery thresholds the places
    of the batches that do not reach a receiver hold empty records; and the generated records, tallied, give what the host's
    records give on the oracle."""
    n, K, H, L = 300, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K, list(range(0, n - 20)))
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs, member, cfg, 10, 8, H, L, materialise=False)
    rx = sc.receivers[::17]
    rng = np.random.default_rng(4)
    for seed, stale_at, with_keep in ((2, None, False), (12345678901234567, 3, False), (99, None, True)):
        bs = sc.batches
        if stale_at is not None:
            recs = bs.recs.copy()
            recs["cfg_id"][stale_at] = cfg + 1
            bs = S.BatchSet(recs, bs.off, bs.sender)
        keep = None
        if with_keep:
            keep = np.where(rng.random(bs.n_batches) < 0.3, rng.integers(0, 1 << 32, size=bs.n_batches), 0xFFFFFFFF).astype(np.uint32)
            keep[0] = 0  # (practically never delivered)
        want, want_off, nb = S.deliver_hashed(bs, rx, seed, keep=keep)
        A = int(bs.off[-1])
        got, rec_off = pyemu.generate(bs, rx, seed, cfg, n, keep=keep, boundary=True)
        assert np.array_equal(rec_off, want_off)
        assert got.tobytes() == want.tobytes()
        ix = pyemu.build_round_index(bs.recs, n, K, L, np.asarray(obs), member)
        entries = pyemu.dict_entries(ix, n)
        ent, words, rec_off = pyemu.generate(bs, rx, seed, cfg, n, entries=entries, keep=keep)
        assert np.array_equal(rec_off, want_off)
        empty = (want["ring_mask"] == 0) & (want["cfg_id"] == 0)
        stale = (want["cfg_id"] != cfg) & ~empty
        assert stale.any() == (stale_at is not None) and empty.any() == with_keep
        assert np.array_equal(words, np.where(empty, 0, _core_words(want)).astype(np.uint32))
        want_ent = np.where(empty, 0, np.where(stale, entries[n], entries[np.minimum(want["dst"], n)])).astype(np.uint32)
        assert np.array_equal(ent, want_ent)
        if keep is None:
            # every receiver got every batch once, and two receivers got them in different orders
            assert all(sorted(want["src"][r * A:(r + 1) * A].tolist()) == sorted(bs.recs["src"].tolist()) for r in range(len(rx)))
            assert not np.array_equal(want["src"][:A], want["src"][A:2 * A])
        else:
            assert nb.min() < bs.n_batches and int((want["flags"] & 1).sum()) == int(nb.sum())
        # the generated streams through the emulated tally (boundary records, tables looked up) = the oracle on the host's records
        _check(got, rec_off, n, K, H, L, cfg, obs, subj, member, seed=int(seed) & 0xFFFF)


def test_generator_edge_cases_and_the_late_delivery_model():
    """One batch / one receiver / single-alert batches / a tile boundary through the emulated generator; and
    scenarios.deliver's late deliveries: additional whole batches carrying the previous configuration id -- every batch of the
    round still arrives exactly once."""
    n, K, H, L = 60, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs, member, cfg, 3, 0, H, L, materialise=False)
    one = S.BatchSet(sc.batches.recs[: sc.batches.off[1]].copy(), sc.batches.off[:2].copy(), sc.batches.sender[:1].copy())
    # more batches than one tile of the lay-down (1,024 deliveries), of one to three alerts each
    lens = np.random.default_rng(1).integers(1, 4, size=1100)
    big = np.zeros(int(lens.sum()), dtype=S.ALERT_DTYPE)
    big["cfg_id"], big["dst"], big["ring_mask"], big["status"] = cfg, np.arange(len(big)) % n, 1 + (np.arange(len(big)) % 7), S.DOWN
    big["src"] = np.repeat(np.arange(1100), lens)
    many = S.BatchSet(big, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), np.arange(1100, dtype=np.int32))
    for bs, rx in ((one, sc.receivers[:1]), (one, sc.receivers[:5]), (sc.batches, sc.receivers[:1]), (many, sc.receivers[:3])):
        want, want_off, nb = S.deliver_hashed(bs, rx, 7)
        got, rec_off = pyemu.generate(bs, rx, 7, cfg, n, boundary=True)
        assert got.tobytes() == want.tobytes() and np.array_equal(rec_off, want_off)
        assert int((got["flags"] & 1).sum()) == bs.n_batches * len(rx)
    # late deliveries
    recs, off, nb = S.deliver(sc.batches, sc.receivers[:6], 2, stale_cfg=cfg - 1, stale_rate=0.25)
    B = sc.batches.n_batches
    n_late = max(1, int(round(0.25 * B)))
    assert np.all(nb == B + n_late)
    for r in range(6):
        seg = recs[off[r]:off[r + 1]]
        cur = seg[seg["cfg_id"] == cfg]
        late = seg[seg["cfg_id"] != cfg]
        assert len(late) > 0 and np.all(late["cfg_id"] == cfg - 1)
        # the round's own alerts: every one exactly once, whatever was delivered late
        key = lambda a: sorted(zip(a["src"].tolist(), a["dst"].tolist(), a["ring_mask"].tolist()))
        assert key(cur) == key(sc.batches.recs)
        assert int((seg["flags"] & 1).sum()) == B + n_late  # whole batches, each with its end


def _q4_scene():
    """A view in which the reference's memoised observers of a member are STALE (quirk Q4, R/MembershipView.java:143-152,
    181-195): x is the maximum of ring 0 and was queried (getObserversOf memoised: its ring-0 observer is the ring minimum m0,
    by wrap-around); then m0 leaves -- ringDelete drops the entries of m0's predecessors without wrap-around, and x, the
    maximum, is not one of them -- so the memo keeps naming m0.  -> (pop, oracle view, x, m0, stale row, K, H, L)"""
    K, H, L = 10, 9, 4
    for n in range(300, 340):
        pop = S.Population.make(n)
        reg, oview = oracle_view(pop, K)
        ring0 = oview.getRing(0)
        x, m0 = int(ring0[-1]), int(ring0[0])
        # m0 must not be x's direct successor on another ring (its removal would drop x's entry legitimately)
        if m0 in oview.getObserversOf(x)[1:]:
            continue
        assert oview.getObserversOf(x)[0] == m0
        oview.ringDelete(m0)
        stale, fresh = oview.getObserversOf(x), oview.computeObserversOf(x)
        assert stale[0] == m0 and fresh[0] == int(oview.getRing(0)[0]) != m0 and stale[1:] == fresh[1:]
        return pop, oview, x, m0, stale, K, H, L
    raise AssertionError("no population with the scene found")


def _q4_round(oview, x, m0, n, K, cfg, n_receivers=6, seed=3):
    """Alerts under which the stale entry decides the outcome: m0, gone from the view, is reported UP by its expected observers
    on every ring (a joiner in the making: it enters preProposal / proposal), x is reported DOWN on eight rings other than
    ring 0 -- one short of H = 9.  The reference's invalidateFailingEdges asks for x's observers, gets the stale list, finds m0
    (in proposal) as x's ring-0 observer and credits x with the implicit report: x reaches H and is proposed.  With today's
    observers x's ring-0 observer is the new ring minimum, which nobody reports: x stays at eight and blocks every proposal."""
    obs, subj, member = oview.tables(n)
    recs = []
    for k in range(K):
        recs.append(S_alert(int(obs[m0, k]), m0, S.UP, cfg, k))          # expected observers of the non-member m0
    for k in range(1, 9):
        recs.append(S_alert(int(obs[x, k]), x, S.DOWN, cfg, k))          # x's (fresh == memoised, on these rings) observers
    alerts = np.concatenate(recs)
    alerts["flags"] = 1  # every alert its own BatchedAlertMessage
    rng = np.random.default_rng(seed)
    parts = [alerts[rng.permutation(len(alerts))] for _ in range(n_receivers)]
    off = np.arange(n_receivers + 1, dtype=np.int64) * len(alerts)
    return alerts, np.concatenate(parts), off, obs, subj, member


def S_alert(src, dst, status, cfg, ring):
    a = np.zeros(1, dtype=S.ALERT_DTYPE)
    a["src"], a["dst"], a["status"], a["cfg_id"], a["ring_mask"] = src, dst, status, cfg, 1 << ring
    return a


def test_q4_stale_observer_memo_decides_a_round_as_in_the_reference():
    """Quirk Q4 reproduced: the faithful oracle with its observer cache as the view changes left it, against the emulated tally
    over a round index whose row for the stale member is the MEMOISED one -- equal; over today's table -- different (nobody
    proposes).  And the index kernel itself, given the memo arrays: a hot member without an entry is memoised from today's
    table, a hot member with a stale entry is read from the memo and the round is marked (info[2] bit 2)."""
    pop, oview, x, m0, stale, K, H, L = _q4_scene()
    n = pop.n
    cfg = oview.getCurrentConfigurationId()
    alerts, records, rec_off, obs, subj, member = _q4_round(oview, x, m0, n, K, cfg)
    oe, on, oo, op = O.sim_run(oview, K, H, L, pop.id_hi, pop.id_lo, records, rec_off, prewarm_observers=False)
    assert np.all(oe >= 0) and all(sorted(op[oo[r]:oo[r + 1]].tolist()) == sorted([x, m0]) for r in range(len(oe)))
    # (the oracle's tables() go through getObserversOf: they ARE the memoised rows; today's row of x comes from computeObserversOf)
    obs_memo = np.asarray(obs).copy()
    assert obs_memo[x].tolist() == stale
    obs = obs_memo.copy()
    obs[x] = oview.computeObserversOf(x)
    for trusted in (False, True):
        emit, nprop, pcount, fpr, props, stats = pyemu.tally(records, rec_off, n, K, H, L, cfg, obs_memo, subj, member, trusted=trusted)
        assert np.array_equal(emit, oe) and np.array_equal(nprop, on) and np.array_equal(pcount, np.diff(oo))
        assert all(props[r, : pcount[r]].tolist() == sorted([x, m0]) for r in range(len(oe)))
        emit_f, _, pcount_f, *_ = pyemu.tally(records, rec_off, n, K, H, L, cfg, obs, subj, member, trusted=trusted)
        # today's observers: x never reaches H and blocks the proposal (a receiver that has m0 complete before x reaches L
        # proposes m0 alone, with or without the memo)
        assert np.all(pcount_f <= 1) and np.array_equal(emit_f >= 0, np.diff(oo) == 1) and not np.array_equal(emit_f, oe)
    # the index kernels with the memo
    want_memo = pyemu.build_round_index(alerts, n, K, L, obs_memo, member)
    want_fresh = pyemu.build_round_index(alerts, n, K, L, np.asarray(obs), member)
    assert want_memo["n_adj"] == want_fresh["n_adj"] + 1
    for chunked, fused in ((False, False), (True, False), (False, True)):
        rows, valid = np.full((n, K), -7, dtype=np.int32), np.zeros(n, dtype=np.uint8)
        got = pyemu.index_run(alerts, n, K, L, cfg, obs, member, chunked=chunked, fused=fused, q4=(rows, valid))
        assert got["info"][2] == 0 and np.array_equal(got["pairs"][: want_fresh["n_adj"]], want_fresh["adj"][: want_fresh["n_adj"]])
        assert valid[x] == 1 and rows[x].tolist() == np.asarray(obs)[x].tolist() and valid[m0] == 0 and valid.sum() == 1  # members only
        rows[x] = stale
        got = pyemu.index_run(alerts, n, K, L, cfg, obs, member, chunked=chunked, fused=fused, q4=(rows, valid))
        assert got["info"][2] == 4 and got["info"][3] == want_memo["n_adj"]
        assert np.array_equal(got["pairs"][: want_memo["n_adj"]], want_memo["adj"][: want_memo["n_adj"]])
        assert np.array_equal(got["smask"][: want_memo["n_hot"]], want_memo["adj_off"][: want_memo["n_hot"]])
        assert rows[x].tolist() == stale  # a stale entry stays what it is


def test_stream_offsets_that_cannot_be_followed_void_the_round():
    """The offsets of an attached stream set are checked by the wave that is about to follow them (TallyParams::stream_bytes): a
    stream whose offsets descend, or run past the records, is not read -- the emulator's bounds model would catch a read -- and
    the launch reports it (error flag bit 1 -> RAPID_EINVAL on the device path)."""
    n, K, H, L = 40, 5, 4, 2
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    rng = np.random.default_rng(5)
    lens = [30, 300, 12, 700, 64]
    parts = [random_stream(rng, n, K, member, cfg, m, [1, 2, 3, 4, 5]) for m in lens]
    records, off = np.concatenate(parts), np.cumsum([0] + lens)
    for bad in ("descending", "past the end"):
        boff = off.copy()
        if bad == "descending":
            boff[2], boff[3] = boff[3], boff[2]
        else:
            boff[-1] += 5
        for mode in (1, 3):
            *res, covered = pyemu.tally(records, boff, n, K, H, L, cfg, obs, subj, member, tables_in_lds=mode, declared=records)
            assert not covered, (bad, mode)
    *res, covered = pyemu.tally(records, off, n, K, H, L, cfg, obs, subj, member, declared=records)
    assert covered


def test_ring_merge_orders_equal_keys_by_node_index():
    """Two endpoints with the same 64-bit ring key (one chance in 2^64 per pair; the reference's TreeSet would refuse the second,
    R/MembershipView.java:123-141) stand in the order of their node indices -- what the stable sort of a full build produces -- on
    both sides of the incremental merge (ring_scatter_kernel / ring_join_kernel): keys drawn from a handful of values, so that
    survivors tie with survivors, joiners with joiners and joiners with survivors, across chunk boundaries."""
    rng = np.random.default_rng(5)
    for n, m_old, J, n_keys in ((3000, 2500, 300, 40), (200, 150, 50, 3), (64, 1, 40, 2), (2200, 2100, 0, 7)):
        perm = rng.permutation(n)
        old, join = perm[:m_old], perm[m_old:m_old + J]
        key = rng.integers(0, n_keys, size=n).astype(np.uint64) * np.uint64(0x1111111111111111)
        order = lambda nodes: np.array(sorted(nodes.tolist(), key=lambda v: (int(key[v]), v)), dtype=np.int32)
        ring = order(old)
        member = np.zeros(n, dtype=np.uint8)
        stay = old[rng.random(m_old) < 0.8]
        member[stay] = 1
        member[join] = 1
        jn = order(join)
        want = order(np.concatenate([stay, join]))
        got_r, got_k = pyemu.ring_merge(ring, key[ring], member, jn, key[jn], seed=int(n))
        assert np.array_equal(got_r, want) and np.array_equal(got_k, key[want])


@pytest.mark.parametrize("seed", range(4))
def test_view_change_flag_kernels(seed):
    """q4_invalidate_kernel (what ringAdd / ringDelete drop from the observer memo, R/MembershipView.java:143-152, 181-195: the node's
    own entry when it leaves, and its ring predecessors WITHOUT wrap-around; the member flags of nodes that leave cleared on the way)
    and member_patch_kernel, emulated, against their statement in numpy."""
    import ctypes as C
    rng = np.random.default_rng(5100 + seed)
    n_nodes, K = int(rng.integers(5, 400)), int(rng.integers(3, 11))
    subj = rng.integers(-1, n_nodes, size=(n_nodes, K)).astype(np.int32)   # ring predecessors (-1: none)
    ring = rng.integers(0, n_nodes, size=(K, 3)).astype(np.int32)          # ring[k][0] = the ring minimum: no predecessor below it
    nodes = rng.choice(n_nodes, size=int(rng.integers(0, n_nodes // 2 + 1)), replace=False).astype(np.int32)
    valid0 = rng.integers(0, 2, size=n_nodes).astype(np.uint8)
    member0 = rng.integers(0, 2, size=n_nodes).astype(np.uint8)
    rest = np.setdiff1d(np.arange(n_nodes), nodes)
    joined = rng.choice(rest, size=int(rng.integers(0, len(rest) // 2 + 1)), replace=False).astype(np.int32) if len(rest) else np.zeros(0, np.int32)
    L_ = pyemu.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for self_flag, clear in ((1, True), (0, False), (1, False)):
        valid, member = valid0.copy(), member0.copy()
        gone = np.zeros(0, np.int32) if clear else nodes  # (cleared by the memo kernel, or by the patch kernel)
        rc = L_.emu_view_flags(p(subj), p(ring), 3, p(nodes) if len(nodes) else None, len(nodes), n_nodes, K, p(valid), self_flag,
                               p(member) if clear else None, p(member), p(gone) if len(gone) else None, len(gone),
                               p(joined) if len(joined) else None, len(joined), C.c_ulonglong(seed))
        assert rc == 0
        want_valid, want_member = valid0.copy(), member0.copy()
        for node in nodes.tolist():
            if self_flag:
                want_valid[node] = 0
            for k in range(K):
                if ring[k, 0] != node and 0 <= subj[node, k] < n_nodes:
                    want_valid[subj[node, k]] = 0
        want_member[nodes] = 0
        want_member[joined] = 1
        assert np.array_equal(valid, want_valid) and np.array_equal(member, want_member), (self_flag, clear)


@pytest.mark.parametrize("n,K", [(1, 3), (700, 2), (1024, 1), (1025, 2), (2600, 3)])
def test_joiners_sorted_per_ring_in_runs(n, K):
    """ring_sort_runs_kernel + ring_merge_runs_kernel (a cut's joiners, ring by ring, by (ring key, node index)): runs of 1,024 pairs
    sorted in LDS, every pair placed among the other runs by binary search -- against numpy, with EQUAL keys among the pairs (which the
    64-bit hashes of real endpoints never produce; the node index decides) and run boundaries that do not divide n."""
    import ctypes as C
    rng = np.random.default_rng(900 + n)
    keys = rng.integers(0, 2**63, size=(K, n), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(K, n)).astype(np.uint64)
    keys[:, ::5] = keys[:, 0:1]  # every fifth pair of a ring shares one key
    nodes = np.stack([rng.permutation(n + 50)[:n] for _ in range(K)]).astype(np.int32)
    want_k, want_n = np.empty_like(keys), np.empty_like(nodes)
    for k in range(K):
        order = np.lexsort((nodes[k], keys[k]))
        want_k[k], want_n[k] = keys[k][order], nodes[k][order]
    kk, nn = keys.copy(), nodes.copy()
    out_k, out_n = np.zeros_like(keys), np.full_like(nodes, -1)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L_ = pyemu.lib()
    L_.emu_join_sort.restype = C.c_int
    assert L_.emu_join_sort(p(kk), p(nn), n, K, p(out_k), p(out_n), C.c_ulonglong(n)) == 0
    assert np.array_equal(out_k, want_k) and np.array_equal(out_n, want_n)


def test_joiner_identifiers_put_in_order_on_the_host():
    """node_id_sort.h (the host's ordering of a cut's joiner NodeIds, in front of the identifiersSeen check and the merge): the order
    of std::sort over (high, low) signed pairs -- for spread UUIDs (the counting pass), for identifiers that share their top bits
    (full buckets), with duplicates and ties in the high word, below and above the size where the counting pass starts."""
    import ctypes as C
    L_ = pyemu.lib()
    rng = np.random.default_rng(17)
    cases = []
    for n in (1, 2, 255, 256, 257, 5000, 70000):
        cases.append(rng.integers(-2**63, 2**63 - 1, size=(n, 2), dtype=np.int64))                  # UUIDs
    seq = np.stack([np.full(3000, 7, dtype=np.int64), np.arange(3000, 0, -1, dtype=np.int64)], axis=1)  # one high word
    cases.append(seq)
    few = np.stack([rng.integers(-3, 3, size=4000), rng.integers(-5, 5, size=4000)], axis=1).astype(np.int64)  # ties and duplicates
    cases.append(few)
    edge = np.array([[2**63 - 1, -1], [-2**63, 0], [0, 0], [-1, 2**63 - 1], [-1, -2**63]] * 80, dtype=np.int64)
    cases.append(edge)
    for ids in cases:
        hi = np.ascontiguousarray(ids[:, 0]); lo = np.ascontiguousarray(ids[:, 1])
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        L_.emu_sort_node_ids.restype = C.c_int
        assert L_.emu_sort_node_ids(p(hi), p(lo), len(ids)) == 0
        want = sorted(map(tuple, ids.tolist()))
        assert list(zip(hi.tolist(), lo.tolist())) == want


def test_votes_accumulated_across_the_tiles_of_a_round():
    """vote_acc_pick / _count / _finish (csrc/vote_kernels.h; rapid_sim_round_tiled runs them once per tile): the candidate is the
    round's FIRST voter's proposal (or, in a counting pass, the first voter holding a given fingerprint), every voter holding its
    fingerprint is compared with it bit for bit -- a voter with the fingerprint and another bitmap is an offender, not a vote --,
    votes and voters add up over the tiles whatever the tiling, and the block left for the all-gather merges like the block of a
    population counted in one launch (R/FastPaxos.java:141-150: a count per proposal is all that is kept)."""
    rng = np.random.default_rng(5)
    R, cap, words = 120, 40, 3  # (64 emulated lanes per receiver and pass: kept small)
    A = sorted(rng.choice(150, size=12, replace=False).tolist())
    B = sorted(rng.choice(150, size=9, replace=False).tolist())

    def bitmap(cut):
        w = np.zeros(words, dtype=np.uint64)
        for x in cut:
            w[x // 64] |= np.uint64(1) << np.uint64(x % 64)
        return w

    fpA, fpB = np.uint64(0x1111222233334444), np.uint64(0x9999AAAABBBBCCCC)
    kind = rng.choice(3, size=R, p=[0.2, 0.65, 0.15])  # 0: does not vote, 1: proposes A, 2: proposes B
    for first_kind in (1, 2):
        kind[:5] = 0
        kind[5] = first_kind
        fp = np.where(kind == 1, fpA, np.where(kind == 2, fpB, np.uint64(0))).astype(np.uint64)
        pc = np.where(kind == 1, len(A), np.where(kind == 2, len(B), 0)).astype(np.int32)
        props = np.zeros((R, cap), dtype=np.int32)
        bits = np.zeros((R, words), dtype=np.uint64)
        for r in range(R):
            cut = A if kind[r] == 1 else B if kind[r] == 2 else []
            props[r, : len(cut)] = cut
            bits[r] = bitmap(cut)
        liar = int(np.flatnonzero(kind == first_kind)[3])  # holds the candidate's fingerprint over another proposal
        bits[liar, 0] ^= np.uint64(1) << np.uint64(63)
        cand, other = (A, B) if first_kind == 1 else (B, A)
        n_cand, n_voters = int((kind == first_kind).sum()), int((kind != 0).sum())
        blocks = {}
        for tile in (R, 50, 33):
            res, ref, block = pyemu.vote_acc(fp, pc, props, bits, tile, seed=tile)
            assert ref == cand and int(res[1]) == int(res[7]) == n_cand - 1 and int(res[2]) == n_voters and int(res[6]) == 1, (tile, res)
            assert int(res[3]) == 2 and res[4] == (fpA if first_kind == 1 else fpB) and res[5] == ~res[4] and int(res[8]) == 0
            blocks[tile] = block
        assert all(np.array_equal(blocks[R], b) for b in blocks.values())  # the tiling does not show in the answer
        # a counting pass for the OTHER proposal (what the exact plurality asks for when the first candidate has no quorum)
        res, ref, _ = pyemu.vote_acc(fp, pc, props, bits, 70, target=int(fpB if first_kind == 1 else fpA))
        assert ref == other and int(res[1]) == int((kind == (3 - first_kind)).sum()) and int(res[2]) == n_voters and int(res[6]) == 0
        # two "ranks" (the receivers cut in two, each taken tile by tile) merged = the whole population's answer; a rank that holds an
        # offender -- or another candidate -- is not merged: the general count would run
        half = R // 2
        _, _, b0 = pyemu.vote_acc(fp[:half], pc[:half], props[:half], bits[:half], 40)
        _, _, b1 = pyemu.vote_acc(fp[half:], pc[half:], props[half:], bits[half:], 40)
        assert pyemu.vote_merge([b0, b1], cap, quorum=n_cand - 1)[0] == 2
        bits[liar] = bitmap(cand)
        _, _, b0 = pyemu.vote_acc(fp[:half], pc[:half], props[:half], bits[:half], 40)
        _, _, b1 = pyemu.vote_acc(fp[half:], pc[half:], props[half:], bits[half:], 40)
        status, votes, voters, cut, _ = pyemu.vote_merge([b0, b1], cap, quorum=n_cand)
        if kind[half:][np.flatnonzero(kind[half:] != 0)[0]] == first_kind:  # both ranks' first voters hold the same proposal
            assert (status, votes, voters, cut) == (1, n_cand, n_voters, cand)
        else:
            assert status == 2
    # nobody votes; a tally error flag of a tile travels with the block
    res, ref, _ = pyemu.vote_acc(np.zeros(50, dtype=np.uint64), np.zeros(50, dtype=np.int32), np.zeros((50, cap), dtype=np.int32),
                                 np.zeros((50, words), dtype=np.uint64), 16, tally_error=1)
    assert ref == [] and int(res[1]) == int(res[2]) == int(res[3]) == 0 and int(res[8]) == 1


def test_identifier_check_answered_into_the_mailbox():
    """ids_contains_publish_kernel (the check in front of every view change that admits joiners): one thread per joiner NodeId, a
    binary search in the sorted identifiersSeen, the answer -- 2 x the call's sequence number, + 1 if any was seen before
    (R/MembershipView.java:127-129) -- published by the last workgroup, which leaves the accumulators zero."""
    rng = np.random.default_rng(9)
    old = sorted({(int(h), int(l)) for h, l in rng.integers(-2**62, 2**62, size=(3000, 2))})
    fresh = [(int(h), int(l)) for h, l in rng.integers(-2**62, 2**62, size=(700, 2))]
    fresh = [x for x in fresh if x not in set(old)]
    assert pyemu.ids_contains_publish(old, fresh, seq=21) == 42
    assert pyemu.ids_contains_publish(old, fresh[:1], seq=5) == 10
    for where in (0, 299, len(fresh) - 1):  # one seen identifier, in the first / a middle / the last workgroup's share
        new = list(fresh)
        new[where] = old[(where * 7) % len(old)]
        assert pyemu.ids_contains_publish(old, new, seq=33) == 67
    assert pyemu.ids_contains_publish(old, [old[0]], seq=1) == 3 and pyemu.ids_contains_publish(old, [old[-1]], seq=1) == 3
    assert pyemu.ids_contains_publish(old, [(old[0][0], old[0][1] + 1)], seq=1) in (2, 3)  # (+ 1 only if that id happens to exist)


def test_fast_round_settled_inside_the_tally_launch():
    """TallyParams::vote_cand: the launch itself names a candidate (the first voter of the first workgroup to finish), compares every
    voter's bitmap with it and lays down the complete answer block -- pyemu.tally checks that block against the per-receiver results
    after EVERY run of this file.  Here the paths a sequential emulator does not reach by itself: a candidate that is claimed but
    published LATE (the workgroups in between hand their voters to the deferred list, the last one compares them: selector bit 17),
    a deferred list that overflows (bit 18: no quorum may be claimed then), and the round-5 statistics alone (bit 16)."""
    n, K, H, L = 300, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario("C3b", subj, cfg, n=n, f=15, H=H, L=L, receivers=np.arange(0, n, 5))
    fe, fn, fo, fp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off)
    assert int((np.diff(fo) != 0).sum()) > 30  # many voters, spread over the workgroups below
    for knob in (0, 131072, 131072 | 262144, 65536):
        for grid, waves in ((4, 3), (5, 2)):
            emit, nprop, pcount, fpr, props, stats = pyemu.tally(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member, force_exact=knob, trusted=True,
                                                                 grid=grid, waves=waves)
            assert np.array_equal(emit, fe) and np.array_equal(pcount, np.diff(fo))
            assert np.array_equal(fpr, proposal_fingerprints(fo, fp, fe >= 0))
    # a round in which two proposals are held (a dissenting receiver): the candidate's votes are the voters holding ITS fingerprint
    recs, off = sc.records.copy(), sc.rec_off
    r_diss = int(np.flatnonzero(np.diff(fo) != 0)[3])
    recs["ring_mask"][off[r_diss]: off[r_diss + 1]] &= 1  # this receiver hears of one ring only: it announces later, or something else
    fe2, fn2, fo2, fp2 = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, recs, off)
    for knob in (0, 131072):
        emit, nprop, pcount, fpr, props, stats = pyemu.tally(recs, off, n, K, H, L, cfg, obs, subj, member, force_exact=knob, grid=4, waves=3)
        assert np.array_equal(emit, fe2) and np.array_equal(fpr, proposal_fingerprints(fo2, fp2, fe2 >= 0))


def test_packed_rounds_with_thousands_of_hot_subjects():
    """The packed instantiations at the scale they exist for, as far as the emulator can go: 2,760 hot subjects (C3b's kind of round
    at 54,000 nodes) -- the sweep over the hot slots takes eight slots per lane and step and needs a second turn of its outer loop
    (more than 2,560 slots), a receiver's proposal is written in six pipelined batches of 512 slots, the last of them incomplete --
    against the oracle, on boundary records looked up in memory and on resolved records."""
    n, f, K, H, L = 54000, 2700, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario("C3b", subj, cfg, n=n, f=f, H=H, L=L, receivers=np.array([5, n - 99]))
    fe, fn, fo, fp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off)
    assert (fe >= 0).all() and np.diff(fo).min() > 2560
    for mode in (0, 3):
        emit, nprop, pcount, fpr, props, stats = pyemu.tally(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member, tables_in_lds=mode,
                                                            packed=True, trusted=True)
        assert np.array_equal(emit, fe) and np.array_equal(nprop, fn) and np.array_equal(pcount, np.diff(fo))
        for r in range(len(fe)):
            assert props[r, : pcount[r]].tolist() == fp[fo[r]: fo[r + 1]].tolist(), (mode, r)
        assert np.array_equal(fpr, proposal_fingerprints(fo, fp, fe >= 0))
        assert int(stats[2]) > 0  # (sweeps over the hot slots happened)
