"""The error contract at the drop-in boundary, on the device (-m gpu): what a host must get back as an error CODE where a
kernel reading the argument would otherwise fault -- and a device fault ends the process, in a deployment the JVM
(INTEGRATION.md section 6).  The reference's own contract here is exceptions, never a crash: R/MembershipView.java:502-519,
R/MultiNodeCutDetector.java:52-55."""
import ctypes as C

import numpy as np
import pytest

from rapid_amd import scenarios as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from rapid_amd import engine
    if engine.device_count() < 1:
        pytest.fail("no gfx950 device visible: the product has no CPU fallback")
    return engine


@pytest.fixture(scope="module")
def hip():
    h = C.CDLL("libamdhip64.so")
    h.hipMalloc.argtypes, h.hipMemcpy.argtypes, h.hipFree.argtypes = [C.POINTER(C.c_void_p), C.c_size_t], [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int], [C.c_void_p]
    h.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    h.hipHostFree.argtypes = [C.c_void_p]
    return h


def to_device(hip, a, slack=0):
    a = np.ascontiguousarray(a)
    ptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(ptr), max(a.nbytes + slack, 16)) == 0
    assert hip.hipMemcpy(ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
    return ptr


def test_borrowed_device_buffers_are_checked_before_a_kernel_reads_them(E, hip):
    n, K, H, L = 300, 10, 9, 4
    pop = S.Population.make(n)
    eng = E.Engine(n_max=n, K=K, H=H, L=L)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    obs, subj, member = view.tables()
    sc = S.build_scenario("C2", subj, view.getCurrentConfigurationId(), n=n, f=5, H=H, L=L)
    sim = E.ClusterSimulation(eng)
    sim.load_streams(sc.records, sc.rec_off)
    sim.tally()
    want = [a.copy() for a in sim.results()]
    raw = np.ascontiguousarray(sc.records).view(np.uint8).reshape(-1)
    off = np.ascontiguousarray(sc.rec_off, dtype=np.int64)
    R = len(off) - 1
    d_rec, d_off = to_device(hip, raw), to_device(hip, off)
    # ordinary host memory where a device pointer is required: an error code, not a fault in the tally kernel
    for fn in (sim.attach_streams_device, sim.load_streams_device):
        with pytest.raises(E.IllegalArgumentException, match="host memory|not memory"):
            fn(raw.ctypes.data, raw.nbytes, d_off.value, R)
        with pytest.raises(E.IllegalArgumentException, match="host memory|not memory"):
            fn(d_rec.value, raw.nbytes, off.ctypes.data, R)
        # a length that runs past the allocation (the tally would read the neighbour's memory, or an unmapped page)
        with pytest.raises(E.IllegalArgumentException, match="run past the end"):
            fn(d_rec.value, raw.nbytes + (64 << 20), d_off.value, R)
        # fewer offsets in the buffer than receivers + 1
        with pytest.raises(E.IllegalArgumentException, match="run past the end"):
            fn(d_rec.value, raw.nbytes, d_off.value, R + (8 << 20))
    # the refused calls changed nothing: the loaded streams are still the loaded streams
    sim.new_round()
    sim.tally()
    assert all(np.array_equal(a, b) for a, b in zip(want, sim.results()))
    al = np.ascontiguousarray(sc.batches.recs).view(np.uint8).reshape(-1)
    d_al = to_device(hip, al)
    sim.attach_streams_device(d_rec.value, raw.nbytes, d_off.value, R)
    with pytest.raises(E.IllegalArgumentException, match="host memory|not memory"):
        sim.set_alert_set_device(al.ctypes.data, len(al) // 20)
    with pytest.raises(E.IllegalArgumentException):
        sim.set_alert_set_device(d_al.value, len(al) // 20 + (4 << 20), alerts_bytes=20 * (len(al) // 20 + (4 << 20)))
    # host-mapped pinned memory IS device-readable: accepted (a producer may leave a small round there)
    hp = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(hp), max(raw.nbytes, 4096), 0x2 | 0x40000000) == 0  # hipHostMallocMapped | hipHostMallocCoherent
    C.memmove(hp, raw.ctypes.data, raw.nbytes)
    sim.attach_streams_device(hp.value, raw.nbytes, d_off.value, R)
    sim.tally()
    assert all(np.array_equal(a, b) for a, b in zip(want, sim.results()))
    sim.attach_streams_device(d_rec.value, raw.nbytes, d_off.value, R)
    sim.tally()
    assert all(np.array_equal(a, b) for a, b in zip(want, sim.results()))
    eng.close()
    assert hip.hipHostFree(hp) == 0
    for p in (d_rec, d_off, d_al):
        assert hip.hipFree(p) == 0


def test_self_test_leaves_the_engine_alone(E):
    """rapid_engine_self_test runs on a private engine: the caller's view, loaded streams and results are what they were."""
    n, K, H, L = 200, 10, 9, 4
    pop = S.Population.make(n)
    eng = E.Engine(n_max=n, K=K, H=H, L=L)
    eng.self_test()  # before a view exists
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    cfg = view.getCurrentConfigurationId()
    obs, subj, member = view.tables()
    sc = S.build_scenario("C2", subj, cfg, n=n, f=4, H=H, L=L)
    sim = E.ClusterSimulation(eng)
    sim.load_streams(sc.records, sc.rec_off)
    sim.tally()
    want = [a.copy() for a in sim.results()]
    eng.self_test()
    assert view.getCurrentConfigurationId() == cfg
    assert all(np.array_equal(a, b) for a, b in zip(want, sim.results()))
    rr, new_cfg = sim.round(apply=True)
    assert rr.decided == 1 and sorted(sim.decided_cut()) == sc.faulty.tolist()
    eng.self_test()
    assert view.getCurrentConfigurationId() == new_cfg


def test_view_errors_name_the_call_and_leave_the_view(E):
    """RAPID_EINVAL paths of rapid_view_build leave a built view as it was (argument checks come before any device work)."""
    n, K = 40, 5
    pop = S.Population.make(n)
    eng = E.Engine(n_max=n, K=K, H=4, L=2)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    cfg = view.getCurrentConfigurationId()
    with pytest.raises(E.IllegalArgumentException):
        E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo, members=[0, 1, n + 3])
    big = S.Population.make(n + 1)
    with pytest.raises(E.IllegalArgumentException):
        E.MembershipView(eng).build(big.hostnames, big.ports, big.id_hi, big.id_lo)  # over n_max
    assert view.getCurrentConfigurationId() == cfg and view.getMembershipSize() == n


def test_fast_round_settled_inside_the_tally_equals_the_separate_count(E):
    """The fast round settled by the tally launch itself (tally_kernel.h: TallyParams::vote_cand; opt-in, knob bit 23) against the
    default -- vote statistics from the tally, counting / verification in launches of their own -- and against the general
    count (knob 512), on the three kinds of round: unanimous, dissenters under a quorum, conflicting proposals without one
    (R/FastPaxos.java:125-156).  Same decision, same cut, same votes; and again on the same engine, round after round (the
    candidate's words and the deferred list are left zeroed by every launch)."""
    ns = {}
    from tests.test_gpu_two_ranks import COMMON
    exec(COMMON, ns)
    N_, K, H, L = ns["N"], ns["K"], ns["H"], ns["L"]
    pop = S.Population.make(N_)
    eng = E.Engine(n_max=N_, K=K, H=H, L=L)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sim = E.ClusterSimulation(eng)
    for rep in range(2):
        for name, (records, rec_off, declared) in ns["rounds"](obs, member, cfg).items():
            got = {}
            for knob in (8388608, 0, 512):
                sim.set_force_exact(knob)
                sim.load_streams(records, rec_off)
                sim.set_alert_set(declared)
                sim.tally()
                rr = sim.count_votes()
                got[knob] = (int(rr.decided), int(rr.votes_total), int(rr.quorum), int(rr.cut_size) if rr.decided else -1,
                             int(rr.votes_winner) if rr.decided else -1, sim.decided_cut() if rr.decided else None)
            sim.set_force_exact(0)
            assert got[0] == got[8388608] == got[512], (name, got)
            assert got[0][0] == (0 if name == "conflict" else 1), (name, got[0])
    # nobody votes: an answer all the same (no candidate), round after round
    quiet = np.zeros(0, dtype=S.ALERT_DTYPE)
    sim.load_streams(quiet, np.zeros(11, dtype=np.int64))
    sim.tally()
    rr = sim.count_votes()
    assert rr.decided == 0 and rr.votes_total == 0


def test_a_round_in_one_call_equals_the_five_calls(E, hip):
    """rapid_sim_round_device = attach + declare + trust + round (include/rapid_mi355x.h): same decision, same cut, same configuration
    id after applying it as the call sequence it stands for; a refused argument refuses the whole call and leaves the loaded round."""
    n, K, H, L = 1200, 10, 9, 4
    pop = S.Population.make(n)
    cuts, cfgs = [], []
    for one_call in (False, True):
        eng = E.Engine(n_max=n, K=K, H=H, L=L)
        view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
        obs, subj, member = view.tables()
        cfg = view.getCurrentConfigurationId()
        sc = S.build_scenario("C2", subj, cfg, n=n, f=12, H=H, L=L)
        raw = np.ascontiguousarray(sc.records).view(np.uint8).reshape(-1)
        off = np.ascontiguousarray(sc.rec_off, dtype=np.int64)
        al = np.ascontiguousarray(sc.batches.recs).view(np.uint8).reshape(-1)
        d_rec, d_off, d_al = to_device(hip, raw), to_device(hip, off), to_device(hip, al)
        sim = E.ClusterSimulation(eng)
        R = len(off) - 1
        if one_call:
            with pytest.raises(E.IllegalArgumentException):  # host memory for the alerts: nothing of the call happens
                sim.round_device(d_rec.value, raw.nbytes, d_off.value, R, al.ctypes.data, len(al) // 20, trust=1, apply=True)
            with pytest.raises(E.IllegalArgumentException):
                sim.round_device(d_rec.value, raw.nbytes, d_off.value, R, d_al.value, len(al) // 20, trust=3, apply=True)
            assert view.getCurrentConfigurationId() == cfg
            rr, new_cfg = sim.round_device(d_rec.value, raw.nbytes, d_off.value, R, d_al.value, len(al) // 20, trust=1, apply=True)
        else:
            sim.attach_streams_device(d_rec.value, raw.nbytes, d_off.value, R)
            sim.set_alert_set_device(d_al.value, len(al) // 20, trust_copies=1)
            rr, new_cfg = sim.round(apply=True)
        assert rr.decided == 1 and sorted(sim.decided_cut()) == sc.faulty.tolist()
        assert view.getCurrentConfigurationId() == new_cfg != cfg
        cuts.append(sim.decided_cut())
        cfgs.append((new_cfg, int(rr.votes_winner), int(rr.quorum)))
        # nothing declared (d_alerts NULL, n_alerts 0): the round scans the delivered records themselves -- in the NEW view these
        # deliveries are of an earlier configuration: every one dropped, nobody proposes
        rr2, _ = sim.round_device(d_rec.value, raw.nbytes, d_off.value, R, None, 0, trust=0, apply=False)
        assert rr2.decided == 0 and rr2.votes_total == 0
        eng.close()
        for p in (d_rec, d_off, d_al):
            assert hip.hipFree(p) == 0
    assert cuts[0] == cuts[1] and cfgs[0] == cfgs[1]


def test_bench_line_of_a_small_configuration_in_the_driver_s_process_form(E):
    """`python bench.py` as the driver starts it (a process of its own, one JSON line on stdout), on the smallest configuration: the
    contract's keys, the roofline and cpu_baseline objects, the receive buffers chosen among the candidates (roofline.placement) --
    and the same without the choice.  The line checks itself against the oracle (a mismatch is a non-zero exit)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--config", "C2", "--steps", "4", "--warmup", "2", "--no-pmc", "--reps", "1"]
    for extra, chosen in ((["--placement-candidates", "2"], True), (["--placement-candidates", "0", "--no-cpu-baseline", "--no-extras"], False)):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                    "data", "config", "roofline"):
            assert key in line, key
        assert line["metric"] == "alert-batches/sec" and line["n_gpus"] == 1 and line["steps"] == 4 and line["value"] > 0
        roof = line["roofline"]
        assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0.0 < roof["frac"] < 1.0 and len(roof["kernel_ms_by_stream_set"]) == 2
        if chosen:
            assert len(roof["placement"]["candidates_ms"]) == 4 and len(roof["placement"]["kept"]) == 2
            assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
            assert line["parity_checked"]["mismatches"] == 0
        else:
            assert roof["placement"] is None
