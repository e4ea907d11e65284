"""N>1 path on CPU: world_size-2 `gloo` processes shard the receivers of one simulated cluster, build their local
vote histograms (from oracle-computed proposals -- the HIP tally needs a GPU), all-reduce them and must reach the
same decision as the single-process run.  Covers rapid_amd.parallel (the host-side statement of what
librapid_mi355x.so does with RCCL)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from rapid_amd import parallel as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in (0, 1, 7, 9487, 99000):
        for world in (1, 2, 3, 8):
            spans = [P.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        P.shard_range(10, 2, 2)


def test_fast_quorum_table():
    """FastPaxosWithoutFallbackTests.java:85-90."""
    for n, q in [(6, 5), (48, 37), (50, 38), (100, 76), (102, 77), (5, 4), (51, 39), (49, 37), (99, 75), (101, 76)]:
        assert P.fast_quorum(n) == q


WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %(root)r)
    import torch.distributed as dist
    from oracle import pyoracle as O
    from rapid_amd import parallel as P, scenarios as S
    from tests.helpers import oracle_view
    dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=int(sys.argv[2]))
    rank, world = dist.get_rank(), dist.get_world_size()
    n, K, H, L = 400, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario("C2", subj, cfg, n=n, f=10, H=H, L=L, materialise=False)
    lo, hi = P.shard_range(len(sc.receivers), rank, world)
    records, rec_off, nb = S.deliver(sc.batches, sc.receivers[lo:hi], 2)
    e, npr, off, props = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, records, rec_off)
    # fingerprint stand-in: any function of the proposal set that is equal for equal sets
    fps = [hash(tuple(props[off[r]:off[r + 1]].tolist())) & ((1 << 64) - 1) if e[r] >= 0 else 0 for r in range(len(e))]
    hist = P.all_reduce_histogram(P.local_histogram(fps, np.diff(off)), dist)
    b, votes, total, ok = P.decide_from_histogram(hist, n)
    # the one-collective form of the same round: candidates all-gathered and merged by quorum
    settled, cvotes, cvoters, cut = P.count_votes_sharded(n, e, np.array(fps, dtype=np.uint64), lambda r: props[off[r]:off[r + 1]].tolist(), dist)
    # ... and of a round with a few dissenters at the END of the last rank's receivers (they must not stop the quorum) and at
    # the FRONT of the first rank's (they become its candidate: the merge hands over to the general count)
    def with_dissenters(front):
        e2, f2 = e.copy(), np.array(fps, dtype=np.uint64)
        mine = (rank == 0) if front else (rank == world - 1)
        idx = list(range(3)) if front else list(range(len(e2) - 3, len(e2)))
        if mine:
            for i in idx:
                e2[i], f2[i] = 0, np.uint64(12345)
        prop = lambda r: [0, 1, 2] if (mine and r in idx) else props[off[r]:off[r + 1]].tolist()
        return P.count_votes_sharded(n, e2, f2, prop, dist)
    tail, front = with_dissenters(False), with_dissenters(True)
    print(json.dumps({"rank": rank, "lo": lo, "hi": hi, "bucket": b, "votes": votes, "total": total, "decided": bool(ok),
                      "merged": [bool(settled), cvotes, cvoters, cut], "tail": [bool(tail[0]), tail[1], tail[2], tail[3]],
                      "front": [bool(front[0]), front[1], front[2]]}))
    dist.destroy_process_group()
""")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world):
    import json
    code = WORKER % {"root": ROOT, "port": free_port()}
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(world)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, cwd=ROOT) for r in range(world)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    return outs


def test_two_rank_gloo_decision_equals_single_rank():
    one = run_world(1)[0]
    two = run_world(2)
    assert one["decided"] and one["votes"] >= P.fast_quorum(400)
    assert two[0]["hi"] == two[1]["lo"] and two[0]["lo"] == 0
    for t in two:
        assert (t["bucket"], t["votes"], t["total"], t["decided"]) == (one["bucket"], one["votes"], one["total"], one["decided"])
        # the one-collective merge says the same, on every rank, and hands out the decided cut itself
        assert t["merged"] == one["merged"] and t["merged"][:3] == [True, one["votes"], one["total"]] and len(t["merged"][3]) == 10
        # three dissenting voters behind the others: still settled, their votes only count as voters
        assert t["tail"] == one["tail"] and t["tail"][0] and t["tail"][3] == t["merged"][3]
        assert t["tail"][2] == one["total"] and t["tail"][1] >= P.fast_quorum(400)
        # dissenters that are a rank's lowest voters make the ranks' candidates differ: not merged
        assert t["front"][0] is False
    assert one["front"][0] is False  # (a single rank whose lowest voters dissent: three votes, no quorum, not unanimous)


RECOVERY_WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %(root)r)
    import torch.distributed as dist
    from oracle import pyoracle as O
    from rapid_amd import parallel as P, scenarios as S
    from tests.helpers import oracle_view
    dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=int(sys.argv[2]))
    rank, world = dist.get_rank(), dist.get_world_size()
    n, K, H, L = 400, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    # one-way failures without closing the fault set: healthy subjects stuck between L and H make the proposals differ
    sc = S.build_scenario("C3a", subj, cfg, n=n, f=20, H=H, L=L, kind="ingress", materialise=False)
    lo, hi = P.shard_range(len(sc.receivers), rank, world)
    records, rec_off, nb = S.deliver(sc.batches, sc.receivers[lo:hi], 2, loss=0.02)
    e, npr, off, props = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, records, rec_off)
    fps = np.array([hash(tuple(props[off[r]:off[r + 1]].tolist())) & ((1 << 64) - 1) if e[r] >= 0 else 0 for r in range(len(e))],
                   dtype=np.uint64)
    hist = P.all_reduce_histogram(P.local_histogram(fps, np.diff(off)), dist)
    b, votes, total, ok = P.decide_from_histogram(hist, n)
    arrival = np.random.default_rng(5).permutation(len(sc.receivers))
    res, cut = P.classic_round_sharded(n, e, fps, lambda r: props[off[r]:off[r + 1]].tolist(), dist, arrival)
    print(json.dumps({"rank": rank, "fast": bool(ok), "votes": votes, "total": total, "res": res, "cut": cut}))
    dist.destroy_process_group()
""")


def run_recovery(world):
    import json
    code = RECOVERY_WORKER % {"root": ROOT, "port": free_port()}
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(world)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, cwd=ROOT) for r in range(world)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    return outs


def test_two_rank_gloo_classic_round_equals_single_rank():
    """No fast quorum -> the recovery round over the sharded receivers decides the cut the single-process run decides."""
    one = run_recovery(1)[0]
    two = run_recovery(2)
    assert not one["fast"] and one["votes"] < P.fast_quorum(400) and one["total"] > 200  # a majority voted, no fast quorum
    assert one["res"]["decided"] and one["cut"]
    for t in two:
        assert (t["fast"], t["votes"], t["total"], t["res"], t["cut"]) == (one["fast"], one["votes"], one["total"], one["res"], one["cut"])
