"""Two PROCESSES, one GPU, real answer blocks (the exchange that replaces the vote fan-out of
R/FastPaxos.java:104 / R/UnicastToAllBroadcaster.java:46-52).

Each rank owns an engine on cuda:0, simulates its shard of the receivers (parallel.shard_range), and contributes the block
rapid_sim_count_votes would put into the all-gather (rapid_debug_vote_segment).  The blocks cross the process boundary
through torch.distributed (gloo) all_gather and BOTH ranks run the device merge (rapid_debug_vote_merge) over the gathered
data: decision, cut and the configuration id after applying it must be equal on both ranks and equal to what ONE engine
holding the whole population decides -- for a unanimous round, a round with dissenters under a quorum, and a round whose
ranks hold conflicting proposals (recognised as such by the merge on both ranks alike).  One GPU cannot run a two-rank RCCL
communicator; this is the product's answer blocks and merge kernel meeting across processes, with gloo standing in for the
transport only."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from rapid_amd import scenarios as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

COMMON = textwrap.dedent("""
    import numpy as np
    from rapid_amd import scenarios as S

    N, K, H, L = 2000, 10, 9, 4

    def shard(records, rec_off, a, b):
        return records[rec_off[a]: rec_off[b]], (rec_off[a: b + 1] - rec_off[a]).astype(np.int64)

    def rounds(obs, member, cfg):
        '''-> {name: (records, rec_off, declared alert set)} over the same view'''
        sc = S.build_churn_scenario(obs, member, cfg, 20, 0, H, L)
        sc3 = S.build_churn_scenario(obs, member, cfg, 20, 0, H, L, seed_fault=5)
        both = np.concatenate([sc.batches.recs, sc3.batches.recs])
        out = {"unanimous": (sc.records, sc.rec_off, sc.batches.recs)}
        # 1,700 receivers see this round's crashes, 60 (the LAST ones: the candidate is the lowest voter's proposal) a different
        # fault set: the common proposal has its quorum of 1,501 whatever the others hold
        ra, oa = shard(sc.records, sc.rec_off, 0, 1700)
        rb, ob = shard(sc3.records, sc3.rec_off, 0, 60)
        out["dissent"] = (np.concatenate([ra, rb]), np.concatenate([oa, ob[1:] + oa[-1]]), both)
        # 600 against 500: nobody has a quorum, and the two halves of the population hold different proposals
        ra, oa = shard(sc.records, sc.rec_off, 0, 600)
        rb, ob = shard(sc3.records, sc3.rec_off, 0, 500)
        out["conflict"] = (np.concatenate([ra, rb]), np.concatenate([oa, ob[1:] + oa[-1]]), both)
        return out
""")

WORKER = COMMON + textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %(root)r)
    import torch
    import torch.distributed as dist
    from rapid_amd import engine as E, parallel as P, _native as NATIVE
    NATIVE.use_test_build()   # (rapid_debug_vote_segment / _merge)
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
    pop = S.Population.make(N)
    eng = E.Engine(n_max=N, K=K, H=H, L=L, device_id=0)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    out = {}
    for name, (records, rec_off, declared) in rounds(obs, member, cfg).items():
        R = len(rec_off) - 1
        lo, hi = P.shard_range(R, rank, world)
        recs, off = shard(records, rec_off, lo, hi)
        sim = E.ClusterSimulation(eng)
        sim.load_streams(recs, off)
        sim.set_alert_set(declared)
        sim.tally()
        mine = torch.from_numpy(sim.vote_segment().copy())
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)                 # the one exchange of the round: the ranks' answer blocks
        status, rr = sim.merge_vote_segments([g.numpy() for g in gathered])
        res = {"status": status, "decided": int(rr.decided), "votes_total": int(rr.votes_total), "votes_winner": int(rr.votes_winner),
               "quorum": int(rr.quorum), "cut_size": int(rr.cut_size), "shard": [lo, hi]}
        if status == 1 and rr.decided:
            res["cut"] = sim.decided_cut()
        out[name] = res
    # the decided cut of the unanimous round, applied by this rank to its own replica of the view
    out["new_cfg"] = E.ClusterSimulation(eng).apply_cut(out["unanimous"]["cut"])
    out["cfg"] = cfg
    print("RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_processes_exchange_their_answer_blocks_and_merge_alike():
    from rapid_amd import engine as E
    if E.device_count() < 1:
        pytest.fail("no gfx950 device visible: the product has no CPU fallback")
    # ---- what ONE engine holding the whole population decides ----
    ns = {}
    exec(COMMON, ns)
    N, K, H, L = ns["N"], ns["K"], ns["H"], ns["L"]
    pop = S.Population.make(N)
    eng = E.Engine(n_max=N, K=K, H=H, L=L)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    want = {}
    for name, (records, rec_off, declared) in ns["rounds"](obs, member, cfg).items():
        sim = E.ClusterSimulation(eng)
        sim.load_streams(records, rec_off)
        sim.set_alert_set(declared)
        sim.tally()
        rr = sim.count_votes()
        want[name] = {"decided": int(rr.decided), "votes_total": int(rr.votes_total), "votes_winner": int(rr.votes_winner),
                      "quorum": int(rr.quorum), "cut": sim.decided_cut() if rr.decided else None, "R": len(rec_off) - 1}
    assert want["unanimous"]["decided"] == 1 and want["dissent"]["decided"] == 1 and want["conflict"]["decided"] == 0
    assert want["dissent"]["votes_winner"] == 1700 and want["dissent"]["votes_total"] == 1760
    want_cfg = E.ClusterSimulation(eng).apply_cut(want["unanimous"]["cut"])
    eng.close()

    # ---- two ranks ----
    port = free_port()
    src = WORKER % {"root": ROOT, "port": port}
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, "-c", src, str(r), "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
             for r in range(2)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-3000:]
        line = [ln for ln in so.splitlines() if ln.startswith("RESULT ")]
        assert line, so[-2000:] + se[-2000:]
        outs.append(json.loads(line[-1][len("RESULT "):]))
    a, b = outs
    assert a["cfg"] == b["cfg"] == cfg
    for name in ("unanimous", "dissent", "conflict"):
        ra, rb = dict(a[name]), dict(b[name])
        sa, sb = ra.pop("shard"), rb.pop("shard")
        assert sa[1] == sb[0] and sa[0] == 0 and sb[1] == want[name]["R"]   # the shards partition the receivers
        assert ra == rb, (name, ra, rb)                                     # both ranks hold the same answer
    for name in ("unanimous", "dissent"):
        assert a[name]["status"] == 1 and a[name]["decided"] == 1
        assert a[name]["cut"] == want[name]["cut"]
        assert a[name]["quorum"] == want[name]["quorum"] and a[name]["votes_winner"] == want[name]["votes_winner"]
    assert a["unanimous"]["votes_total"] == want["unanimous"]["votes_total"]
    # the merge counts the voters of every rank (dissenters included) the way the single engine does
    assert a["dissent"]["votes_total"] == want["dissent"]["votes_total"]
    # conflicting proposals: the merge says so on both ranks (the general histogram count is what would run then)
    assert a["conflict"]["status"] == 2 and b["conflict"]["status"] == 2
    # ... and each rank, applying the decided cut to its replica of the view, arrives at the same next configuration
    assert a["new_cfg"] == b["new_cfg"] == want_cfg
