"""N > 1 ranks over a real RCCL communicator, one process per GPU -- skipped unless at least two gfx950 devices are
visible.  (tests/test_gpu_two_ranks.py covers what ONE GPU can: two processes exchanging the real answer blocks over gloo.)

1. The three kinds of round of test_gpu_two_ranks.py -- unanimous, dissenters under a quorum, conflicting proposals without a
   quorum -- through rapid_sim_count_votes itself: every rank settles its own voters, ONE ncclAllGather moves the answer blocks,
   every rank merges (the conflict falls back to the all-reduce of the vote histogram): decision, cut, votes and the
   configuration id after applying the cut are equal on both ranks and equal to ONE engine holding the whole population.
2. the plain `python bench.py --gpus 2` (it launches its two ranks itself), strong (C3b at a reduced N) and weak (`--config C4`: shards 0 and 1 of the
   eight): one JSON line, n_ranks_seen == 2, the contract's fields present."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from rapid_amd import scenarios as S
from tests.test_gpu_two_ranks import COMMON, ROOT, free_port

pytestmark = pytest.mark.gpu


def _two_devices():
    from rapid_amd import engine as E
    return E.device_count() >= 2


WORKER = COMMON + textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %(root)r)
    import torch
    import torch.distributed as dist
    from rapid_amd import engine as E, parallel as P
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
    pop = S.Population.make(N)
    eng = E.Engine(n_max=N, K=K, H=H, L=L, device_id=rank)
    uid = [E.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    eng.comm_init(uid[0], rank, world)   # the RCCL communicator inside the library: one rank per GPU
    assert eng.comm_info() == (rank, world)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    out = {}
    for name, (records, rec_off, declared) in rounds(obs, member, cfg).items():
        R = len(rec_off) - 1
        lo, hi = P.shard_range(R, rank, world)
        recs, off = shard(records, rec_off, lo, hi)
        for knob in (0, 512):   # 512: always the general count (histogram all-reduce + the three small reductions)
            sim = E.ClusterSimulation(eng)
            sim.set_force_exact(knob)
            sim.load_streams(recs, off)
            sim.set_alert_set(declared)
            sim.tally()
            rr = sim.count_votes()   # the collective(s) of the round run in here
            res = {"decided": int(rr.decided), "votes_total": int(rr.votes_total), "votes_winner": int(rr.votes_winner),
                   "quorum": int(rr.quorum), "cut_size": int(rr.cut_size), "shard": [lo, hi]}
            if rr.decided:
                res["cut"] = sim.decided_cut()
            out["%%s/%%d" %% (name, knob)] = res
        sim.set_force_exact(0)
    out["new_cfg"] = E.ClusterSimulation(eng).apply_cut(out["unanimous/0"]["cut"])
    out["cfg"] = cfg
    print("RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.skipif(not _two_devices(), reason="needs two gfx950 devices")
def test_vote_count_over_a_two_rank_rccl_communicator():
    from rapid_amd import engine as E
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
    want_cfg = E.ClusterSimulation(eng).apply_cut(want["unanimous"]["cut"])
    eng.close()

    src = WORKER % {"root": ROOT, "port": free_port()}
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, "-c", src, str(r), "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
             for r in range(2)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, se[-3000:]
        line = [ln for ln in so.splitlines() if ln.startswith("RESULT ")]
        assert line, so[-2000:] + se[-2000:]
        outs.append(json.loads(line[-1][len("RESULT "):]))
    a, b = outs
    assert a["cfg"] == b["cfg"] == cfg
    for name in ("unanimous", "dissent", "conflict"):
        for knob in (0, 512):
            ra, rb = dict(a["%s/%d" % (name, knob)]), dict(b["%s/%d" % (name, knob)])
            sa, sb = ra.pop("shard"), rb.pop("shard")
            assert sa[1] == sb[0] and sa[0] == 0 and sb[1] == want[name]["R"]
            assert ra == rb, (name, knob, ra, rb)  # both ranks hold the same answer
            w = want[name]
            assert ra["decided"] == w["decided"] and ra["quorum"] == w["quorum"], (name, knob)
            assert ra["votes_total"] == w["votes_total"], (name, knob)
            if w["decided"]:
                assert ra["cut"] == w["cut"] and ra["votes_winner"] == w["votes_winner"], (name, knob)
    assert a["new_cfg"] == b["new_cfg"] == want_cfg


@pytest.mark.skipif(not _two_devices(), reason="needs two gfx950 devices")
@pytest.mark.parametrize("extra", [["--n", "3000"], ["--config", "C4"]])
def test_bench_runs_on_two_ranks(extra):
    # the plain command: bench.py starts its own ranks (bench.self_launch: torch.distributed.run, one rank per GPU, dmabuf IPC)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pmc", "--no-extras"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert d["metric"] == "alert-batches/sec" and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    if extra[0] == "--config":
        assert d["scaling"] == "weak" and d["decided"] == 0  # two of the eight shards: no quorum of 75,001 votes yet
    else:
        assert d["scaling"] == "strong" and d["decided"] == 1 and d["time_to_stable_cut_ms"] > 0
