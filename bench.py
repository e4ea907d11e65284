#!/usr/bin/env python3
"""bench.py -- alert-batches/sec of the cut-detection hot path on MI355X (BASELINE.json metric, N=10k K=10).

One *step* = one FRESH ROUND of the hot path: a stream set the engine has not seen in the previous step -- the 20-byte alert
records of every receiver's deliveries, exactly as they cross the C ABI, resident in HBM -- is attached in place
(rapid_sim_attach_streams_device: no copy), the round's distinct alerts -- resident like the streams -- are declared in place
(rapid_sim_set_alert_set_device), and then everything a round costs runs: the per-round index (which subjects can reach the L watermark, slot dictionary, hot adjacency, validation of
the declared alerts against the view), the alert-tally kernel over every receiver (MembershipService.handleMessage(
BatchedAlertMessage) semantics) -- the ONLY pass that touches a delivered record: it compares the record's configuration id,
looks its subject up (tables in LDS) and tallies it on the record's way through the registers -- and the fast-round vote count
over the proposals (every rank counts and verifies its own voters; one all-gather + merge across ranks; quorum test) with the
decision read back to the host.  Nothing about a stream set survives a step: the steps alternate between `--stream-sets`
(default 2) resident sets with different delivery orders, 1.9 GB each at C3b -- far beyond the 256 MB of MALL, so every step
reads its records from HBM.  The view is NOT changed inside the timed loop so that every step does identical work; one extra
round that also applies the cut gives `time_to_stable_cut_ms` (attach + declare + index + tally + votes + apply cut + new
configuration id).

`roofline`: the tally kernel priced in SURVEY 8(d)'s unit -- 20 B per delivered record consumed, the record as it crosses the
boundary -- over the kernel's own duration (HIP events on the engine's stream, back-to-back launches); `traffic` is the
PMC-measured HBM traffic of the same launch and must agree.  `ms_per_step` is the mean the contract asks for;
`ms_per_step_min` / `_median` over the same steps are reported next to it.

Launch: `python bench.py --gpus N` (for N > 1 the process turns itself into the launcher of its N ranks: self_launch) or, as the
driver does, `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N`.
The simulated receivers are sharded across ranks (strong scaling of ONE cluster); the only data-path collective is
the per-round RCCL all-gather of the ranks' local vote counts inside librapid_mi355x.so (the histogram all-reduces of
the general count only run for a round whose voters disagree).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3b", help="C2 | C3a | C3b (headline: C3b = BASELINE configs[2], closed fault set) | C4 (BASELINE configs[3]: N = 100,000 in EIGHT "
                         "shards, rank r simulates shard r -- weak scaling: --gpus 1 measures one shard, --gpus 8 the whole cluster) | C5 "
                         "(BASELINE configs[4]: N = 1,000,000, continuous churn, eight shards like C4; a step = one whole tiled round of the shard)")
    ap.add_argument("--n", type=int, default=None, help="override population size (testing only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 FETCH_SIZE pass that measures roofline.traffic")
    ap.add_argument("--placement-candidates", type=int, default=3,
                    help="allocations tried per stream set for the receive buffers; the fastest are kept (0/1 = first come)")
    ap.add_argument("--stream-sets", type=int, default=2, help="resident stream sets (different delivery orders) the steps alternate between")
    ap.add_argument("--no-extras", action="store_true", help="skip the measurements beside the line (generator, per-delivery filter, probe)")
    ap.add_argument("--ttsc-trials", type=int, default=5, help="trials of the time-to-stable-cut measurement (the view is rebuilt between them)")
    ap.add_argument("--tile", type=int, default=0, help="C5: receivers per launch of the tiled round (default 4096)")
    ap.add_argument("--seed-fault", type=int, default=1, help="SURVEY 8(d): seed of the fault set")
    ap.add_argument("--seed-delivery", type=int, default=2, help="SURVEY 8(d): seed of the per-receiver delivery orders (stream set k uses seed + k)")
    ap.add_argument("--reps", type=int, default=5, help="SURVEY 8(d) 'Seeds': repetitions with seed_fault + rep, seed_delivery + rep (rep 0 = the line itself; "
                                                         "the others: one stream set, --rep-steps steps) + one run at H = 8, L = 2; 1 = none")
    ap.add_argument("--rep-steps", type=int, default=10)
    ap.add_argument("--parity-receivers", type=int, default=64, help="receivers per stream set checked against the oracle after the timed loop")
    ap.add_argument("--no-parity", action="store_true")
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch_command(gpus, argv, port):
    """The command line and environment `python3 bench.py --gpus N` turns itself into when nobody launched its ranks: one process
    per GPU under torch.distributed.run on this node, rendezvous on 127.0.0.1 (the container's hostname may not resolve).
    HSA_ENABLE_IPC_MODE_LEGACY=0: the ranks' RCCL communicator inside librapid_mi355x.so maps its peers' buffers through
    hipIpcGetMemHandle, and this platform's host driver only supports the dmabuf form of it -- with the legacy form the
    communicator's setup fails with "hipIpcGetMemHandle: invalid argument" (tests/test_gpu_multi.py sets the same)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    return cmd, env


def self_launch(args, execve=os.execve, device_count=None):
    """Replaces this process by the launcher of its ranks (the fan-out of the reference being replaced: one unicast per member,
    R/UnicastToAllBroadcaster.java:46-52, R/FastPaxos.java:104 -- here one rank per GPU).  Exits non-zero when the node does not
    have the devices asked for: a line from fewer ranks than --gpus would be a wrong line."""
    if device_count is None:
        from rapid_amd import engine as E
        device_count = E.device_count
    have = device_count()
    if have < args.gpus:
        raise SystemExit("bench.py: --gpus %d but %d gfx950 device(s) are visible" % (args.gpus, have))
    cmd, env = self_launch_command(args.gpus, sys.argv[1:], free_port())
    sys.stdout.flush()
    sys.stderr.flush()
    execve(cmd[0], cmd, env)
    raise SystemExit("bench.py: could not start %s" % " ".join(cmd))  # (execve returned: only the mocked one of the tests does)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            self_launch(args)  # `python3 bench.py --gpus N`: this process becomes the launcher of its N ranks (does not return)
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    # (every rank, however it was launched: RCCL's peer mappings need the dmabuf form of hipIpcGetMemHandle here -- see self_launch_command)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    from rapid_amd import engine as E
    from rapid_amd import parallel as P
    from rapid_amd import scenarios as S

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    cfgname = args.config
    if cfgname == "C5":
        return bench_c5(args, rank, world, local_rank, torch, dist, E, P, S)
    spec = dict(S.CONFIGS[cfgname])
    n = args.n or spec["n"]
    K, H, L = spec["K"], spec["H"], spec["L"]
    f = spec["f"] if args.n is None else max(1, spec["f"] * n // spec["n"])

    # ---- population + view (every rank builds the same deterministic view) ----
    t0 = time.time()
    pop = S.Population.make(n)
    eng = E.Engine(n_max=n, K=K, H=H, L=L, device_id=local_rank)
    if world > 1:
        uid = [E.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(uid[0], rank, world)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    obs, subj, member = view.tables()
    cfg_id = view.getCurrentConfigurationId()
    sc = S.build_scenario(cfgname, subj, cfg_id, n=n, f=f, materialise=False, seed_fault=args.seed_fault)
    # C3b (default): ONE cluster, its receivers cut into `world` shards (strong scaling).  C4: the cluster BASELINE runs on eight
    # GPUs, always cut into eight shards -- every rank carries the same load whatever the number of ranks (weak scaling)
    shards = 8 if cfgname == "C4" else world
    if world > shards:
        raise SystemExit("%s is defined on %d shards; --gpus %d" % (cfgname, shards, world))
    lo, hi = P.shard_range(len(sc.receivers), rank, shards)
    my_rx = sc.receivers[lo:hi]
    # `--stream-sets` delivered stream sets of the SAME round (same alert set, per-receiver delivery orders from different seeds),
    # resident in HBM as the 20-byte records of the boundary before anything is timed; set 0 (seed 2) is the workload of the
    # earlier rounds' benches and of tests/test_gpu_parity.py::test_full_size_c3_against_fast_oracle
    n_sets = max(1, args.stream_sets)
    sets = []
    host_sets = []
    parity_samples = []  # per stream set: (receiver indices, their delivered records, offsets) for the untimed check against the oracle
    for k in range(n_sets):
        recs_k, off_k, nb_k = S.deliver(sc.batches, my_rx, seed_delivery=args.seed_delivery + k)
        host_sets.append((recs_k.view(np.uint8).reshape(-1), np.ascontiguousarray(off_k, dtype=np.int64)))
        parity_samples.append(take_sample(recs_k, off_k, args.parity_receivers))
        if k == 0:
            records, rec_off, nb = recs_k, off_k, nb_k
        del recs_k
    # the round's distinct alerts, resident like the streams (a copy per stream set: a round's input is new in every part)
    alert_set = np.ascontiguousarray(sc.batches.recs)
    d_alert_sets = [torch.from_numpy(alert_set.view(np.uint8).reshape(-1).copy()).cuda() for _ in range(n_sets)]
    sim = E.ClusterSimulation(eng)
    # ---- where the receive buffers lie.  The tally streams a records buffer at a rate that depends on the PHYSICAL memory the
    # allocation got (0.379 .. 0.401 ms for the same bytes; one allocation mapped at six virtual addresses measures the same at all
    # six: profiles/r06_measurements.md section 5).  A host keeps its receive buffers for the life of the process, so it chooses them
    # once: `--placement-candidates` allocations per stream set are made, each is timed holding set 0's records, the fastest are kept
    # as the receive buffers and the rest are freed.  0 = take the first allocations as they come (the earlier rounds' bench).
    placement = None
    cand = max(0, args.placement_candidates)
    if cand > 1:
        nbytes = max(len(h[0]) for h in host_sets)
        pool = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(cand * n_sets)]
        h_rec0, h_off0 = host_sets[0]
        d_off0 = torch.from_numpy(h_off0).cuda()
        pool[0][:len(h_rec0)].copy_(torch.from_numpy(h_rec0))
        for b in pool[1:]:
            b[:len(h_rec0)].copy_(pool[0][:len(h_rec0)])
        torch.cuda.synchronize()
        cand_ms = []
        for b in pool:
            sim.attach_streams_device(b.data_ptr(), len(h_rec0), d_off0.data_ptr(), len(h_off0) - 1, keepalive=(pool, d_off0))
            sim.set_alert_set_device(d_alert_sets[0].data_ptr(), len(alert_set), trust_copies=True, keepalive=d_alert_sets)
            cand_ms.append(sim.time_tally(3))
        order = sorted(range(len(pool)), key=lambda j: cand_ms[j])
        placement = {"candidates_ms": [round(x, 4) for x in cand_ms], "kept": order[:n_sets],
                     "note": "receive buffers chosen once, before anything is timed, among %d allocations of %d MB each by the tally's "
                             "time over set 0's records in them; the others freed" % (len(pool), nbytes >> 20)}
        for k in range(n_sets):
            h_rec, h_off = host_sets[k]
            d_rec = pool[order[k]][:len(h_rec)]
            d_rec.copy_(torch.from_numpy(h_rec))
            sets.append((d_rec, torch.from_numpy(h_off).cuda(), len(h_off) - 1))
        del pool, b
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    else:
        for h_rec, h_off in host_sets:
            sets.append((torch.from_numpy(h_rec).cuda(), torch.from_numpy(h_off).cuda(), len(h_off) - 1))
    del host_sets
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    my_batches = int(nb.sum())
    my_records = int(len(records))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    def fresh_round(i):
        """A round's deliveries and its distinct alerts arrive: attached where they lie, declared (a new round: index rebuilt)."""
        d_rec, d_off, n_rx = sets[i % n_sets]
        sim.attach_streams_device(d_rec.data_ptr(), d_rec.numel(), d_off.data_ptr(), n_rx, keepalive=sets)
        sim.set_alert_set_device(d_alert_sets[i % n_sets].data_ptr(), len(alert_set), trust_copies=True, keepalive=d_alert_sets)

    def step(i):
        """One fresh round through the library's one-call form (rapid_sim_round_device = attach in place + declare in place +
        trust level 1 + index + tally + vote count; nothing applied): what fresh_round() + tally() + count_votes() do in five calls."""
        d_rec, d_off, n_rx = sets[i % n_sets]
        rr, _ = sim.round_device(d_rec.data_ptr(), d_rec.numel(), d_off.data_ptr(), n_rx, d_alert_sets[i % n_sets].data_ptr(),
                                 len(alert_set), trust=1, apply=False, keepalive=(sets, d_alert_sets))
        return rr  # (returned when the decision is on the host)

    for i in range(args.warmup):
        rr = step(i)
    barrier()
    per_step = []
    t1 = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        rr = step(args.warmup + i)
        per_step.append(time.perf_counter() - ts)
    barrier()
    elapsed = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([my_batches, my_records], dtype=torch.float64, device="cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_batches, tot_records = int(cnt[0].item()), int(cnt[1].item())
    else:
        tot_batches, tot_records = my_batches, my_records
    ms_per_step = 1e3 * elapsed / args.steps
    value = tot_batches * args.steps / elapsed

    # ---- dominant kernel: the alert tally.  HIP events on the engine's own stream, back-to-back launches over EVERY resident stream
    # set the timed steps alternated between -- the kernel's average launch duration over the timed region's launches.  (The sets hold
    # the same round in different delivery orders and lie in different allocations: the same kernel measures 0.385 or 0.400 ms by the
    # allocation the records happen to lie in, profiles/r06_measurements.md section 5 -- one set alone would be a draw of that.)
    sim.index_info()  # (asks for the device time of the NEXT index build: the rounds of the timed loop above carried no timing events)
    kern_by_set, consumed_by_set = [], []
    for k in range(n_sets - 1, -1, -1):  # (set 0 last: the measurements beside the line go on with it)
        fresh_round(k)
        kern_by_set.insert(0, sim.time_tally(args.kernel_reps))
        st = sim.stats()  # (the last pass, over set 0, is the one whose event counters the line carries)
        consumed_by_set.insert(0, st["records_consumed"] // (args.kernel_reps + 1))
    kern_ms = sum(kern_by_set) / len(kern_by_set)
    consumed = sum(consumed_by_set) // len(consumed_by_set)
    index = sim.index_info(timed=False)
    kern_filter_ms = kern_nolate_ms = step_filter_ms = None
    if not args.no_extras:
        # the same kernel with the per-delivery filter forced on (the instantiation that runs when the round's alerts do not
        # all validate against the view, or deliveries are not vouched for)
        sim.set_force_exact(64)
        kern_filter_ms = sim.time_tally(args.kernel_reps)
        # ... and whole steps in that form (untimed region of the line: the line's `value` is the vouched-for form's; this is what
        # it would be with every delivery filtered)
        for i in range(2):
            step(i)
        t_f = time.perf_counter()
        for i in range(max(5, min(args.steps, 20))):
            step(i)
        eng.sync()
        step_filter_ms = 1e3 * (time.perf_counter() - t_f) / max(5, min(args.steps, 20))
        sim.set_force_exact(0)
        fresh_round(0)
        # ... and with the caller's word that no late delivery is among the records (rapid_sim_trust_alert_copies level 2: true of
        # these streams): the configuration ids are not read at all.  NOT the line's kernel: level 1 rests on verified facts only.
        sim.set_alert_set_device(d_alert_sets[0].data_ptr(), len(alert_set), trust_copies=2, keepalive=d_alert_sets)
        kern_nolate_ms = sim.time_tally(args.kernel_reps)
        assert sim.index_info(timed=False)["configuration_ids_known_current"] == 1
        fresh_round(0)
    traffic, traffic_source = (None, "not measured at %d ranks" % world)
    if world == 1 and rank == 0 and not args.no_pmc:
        per_byte, traffic_source = measure_traffic(cfgname)
        traffic = int(per_byte * 20.0 * my_records) if per_byte else None
    # Accounting: SURVEY 8(d)'s unit -- 20 B per delivered record, the record as it crosses the boundary; every one of those
    # bytes is in a cache line the kernel pulls from HBM (src, 4 of the 20, is never loaded into a register), cross-checked by
    # the PMC traffic of the same launch.
    rec_b = 20.0
    achieved = rec_b * consumed / (kern_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "kernel": "tally_population_kernel<%s, %s, kFmtBoundary%s>" % ({0: "kDictMemory", 1: "kDictDirect", 2: "kDictCompressed"}.get(index["dict_mode"], "?"),
                                                                                 "trusted" if index["alerts_prevalidated"] else "filter",
                                                                                 ", ids known current" if index.get("configuration_ids_known_current") else ""),
                "kernel_ms": round(kern_ms, 4), "kernel_ms_by_stream_set": [round(x, 4) for x in kern_by_set],
                "bytes_per_launch": int(rec_b * consumed), "bytes_per_record": 20, "records_consumed_per_launch": int(consumed),
                "records_delivered_per_launch": my_records,
                "traffic_over_bytes": round(traffic / (rec_b * consumed), 3) if traffic else None,
                "kernel_ms_filter_per_delivery": round(kern_filter_ms, 4) if kern_filter_ms else None,
                "frac_filter_per_delivery": round(rec_b * consumed / (kern_filter_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kern_filter_ms else None,
                "ms_per_step_filter_per_delivery": round(step_filter_ms, 4) if kern_filter_ms else None,
                "value_filter_per_delivery": round(my_batches / (step_filter_ms * 1e-3), 1) if kern_filter_ms and world == 1 else None,
                "kernel_ms_no_late_deliveries_vouched": round(kern_nolate_ms, 4) if kern_nolate_ms else None,
                "passes_over_a_delivered_record": 1, "placement": placement}

    if world > 1:  # every rank's own tally kernel: duration and roofline fraction (the line's `roofline` is rank 0's)
        mine = torch.tensor([kern_ms, float(consumed)], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        roofline["per_rank"] = [{"rank": i, "kernel_ms": round(float(t[0]), 4), "records_consumed": int(t[1]),
                                 "frac": round(rec_b * float(t[1]) / (float(t[0]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for i, t in enumerate(allr)]

    # ---- the same round's deliveries GENERATED on the device (rapid_sim_generate): every receiver gets every batch in a seeded
    # permutation of its own, written directly as resolved 8-byte records (the 20-byte records never exist; the tally reads 8 B
    # per record and looks nothing up) or as the 20-byte boundary records.  (The delivery order is the generator's own, not the
    # numpy permutation of the loaded streams: same distribution, other streams; the round must decide the same cut.)
    gen = None
    if not args.no_extras:
        try:
            sim3 = E.ClusterSimulation(eng)
            out_g = {}
            for form, boundary in (("resolved", False), ("boundary", True)):
                ts_, dev_ = [], []
                for k in range(3):
                    t_ = time.perf_counter()
                    sim3.generate(sc.batches, my_rx, seed=2 + k, trust_copies=True, boundary=boundary)
                    eng.sync()
                    ts_.append(1e3 * (time.perf_counter() - t_))
                    dev_.append(sim3.index_info(timed=False)["generate_ms"])
                t_ = time.perf_counter()
                sim3.tally()
                rr_g = sim3.count_votes()
                round_ms = 1e3 * (time.perf_counter() - t_)
                kms = sim3.time_tally(5)
                out_g[form] = {"generate_ms": round(min(ts_), 4), "generate_device_ms": round(min(dev_), 4),
                               "records_per_s": round(my_records / (min(dev_) * 1e-3), 1),
                               "bytes_written": int(my_records * (20 if boundary else 8)),
                               "tally_kernel_ms": round(kms, 4), "round_after_generation_ms": round(round_ms, 4),
                               "decided": int(rr_g.decided), "cut_size": int(rr_g.cut_size), "votes_winner": int(rr_g.votes_winner)}
            gen = dict(out_g, records=my_records,
                       note="generate_ms = host wall time of rapid_sim_generate (upload of the round's distinct alerts, index, lay-down); "
                            "generate_device_ms = the lay-down kernels on the device; resolved: 8-byte records, tally<kDictResolved>; "
                            "boundary: the 20-byte records a load would have been handed")
            del sim3
        except Exception as e:  # (a measurement beside the line, not the line)
            sys.stderr.write("generator not timed: %s\n" % str(e)[:200])

    # ---- the PCIe-inclusive way in (never `value`): host records -> rapid_sim_load_streams (copy into the engine's buffer)
    load_host_ms = None
    if not args.no_extras and world == 1:
        try:
            sim4 = E.ClusterSimulation(eng)
            t_ = time.perf_counter()
            sim4.load_streams(records, rec_off)
            eng.sync()
            load_host_ms = 1e3 * (time.perf_counter() - t_)
            del sim4
        except Exception as e:
            sys.stderr.write("host load not timed: %s\n" % str(e)[:200])

    # ---- one full round including decideViewChange: time-to-stable-cut = a round's deliveries resident in HBM -> decided cut +
    # new configuration id on the host: attach + declare + per-round index + tally + vote count + apply cut (rings, tables,
    # configuration id)
    # (several trials: the view is built again between them -- untimed; the same membership, hence the same configuration id the
    # resident streams carry -- and the first one pays the first launch of every view-change kernel in the process)
    ttsc_trials = []
    for k in range(max(1, args.ttsc_trials)):
        if k > 0:
            view.build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
            assert view.getCurrentConfigurationId() == cfg_id
        barrier()
        t2 = time.perf_counter()
        fresh_round(1)
        rr_full, new_cfg = sim.round(apply=True)  # (returns with the decided cut and the new configuration id on the host)
        t_k = 1e3 * (time.perf_counter() - t2)
        eng.sync()
        if world > 1:
            tt = torch.tensor([t_k], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_k = float(tt.item())
        ttsc_trials.append(t_k)
    ttsc_ms = float(np.median(ttsc_trials))

    # ---- the line verifies itself (untimed): the decided cut is the scenario's fault set, and for EVERY stream set of the timed loop a
    # sample of receivers -- announce batch, getNumProposals(), proposal size, fingerprint of the proposal -- equals the CPU
    # restatement (oracle/fast_cut.hpp) fed the same delivered bytes.  A mismatch is a non-zero exit, not a line.
    parity = None
    if not args.no_parity:
        view.build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)  # (the last time-to-stable-cut trial applied its cut: back to the round's view)
        assert view.getCurrentConfigurationId() == cfg_id
        parity = {"receivers_per_stream_set": int(len(parity_samples[0][0])), "stream_sets": n_sets, "checker": "oracle/fast_cut.hpp",
                  "fields": ["emit_batch", "num_proposals", "proposal_size", "fingerprint"], "mismatches": 0}
        for k in range(n_sets):
            fresh_round(k)
            sim.tally()
            bad = parity_mismatches(sim.results(), parity_samples[k], pop.n, K, H, L, cfg_id, obs, subj, member)
            parity["mismatches"] += bad
        fresh_round(0)
        rr_chk = None
        if cfgname in ("C2", "C3b"):
            sim.tally()
            rr_chk = sim.count_votes()
            cut_ok = bool(rr_chk.decided) and sorted(sim.decided_cut()) == sc.faulty.tolist()
            if world == 1 or rr_chk.decided:
                parity["decided_cut_is_the_fault_set"] = cut_ok
                if not cut_ok:
                    parity["mismatches"] += 1
        if parity["mismatches"]:
            sys.stderr.write("bench.py: PARITY MISMATCH %r\n" % parity)
            raise SystemExit(3)

    # ---- SURVEY 8(d) "Seeds": repetitions with seed_fault + rep, seed_delivery + rep (rep 0 is the line itself), and the H = 8, L = 2
    # thresholds of the reference's own tests as one more -- each a fresh engine, one stream set, fewer steps, its own parity sample
    repetitions = None
    if world == 1 and args.reps > 1 and cfgname in ("C2", "C3a", "C3b") and not args.no_extras:
        runs = [{"rep": 0, "seed_fault": args.seed_fault, "seed_delivery": args.seed_delivery, "ms_per_step": round(ms_per_step, 4),
                 "time_to_stable_cut_ms": round(ttsc_ms, 3) if rr_full.decided else None, "decided": int(rr_full.decided), "cut_size": int(rr_full.cut_size)}]
        for rep in range(1, args.reps):
            runs.append(dict(light_run(E, S, torch, pop, K, H, L, cfgname, n, f, args.seed_fault + rep, args.seed_delivery + rep, local_rank, args), rep=rep))
        # (the crash form of the fault set: closing an ingress-loss set under "two faulty observers" swallows the cluster)
        h8 = dict(light_run(E, S, torch, pop, K, 8, 2, cfgname, n, f, args.seed_fault, args.seed_delivery, local_rank, args, kind="crash"),
                  scenario="the same %d nodes crash (every healthy observer reports them), thresholds H=8 L=2 of the reference's own tests" % f)
        ms = [r["ms_per_step"] for r in runs]
        tt = [r["time_to_stable_cut_ms"] for r in runs if r["time_to_stable_cut_ms"] is not None]
        repetitions = {"reps": len(runs), "ms_per_step": {"min": min(ms), "median": round(float(np.median(ms)), 4), "max": max(ms)},
                       "time_to_stable_cut_ms": {"min": min(tt), "median": round(float(np.median(tt)), 3), "max": max(tt)} if tt else None,
                       "runs": runs, "h8_l2": h8}
        if any(r.get("parity_mismatches") for r in runs + [h8]):
            sys.stderr.write("bench.py: PARITY MISMATCH in a repetition %r\n" % repetitions)
            raise SystemExit(3)

    out = {
        "metric": "alert-batches/sec", "value": round(value, 1), "unit": "alert-batches/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak" if cfgname == "C4" else "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%s: N=%d K=%d H=%d L=%d, %d ingress-loss nodes (5%% one-way failures, fault set closed "
                               "under >=L faulty observers), %d receivers, per-receiver seeded delivery order"
                               % (cfgname, n, K, H, L, len(sc.faulty), len(sc.receivers)) if cfgname == "C3b" else
                   "%s: N=%d K=%d H=%d L=%d faults=%d receivers=%d%s" % (cfgname, n, K, H, L, len(sc.faulty), len(sc.receivers),
                                                                            " (shards %d..%d of 8 simulated)" % (0, world - 1) if cfgname == "C4" else ""),
                   "parallelism": "receivers sharded over %d GPU(s); per round ONE all-gather (RCCL) of the ranks' local vote counts, merged on every rank" % world,
                   "step": "a FRESH round per step: one of %d resident stream sets (20-byte boundary records, %d MB each) attached in place, "
                           "alert set declared, index + tally (the one pass over the records) + vote count; nothing is kept "
                           "between steps%s" % (n_sets, int(my_records * 20 / 1e6),
                                              "; the receive buffers were chosen among %d allocations before timing (roofline.placement)"
                                              % (cand * n_sets) if placement else ""),
                   "alert_set": "the round's distinct alerts are declared and validated once per round against the view and the "
                                "deliveries are vouched for as copies (rapid_sim_trust_alert_copies); the configuration id of "
                                "every delivered record is still compared by the kernel; roofline.kernel_ms_filter_per_delivery "
                                "is the same kernel filtering every delivery",
                   "baseline_config": "BASELINE.json configs[3] (100,000 nodes, K=10, 1% crashes, 8 GPUs)" if cfgname == "C4" else
                                      "BASELINE.json configs[2] (10,000 nodes, K=10, 5% asymmetric one-way edge failures)"},
        "ms_per_step_min": round(1e3 * min(per_step), 4), "ms_per_step_median": round(1e3 * float(np.median(per_step)), 4),
        "n_ranks_seen": eng.comm_info()[1],
        "generated_streams": gen,
        "load_from_host_ms": round(load_host_ms, 3) if load_host_ms is not None else None,
        "alert_records_per_s": round(tot_records * args.steps / elapsed, 1),
        "time_to_stable_cut_ms": round(ttsc_ms, 3) if rr_full.decided else None,
        "time_to_stable_cut_trials_ms": [round(t, 3) for t in ttsc_trials],  # (the headline is their median; the first is the cold one)
        "decided": int(rr_full.decided), "cut_size": int(rr_full.cut_size), "votes_winner": int(rr_full.votes_winner),
        "quorum": int(rr_full.quorum), "kernel_stats": st, "round_index": index, "setup_s": round(setup_s, 1),
        "parity_checked": parity, "repetitions": repetitions,
        "roofline": roofline,
    }

    # ---- CPU baseline beside it (rank 0, N=1 only): the oracle = CPU restatement of the reference's Java path ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline and n <= 20000:  # (the faithful oracle is quadratic per batch: one C4 receiver takes minutes)
        out["cpu_baseline"], out["cpu_optimized"] = cpu_baseline(pop, K, H, L, cfg_id, obs, subj, member, records, rec_off,
                                                                 nb, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if out["n_ranks_seen"] != args.gpus:  # the communicator inside the library must span exactly the ranks asked for
        raise SystemExit("bench.py: the engine's communicator sees %d rank(s), --gpus %d" % (out["n_ranks_seen"], args.gpus))


def bench_c5(args, rank, world, local_rank, torch, dist, E, P, S):
    """BASELINE configs[4]: 1,000,000 members, K = 10, continuous churn (every round 1 % of the members crash, 0.5 % join, 1 % late
    deliveries of the previous configuration), 8 x MI355X.  The cluster is ALWAYS cut into eight shards of receivers and rank r
    simulates shard r (weak scaling, like C4): --gpus 1 measures one shard, --gpus 8 the whole cluster -- the fast quorum of
    750,001 votes exists only there.  A shard is ~123,000 receivers x ~150,000 deliveries = 1.8 x 10^10 delivered alerts per round
    (370 GB as 20-byte records): they never exist at once.  One STEP = one whole round of the shard through rapid_sim_round_tiled:
    per tile the deliveries are made on the device from the round's resident alert set (8-byte resolved records), tallied, and
    the fast-round votes accumulated across the tiles' launches; one all-gather across the ranks at the end.  The generator is
    INSIDE the timed step (a delivered record of this configuration exists nowhere else); the tally kernel alone is priced by
    `roofline` on one resident tile of 20-byte boundary records, as for the other configurations."""
    spec = dict(S.CONFIGS["C5"])
    n_mem = args.n or spec["n"]
    K, H, L = spec["K"], spec["H"], spec["L"]
    shards = 8
    if world > shards:
        raise SystemExit("C5 is defined on %d shards; --gpus %d" % (shards, world))
    t0 = time.time()
    pop = S.Population.make(n_mem + int(0.006 * n_mem * 3) + 64)
    eng = E.Engine(n_max=pop.n, K=K, H=H, L=L, device_id=local_rank, max_cut=max(4096, int(0.02 * n_mem)))
    if world > 1:
        uid = [E.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(uid[0], rank, world)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo, members=np.arange(n_mem, dtype=np.int32))
    sim = E.ClusterSimulation(eng)
    st = S.StreamingChurn(H, L)
    tile = args.tile if args.tile > 0 else 4096

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    def next_round():
        obs, subj, member = view.tables()
        cfg = view.getCurrentConfigurationId()
        sc, deliver_set = st.next_round_batches(obs, member, cfg)
        lo, hi = P.shard_range(len(sc.receivers), rank, shards)
        return sc, deliver_set, sc.receivers[lo:hi], cfg

    # round 0 (untimed): the configuration the timed round's late deliveries come from.  Its cut is applied as decided by the
    # round itself where all eight shards are present; a partial cluster applies the round's fault set (what the whole cluster
    # decides: the first round loses nothing).
    sc0, set0, rx0, cfg0 = next_round()
    rr0 = sim.round_tiled(set0, rx0, seed=1000, tile_receivers=tile)
    cut0 = sim.decided_cut() if rr0.decided else sc0.faulty.tolist()
    assert sorted(cut0) == sc0.faulty.tolist(), "round 0 decided something else than its fault set"
    sim.apply_cut(np.asarray(cut0, dtype=np.int32))
    sc, deliver_set, my_rx, cfg_id = next_round()
    setup_s = time.time() - t0
    A = len(deliver_set.recs)
    my_batches = len(my_rx) * deliver_set.n_batches
    my_records = len(my_rx) * A

    def step(i):
        return sim.round_tiled(deliver_set, my_rx, seed=2000, tile_receivers=tile)

    for i in range(args.warmup):
        rr = step(i)
    barrier()
    per_step = []
    t1 = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        rr = step(args.warmup + i)
        per_step.append(time.perf_counter() - ts)
    barrier()
    elapsed = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([my_batches, my_records], dtype=torch.float64, device="cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_batches, tot_records = int(cnt[0].item()), int(cnt[1].item())
    else:
        tot_batches, tot_records = my_batches, my_records
    ms_per_step = 1e3 * elapsed / args.steps
    value = tot_batches * args.steps / elapsed
    emit, nprop, pcount, fp = sim.results()
    proposing = int((emit >= 0).sum())
    distinct = int(len(np.unique(fp[emit >= 0])))
    tiled = sim.round_tiled_info()

    # ---- the tally kernel alone, on ONE resident tile (HIP events on the engine's stream, back-to-back launches): the 20-byte boundary
    # records (SURVEY 8d's unit: `roofline`) and the resolved 8-byte records the timed rounds run on
    tile_rx = my_rx[:min(len(my_rx), 1024)]
    forms = {}
    for form, boundary in (("boundary", True), ("resolved", False)):
        sim.generate(deliver_set, tile_rx, seed=2000, trust_copies=True, boundary=boundary)
        kms = sim.time_tally(args.kernel_reps)
        stt = sim.stats()
        consumed = stt["records_consumed"] // (args.kernel_reps + 1)
        info = sim.index_info(timed=False)
        rec_b = 20.0 if boundary else 8.0
        forms[form] = {"kernel_ms": round(kms, 4), "records_consumed_per_launch": int(consumed), "bytes_per_record": int(rec_b),
                       "records_per_s": round(consumed / (kms * 1e-3), 1), "achieved_GBps": round(rec_b * consumed / (kms * 1e-3) / 1e9, 1),
                       "frac": round(rec_b * consumed / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "generate_device_ms": info["generate_ms"],
                       "dict_mode": info["dict_mode"], "waves_per_workgroup": info["waves_per_workgroup"], "workgroups": info["workgroups"],
                       "hot_subjects": info["hot_subjects"], "alerts_prevalidated": info["alerts_prevalidated"]}
    b = forms["boundary"]
    traffic, traffic_source = (None, "not measured at %d ranks" % world)
    if world == 1 and rank == 0 and not args.no_pmc:
        per_byte, traffic_source = measure_traffic_c5()
        traffic = int(per_byte * 20.0 * b["records_consumed_per_launch"]) if per_byte else None
    roofline = {"bound": "hbm", "achieved": b["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b["frac"], "traffic": traffic,
                "traffic_source": traffic_source,
                "traffic_over_bytes": round(traffic / (20.0 * b["records_consumed_per_launch"]), 3) if traffic else None,
                "kernel": "tally_population_kernel<%s, %s, kFmtBoundary, packed>" % ({0: "kDictMemory", 4: "kDictHashed"}.get(b["dict_mode"], "?"),
                                                                                       "trusted" if b["alerts_prevalidated"] else "filter"),
                "kernel_ms": b["kernel_ms"], "bytes_per_launch": int(20 * b["records_consumed_per_launch"]), "bytes_per_record": 20,
                "records_consumed_per_launch": b["records_consumed_per_launch"], "tile_receivers": int(len(tile_rx)),
                "resolved_records": forms["resolved"]}
    if world > 1:
        mine = torch.tensor([b["kernel_ms"], float(b["records_consumed_per_launch"])], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        roofline["per_rank"] = [{"rank": i, "kernel_ms": round(float(t[0]), 4), "records_consumed": int(t[1]),
                                 "frac": round(20.0 * float(t[1]) / (float(t[0]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for i, t in enumerate(allr)]

    # ---- one round + its view change: time-to-stable-cut (only a whole cluster decides; a partial one applies the fault set)
    barrier()
    t2 = time.perf_counter()
    rr_full = sim.round_tiled(deliver_set, my_rx, seed=2000, tile_receivers=tile)
    cut = sim.decided_cut() if rr_full.decided else None
    t_round = 1e3 * (time.perf_counter() - t2)
    t3 = time.perf_counter()
    new_cfg = sim.apply_cut(np.asarray(cut if cut is not None else sc.faulty, dtype=np.int32))
    t_apply = 1e3 * (time.perf_counter() - t3)
    eng.sync()
    if cut is not None and sorted(cut) != sc.faulty.tolist():
        raise SystemExit("bench.py: the decided cut is not the round's fault set")

    # ---- beside the line, one GPU only: the WHOLE population on this GPU, shard after shard (the votes of all ~985,000 receivers
    # accumulated across ~240 launches): a decided round of BASELINE configs[4] without the other seven GPUs
    whole = None
    if world == 1 and not args.no_extras:
        # (the view has moved on by one cut: the whole-population round runs in the NEXT configuration, with a round of its own)
        sc2, set2, _, cfg2 = next_round()
        t4 = time.perf_counter()
        rr_w = sim.round_tiled(set2, sc2.receivers, seed=3000, tile_receivers=tile)
        t_w = 1e3 * (time.perf_counter() - t4)
        cut_w = sim.decided_cut() if rr_w.decided else []
        whole = {"receivers": int(len(sc2.receivers)), "records_delivered": int(len(sc2.receivers)) * int(len(set2.recs)), "round_ms": round(t_w, 1),
                 "records_per_s": round(len(sc2.receivers) * len(set2.recs) / (t_w * 1e-3), 1), "decided": int(rr_w.decided),
                 "votes_winner": int(rr_w.votes_winner), "quorum": int(rr_w.quorum), "cut_size": int(rr_w.cut_size),
                 "cut_is_the_fault_set": bool(sorted(cut_w) == sc2.faulty.tolist()), "tiles": sim.round_tiled_info()["tiles"]}
        if not (rr_w.decided and whole["cut_is_the_fault_set"]):
            raise SystemExit("bench.py: the whole-population round did not decide its fault set: %r" % whole)

    out = {
        "metric": "alert-batches/sec", "value": round(value, 1), "unit": "alert-batches/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic, deliveries made on the device inside the step",
        "config": {"workload": "C5: N=%d K=%d H=%d L=%d, round of continuous churn: %d crash + %d join, %d alerts in %d batches (+ %d late "
                               "batches of the previous configuration), %d receivers in the cluster (shards 0..%d of 8 simulated: %d receivers)"
                               % (n_mem, K, H, L, len(sc.crashed), len(sc.joiners), len(sc.batches.recs), sc.batches.n_batches,
                                  deliver_set.n_batches - sc.batches.n_batches, len(sc.receivers), world - 1, len(my_rx) if world == 1 else -1),
                   "parallelism": "receivers in eight shards, rank r = shard r; per round ONE all-gather (RCCL) of the ranks' accumulated vote counts",
                   "step": "one whole round of the shard, tile by tile (%d receivers per launch): deliveries generated on the device as 8-byte "
                           "resolved records, tallied, votes accumulated across the launches" % tile,
                   "baseline_config": "BASELINE.json configs[4] (1,000,000 nodes, K=10, streaming alert batches, continuous churn, 8 GPUs)"},
        "ms_per_step_min": round(1e3 * min(per_step), 3), "ms_per_step_median": round(1e3 * float(np.median(per_step)), 3),
        "n_ranks_seen": eng.comm_info()[1],
        "alert_records_per_s": round(tot_records * args.steps / elapsed, 1),
        "records_delivered_per_step": tot_records, "tiles_per_step": tiled["tiles"], "passes_per_step": tiled["passes"],
        "time_to_stable_cut_ms": round(t_round + t_apply, 3) if rr_full.decided else None,
        "round_ms": round(t_round, 3), "apply_cut_ms": round(t_apply, 3),
        "decided": int(rr_full.decided), "cut_size": int(rr_full.cut_size), "votes_winner": int(rr_full.votes_winner), "votes_total": int(rr_full.votes_total),
        "quorum": int(rr_full.quorum), "proposing": proposing, "distinct_proposals": distinct,
        "whole_population_on_one_gpu": whole, "setup_s": round(setup_s, 1), "roofline": roofline,
    }
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if out["n_ranks_seen"] != args.gpus:
        raise SystemExit("bench.py: the engine's communicator sees %d rank(s), --gpus %d" % (out["n_ranks_seen"], args.gpus))


def take_sample(records, rec_off, k):
    """k receivers spread evenly over a stream set: (their indices, their delivered records back to back, offsets)."""
    R = len(rec_off) - 1
    idx = np.unique(np.linspace(0, max(R - 1, 0), num=min(k, R), dtype=np.int64)) if R else np.zeros(0, dtype=np.int64)
    parts = [records[rec_off[r]:rec_off[r + 1]] for r in idx]
    off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    return idx, (np.concatenate(parts) if parts else records[:0]).copy(), off


def fingerprints_of(prop_off, props, emitted):
    """The kernel's proposal fingerprint (tally_kernel.h: sum of mix64(node) + mix64(0x5EED + count), 0 reserved) of the oracle's lists."""
    from rapid_amd.scenarios import mix64
    out = np.zeros(len(prop_off) - 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = mix64(np.asarray(props, dtype=np.uint64))
        csum = np.concatenate([[np.uint64(0)], np.cumsum(x, dtype=np.uint64)])
        for r in np.flatnonzero(emitted):
            cnt = int(prop_off[r + 1] - prop_off[r])
            v = (csum[prop_off[r + 1]] - csum[prop_off[r]]) + mix64(np.uint64(0x5EED + cnt))
            out[r] = v if v != 0 else np.uint64(1)
    return out


def parity_mismatches(results, sample, n_nodes, K, H, L, cfg_id, obs, subj, member):
    """Receivers of the sample whose device results differ from oracle/fast_cut.hpp fed the same delivered records (the checker,
    after the timed region: never the thing measured)."""
    from oracle import pyoracle as O
    idx, recs, off = sample
    if len(idx) == 0:
        return 0
    emit, nprop, pcount, fp = results
    fe, fn, fo, fpp = O.fast_sim_run(n_nodes, K, H, L, cfg_id, obs, subj, member, recs, off, nthreads=min(16, os.cpu_count() or 1))
    want_fp = fingerprints_of(fo, fpp, fe >= 0)
    bad = (emit[idx] != fe) | (nprop[idx] != fn) | (pcount[idx] != np.diff(fo)) | (fp[idx] != want_fp)
    return int(bad.sum())


def light_run(E, S, torch, pop, K, H, L, cfgname, n, f, seed_fault, seed_delivery, device_id, args, kind=None):
    """One repetition beside the line: a fresh engine, the scenario under other seeds (or thresholds), ONE resident stream set, the
    same fresh-round step, time-to-stable-cut, and the parity sample.  -> dict."""
    eng = E.Engine(n_max=n, K=K, H=H, L=L, device_id=device_id)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    obs, subj, member = view.tables()
    cfg_id = view.getCurrentConfigurationId()
    sc = S.build_scenario(cfgname, subj, cfg_id, n=n, f=f, H=H, L=L, materialise=False, seed_fault=seed_fault, kind=kind)
    recs, off, nb = S.deliver(sc.batches, sc.receivers, seed_delivery=seed_delivery)
    sample = take_sample(recs, off, args.parity_receivers)
    d_rec = torch.from_numpy(recs.view(np.uint8).reshape(-1)).cuda()
    d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.int64)).cuda()
    alert_set = np.ascontiguousarray(sc.batches.recs)
    d_al = torch.from_numpy(alert_set.view(np.uint8).reshape(-1).copy()).cuda()
    n_rec = len(recs)
    del recs
    torch.cuda.synchronize()
    sim = E.ClusterSimulation(eng)

    def fresh():
        sim.attach_streams_device(d_rec.data_ptr(), d_rec.numel(), d_off.data_ptr(), len(off) - 1, keepalive=(d_rec, d_off))
        sim.set_alert_set_device(d_al.data_ptr(), len(alert_set), trust_copies=True, keepalive=d_al)

    for _ in range(3):
        fresh()
        sim.tally()
        sim.count_votes()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(args.rep_steps):
        fresh()
        sim.tally()
        rr = sim.count_votes()
    eng.sync()
    ms = 1e3 * (time.perf_counter() - t0) / max(args.rep_steps, 1)
    fresh()
    sim.tally()
    bad = parity_mismatches(sim.results(), sample, pop.n, K, H, L, cfg_id, obs, subj, member) if not args.no_parity else 0
    trials = []
    for k in range(3):
        if k > 0:
            view.build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
        eng.sync()
        t1 = time.perf_counter()
        fresh()
        rr_full, new_cfg = sim.round(apply=True)
        trials.append(1e3 * (time.perf_counter() - t1))
        eng.sync()
    cut_ok = None
    if rr_full.decided:
        cut_ok = sorted(sim.decided_cut()) == sc.faulty.tolist()
        bad += 0 if cut_ok else 1
    out = {"seed_fault": seed_fault, "seed_delivery": seed_delivery, "H": H, "L": L, "ms_per_step": round(ms, 4),
           "alert_batches_per_s": round(float(nb.sum()) / (ms * 1e-3), 1), "records": int(n_rec), "faulty": int(len(sc.faulty)),
           "time_to_stable_cut_ms": round(float(np.median(trials)), 3) if rr_full.decided else None, "decided": int(rr_full.decided),
           "cut_size": int(rr_full.cut_size), "votes_winner": int(rr_full.votes_winner), "cut_is_the_fault_set": cut_ok,
           "proposing": int((sim.results()[0] >= 0).sum()) if False else None, "parity_mismatches": int(bad)}
    out.pop("proposing")
    del sim
    eng.close()
    del d_rec, d_off, d_al
    torch.cuda.empty_cache()
    return out


def measure_traffic_c5():
    """HBM bytes per delivered byte of the C5 tile tally (20-byte boundary records, packed detector state), measured in this run: a
    `rocprofv3 --pmc FETCH_SIZE` pass (its own process, counters only) over scripts/c5_probe.py -- one tile of 1,024 receivers at
    10^6 members.  FETCH_SIZE (KiB) under-reports wide coalesced reads on gfx950; the same pass holds its own calibration: the tally of
    the RESOLVED records of the same tile reads a known byte count (8 B per record) with the same load instructions.
    -> (HBM bytes per delivered byte, or None; how it was obtained)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            cmd = [exe, "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
                   os.path.join(ROOT, "scripts", "c5_probe.py"), "1000000", "1024", "2"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", RAPID_AB_ONLY="default"), capture_output=True, text=True, timeout=600)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, "rocprofv3 --pmc produced no counter file (rc %d)" % r.returncode
            boundary, resolved = [], []
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != "FETCH_SIZE" or "tally_population" not in row["Kernel_Name"]:
                    continue
                (resolved if "<3," in row["Kernel_Name"] else boundary).append(float(row["Counter_Value"]))
            if not boundary or not resolved:
                return None, "counter file without the two tally kernels"
            # the same tile, the same number n of records: the resolved tally reads 8 n known bytes, the boundary tally 20 n algorithmic
            # ones -- bytes fetched per delivered byte = (counter ratio) x 8 / 20
            return (sum(boundary) / len(boundary)) / (sum(resolved) / len(resolved)) * 8.0 / 20.0, \
                "rocprofv3 --pmc FETCH_SIZE in this run over one C5 tile (scripts/c5_probe.py), calibrated on the tally of the same tile's resolved 8-byte records"
    except Exception as e:  # a measurement aid must not take the bench line down
        return None, "PMC pass failed: %s" % str(e)[:120]


def measure_traffic(cfgname):
    """HBM bytes per tally launch, measured in THIS run: a `rocprofv3 --pmc FETCH_SIZE` pass (its own process, counters only,
    no tracing domains) over scripts/prof_tally.py, which replays the same workload and also runs the stream probe -- same
    access pattern, known byte count -- that calibrates the counter (MI355X_MICROARCH.md, HBM section: FETCH_SIZE is in
    KiB and under-reports wide coalesced reads on gfx950 by ~2x; calibrate on a known byte count of your own pattern).
    -> (HBM bytes per delivered byte of the profiled launch, or None; how it was obtained)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            cmd = [exe, "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
                   os.path.join(ROOT, "scripts", "prof_tally.py"), cfgname, "3"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=300)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, "rocprofv3 --pmc produced no counter file (rc %d)" % r.returncode
            stream_bytes = None
            for line in r.stdout.splitlines():
                t = line.split()
                if t and t[0] == "workload" and "stream_bytes" in t:
                    stream_bytes = float(t[t.index("stream_bytes") + 1])
            tally, probe = [], []
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != "FETCH_SIZE":
                    continue
                if "tally_population" in row["Kernel_Name"]:
                    tally.append(float(row["Counter_Value"]))
                elif "stream_probe" in row["Kernel_Name"]:
                    probe.append(float(row["Counter_Value"]))
            if not tally or not probe or not stream_bytes or sum(probe) == 0:
                return None, "counter file without tally / probe rows"
            factor = stream_bytes / (1024.0 * sum(probe) / len(probe))
            # (per delivered byte of the profiled launch: scripts/prof_tally.py holds the configuration's WHOLE population, a rank
            # of a sharded configuration one shard of it -- the caller scales by its own launch)
            return 1024.0 * sum(tally) / len(tally) * factor / stream_bytes, \
                "rocprofv3 --pmc FETCH_SIZE in this run, calibrated x%.3f on the stream probe (known byte count)" % factor
    except Exception as e:  # a measurement aid must not take the bench line down
        return None, "PMC pass failed: %s" % str(e)[:120]


def cpu_baseline(pop, K, H, L, cfg_id, obs, subj, member, records, rec_off, nb, budget_s):
    """Times oracle/ on a bounded sample of the SAME resident workload.  `cpu_baseline` = the faithful restatement
    of the Java (HashMap/TreeSet structure, full preProposal re-scan after every batch), one thread.
    `cpu_optimized` = the dense-mask / incremental-invalidation CPU formulation on all host cores."""
    from oracle import pyoracle as O
    cores = os.cpu_count() or 1
    reg = O.Registry()
    for i in range(pop.n):
        reg.intern(pop.hostnames[i], int(pop.ports[i]))
    oview = O.MembershipView(reg, K, list(zip(pop.id_hi.tolist(), pop.id_lo.tolist())), list(range(pop.n)))
    assert oview.getCurrentConfigurationId() == cfg_id

    def sub(k):
        return records[: rec_off[k]], rec_off[: k + 1].copy(), int(nb[:k].sum())

    # faithful: grow the sample until it has used a fair share of the budget
    k, spent, done_b, done_t = 1, 0.0, 0, 0.0
    while True:
        r, o, b = sub(k)
        t = time.perf_counter()
        O.sim_run(oview, K, H, L, pop.id_hi, pop.id_lo, r, o, nthreads=1)
        dt = time.perf_counter() - t
        spent += dt
        done_b, done_t, done_k = b, dt, k
        if spent > budget_s or k >= len(nb):
            break
        k = min(len(nb), max(k + 1, int(k * min(4.0, 0.6 * budget_s / max(dt, 1e-3)))))
    base = {"value": round(done_b / done_t, 1), "unit": "alert-batches/s", "cores": 1, "kind": "port",
            "sample": "first %d receivers of the same resident workload (%d alert batches), oracle/rapid_oracle.hpp "
                      "(statement-by-statement restatement of the Java), %.1f s" % (done_k, done_b, done_t)}
    k2 = min(len(nb), max(64, cores * 64))
    r, o, b = sub(k2)
    t = time.perf_counter()
    O.fast_sim_run(pop.n, K, H, L, cfg_id, obs, subj, member, r, o, nthreads=cores)
    dt = time.perf_counter() - t
    opt = {"value": round(b / dt, 1), "unit": "alert-batches/s", "cores": cores, "kind": "port",
           "sample": "first %d receivers (%d alert batches), oracle/fast_cut.hpp dense-mask formulation, %d threads, %.2f s"
                     % (k2, b, cores, dt)}
    return base, opt


if __name__ == "__main__":
    main()
