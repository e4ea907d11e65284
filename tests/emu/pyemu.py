"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper over tests/emu/_build/libtally_emu.so: the unmodified HIP tally
kernels compiled with g++ against the SIMT emulator (tests/emu/hip/hip_runtime.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        variant = os.environ.get("RAPID_EMU_VARIANT", "")  # "q1" / "q2": smaller windows; "ubsan": sanitizer build
        import fcntl
        os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
        with open(os.path.join(_HERE, "_build", ".lock"), "w") as lk:  # one builder at a time (pytest-xdist workers)
            fcntl.flock(lk, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-C", _HERE, "-s", "VARIANT=" + variant], stderr=subprocess.DEVNULL)
        _LIB = C.CDLL(os.path.join(_HERE, "_build", "libtally_emu%s.so" % ("_" + variant if variant else "")))
    return _LIB


def window_records():
    """Records per window of the emulated build (RAPID_QUARTERS x 64)."""
    return int(lib().emu_window_records())


def build_round_index(records, n_nodes, K, L, obs, member):
    """Independent numpy statement of rapid_amd/csrc/index_kernels.h: touched / hot subjects, slot numbering (hot
    first, ascending node index), state template and the hot-hot adjacency in CSR form."""
    recs = np.asarray(records)
    kmask = (1 << K) - 1
    gmask = np.zeros(n_nodes, dtype=np.int64)
    ok = recs["dst"] < n_nodes
    np.bitwise_or.at(gmask, recs["dst"][ok].astype(np.int64), recs["ring_mask"][ok].astype(np.int64) & kmask)
    pop = np.array([bin(int(x)).count("1") for x in gmask])
    hot_nodes = np.flatnonzero(pop >= L)
    node_of_slot = hot_nodes.astype(np.int32)
    n_hot = len(hot_nodes)
    assert n_hot <= 16318
    slot_of = np.full(n_nodes, 0x3FFF, dtype=np.int64)
    slot_of[node_of_slot] = np.arange(n_hot)
    # the hot adjacency as index_build_block_kernel produces it: one triple (subject slot | observer slot << 14 | ring << 28)
    # per ring on which a hot node observes a hot slot, and the per-slot mask of those rings
    pairs, smask = [], np.zeros(n_hot + 1 + 8, dtype=np.uint16)  # (+ 8: the packed sweep reads the masks 16 bytes at a time, as engine.hip sizes the table)
    dict_ = slot_of | np.where(np.asarray(member) != 0, 0x8000, 0)
    for e in range(n_hot):
        s_node = int(node_of_slot[e])
        for k in range(K):
            o = int(obs[s_node, k])
            if o < 0:
                continue
            eo = int(slot_of[o])
            if eo >= n_hot:
                continue
            pairs.append(e | (eo << 14) | (k << 28))
            smask[e] |= 1 << k
        if smask[e]:
            dict_[s_node] |= 0x4000
    adj_off, adj = smask, np.array(pairs + [0], dtype=np.uint32)
    dict_ = dict_.astype(np.uint16)
    decl = (np.where(slot_of < n_hot, 0x3FFF, gmask & 0x3FFF) | np.where(np.asarray(member) != 0, 0x8000, 0)).astype(np.uint16)
    # the compressed form (index_build_block_kernel): one bit per node named by the alert set, touched nodes before each
    # 32-node word, (decl << 16 | slot) per touched node
    touched = (decl & 0x3FFF) != 0
    n_words = (n_nodes + 31) // 32
    padded = np.zeros(n_words * 32, dtype=bool)
    padded[:n_nodes] = touched
    tbits = (padded.reshape(n_words, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
    per_word = padded.reshape(n_words, 32).sum(axis=1)
    trank = np.concatenate([[0], np.cumsum(per_word)[:-1]]).astype(np.uint16)
    tn = np.flatnonzero(touched)
    dt, st = decl[tn].astype(np.uint32), dict_[tn].astype(np.uint32) & 0x3FFF  # tally_kernel.h: dict_entry(decl entry, slot)
    tent = (((~dt) & 0x3FFF) | np.where((dt >> 15) != 0, 1 << 15, 1 << 14).astype(np.uint32) | (st << 17)).astype(np.uint32)
    tent = np.concatenate([tent, np.zeros(1, dtype=np.uint32)])
    pad = np.zeros(8, dtype=np.uint16)  # the kernel stages both tables 16 bytes at a time: allocated with slack, as on the device
    dict_, decl = np.concatenate([dict_.astype(np.uint16), pad]), np.concatenate([np.asarray(decl).astype(np.uint16), pad])
    return dict(dict=dict_, decl=decl, tbits=tbits, trank=trank, tent=tent, n_touched=len(tn), node_of_slot=np.concatenate([node_of_slot, [0]]).astype(np.int32), adj_off=adj_off,
                adj=adj, n_hot=n_hot, n_adj=len(pairs))


def index_run(records, n_nodes, K, L, cfg_id, obs, member, chunked=False, direct_budget=-1, seed=1, q4=None, fused=False):
    """Runs the round-index KERNELS (rapid_amd/csrc/index_kernels.h: touch, then one workgroup -- or count / assign / one
    workgroup, the form of large populations) under the emulator.  -> dict shaped like build_round_index's, plus `info`
    (the eight words the host reads from the mailbox)."""
    L_ = lib()
    recs = np.ascontiguousarray(records)
    raw = np.zeros(recs.nbytes + 32, dtype=np.uint8)
    raw[: recs.nbytes] = recs.view(np.uint8).reshape(-1)
    member = np.ascontiguousarray(member, dtype=np.uint8)
    obs = np.ascontiguousarray(obs, dtype=np.int32)
    n_words = (n_nodes + 31) // 32
    adj_cap, tent_cap = 65536, 16384
    out = dict(dict=np.full(n_nodes + 40, 0xEEEE, dtype=np.uint16), decl=np.full(n_nodes + 40, 0xEEEE, dtype=np.uint16),
               node_of_slot=np.full(n_nodes + 1, -7, dtype=np.int32), smask=np.full(n_nodes + 1 + 8, 0xEEEE, dtype=np.uint16),
               pairs=np.zeros(adj_cap + 1, dtype=np.uint32), tbits=np.full(n_words + 1, 0xEEEEEEEE, dtype=np.uint32),
               trank=np.full(n_words + 1, 0xEEEE, dtype=np.uint16), tent=np.full(tent_cap, 0xEEEEEEEE, dtype=np.uint32),
               info=np.zeros(8, dtype=np.int32), entries=np.full(n_nodes + 1, 0xEEEEEEEE, dtype=np.uint32))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L_.emu_index_run.restype = C.c_int
    rc = L_.emu_index_run(p(raw), C.c_longlong(len(recs)), n_nodes, K, L, C.c_longlong(cfg_id), p(member), p(obs), 2 if fused else (1 if chunked else 0),
                          direct_budget, p(out["dict"]), p(out["decl"]), p(out["node_of_slot"]), p(out["smask"]), p(out["pairs"]), adj_cap,
                          p(out["tbits"]), p(out["trank"]), p(out["tent"]), tent_cap, p(out["info"]), C.c_ulonglong(seed),
                          None if q4 is None else p(q4[0]), None if q4 is None else p(q4[1]),  # q4 = (rows int32 [n][K], valid uint8 [n]): the memo
                          p(out["entries"]))  # (written by the one-launch form only)
    assert rc == 0, rc
    return out


def validate_alerts(records, n_nodes, K, cfg_id, member):
    """(every record passes the filter of MembershipService.java:644-675, every record is DOWN) -- what
    index_touch_kernel computes once per round."""
    r = np.asarray(records)
    if len(r) == 0:
        return True, True
    dst = r["dst"].astype(np.int64)
    inr = dst < n_nodes
    mem = np.asarray(member)[np.where(inr, dst, 0)] != 0
    down = r["status"] != 0
    # (an alert of another configuration is not validated: its copies are dropped whole per delivery, index_kernels.h: index_touch_kernel)
    ok = (r["cfg_id"] != cfg_id) | (inr & ((r["ring_mask"] & ((1 << K) - 1)) != 0) & (mem == down))
    return bool(ok.all()), bool(down.all())


CURRENT_RUNS = 0  # runs of a kCurrent instantiation so far (tally() below)


def tally(records, rec_off, n_nodes, K, H, L, cfg_id, obs, subj, member, prop_cap=None, force_exact=0, seed=1, waves=3,
          grid=2, tables_in_lds=1, trusted=False, declared=None, pool=False, fmt=None, packed=False):
    """`declared`: build the round index from these records (the round's distinct alert set) instead of the delivered
    ones; the call then returns (results ..., covered) with covered = False if the kernel found a delivered report that
    the declared set does not contain.  pool: only the first deal (grid x waves receivers) is static, the rest is claimed
    from the kernel's common pool."""
    if pool:
        force_exact = force_exact | 512
    if fmt is None:  # the product's pairing: 20-byte boundary records are looked up by the tally (tables_in_lds 0 / 1 / 2), resident
        fmt = 0 if tables_in_lds == 3 else 1  # 8-byte records (what the generator writes) carry their subjects' entries (3)
    L_ = lib()
    recs = np.ascontiguousarray(records)
    raw = np.zeros(((recs.nbytes + 15) // 16) * 16 + 32, dtype=np.uint8)
    raw[: recs.nbytes] = recs.view(np.uint8).reshape(-1)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.int64)
    R = len(rec_off) - 1
    prop_cap = n_nodes if prop_cap is None else prop_cap
    ix = build_round_index(recs if declared is None else np.ascontiguousarray(declared), n_nodes, K, L, np.asarray(obs), member)
    if trusted:
        ok, all_down = validate_alerts(recs, n_nodes, K, cfg_id, member)
        assert ok, "trusted variant requested for a stream with alerts that fail the filter"
        force_exact = force_exact | 256
    emit = np.full(R, -99, dtype=np.int32)
    nprop = np.full(R, -99, dtype=np.int32)
    pcount = np.full(R, -99, dtype=np.int32)
    fp = np.zeros(R, dtype=np.uint64)
    props = np.full((R, prop_cap), -1, dtype=np.int32)
    stats = np.zeros((grid, 8), dtype=np.uint64)  # one row per workgroup, as the kernel writes them
    vote_res = np.zeros(11 + (prop_cap + 2) // 2 + 1, dtype=np.uint64)  # res[0..9], "settled by the launch", the candidate's {size, list}
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    hashed = None
    if tables_in_lds == 4:  # the hashed dictionary of packed rounds: built by index_hash_kernel, which renumbers the slots
        assert packed and ix["n_touched"] == ix["n_hot"], "kDictHashed: packed rounds whose every named subject is hot"
        hashed = hash_build(ix, n_nodes, member, seed=seed)
        ix = dict(ix, node_of_slot=hashed["node_of_slot"], adj_off=hashed["smask"], adj=hashed["pairs"])
    # pre-validated boundary records that all carry the engine's configuration id: the product runs the instantiation that leaves
    # the ids in their cache lines (kCurrent) -- run here too, first, and held against the run that compares them
    if trusted and fmt == 1 and tables_in_lds in (0, 1, 2) and not (force_exact & 16384) and len(recs) and bool(np.all(recs["cfg_id"] == cfg_id)):
        want = tally(records, rec_off, n_nodes, K, H, L, cfg_id, obs, subj, member, prop_cap=prop_cap, force_exact=(force_exact & ~256) | 16384 | 32768,
                     seed=seed, waves=waves, grid=grid, tables_in_lds=tables_in_lds, trusted=True, declared=declared, pool=False, fmt=fmt, packed=packed)
    else:
        want = None
    if force_exact & 32768:  # (this call IS that run)
        global CURRENT_RUNS
        CURRENT_RUNS += 1
    force_exact &= ~32768
    rc = L_.emu_tally_run(p(raw), C.c_ulonglong(recs.nbytes), p(rec_off), R, n_nodes, K, H, L, C.c_longlong(cfg_id),
                          p(ix["dict"]), p(ix["decl"]), p(ix["node_of_slot"]), p(ix["adj_off"]), p(ix["adj"]),
                          ix["n_hot"], ix["n_adj"], p(emit), p(nprop), p(pcount), p(fp), p(props), prop_cap, p(stats),
                          force_exact, waves, grid, tables_in_lds, C.c_ulonglong(seed), p(ix["tbits"]), p(ix["trank"]), p(ix["tent"]),
                          ix["n_touched"], p(vote_res), int(fmt), 1 if packed else 0,
                          p(hashed["hoff"]) if hashed else None, p(hashed["hrem"]) if hashed else None, p(hashed["hmem"]) if hashed else None,
                          C.c_uint(hashed["mul"] if hashed else 0))
    if hashed:  # (the kernel lists a proposal in slot order; slots are in hash order there: back to ascending node index)
        for r in range(R):
            if pcount[r] > 0:
                props[r, : pcount[r]] = np.sort(props[r, : pcount[r]])
    # the vote statistics the kernel gathers next to the proposals (TallyParams::vote_res) against the results themselves
    voters = np.flatnonzero(pcount != 0)
    assert int(vote_res[2]) == len(voters), (vote_res[:11], len(voters))
    if int(vote_res[10]) == 0:  # the earlier statistics: lowest voter + voters, the rest left to vote_verify_kernel
        assert int(vote_res[0]) == (int(voters[0]) if len(voters) else 0xFFFFFFFF), vote_res[:11]
        assert not vote_res[[1, 3, 4, 5, 6, 7, 9]].any(), vote_res[:11]
    else:
        # settled by the launch (TallyParams::vote_cand): the candidate is SOME voter's proposal (whoever finished first), every voter
        # holding its fingerprint is counted, none of them differs bit for bit, and the answer carries the candidate's node list
        ref32 = vote_res[11:].view(np.int32)
        if len(voters) == 0:
            assert int(vote_res[0]) == 0xFFFFFFFF and not vote_res[[1, 3, 4, 6, 7, 9]].any() and int(vote_res[5]) == 0xFFFFFFFFFFFFFFFF and ref32[0] == 0, vote_res[:11]
        else:
            own = int(vote_res[0])
            assert own in set(voters.tolist()), (own, voters[:8])
            same = int(np.count_nonzero((pcount != 0) & (fp == fp[own])))
            assert int(vote_res[4]) == int(fp[own]) and int(vote_res[5]) == int(fp[own]) ^ 0xFFFFFFFFFFFFFFFF, vote_res[:11]
            deferred_lost = int(vote_res[1]) == 0 and same > 0  # (more deferred voters than the emulator's short list holds: no quorum is claimed)
            if not deferred_lost:
                assert int(vote_res[1]) == same and int(vote_res[7]) == same, (vote_res[:11], same)
                assert int(vote_res[3]) == (1 if same == len(voters) else 2)
            assert int(vote_res[6]) == 0 and int(vote_res[9]) == 0, vote_res[:11]
            assert int(ref32[0]) == int(pcount[own])
            if pcount[own] > 0:
                want_list = np.sort(props[own, : pcount[own]]) if hashed else props[own, : pcount[own]]
                got_list = np.sort(ref32[1: 1 + pcount[own]]) if hashed else ref32[1: 1 + pcount[own]]
                assert np.array_equal(got_list, want_list), (got_list[:8], want_list[:8])
    if want is not None:  # (statistics aside: the two instantiations take the same windows the same way, but that is not a promise)
        for a, b in zip(want[:5], (emit, nprop, pcount, fp, props)):
            assert np.array_equal(a, b), "kCurrent differs from the instantiation that compares the configuration ids"
        if declared is not None:
            assert want[6] == (rc == 0)
    if declared is not None:
        assert rc in (0, -1), rc
        return emit, nprop, pcount, fp, props, stats.sum(axis=0), rc == 0
    assert rc == 0, rc
    return emit, nprop, pcount, fp, props, stats.sum(axis=0)


def hash_key_bits(n_nodes):
    b = 10
    while (1 << b) < n_nodes:
        b += 1
    return b


def hash_build(ix, n_nodes, member, seed=1):
    """index_hash_kernel under the emulator, checked against its statement here: the hot nodes ordered by (bucket, remainder) of
    x = (node * mul) mod 2^bits under the first multiplier that keeps every bucket within 16 keys; hoff / hrem / hmem; node_of_slot,
    smask and the triples in the new numbering.  -> dict of the new tables + `mul`."""
    L_ = lib()
    n_hot, n_adj = ix["n_hot"], ix["n_adj"]
    bits = hash_key_bits(n_nodes)
    nb = 1 << (bits - 8)
    nos = np.ascontiguousarray(ix["node_of_slot"], dtype=np.int32)
    smask = np.ascontiguousarray(ix["adj_off"], dtype=np.uint16)
    pairs = np.ascontiguousarray(ix["adj"], dtype=np.uint32)
    member = np.ascontiguousarray(member, dtype=np.uint8)
    entries = dict_entries(ix, n_nodes).copy()
    dict_ = np.ascontiguousarray(ix["dict"], dtype=np.uint16).copy()
    out = dict(hoff=np.full(nb + 4, 0xEEEE, dtype=np.uint16), hrem=np.full(n_hot + 32 + 8, 0xEE, dtype=np.uint8), hmem=np.full((n_hot + 31) // 32 + 1, 0xEEEEEEEE, dtype=np.uint32),
               node_of_slot=np.full(n_hot + 1, -7, dtype=np.int32), smask=np.full(n_hot + 1 + 8, 0xEEEE, dtype=np.uint16),
               pairs=np.zeros(max(n_adj, 1) + 1, dtype=np.uint32), new_of_old=np.full(n_hot + 1, 0xEEEE, dtype=np.uint16))
    ans = np.zeros(2, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L_.emu_hash_build.restype = C.c_int
    L_.emu_hash_multiplier.restype = C.c_uint
    rc = L_.emu_hash_build(p(nos), p(smask), p(pairs), n_hot, n_adj, n_nodes, p(member), p(out["hoff"]), p(out["hrem"]), p(out["hmem"]), p(out["node_of_slot"]),
                           p(out["smask"]), p(out["pairs"]), p(out["new_of_old"]), p(entries), p(dict_), p(ans), C.c_ulonglong(seed))
    assert rc == 0 and ans[0] == 1, (rc, ans)
    mul = int(L_.emu_hash_multiplier(int(ans[1])))
    # the statement: keys, their order, the tables
    x = (nos[:n_hot].astype(np.uint64) * np.uint64(mul)) & np.uint64((1 << bits) - 1)
    order = np.argsort(x, kind="stable")  # (bucket, remainder) = the key itself; keys are distinct
    new_of_old = np.empty(n_hot, dtype=np.int64)
    new_of_old[order] = np.arange(n_hot)
    counts = np.bincount((x >> np.uint64(8)).astype(np.int64), minlength=nb)
    assert counts.max() <= 16
    for t in range(int(ans[1])):  # an earlier multiplier was passed over only because a bucket overflowed under it
        xt = (nos[:n_hot].astype(np.uint64) * np.uint64(int(L_.emu_hash_multiplier(t)))) & np.uint64((1 << bits) - 1)
        assert np.bincount((xt >> np.uint64(8)).astype(np.int64), minlength=nb).max() > 16
    assert np.array_equal(out["new_of_old"][:n_hot], new_of_old.astype(np.uint16))
    assert np.array_equal(out["hoff"][: nb + 1], np.concatenate([[0], np.cumsum(counts)]).astype(np.uint16))
    assert np.array_equal(out["hrem"][:n_hot], (x[order] & np.uint64(255)).astype(np.uint8)) and not out["hrem"][n_hot: n_hot + 32].any()
    assert np.array_equal(out["node_of_slot"][:n_hot], nos[:n_hot][order]) and np.array_equal(out["smask"][:n_hot], smask[:n_hot][order])
    mem_bits = member[nos[:n_hot][order]] != 0
    want_mem = np.zeros((n_hot + 31) // 32, dtype=np.uint32)
    for i in np.flatnonzero(mem_bits):
        want_mem[i >> 5] |= np.uint32(1 << (int(i) & 31))
    assert np.array_equal(out["hmem"][: len(want_mem)], want_mem)
    pr = pairs[:n_adj].astype(np.int64)
    want_pairs = new_of_old[pr & 0x3FFF] | (new_of_old[(pr >> 14) & 0x3FFF] << 14) | (pr & 0xF0000000)
    assert np.array_equal(out["pairs"][:n_adj], want_pairs.astype(np.uint32))
    e0 = dict_entries(ix, n_nodes)
    hot_nodes = nos[:n_hot]
    want_e = e0.copy()
    want_e[hot_nodes] = (e0[hot_nodes] & 0x1FFFF) | (new_of_old.astype(np.uint32) << 17)
    assert np.array_equal(entries, want_e)
    out["mul"] = mul
    return out


class CdInstance:
    """rapid_cd_* semantics through cd_instance_kernel under the emulator."""

    def __init__(self, n_nodes, K, H, L, obs, subj, member):
        self.n, self.K, self.H, self.L = n_nodes, K, H, L
        self.state = np.zeros(((n_nodes + 7) // 8) * 8, dtype=np.uint16)
        self.scal = np.zeros(4, dtype=np.int32)
        self.obs = np.ascontiguousarray(obs, dtype=np.int32)
        self.subj = np.ascontiguousarray(subj, dtype=np.int32)

    def _run(self, alerts, mode):
        n = len(alerts)
        out = np.full(max(self.n, 1), -1, dtype=np.int32)
        counts = np.zeros(max(n, 1), dtype=np.int32)
        out_n = np.zeros(1, dtype=np.int32)
        raw = np.ascontiguousarray(alerts).view(np.uint8).reshape(-1) if n else np.zeros(1, dtype=np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        lib().emu_cd_run(p(self.state), p(self.scal), p(raw), n, self.n, self.K, self.H, self.L, p(self.obs),
                         p(out), len(out), p(counts), p(out_n), mode, C.c_ulonglong(7))
        return out[: out_n[0]].tolist(), counts[:n].tolist()

    def aggregate(self, alerts):
        return self._run(alerts, 0)

    def invalidate(self):
        return self._run(np.zeros(0, dtype=np.uint8), 1)[0]

    def num_proposals(self):
        return int(self.scal[1])


RES_WORDS = 10


def vote_settle(fp, prop_count, props, mode, salt=0, seed=1, bits=None):
    """vote_count_local_kernel + vote_verify_kernel (mode 0) or vote_verify_kernel on the statistics the tally kernel leaves
    (mode 1: candidate = the lowest voter's proposal) under the emulator.  -> (res[10] as published, ref list or None if it
    overflowed)."""
    L_ = lib()
    fp = np.ascontiguousarray(fp, dtype=np.uint64)
    prop_count = np.ascontiguousarray(prop_count, dtype=np.int32)
    props = np.ascontiguousarray(props, dtype=np.int32)
    R, cap = props.shape
    block = np.zeros(RES_WORDS + (cap + 2) // 2 + 1, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L_.emu_vote_settle.restype = C.c_int
    # mode 2: the comparison on bits[R, words] (the voters' proposals as bitmaps over the round's hot slots)
    if bits is not None:
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        assert mode in (2, 3) and bits.shape[0] == R
    rc = L_.emu_vote_settle(p(fp), p(prop_count), p(props), cap, R, C.c_ulonglong(salt), mode, p(block), C.c_ulonglong(seed),
                            p(bits) if bits is not None else None, int(bits.shape[1]) if bits is not None else 0)
    assert rc == 0, rc
    ref = block[RES_WORDS:].view(np.int32)
    n = int(ref[0])
    return block[:RES_WORDS].copy(), (ref[1: 1 + n].tolist() if n >= 0 else None)


def vote_acc(fp, prop_count, props, bits, tile, target=0, tally_error=0, seed=1):
    """The accumulated vote count of a round taken tile by tile (vote_acc_pick / _count / _finish kernels).  -> (res[10], the
    candidate's list or None): the block a rank contributes to the all-gather."""
    L_ = lib()
    fp = np.ascontiguousarray(fp, dtype=np.uint64)
    prop_count = np.ascontiguousarray(prop_count, dtype=np.int32)
    props = np.ascontiguousarray(props, dtype=np.int32)
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    R, cap = props.shape
    block = np.zeros(RES_WORDS + (cap + 2) // 2 + 1, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L_.emu_vote_acc.restype = C.c_int
    rc = L_.emu_vote_acc(p(fp), p(prop_count), p(props), cap, p(bits), int(bits.shape[1]), R, int(tile), C.c_ulonglong(int(target)),
                         C.c_uint(tally_error), p(block), C.c_ulonglong(seed))
    assert rc == 0, rc
    ref = block[RES_WORDS:].view(np.int32)
    n = int(ref[0])
    return block[:RES_WORDS].copy(), (ref[1: 1 + n].tolist() if n >= 0 else None), block.copy()


def ids_contains_publish(old, new, seq, seed=1):
    """ids_contains_publish_kernel: is any of `new` (NodeIds as (high, low)) in the sorted list `old`?  -> the published word."""
    L_ = lib()
    oh = np.array([h for h, _ in old], dtype=np.int64)
    ol = np.array([l for _, l in old], dtype=np.int64)
    nh = np.array([h for h, _ in new], dtype=np.int64)
    nl = np.array([l for _, l in new], dtype=np.int64)
    word = C.c_uint(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L_.emu_ids_contains_publish.restype = C.c_int
    rc = L_.emu_ids_contains_publish(p(oh), p(ol), len(old), p(nh), p(nl), len(new), C.c_uint(seq), C.byref(word), C.c_ulonglong(seed))
    assert rc == 0, rc
    return int(word.value)


def vote_block(votes, voters, fp, cut, prop_cap, bad=0, err=0):
    """A rank's answer block as rapid_sim_count_votes all-gathers it: res[10] + ref[1 + prop_cap], padded to 8 bytes."""
    seg_words = RES_WORDS + (prop_cap + 2) // 2 + 1
    b = np.zeros(seg_words, dtype=np.uint64)
    M = (1 << 64) - 1
    b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8] = votes, voters, (0 if voters == 0 else 1 if votes == voters else 2), fp, (~fp) & M, bad, votes, err
    ref = b[RES_WORDS:].view(np.int32)
    ref[0] = len(cut) if cut is not None else 0
    if cut:
        ref[1: 1 + len(cut)] = cut
    return b


def vote_merge(blocks, prop_cap, quorum, seed=1):
    """vote_merge_kernel over the ranks' blocks.  -> (status, votes, voters, cut)."""
    L_ = lib()
    gathered = np.ascontiguousarray(np.concatenate(blocks))
    seg_words = len(blocks[0])
    out = np.zeros(seg_words, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L_.emu_vote_merge.restype = C.c_int
    rc = L_.emu_vote_merge(p(gathered), len(blocks), seg_words, prop_cap, C.c_longlong(quorum), p(out), C.c_ulonglong(seed))
    assert rc == 0, rc
    ref = out[RES_WORDS:].view(np.int32)
    n = int(ref[0])
    return int(out[9]), int(out[1]), int(out[2]), (ref[1: 1 + n].tolist() if n > 0 else None), out[:RES_WORDS].copy()


def view_build(hostnames, ports, id_hi, id_lo, K, members, keep=None, seed=1):
    """The view kernels of rapid_amd/csrc/view_kernels.h under the emulator (std::stable_sort standing in for the device's
    segmented radix sort): ring keys, rings, observer / subject tables, configuration id of the view over `members`; with
    `keep` (bool per node = the member flags after a view change) also the rings of that change, obtained from the old rings
    by compaction + merge of the joiners (ring_count / ring_scatter / ring_join kernels)."""
    L_ = lib()
    n = len(hostnames)
    blob = np.frombuffer(b"".join(hostnames) + b"\0" * 8, dtype=np.uint8).copy()
    off = np.zeros(n + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(h) for h in hostnames])
    ports = np.ascontiguousarray(ports, dtype=np.int32)
    members = np.ascontiguousarray(members, dtype=np.int32)
    M = len(members)
    ids = sorted((int(id_hi[m]), int(id_lo[m])) for m in members)  # signed order: high, then low (R/MembershipView.java:474-500)
    hi = np.array([a for a, _ in ids] + [0], dtype=np.int64)
    lo = np.array([b for _, b in ids] + [0], dtype=np.int64)
    out = dict(keys=np.zeros(K * n + 1, dtype=np.int64), ring=np.full(K * max(M, 1), -1, dtype=np.int32),
               obs=np.full(n * K + 1, -9, dtype=np.int32), subj=np.full(n * K + 1, -9, dtype=np.int32), cfg=np.zeros(1, dtype=np.int64),
               ring2=np.full(K * max(n, 1), -1, dtype=np.int32), m2=np.zeros(1, dtype=np.int32),
               obs2=np.full(n * K + 1, -9, dtype=np.int32), subj2=np.full(n * K + 1, -9, dtype=np.int32))
    keep_a = None if keep is None else np.ascontiguousarray(np.concatenate([np.asarray(keep, dtype=np.uint8), np.zeros(1, dtype=np.uint8)]))
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    L_.emu_view_build.restype = C.c_int
    rc = L_.emu_view_build(p(blob), p(off), p(ports), n, K, p(members), M, p(hi), p(lo), M, p(keep_a), p(out["keys"]), p(out["ring"]),
                           p(out["obs"]), p(out["subj"]), p(out["cfg"]), p(out["ring2"]), p(out["m2"]), C.c_ulonglong(seed),
                           p(out["obs2"]) if keep is not None else None, p(out["subj2"]) if keep is not None else None)
    assert rc == 0, rc
    out["keys"] = out["keys"][: K * n].reshape(K, n)
    out["ring"] = out["ring"][: K * M].reshape(K, M) if M else np.zeros((K, 0), dtype=np.int32)
    out["obs"], out["subj"] = out["obs"][: n * K].reshape(n, K), out["subj"][: n * K].reshape(n, K)
    if keep is not None:
        m2 = int(out["m2"][0])
        out["ring2"] = out["ring2"][: K * m2].reshape(K, m2)
        out["obs2"], out["subj2"] = out["obs2"][: n * K].reshape(n, K), out["subj2"][: n * K].reshape(n, K)
    return out


def ring_merge(ring, keys, member, join_nodes, join_keys, seed=1):
    """ring_count / ring_scatter / ring_join on ONE ring with hand-made (possibly equal) keys: old ring and joiners, each sorted by
    (key, node); member[node] = stays or joins -> (new ring, its keys)."""
    L_ = lib()
    ring = np.ascontiguousarray(np.concatenate([ring, [0]]), dtype=np.int32)
    keys = np.ascontiguousarray(np.concatenate([np.asarray(keys, dtype=np.uint64), np.zeros(1, dtype=np.uint64)]))
    jn = np.ascontiguousarray(np.concatenate([join_nodes, [0]]), dtype=np.int32)
    jk = np.ascontiguousarray(np.concatenate([np.asarray(join_keys, dtype=np.uint64), np.zeros(1, dtype=np.uint64)]))
    member = np.ascontiguousarray(np.concatenate([np.asarray(member, dtype=np.uint8), [0]]), dtype=np.uint8)
    m_old, J = len(ring) - 1, len(jn) - 1
    m_new = int(member[ring[:m_old]].sum()) + J
    out_r, out_k = np.full(m_new + 1, -1, dtype=np.int32), np.zeros(m_new + 1, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L_.emu_ring_merge.restype = C.c_int
    rc = L_.emu_ring_merge(p(ring), p(keys), m_old, p(member), p(jn), p(jk), J, p(out_r), p(out_k), m_new, C.c_ulonglong(seed))
    assert rc == 0, rc
    return out_r[:m_new], out_k[:m_new]


def ids_merge(old, new, seed=1):
    """ids_merge_kernel: two lists of (high, low) NodeIds sorted the way identifiersSeen is -> the merged list."""
    L_ = lib()
    oh = np.array([a for a, _ in old] + [0], dtype=np.int64)
    ol = np.array([b for _, b in old] + [0], dtype=np.int64)
    nh = np.array([a for a, _ in new] + [0], dtype=np.int64)
    nl = np.array([b for _, b in new] + [0], dtype=np.int64)
    n = len(old) + len(new)
    out_h, out_l = np.zeros(n + 1, dtype=np.int64), np.zeros(n + 1, dtype=np.int64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L_.emu_ids_merge.restype = C.c_int
    rc = L_.emu_ids_merge(p(oh), p(ol), len(old), p(nh), p(nl), len(new), p(out_h), p(out_l), C.c_ulonglong(seed))
    assert rc == 0, rc
    return list(zip(out_h[:n].tolist(), out_l[:n].tolist()))


def dict_entries(ix, n_nodes):
    """dict_entry per node + the poison entry (index_kernels.h: dict_entries_kernel) from build_round_index's tables."""
    n_hot = ix["n_hot"]
    d, decl = ix["dict"][:n_nodes].astype(np.uint32), ix["decl"][:n_nodes].astype(np.uint32)
    sl = d & 0x3FFF
    sl = np.where(sl == 0x3FFF, n_hot + (np.arange(n_nodes, dtype=np.uint32) & 63), sl).astype(np.uint32)
    e = ((~decl) & 0x3FFF) | np.where((decl >> 15) != 0, 1 << 15, 1 << 14).astype(np.uint32) | (sl << 17)
    return np.concatenate([e.astype(np.uint32), np.array([0xFFFF | (n_hot << 17)], dtype=np.uint32)])


def generate(batches, receivers, seed, cfg_id, n_nodes, entries=None, keep=None, boundary=False):
    """rapid_sim_generate's kernels under the emulator.  boundary: -> (records as ALERT_DTYPE, rec_off); else
    -> (entries, core words, rec_off) of the resolved 8-byte records (`entries` = dict_entries(...) of the round's index)."""
    from rapid_amd.scenarios import ALERT_DTYPE
    L_ = lib()
    recs = np.ascontiguousarray(batches.recs)
    raw = np.concatenate([recs.view(np.uint8).reshape(-1), np.zeros(32, dtype=np.uint8)])
    off = np.ascontiguousarray(batches.off, dtype=np.int64)
    rx = np.ascontiguousarray(receivers, dtype=np.int32)
    R, A = len(rx), int(off[-1])
    n = R * A
    out = np.full(n * (20 if boundary else 8) + 8, 0xEE, dtype=np.uint8)
    rec_off = np.zeros(R + 1, dtype=np.int64)
    kp = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint32)
    ent = None if entries is None else np.ascontiguousarray(entries, dtype=np.uint32)
    assert boundary or ent is not None
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    L_.emu_generate.restype = C.c_int
    rc = L_.emu_generate(p(raw), p(off), len(off) - 1, p(kp), p(rx), R, C.c_ulonglong(int(seed) & ((1 << 64) - 1)), C.c_longlong(cfg_id), n_nodes,
                         p(ent), p(out), 1 if boundary else 0, p(rec_off), C.c_ulonglong(3))
    assert rc == 0, rc
    assert (out[n * (20 if boundary else 8):] == 0xEE).all()  # nothing written behind the last stream
    if boundary:
        return out[: n * 20].view(ALERT_DTYPE).copy(), rec_off
    core = out[: n * 8].view(np.uint32).reshape(n, 2)
    return core[:, 0].copy(), core[:, 1].copy(), rec_off
