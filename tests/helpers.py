"""Shared helpers for the CPU-side tests (oracle construction, random adversarial streams)."""
import numpy as np

from oracle import pyoracle as O
from rapid_amd import scenarios as S


def oracle_view(pop, K, members=None):
    """Registry with every endpoint of `pop` interned (handle == node index) + a view of `members`."""
    reg = O.Registry()
    for i in range(pop.n):
        h = reg.intern(pop.hostnames[i], int(pop.ports[i]))
        assert h == i
    members = range(pop.n) if members is None else members
    ids = [(int(pop.id_hi[i]), int(pop.id_lo[i])) for i in members]
    view = O.MembershipView(reg, K, ids, list(members))
    return reg, view


def random_stream(rng, n_nodes, K, member, cfg_id, n_rec, hot, p_bad_cfg=0.05, p_bad_status=0.05, p_multi=0.15,
                  p_eob=0.5):
    """One receiver's adversarial stream: duplicates, stale configuration ids, wrong-status alerts, multi-ring
    masks, random batch boundaries."""
    recs = np.zeros(n_rec, dtype=S.ALERT_DTYPE)
    dst = rng.choice(hot, size=n_rec)
    recs["dst"] = dst
    recs["src"] = rng.integers(0, n_nodes, size=n_rec)
    mask = (1 << rng.integers(0, K, size=n_rec)).astype(np.uint16)
    multi = rng.random(n_rec) < p_multi
    mask[multi] |= (1 << rng.integers(0, K, size=int(multi.sum()))).astype(np.uint16)
    multi2 = rng.random(n_rec) < p_multi / 3
    mask[multi2] |= rng.integers(0, 1 << K, size=int(multi2.sum())).astype(np.uint16)
    recs["ring_mask"] = mask
    status = np.where(member[dst] != 0, S.DOWN, S.UP).astype(np.uint8)
    flip = rng.random(n_rec) < p_bad_status
    status[flip] ^= 1
    recs["status"] = status
    recs["cfg_id"] = cfg_id
    bad = rng.random(n_rec) < p_bad_cfg
    recs["cfg_id"][bad] = cfg_id + 1
    recs["flags"] = (rng.random(n_rec) < p_eob).astype(np.uint8)
    return recs


def props_list(poff, props, r):
    return props[poff[r]: poff[r + 1]].tolist()


_M64 = (1 << 64) - 1


def _mix64(x):
    """splitmix64 finaliser, as rapid::mix64 in csrc/tally_kernel.h."""
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def proposal_fingerprints(poff, props, emitted):
    """The kernel's per-receiver proposal fingerprint restated on the host (sum of mix64(node) + mix64(0x5EED + count),
    0 reserved for "no proposal"), computed from the ORACLE's proposal lists: comparing the whole array with the
    device's checks every receiver's proposal contents at full size."""
    R = len(poff) - 1
    out = np.zeros(R, dtype=np.uint64)
    props = np.asarray(props, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = props + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
        csum = np.concatenate([[np.uint64(0)], np.cumsum(x, dtype=np.uint64)])
        for r in range(R):
            if not emitted[r]:
                continue
            cnt = int(poff[r + 1] - poff[r])
            v = (int(csum[poff[r + 1]]) - int(csum[poff[r]]) + _mix64(0x5EED + cnt)) & _M64
            out[r] = v if v != 0 else 1
    return out
