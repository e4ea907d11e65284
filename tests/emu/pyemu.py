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
        subprocess.check_call(["make", "-C", _HERE, "-s"], stderr=subprocess.DEVNULL)
        _LIB = C.CDLL(os.path.join(_HERE, "_build", "libtally_emu.so"))
        _LIB.emu_tally_lds_bytes.restype = C.c_int
    return _LIB


def state_template(member):
    n = len(member)
    t = np.zeros(((n + 7) // 8) * 8, dtype=np.uint16)
    t[:n] = np.where(np.asarray(member) != 0, 0x8000, 0).astype(np.uint16)
    return t


def tally(records, rec_off, n_nodes, K, H, L, cfg_id, obs, subj, member, prop_cap=None, force_exact=0, seed=1):
    L_ = lib()
    recs = np.ascontiguousarray(records)
    raw = np.zeros(((recs.nbytes + 15) // 16) * 16 + 32, dtype=np.uint8)
    raw[: recs.nbytes] = recs.view(np.uint8).reshape(-1)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.int64)
    R = len(rec_off) - 1
    prop_cap = n_nodes if prop_cap is None else prop_cap
    tpl = state_template(member)
    obs = np.ascontiguousarray(obs, dtype=np.int32)
    subj = np.ascontiguousarray(subj, dtype=np.int32)
    emit = np.full(R, -99, dtype=np.int32)
    nprop = np.full(R, -99, dtype=np.int32)
    pcount = np.full(R, -99, dtype=np.int32)
    fp = np.zeros(R, dtype=np.uint64)
    props = np.full((R, prop_cap), -1, dtype=np.int32)
    stats = np.zeros(8, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L_.emu_tally_run(p(raw), C.c_ulonglong(raw.nbytes), p(rec_off), R, n_nodes, K, H, L, C.c_longlong(cfg_id),
                          p(tpl), p(obs), p(subj), p(emit), p(nprop), p(pcount), p(fp), p(props), prop_cap, p(stats),
                          force_exact, C.c_ulonglong(seed))
    assert rc == 0, rc
    return emit, nprop, pcount, fp, props, stats


class CdInstance:
    """rapid_cd_* semantics through cd_instance_kernel under the emulator."""

    def __init__(self, n_nodes, K, H, L, obs, subj, member):
        self.n, self.K, self.H, self.L = n_nodes, K, H, L
        self.state = state_template(member).copy()
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
                         p(self.subj), p(out), len(out), p(counts), p(out_n), mode, C.c_ulonglong(7))
        return out[: out_n[0]].tolist(), counts[:n].tolist()

    def aggregate(self, alerts):
        return self._run(alerts, 0)

    def invalidate(self):
        return self._run(np.zeros(0, dtype=np.uint8), 1)[0]

    def num_proposals(self):
        return int(self.scal[1])
