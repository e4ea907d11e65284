"""Which allocations are 'fast' for the tally (scripts/placement_probe.py: 0.385 against 0.400 ms by allocation)?  Candidates by size
and by how they are obtained; the records copied to the start of each and tallied in place."""
import sys
import ctypes as C
import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402
from rapid_amd import engine as E, scenarios as S  # noqa: E402

spec = S.CONFIGS["C3b"]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
sc = S.build_scenario("C3b", subj, view.getCurrentConfigurationId(), materialise=False)
alert_set = np.ascontiguousarray(sc.batches.recs)
d_al = torch.from_numpy(alert_set.view(np.uint8).reshape(-1).copy()).cuda()
recs, off, nb = S.deliver(sc.batches, sc.receivers, seed_delivery=2)
raw = torch.from_numpy(recs.view(np.uint8).reshape(-1))
nbytes = raw.numel()
d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.int64)).cuda()
R = len(off) - 1
sim = E.ClusterSimulation(eng)
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


def timed(ptr):
    sim.attach_streams_device(ptr, nbytes, d_off.data_ptr(), R, keepalive=None)
    sim.set_alert_set_device(d_al.data_ptr(), len(alert_set), trust_copies=True, keepalive=d_al)
    return min(sim.time_tally(10) for _ in range(3))


def fill(ptr):
    assert hip.hipMemcpy(ptr, raw.data_ptr(), nbytes, 1) == 0


for label, size in (("exact size", nbytes + 64), ("2 GiB", 2 << 30), ("4 GiB", 4 << 30), ("8 GiB", 8 << 30), ("16 GiB", 16 << 30)):
    res = []
    ptrs = []
    for i in range(5):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), size) == 0
        ptrs.append(p)
        fill(p.value)
        res.append(timed(p.value))
    print("hipMalloc %-10s x5 (all kept): %s" % (label, " ".join("%.4f" % r for r in res)), flush=True)
    # a second position inside the larger ones
    if size >= (4 << 30):
        q = ptrs[0].value + (2 << 30)
        fill(q)
        print("   ... the same records 2 GiB into the first one: %.4f" % timed(q), flush=True)
    for p in ptrs:
        hip.hipFree(p)
