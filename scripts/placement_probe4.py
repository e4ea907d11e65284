"""Tally time by the VIRTUAL address of the records (scripts/placement_probe3.py: five different physical allocations mapped at one
virtual address all measure the same): many allocations, (address, time) pairs for a look at the address bits."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402
from rapid_amd import engine as E, scenarios as S  # noqa: E402

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
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


def timed(ptr):
    sim.attach_streams_device(ptr, nbytes, d_off.data_ptr(), R, keepalive=None)
    sim.set_alert_set_device(d_al.data_ptr(), len(alert_set), trust_copies=True, keepalive=d_al)
    return min(sim.time_tally(8) for _ in range(2))


rng = np.random.default_rng(3)
ptrs = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 48):
    size = nbytes + 64 + int(rng.integers(0, 64)) * (2 << 20)  # (varying sizes: varying address steps)
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), size) == 0
    ptrs.append(p)
    assert hip.hipMemcpy(p, raw.data_ptr(), nbytes, 1) == 0
    for shift in (0, 1 << 21, 3 << 21):
        if shift and shift + nbytes > size:
            continue
        if shift:
            assert hip.hipMemcpy(C.c_void_p(p.value + shift), raw.data_ptr(), nbytes, 1) == 0
        print("VA %x %.4f" % (p.value + shift, timed(p.value + shift)), flush=True)
