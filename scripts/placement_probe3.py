"""Is it the VIRTUAL or the PHYSICAL placement of the records that the tally's time depends on (scripts/placement_probe.py)?
Physical memory created with hipMemCreate, mapped at several virtual addresses (hipMemAddressReserve / hipMemMap), the C3b records
copied in, the tally attached in place."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402
from rapid_amd import engine as E, scenarios as S  # noqa: E402


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Flags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32HandleMetaData", C.c_void_p), ("allocFlags", Flags)]


class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


hip = C.CDLL("libamdhip64.so")
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
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


def timed(ptr):
    sim.attach_streams_device(ptr, nbytes, d_off.data_ptr(), R, keepalive=None)
    sim.set_alert_set_device(d_al.data_ptr(), len(alert_set), trust_copies=True, keepalive=d_al)
    return min(sim.time_tally(10) for _ in range(3))


prop = Prop()
prop.type = 1  # hipMemAllocationTypePinned
prop.location.type = 1  # hipMemLocationTypeDevice
prop.location.id = 0
gran = C.c_size_t(0)
assert hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1) == 0  # recommended
size = ((nbytes + gran.value - 1) // gran.value) * gran.value
size = max(size, 2 << 30)
print("granularity %d, size %d" % (gran.value, size))
acc = Access()
acc.location.type, acc.location.id, acc.flags = 1, 0, 3
handles = []
# ONE physical allocation at several virtual addresses (reserved far apart: spacer reservations in between are never mapped)
hnd = C.c_void_p()
assert hip.hipMemCreate(C.byref(hnd), C.c_size_t(size), C.byref(prop), C.c_ulonglong(0)) == 0
vas = []
for i in range(6):
    va = C.c_void_p()
    assert hip.hipMemAddressReserve(C.byref(va), C.c_size_t(size), C.c_size_t(0), C.c_void_p(0), C.c_ulonglong(0)) == 0
    spacer = C.c_void_p()
    assert hip.hipMemAddressReserve(C.byref(spacer), C.c_size_t(14 << 30), C.c_size_t(0), C.c_void_p(0), C.c_ulonglong(0)) == 0
    rc = hip.hipMemMap(va, C.c_size_t(size), C.c_size_t(0), hnd, C.c_ulonglong(0))
    if rc != 0:
        print("mapping the same physical allocation a second time: rc %d" % rc)
        break
    assert hip.hipMemSetAccess(va, C.c_size_t(size), C.byref(acc), C.c_size_t(1)) == 0
    vas.append(va)
if vas:
    assert hip.hipMemcpy(vas[0], raw.data_ptr(), nbytes, 1) == 0
    print("ONE physical allocation, %d virtual addresses: %s" % (len(vas), " | ".join("va %x: %.4f" % (v.value, timed(v.value)) for v in vas)), flush=True)
# several physical allocations, each at its own (kept) virtual address
others = []
for i in range(6):
    h2 = C.c_void_p()
    assert hip.hipMemCreate(C.byref(h2), C.c_size_t(size), C.byref(prop), C.c_ulonglong(0)) == 0
    va = C.c_void_p()
    assert hip.hipMemAddressReserve(C.byref(va), C.c_size_t(size), C.c_size_t(0), C.c_void_p(0), C.c_ulonglong(0)) == 0
    assert hip.hipMemMap(va, C.c_size_t(size), C.c_size_t(0), h2, C.c_ulonglong(0)) == 0
    assert hip.hipMemSetAccess(va, C.c_size_t(size), C.byref(acc), C.c_size_t(1)) == 0
    assert hip.hipMemcpy(va, raw.data_ptr(), nbytes, 1) == 0
    others.append((h2, va))
    print("physical allocation %d at va %x: %.4f" % (i, va.value, timed(va.value)), flush=True)
# ... and the FIRST physical allocation's records once more through one of the later virtual ranges' neighbours
sys.exit(0)
