"""Does the tally kernel's time depend on WHERE the delivered records lie?  (scripts/ab_variants.py showed consecutive measurements
alternating between 0.385 and 0.405 ms whatever the build: every engine had copied the streams into a fresh allocation.)
One device allocation, the C3b records copied to different offsets inside it and attached in place; then separate allocations.
    python scripts/placement_probe.py"""
import sys
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


def timed(ptr):
    sim.attach_streams_device(ptr, nbytes, d_off.data_ptr(), R, keepalive=None)
    sim.set_alert_set_device(d_al.data_ptr(), len(alert_set), trust_copies=True, keepalive=d_al)
    return min(sim.time_tally(10) for _ in range(3))


big = torch.empty(nbytes + (256 << 20), dtype=torch.uint8, device="cuda")
base = big.data_ptr()
print("one allocation at %x (mod 2 MiB: %x, mod 1 GiB: %x)" % (base, base % (2 << 20), base % (1 << 30)))
for shift in (0, 4, 64, 128, 256, 1024, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 16 << 20, 64 << 20, 128 << 20, 0, 4096):
    big[shift:shift + nbytes].copy_(raw, non_blocking=False)
    torch.cuda.synchronize()
    print("  offset %10d: tally %.4f ms" % (shift, timed(base + shift)), flush=True)
del big
torch.cuda.empty_cache()
bufs = []
for i in range(6):
    b = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda")
    b[:nbytes].copy_(raw)
    torch.cuda.synchronize()
    bufs.append(b)
    print("separate allocation %d at %x (mod 2 MiB %x): tally %.4f ms" % (i, b.data_ptr(), b.data_ptr() % (2 << 20), timed(b.data_ptr())), flush=True)
for i, b in enumerate(bufs):
    print("again, allocation %d: tally %.4f ms" % (i, timed(b.data_ptr())), flush=True)
