"""Mutation fuzzing of the wire decoders (rapid_amd/csrc/host_abi.cpp reads bytes that came from the network) under
AddressSanitizer + UndefinedBehaviorSanitizer: tests/fuzz/wire_fuzz.cpp is compiled with g++ against the host-only
translation unit and fed valid messages built with the Python protobuf runtime.  Host only."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from tests import proto_rapid as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ep(i):
    return P.Endpoint(hostname=b"h%d" % i, port=5000 + i)


def corpus():
    rng = np.random.default_rng(1)
    out = []
    for _ in range(40):
        batch = P.BatchedAlertMessage(sender=ep(int(rng.integers(64))))
        for _ in range(int(rng.integers(0, 9))):
            a = P.AlertMessage(edgeSrc=ep(int(rng.integers(64))), edgeDst=ep(int(rng.integers(64))), edgeStatus=int(rng.integers(2)),
                               configurationId=int(rng.integers(-2**63, 2**63 - 1)))
            a.ringNumber.extend(sorted(set(rng.integers(0, 10, int(rng.integers(1, 5))).tolist())))
            if rng.random() < 0.5:
                a.nodeId.high, a.nodeId.low = int(rng.integers(-2**63, 2**63 - 1)), int(rng.integers(-2**63, 2**63 - 1))
            batch.messages.append(a)
        out.append((0, P.RapidRequest(batchedAlertMessage=batch).SerializeToString()))
        out.append((3, batch.SerializeToString()))
        eps = [ep(int(e)) for e in rng.integers(0, 64, int(rng.integers(0, 12)))]
        cfg = int(rng.integers(-2**63, 2**63 - 1))
        rank = P.Rank(round=int(rng.integers(0, 4)), nodeIndex=int(rng.integers(-2**31, 2**31 - 1)))
        msgs = [
            P.RapidRequest(fastRoundPhase2bMessage=P.FastRoundPhase2bMessage(sender=ep(1), configurationId=cfg, endpoints=eps)),
            P.RapidRequest(phase1aMessage=P.Phase1aMessage(sender=ep(2), configurationId=cfg, rank=rank)),
            P.RapidRequest(phase1bMessage=P.Phase1bMessage(sender=ep(3), configurationId=cfg, rnd=rank, vrnd=P.Rank(round=1, nodeIndex=1), vval=eps)),
            P.RapidRequest(phase2aMessage=P.Phase2aMessage(sender=ep(4), configurationId=cfg, rnd=rank, vval=eps)),
            P.RapidRequest(phase2bMessage=P.Phase2bMessage(sender=ep(5), configurationId=cfg, rnd=rank, endpoints=eps)),
            P.RapidRequest(probeMessage=P.ProbeMessage(sender=ep(6))),
        ]
        out.extend((0, m.SerializeToString()) for m in msgs)
    return out


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_wire_decoders_survive_mutated_messages(tmp_path):
    exe = tmp_path / "wire_fuzz"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "rapid_amd", "csrc"),
                           os.path.join(ROOT, "tests", "fuzz", "wire_fuzz.cpp"), os.path.join(ROOT, "rapid_amd", "csrc", "host_abi.cpp"),
                           "-o", str(exe)])
    path = tmp_path / "corpus.bin"
    with open(path, "wb") as f:
        for kind, data in corpus():
            f.write(struct.pack("<II", kind, len(data)) + data)
    for seed in (1, 2):
        r = subprocess.run([str(exe), str(path), "150000", str(seed)], capture_output=True, text=True, timeout=300,
                           env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1"})
        assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
