"""Pins oracle/xxh64.hpp: standard XXH64 vectors + the independent Python `xxhash` package.

The reference reaches XXH64 through net.openhft:zero-allocation-hashing:0.8 (rapid/pom.xml:79-83); its own
tests pin no literal hash (parity unpinned at this boundary), so the public
specification is the anchor."""
import random
import struct

import pytest

from oracle import pyoracle as O

xxhash = pytest.importorskip("xxhash")


def test_standard_vectors():
    assert O.xxh64(b"", 0) == 0xEF46DB3751D8E999  # well-known empty-input vector
    assert O.xxh64(b"", 1) == 0xD5AFBA1336A3BE4B
    assert O.xxh64(struct.pack("<i", 1234), 0) == 0x275772FECB918454
    assert O.xxh64(b"127.0.0.1", 0) == 0xC08B1587DF65B7A7


@pytest.mark.parametrize("length", list(range(0, 70)) + [95, 96, 97, 127, 128, 129, 255, 256, 1000])
def test_against_python_xxhash_all_tail_lengths(length):
    rng = random.Random(length)
    data = bytes(rng.randrange(256) for _ in range(length))
    for seed in (0, 1, 9, 0x9E3779B1, 2**64 - 1):
        assert O.xxh64(data, seed) == xxhash.xxh64(data, seed=seed).intdigest()


def test_survey_appendix_c_vectors():
    """SURVEY.md Appendix C (derived under the A.1/A.2 semantics)."""
    reg = O.Registry()
    n = reg.intern("127.0.0.1", 1234)
    view = O.MembershipView(reg, 10)
    view.ringAdd(n, (0, 0))
    assert view.ringKey(0, n) == 8660156494884553101
    assert view.ringKey(1, n) == -1048522345525689326
    assert view.ringKey(2, n) == 8804440040588648532
    assert view.getCurrentConfigurationId() == -3053644066016802086

    reg2 = O.Registry()
    view2 = O.MembershipView(reg2, 10)
    nodes = {}
    for i, port in enumerate(range(1234, 1240)):
        nodes[port] = reg2.intern("127.0.0.1", port)
        view2.ringAdd(nodes[port], (i + 1, i + 1))
    inv = {v: k for k, v in nodes.items()}
    assert [inv[x] for x in view2.getRing(0)] == [1237, 1239, 1236, 1238, 1235, 1234]
    assert [inv[x] for x in view2.getRing(1)] == [1236, 1234, 1238, 1237, 1235, 1239]
