// TEST INFRASTRUCTURE ONLY (see oracle/README.md) -- never linked into the product library.
//
// XXH64, restated from the public xxHash specification (xxhash_spec.md, "XXH64 algorithm").
// The reference reaches it through the third-party dependency
//   net.openhft:zero-allocation-hashing:0.8   (rapid/pom.xml:79-83, not vendored under /root/reference)
// via LongHashFunction.xx(seed).{hashBytes,hashInt,hashLong}; call sites:
//   rapid/src/main/java/com/vrg/rapid/MembershipView.java:47, 548-549, 552-553, 568, 580-581.
// hashInt / hashLong hash the 4 / 8 value bytes in little-endian order, hashBytes hashes the
// hostname bytes; all results are reinterpreted as Java signed longs.
//
// PARITY NOTE: the reference's tests pin no literal hash value; this file is validated against the
// standard XXH64 vectors and against the independent Python `xxhash` package (tests/test_oracle_xxh64.py).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace oracle {

namespace xxh_detail {
constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL;
constexpr uint64_t P2 = 0xC2B2AE3D27D4EB4FULL;
constexpr uint64_t P3 = 0x165667B19E3779F9ULL;
constexpr uint64_t P4 = 0x85EBCA77C2B2AE63ULL;
constexpr uint64_t P5 = 0x27D4EB2F165667C5ULL;

inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t rd64(const uint8_t* p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}
inline uint32_t rd32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
inline uint64_t round(uint64_t acc, uint64_t in) {
    acc += in * P2;
    acc = rotl(acc, 31);
    return acc * P1;
}
inline uint64_t merge(uint64_t acc, uint64_t v) {
    acc ^= round(0, v);
    return acc * P1 + P4;
}
}  // namespace xxh_detail

inline uint64_t xxh64(const void* data, size_t len, uint64_t seed) {
    using namespace xxh_detail;
    const uint8_t* p = static_cast<const uint8_t*>(data);
    const uint8_t* const end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t* const limit = end - 32;
        do {
            v1 = round(v1, rd64(p));
            v2 = round(v2, rd64(p + 8));
            v3 = round(v3, rd64(p + 16));
            v4 = round(v4, rd64(p + 24));
            p += 32;
        } while (p <= limit);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        h = merge(h, v1);
        h = merge(h, v2);
        h = merge(h, v3);
        h = merge(h, v4);
    } else {
        h = seed + P5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) {
        h ^= round(0, rd64(p));
        h = rotl(h, 27) * P1 + P4;
        p += 8;
    }
    if (p + 4 <= end) {
        h ^= (uint64_t)rd32(p) * P1;
        h = rotl(h, 23) * P2 + P3;
        p += 4;
    }
    while (p < end) {
        h ^= (uint64_t)(*p) * P5;
        h = rotl(h, 11) * P1;
        ++p;
    }
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return h;
}

// LongHashFunction.hashInt / hashLong: value bytes in little-endian order.
inline uint64_t xxh64_int(int32_t v, uint64_t seed) {
    uint8_t b[4];
    uint32_t u = (uint32_t)v;
    for (int i = 0; i < 4; ++i) b[i] = (uint8_t)(u >> (8 * i));
    return xxh64(b, 4, seed);
}
inline uint64_t xxh64_long(int64_t v, uint64_t seed) {
    uint8_t b[8];
    uint64_t u = (uint64_t)v;
    for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(u >> (8 * i));
    return xxh64(b, 8, seed);
}

}  // namespace oracle
