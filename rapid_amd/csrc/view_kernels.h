// K-ring expander construction on the GPU (R/ = /root/reference/rapid/src/main/java/com/vrg/rapid/):
//   ring_keys_kernel      R/MembershipView.java:562-587  AddressComparator.computeHash, all endpoints x K seeds
//   ring_tables_kernel    R/MembershipView.java:234-257 (successors), :308-322 (predecessors),
//                         :292-303 (expected observers of a non-member = its would-be predecessors)
//   config_id_kernel      R/MembershipView.java:544-556  Configuration.getConfigurationId as an exact
//                         mod-2^64 reduction of (value, multiplier) pairs
// XXH64 is restated from the public specification (the reference reaches it through
// net.openhft:zero-allocation-hashing:0.8, rapid/pom.xml:79-83).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// the workgroup's dynamic LDS segment (tests/emu/ supplies its own definition: a CPU build has no such thing)
#ifndef RAPID_DYNAMIC_LDS
#define RAPID_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

namespace rapid {

// ---- XXH64 ------------------------------------------------------------------------------------------------
constexpr unsigned long long XP1 = 0x9E3779B185EBCA87ull;
constexpr unsigned long long XP2 = 0xC2B2AE3D27D4EB4Full;
constexpr unsigned long long XP3 = 0x165667B19E3779F9ull;
constexpr unsigned long long XP4 = 0x85EBCA77C2B2AE63ull;
constexpr unsigned long long XP5 = 0x27D4EB2F165667C5ull;

__host__ __device__ inline unsigned long long xx_rotl(unsigned long long x, int r) { return (x << r) | (x >> (64 - r)); }
__host__ __device__ inline unsigned long long xx_rd64(const unsigned char* p) {
    unsigned long long v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}
__host__ __device__ inline unsigned int xx_rd32(const unsigned char* p) {
    return (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24);
}
__host__ __device__ inline unsigned long long xx_round(unsigned long long acc, unsigned long long in) {
    acc += in * XP2;
    acc = xx_rotl(acc, 31);
    return acc * XP1;
}
__host__ __device__ inline unsigned long long xx_merge(unsigned long long acc, unsigned long long v) {
    acc ^= xx_round(0, v);
    return acc * XP1 + XP4;
}
__host__ __device__ inline unsigned long long xx_avalanche(unsigned long long h) {
    h ^= h >> 33;
    h *= XP2;
    h ^= h >> 29;
    h *= XP3;
    h ^= h >> 32;
    return h;
}
__host__ __device__ inline unsigned long long xxh64_bytes(const unsigned char* p, int len, unsigned long long seed) {
    const unsigned char* const end = p + len;
    unsigned long long h;
    if (len >= 32) {
        unsigned long long v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        const unsigned char* const limit = end - 32;
        do {
            v1 = xx_round(v1, xx_rd64(p));
            v2 = xx_round(v2, xx_rd64(p + 8));
            v3 = xx_round(v3, xx_rd64(p + 16));
            v4 = xx_round(v4, xx_rd64(p + 24));
            p += 32;
        } while (p <= limit);
        h = xx_rotl(v1, 1) + xx_rotl(v2, 7) + xx_rotl(v3, 12) + xx_rotl(v4, 18);
        h = xx_merge(h, v1);
        h = xx_merge(h, v2);
        h = xx_merge(h, v3);
        h = xx_merge(h, v4);
    } else {
        h = seed + XP5;
    }
    h += (unsigned long long)len;
    while (p + 8 <= end) {
        h ^= xx_round(0, xx_rd64(p));
        h = xx_rotl(h, 27) * XP1 + XP4;
        p += 8;
    }
    if (p + 4 <= end) {
        h ^= (unsigned long long)xx_rd32(p) * XP1;
        h = xx_rotl(h, 23) * XP2 + XP3;
        p += 4;
    }
    while (p < end) {
        h ^= (unsigned long long)(*p) * XP5;
        h = xx_rotl(h, 11) * XP1;
        ++p;
    }
    return xx_avalanche(h);
}
// LongHashFunction.hashInt / hashLong: the value's 4 / 8 little-endian bytes
__host__ __device__ inline unsigned long long xxh64_u32(unsigned int v, unsigned long long seed) {
    unsigned long long h = seed + XP5 + 4ull;
    h ^= (unsigned long long)v * XP1;
    h = xx_rotl(h, 23) * XP2 + XP3;
    return xx_avalanche(h);
}
__host__ __device__ inline unsigned long long xxh64_u64(unsigned long long v, unsigned long long seed) {
    unsigned long long h = seed + XP5 + 8ull;
    h ^= xx_round(0, v);
    h = xx_rotl(h, 27) * XP1 + XP4;
    return xx_avalanche(h);
}

// ---- ring keys: one thread per (endpoint, ring) -------------------------------------------------------------
// keys[k * n_nodes + n] = (int64)(xxh64(hostname, k) * 31 + xxh64(le32(port), k));  sortable[...] = key with the
// sign bit flipped (unsigned order == Java signed Long.compare order).  For k == 0 the two seed-0 hashes are
// also kept for the configuration id.
__global__ void ring_keys_kernel(const unsigned char* blob, const int* host_off, const int* ports, int n_nodes, int K,
                                 long long* keys, unsigned long long* hx_host0, unsigned long long* hx_port0) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= n_nodes * K) return;
    const int k = t / n_nodes;
    const int n = t - k * n_nodes;
    const unsigned long long hh = xxh64_bytes(blob + host_off[n], host_off[n + 1] - host_off[n], (unsigned long long)k);
    const unsigned long long hp = xxh64_u32((unsigned int)ports[n], (unsigned long long)k);
    keys[(long long)k * n_nodes + n] = (long long)(hh * 31ull + hp);
    if (k == 0) {
        hx_host0[n] = hh;
        hx_port0[n] = hp;
    }
}

// sort input for ring k: (sortable key, node index) of every member
__global__ void ring_gather_kernel(const long long* keys, const int* members, int n_members, int n_nodes, int K,
                                   unsigned long long* sort_keys, int* sort_vals) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_members * K) return;
    const int k = (int)(t / n_members);
    const int i = (int)(t - (long long)k * n_members);
    const int n = members[i];
    sort_keys[t] = (unsigned long long)keys[(long long)k * n_nodes + n] ^ 0x8000000000000000ull;
    sort_vals[t] = n;
}

// after the K sorts: ring[k][pos] = node, ring_skeys[k][pos] = sortable key.  Members get successor /
// predecessor rows; non-members get their expected observers (predecessor of their key on every ring).
__global__ void ring_tables_kernel(const int* ring, const unsigned long long* ring_skeys, const long long* keys,
                                   const unsigned char* member, int n_nodes, int n_members, int K, int* pos_scratch,
                                   int* obs, int* subj, int phase) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (phase == 0) {  // scatter ring positions: pos_scratch[k][node]
        if (t >= (long long)n_members * K) return;
        const int k = (int)(t / n_members);
        const int p = (int)(t - (long long)k * n_members);
        pos_scratch[(long long)k * n_nodes + ring[t]] = p;
        return;
    }
    if (t >= (long long)n_nodes * K) return;
    const int n = (int)(t / K);
    const int k = (int)(t - (long long)n * K);
    const int* rk = ring + (long long)k * n_members;
    int o = -1, s = -1;
    if (member[n]) {
        if (n_members > 1) {  // :240-242, :275-277 -- a single member has no observers / subjects
            const int p = pos_scratch[(long long)k * n_nodes + n];
            o = rk[p + 1 == n_members ? 0 : p + 1];
            s = rk[p == 0 ? n_members - 1 : p - 1];
        }
    } else if (n_members > 0) {  // :296-299
        const unsigned long long key = (unsigned long long)keys[(long long)k * n_nodes + n] ^ 0x8000000000000000ull;
        const unsigned long long* sk = ring_skeys + (long long)k * n_members;
        int lo = 0, hi = n_members;  // lower_bound
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sk[mid] < key) lo = mid + 1; else hi = mid;
        }
        o = rk[lo == 0 ? n_members - 1 : lo - 1];  // lower(node), wrapping to last()
    }
    obs[t] = o;
    subj[t] = s;
}

// ---- incremental view change (R/MembershipView.java:123-160 ringAdd, :167-201 ringDelete) ---------------------------------
// A decided cut removes some members and admits some joiners; everybody else keeps its place in every ring.  The new ring k
// is therefore the old ring k without the removed nodes, merged with the (few) joiners sorted by their ring-k key -- a
// stable compaction plus a merge by binary search, spread over workgroups of kRingChunk old positions, instead of sorting
// all K x M keys again.  Four launches:
//   ring_count_kernel    chunk_kept[k][c] = survivors among the old positions of chunk c of ring k
//   ring_chunk_prep_kernel  their exclusive scan per ring + the joiners before every chunk's first position
//   ring_scatter_kernel  survivor at old position p  ->  new position (survivors before p) + (joiners with a smaller key)
//   ring_join_kernel     joiner j of ring k          ->  new position j + (survivors with a smaller key)
constexpr int kRingChunk = 1024;

// A ring is ordered by (key, node index): two endpoints with the same 64-bit ring key -- one chance in 2^64 per pair; the
// reference's TreeSet would silently refuse the second one, R/MembershipView.java:123-141 with the comparator of :562-587 -- stand in
// the order of their node indices, which is what the stable sort of a full build produces (members are handed to it in ascending
// node order).  The merge below uses the same total order on both sides, so the incremental and the full path cannot disagree.
// elements of (keys[], nodes[]) -- sorted by that order -- that are smaller than (key, node)
__device__ inline int lower_bound_key_node(const unsigned long long* keys, const int* nodes, int n, unsigned long long key, int node) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const unsigned long long km = keys[mid];
        if (km < key || (km == key && nodes[mid] < node)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// grid = K x n_chunks (blockIdx.x = k * n_chunks + c), block = kRingChunk
__global__ __launch_bounds__(kRingChunk) void ring_count_kernel(const int* ring_in, int m_old, int n_chunks, const unsigned char* member,
                                                                int* chunk_kept) {
    __shared__ int s_total;
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();
    const int k = (int)blockIdx.x / n_chunks, c = (int)blockIdx.x - k * n_chunks;
    const int p = c * kRingChunk + (int)threadIdx.x;
    const bool keep = p < m_old && member[ring_in[(long long)k * m_old + p]] != 0;
    const unsigned long long b = __ballot(keep);
    if ((threadIdx.x & 63u) == 0u && b != 0ull) atomicAdd(&s_total, __popcll(b));
    __syncthreads();
    if (threadIdx.x == 0) chunk_kept[blockIdx.x] = s_total;
}

// What every chunk's workgroup needs before it can place anything, worked out ONCE per ring instead of by each of the ~10^4
// workgroups for itself (every one summed the counts of the chunks before it and ran two thirteen-step binary searches through memory
// ahead of its first store: ~15 us of dependent round trips per workgroup, 150 us for the scatter of 10^7 positions):
//   chunk_base[k][c] = survivors in the chunks before c (an exclusive scan of chunk_kept), chunk_base[k][n_chunks] = all survivors;
//   chunk_lb[k][c]   = joiners of ring k that sort before the FIRST old position of chunk c, chunk_lb[k][n_chunks] = n_join: the
//                      joiners before any position of chunk c lie in [chunk_lb[c], chunk_lb[c + 1]] (both sides are sorted).
// grid = K, block = 1024.
__global__ __launch_bounds__(1024) void ring_chunk_prep_kernel(const int* ring_in, const unsigned long long* skeys_in, int m_old, int n_chunks,
                                                               const int* chunk_kept, const unsigned long long* join_skeys, const int* join_nodes, int n_join,
                                                               int* chunk_base, int* chunk_lb) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    const int k = (int)blockIdx.x, t = (int)threadIdx.x, lane = t & 63, wv = t >> 6;
    const int* const kept = chunk_kept + (long long)k * n_chunks;
    int* const base = chunk_base + (long long)k * (n_chunks + 1);
    int* const lb = chunk_lb + (long long)k * (n_chunks + 1);
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_chunks; c0 += 1024) {
        const int c = c0 + t;
        const int v = c < n_chunks ? kept[c] : 0;
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        int before = s_carry;
        for (int i = 0; i < wv; ++i) before += s_wave[i];
        if (c < n_chunks) base[c] = before + incl - v;
        __syncthreads();
        if (t == 1023) s_carry = before + incl;
        __syncthreads();
    }
    if (t == 0) {
        base[n_chunks] = s_carry;
        lb[n_chunks] = n_join;
    }
    for (int c = t; c < n_chunks; c += 1024) {
        const long long p = (long long)k * m_old + (long long)c * kRingChunk;
        lb[c] = n_join > 0 ? lower_bound_key_node(join_skeys + (long long)k * n_join, join_nodes + (long long)k * n_join, n_join, skeys_in[p], ring_in[p]) : 0;
    }
}

__global__ __launch_bounds__(kRingChunk) void ring_scatter_kernel(const int* ring_in, const unsigned long long* skeys_in, int m_old, int n_chunks,
                                                                  const unsigned char* member, const int* chunk_base, const int* chunk_lb,
                                                                  const unsigned long long* join_skeys, const int* join_nodes, int n_join,
                                                                  int* ring_out, unsigned long long* skeys_out, int m_new) {
    __shared__ int s_wave[kRingChunk / 64];
    const int k = (int)blockIdx.x / n_chunks, c = (int)blockIdx.x - k * n_chunks;
    const int base = chunk_base[(long long)k * (n_chunks + 1) + c];
    const int p = c * kRingChunk + (int)threadIdx.x;
    int node = 0;
    unsigned long long key = 0ull;
    bool keep = false;
    if (p < m_old) {
        node = ring_in[(long long)k * m_old + p];
        key = skeys_in[(long long)k * m_old + p];
        keep = member[node] != 0;
    }
    // The joiners that sort before a position are a monotone function of the position (both sides are sorted by (key, node)): the
    // answers of this chunk's and the next chunk's first positions (ring_chunk_prep_kernel) bound everybody's, and with a few thousand
    // joiners among a million members the two are usually equal -- a thirteen-step binary search through memory per survivor
    // shrinks to a search inside that interval.
    const unsigned long long* const jk = join_skeys + (long long)k * n_join;
    const int* const jn = join_nodes + (long long)k * n_join;
    const int lo0 = chunk_lb[(long long)k * (n_chunks + 1) + c], hi0 = chunk_lb[(long long)k * (n_chunks + 1) + c + 1];
    const unsigned long long b = __ballot(keep);
    const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6);
    if (lane == 0) s_wave[wv] = __popcll(b);
    __syncthreads();
    int before = base + __popcll(b & ((1ull << lane) - 1ull));
    for (int i = 0; i < wv; ++i) before += s_wave[i];
    if (keep) {
        const int lb = lo0 + (hi0 > lo0 ? lower_bound_key_node(jk + lo0, jn + lo0, hi0 - lo0, key, node) : 0);
        const int at = before + lb;
        if (at < m_new) {
            ring_out[(long long)k * m_new + at] = node;
            skeys_out[(long long)k * m_new + at] = key;
        }
    }
}

// one wavefront per (ring, joiner): grid = ceil(K * n_join / waves per block)
__global__ __launch_bounds__(256) void ring_join_kernel(const int* ring_in, const unsigned long long* skeys_in, int m_old, int n_chunks,
                                                        const unsigned char* member, const int* chunk_base, const unsigned long long* join_skeys,
                                                        const int* join_nodes, int n_join, int K, int* ring_out, unsigned long long* skeys_out,
                                                        int m_new) {
    const int lane = (int)(threadIdx.x & 63u);
    const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= (long long)K * n_join) return;
    const int k = (int)(w / n_join), j = (int)(w - (long long)k * n_join);
    const unsigned long long key = join_skeys[w];
    const int* rk = ring_in + (long long)k * m_old;
    const int p = lower_bound_key_node(skeys_in + (long long)k * m_old, rk, m_old, key, join_nodes[w]);  // old positions that sort before the joiner: [0, p)
    const int c = p / kRingChunk;
    // the survivors among the old positions of the joiner's own chunk before p: up to sixteen steps of 64 positions, every step's two
    // dependent reads (the node at the position, its member flag) issued for all steps together -- two round trips, not thirty-two
    int kept = 0;
    int nodes_[kRingChunk / 64];
#pragma unroll
    for (int s_ = 0; s_ < kRingChunk / 64; ++s_) {
        const int i = c * kRingChunk + s_ * 64 + lane;
        nodes_[s_] = i < p ? rk[i] : -1;
    }
#pragma unroll
    for (int s_ = 0; s_ < kRingChunk / 64; ++s_) kept += (nodes_[s_] >= 0 && member[nodes_[s_]] != 0) ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off, 64);
    kept += chunk_base[(long long)k * (n_chunks + 1) + min(c, n_chunks)];  // (the survivors of the chunks before the joiner's)
    if (lane == 0 && j + kept < m_new) {
        ring_out[(long long)k * m_new + j + kept] = join_nodes[w];
        skeys_out[(long long)k * m_new + j + kept] = key;
    }
}

// The joiners of a cut, ring by ring, sorted by (sortable key, node), in two launches that use the whole GPU for a few thousand pairs:
//   ring_sort_runs_kernel   every run of kJoinRun consecutive pairs of a ring is sorted in place in LDS (a bitonic network over the run);
//   ring_merge_runs_kernel  every pair's final place = its index inside its run + the pairs of every OTHER run of its ring that sort
//                           before it (a binary search per run; pairs are distinct -- a node joins once -- so the places are too).
// (Measured at 10^6 members, 5,000 joiners x 10 rings: the library's segmented radix sort 0.27 ms of fixed launches and scans; one
// workgroup per ring running a bitonic network over 8,192 padded pairs 109 us -- ten CUs busy, 246 idle; every pair counting its
// predecessors among all of its ring's pairs 361 us -- J^2 comparisons are too many even spread over every CU;
// profiles/r05_apply_1m_kernels.txt.)  n <= kJoinSortMax; beyond: the library sort.  keys / nodes: [K][n] as ring_gather_kernel
// leaves them, in any order.
constexpr int kJoinSortMax = 8192;
constexpr int kJoinRun = 1024;
__global__ __launch_bounds__(kJoinRun / 2) void ring_sort_runs_kernel(unsigned long long* keys, int* nodes, int n) {
    __shared__ unsigned long long sk[kJoinRun];
    __shared__ int sn[kJoinRun];
    const int runs = (n + kJoinRun - 1) / kJoinRun;
    const int k = (int)blockIdx.x / runs, r0 = ((int)blockIdx.x - k * runs) * kJoinRun;
    const int cnt = min(kJoinRun, n - r0), T = (int)blockDim.x, t = (int)threadIdx.x;
    unsigned long long* const gk = keys + (long long)k * n + r0;
    int* const gn = nodes + (long long)k * n + r0;
    for (int i = t; i < kJoinRun; i += T) {
        sk[i] = i < cnt ? gk[i] : ~0ull;  // (padding sorts last: behind every real pair, whose node index is below INT_MAX)
        sn[i] = i < cnt ? gn[i] : 0x7FFFFFFF;
    }
    __syncthreads();
    for (int size = 2; size <= kJoinRun; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = t; i < kJoinRun / 2; i += T) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;  // the i-th pair of positions at this stride
                const bool up = (lo & size) == 0;
                const unsigned long long ka = sk[lo], kb = sk[hi];
                const int na = sn[lo], nb = sn[hi];
                const bool a_after_b = ka > kb || (ka == kb && na > nb);
                if (a_after_b == up) {
                    sk[lo] = kb;
                    sk[hi] = ka;
                    sn[lo] = nb;
                    sn[hi] = na;
                }
            }
            __syncthreads();
        }
    }
    for (int i = t; i < cnt; i += T) {
        gk[i] = sk[i];
        gn[i] = sn[i];
    }
}
__global__ void ring_merge_runs_kernel(const unsigned long long* keys, const int* nodes, int n, int K, unsigned long long* keys_out, int* nodes_out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)K * n) return;
    const int k = (int)(t / n), i = (int)(t - (long long)k * n);
    const unsigned long long* const rk = keys + (long long)k * n;
    const int* const rn = nodes + (long long)k * n;
    const unsigned long long key = rk[i];
    const int node = rn[i];
    const int mine = i / kJoinRun;
    int place = i - mine * kJoinRun;
    for (int r0 = 0, r = 0; r0 < n; r0 += kJoinRun, ++r)
        if (r != mine) place += lower_bound_key_node(rk + r0, rn + r0, min(kJoinRun, n - r0), key, node);
    keys_out[(long long)k * n + place] = key;
    nodes_out[(long long)k * n + place] = node;
}

// ---- the observer / subject tables after a cut, PATCHED instead of rebuilt ---------------------------------------------------
// ringAdd / ringDelete change the successor and the predecessor of the changed node's ring neighbours and nothing else
// (R/MembershipView.java:123-201).  So only these rows change: on every ring, the old predecessor and the old successor of a node
// that left (read off the OLD tables: a row of a node that left is not written here), the joiner's own row and the rows of its
// new neighbours.  Every one of them is recomputed from the NEW ring -- the node's position found by a binary search over the
// ring's sorted (key, node) pairs -- so that chains of changes (the neighbour left too, two joiners side by side) need no case
// analysis: whoever writes a row writes what the final ring says.  changed[0 .. n_gone) = the nodes that left, changed[n_gone ..
// n_gone + n_join) = the joiners; member = the NEW flags; ring / skeys = the NEW rings [K][m]; m >= 2.  One thread per (changed
// node, ring): 150,000 threads instead of two passes over 10^7 table entries with scattered reads (0.38 ms at 10^6 members).
__device__ inline void ring_patch_row(const int* rk, const unsigned long long* sk, int m, const long long* keys, int n_nodes, int k, int K, int node,
                                      int* obs, int* subj) {
    const unsigned long long key = (unsigned long long)keys[(long long)k * n_nodes + node] ^ 0x8000000000000000ull;
    const int q = lower_bound_key_node(sk, rk, m, key, node);  // the node's own position: (key, node) is in the ring
    obs[(long long)node * K + k] = rk[q + 1 == m ? 0 : q + 1];
    subj[(long long)node * K + k] = rk[q == 0 ? m - 1 : q - 1];
}
__global__ void ring_patch_kernel(const int* changed, int n_gone, int n_join, const int* ring, const unsigned long long* ring_skeys, int m,
                                  const long long* keys, const unsigned char* member, int n_nodes, int K, int* obs, int* subj) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)(n_gone + n_join) * K) return;
    const int i = (int)(t / K), k = (int)(t - (long long)i * K);
    const int node = changed[i];
    const int* const rk = ring + (long long)k * m;
    const unsigned long long* const sk = ring_skeys + (long long)k * m;
    if (i < n_gone) {  // (its row still holds the old view's neighbours)
        const int pred = subj[(long long)node * K + k], succ = obs[(long long)node * K + k];
        if (pred >= 0 && pred < n_nodes && member[pred] != 0) ring_patch_row(rk, sk, m, keys, n_nodes, k, K, pred, obs, subj);
        if (succ >= 0 && succ < n_nodes && member[succ] != 0) ring_patch_row(rk, sk, m, keys, n_nodes, k, K, succ, obs, subj);
    } else {
        const unsigned long long key = (unsigned long long)keys[(long long)k * n_nodes + node] ^ 0x8000000000000000ull;
        const int q = lower_bound_key_node(sk, rk, m, key, node);
        const int succ = rk[q + 1 == m ? 0 : q + 1], pred = rk[q == 0 ? m - 1 : q - 1];
        obs[(long long)node * K + k] = succ;
        subj[(long long)node * K + k] = pred;
        obs[(long long)pred * K + k] = node;   // (what the final ring says about them, whoever else writes these rows)
        subj[(long long)succ * K + k] = node;
    }
}
// ... and the rows of the NON-members -- their expected observers, R/MembershipView.java:292-322: the ring predecessors of their
// keys; subjects: none -- after the members' rows are final (a node that left reads its old row in ring_patch_kernel).  Every
// registered non-member is recomputed: which of them have a changed neighbourhood is not worth finding out.
// Two launches: the registered non-members gathered into a list (one pass over the member flags: a wave's ballot, one atomic per
// wave; list[0] = their number, in no particular order), then one thread per (non-member, ring) in a grid-stride loop over that
// number -- the host never learns it.  (One thread per (node, ring) of the whole registry, 10^7 of them at 10^6 nodes to find 30,000
// non-members, was 77 us.)
// (ONE atomic per workgroup of 1,024 nodes: with one per wave, the ~7,000 waves that hold one of a cut's 10,000 leavers queued up
// behind each other on list[0] for 126 us)
__global__ __launch_bounds__(1024) void nonmember_list_kernel(const unsigned char* member, int n_nodes, int* list) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int n = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const bool is = n < n_nodes && member[n] == 0;
    const unsigned long long b = __ballot(is);
    const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    if (lane == 0) s_wave[wv] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int i = 0; i < nw; ++i) total += s_wave[i];
        s_base = total > 0 ? atomicAdd(&list[0], total) : 0;
    }
    __syncthreads();
    int before = s_base;
    for (int i = 0; i < wv; ++i) before += s_wave[i];
    if (is) list[1 + before + __popcll(b & ((1ull << lane) - 1ull))] = n;
}
__global__ void ring_nonmember_rows_kernel(const int* ring, const unsigned long long* ring_skeys, int m, const long long* keys, const int* list, int n_nodes,
                                           int K, int* obs, int* subj) {
    const long long total = (long long)list[0] * K;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int n = list[1 + (int)(t / K)], k = (int)(t % K);
    int o = -1;
    if (m > 0) {
        const unsigned long long key = (unsigned long long)keys[(long long)k * n_nodes + n] ^ 0x8000000000000000ull;
        const unsigned long long* sk = ring_skeys + (long long)k * m;
        int lo = 0, hi = m;  // lower_bound by key alone, as ring_tables_kernel does for a non-member
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sk[mid] < key) lo = mid + 1; else hi = mid;
        }
        o = ring[(long long)k * m + (lo == 0 ? m - 1 : lo - 1)];
    }
    obs[(long long)n * K + k] = o;
    subj[(long long)n * K + k] = -1;
    }
}

// identifiersSeen (R/MembershipView.java:474-500: ordered by signed high, then signed low; never pruned) lives sorted on the
// device; the NodeIds a cut admits (sorted the same way on the host, a handful) are merged in: one thread per element
__device__ inline bool id_less(long long ah, long long al, long long bh, long long bl) { return ah < bh || (ah == bh && al < bl); }
// flag[0] |= 1 if any of the `n_new` ids is among the `n_old` sorted ones (UUIDAlreadySeenException, R/MembershipView.java:127-129)
__global__ void ids_contains_kernel(const long long* old_hi, const long long* old_lo, int n_old, const long long* new_hi, const long long* new_lo,
                                    int n_new, unsigned int* flag) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_new) return;
    const long long h = new_hi[t], l = new_lo[t];
    int lo = 0, hi = n_old;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (id_less(old_hi[mid], old_lo[mid], h, l)) lo = mid + 1; else hi = mid;
    }
    if (lo < n_old && old_hi[lo] == h && old_lo[lo] == l) atomicOr(flag, 1u);
}

// The same question for the few thousand NodeIds of one cut, answered into the host-mapped page the host polls (ONE word: the call's
// sequence number doubled, + 1 if any of them was seen before): no copy, no stream synchronisation on a view change's path.
// One thread per NodeId (twenty-one dependent steps through two million sorted ids each: spread over the GPU they take what one of
// them takes; one workgroup walking all 5,000 took 96 us); acc[0] collects the answer, acc[1] counts finished workgroups, and the
// last one publishes and leaves both words zero for the next call.
__global__ __launch_bounds__(256) void ids_contains_publish_kernel(const long long* old_hi, const long long* old_lo, int n_old, const long long* new_hi,
                                                                   const long long* new_lo, int n_new, unsigned int* acc, volatile unsigned int* mail, int word,
                                                                   unsigned int seq) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    bool any = false;
    if (t < n_new) {
        const long long h = new_hi[t], l = new_lo[t];
        int lo = 0, hi = n_old;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (id_less(old_hi[mid], old_lo[mid], h, l)) lo = mid + 1; else hi = mid;
        }
        any = lo < n_old && old_hi[lo] == h && old_lo[lo] == l;
    }
    if (__syncthreads_or(any ? 1 : 0) != 0 && threadIdx.x == 0) atomicOr(&acc[0], 1u);
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&acc[1], 1u) == gridDim.x - 1u) {
            __threadfence();
            const unsigned int a = atomicOr(&acc[0], 0u);
            acc[0] = 0u;
            acc[1] = 0u;
            mail[word] = 2u * seq + (a != 0u ? 1u : 0u);
        }
    }
}

__global__ void ids_merge_kernel(const long long* old_hi, const long long* old_lo, int n_old, const long long* new_hi, const long long* new_lo,
                                 int n_new, long long* out_hi, long long* out_lo) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_old) {
        const long long h = old_hi[t], l = old_lo[t];
        int lo = 0, hi = n_new;  // new ids smaller than this one
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (id_less(new_hi[mid], new_lo[mid], h, l)) lo = mid + 1; else hi = mid;
        }
        out_hi[t + lo] = h;
        out_lo[t + lo] = l;
    } else if (t < (long long)n_old + n_new) {
        const int j = (int)(t - n_old);
        const long long h = new_hi[j], l = new_lo[j];
        int lo = 0, hi = n_old;  // old ids smaller than this one
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (id_less(old_hi[mid], old_lo[mid], h, l)) lo = mid + 1; else hi = mid;
        }
        out_hi[j + lo] = h;
        out_lo[j + lo] = l;
    }
}

// rows[i][0..K) = table[nodes[i]][0..K): a few rows of the observer table for the host (rapid_view_q4_at_risk)
__global__ void gather_rows_kernel(const int* table, const int* nodes, int n, int K, int* rows) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * K) return;
    const int i = (int)(t / K), k = (int)(t - (long long)i * K);
    rows[t] = table[(long long)nodes[i] * K + k];
}

// preds[i][k] = the ring-k predecessor of nodes[i] WITHOUT wrap-around (TreeSet.lower: -1 for the ring minimum) -- the node
// whose memoised observers ringAdd / ringDelete of nodes[i] drop (R/MembershipView.java:143-152, 181-195)
__global__ void gather_lower_kernel(const int* subj, const int* ring, int n_members, const int* nodes, int n, int n_nodes, int K, int* preds) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * K) return;
    const int i = (int)(t / K), k = (int)(t - (long long)i * K);
    const int node = nodes[i];
    preds[t] = (n_members > 0 && ring[(long long)k * n_members] != node) ? subj[(long long)node * K + k] : -1;
}

// The memoised observers of the reference (Q4; index_kernels.h reads them for hot members): valid[node] = 0 for what
// ringAdd / ringDelete of nodes[i] drop -- the node's ring predecessors WITHOUT wrap-around (TreeSet.lower) and, self != 0, the
// node's own entry (R/MembershipView.java:143-152, 181-195).  subj / ring ([K][n_members]): the tables of the view the predecessors
// are taken from -- "has a predecessor without wrap-around" = "is not the ring's first element" (the per-node ring positions this
// used to read are not kept any more: a view change patches the tables instead of walking every position).
// member_clear != nullptr: nodes[] are the nodes that leave in this view change, and their member flags on the device are cleared
// on the way (what member_patch_kernel would do in a launch of its own).
__global__ void q4_invalidate_kernel(const int* subj, const int* ring, int n_members, const int* nodes, int n, int n_nodes, int K, unsigned char* valid,
                                     int self, unsigned char* member_clear = nullptr) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * K) return;
    const int i = (int)(t / K), k = (int)(t - (long long)i * K);
    const int node = nodes[i];
    if (member_clear != nullptr && k == 0) member_clear[node] = 0;
    if (self != 0 && k == 0) valid[node] = 0;
    if (n_members > 0 && ring[(long long)k * n_members] != node) {
        const int pred = subj[(long long)node * K + k];
        if (pred >= 0 && pred < n_nodes) valid[pred] = 0;
    }
}
// rapid_view_q4_at_risk: for every member among nodes[] -- memoise today's observers if nothing is memoised (what the first
// getObserversOf does), else flag[i] = the memo differs from today's table
__global__ void q4_check_kernel(const int* nodes, int n, const unsigned char* member, const int* obs, int K, int* rows, unsigned char* valid,
                                unsigned char* flag) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int node = nodes[i];
    flag[i] = 0;
    if (member[node] == 0) return;
    if (valid[node] == 0) {
        for (int k = 0; k < K; ++k) rows[(long long)node * K + k] = obs[(long long)node * K + k];
        valid[node] = 1;
        return;
    }
    bool differs = false;
    for (int k = 0; k < K; ++k) differs = differs || rows[(long long)node * K + k] != obs[(long long)node * K + k];
    flag[i] = differs ? 1 : 0;
}

// A view change's member flags, patched where they changed (the nodes that left: 0, the nodes that came: 1) -- the flags on the
// device are the ones the rings were built from, and the change is a few thousand nodes of a million.
__global__ void member_patch_kernel(unsigned char* member, const int* gone, int n_gone, const int* joined, int n_joined) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n_gone) member[gone[i]] = 0;
    if (i < n_joined) member[joined[i]] = 1;
}

// ---- configuration id -------------------------------------------------------------------------------------------
// hash = 1; for id in sorted ids: hash = hash*37 + xx0(high); hash = hash*37 + xx0(low);
//           for ep in ring 0:     hash = hash*37 + xx0(hostname); hash = hash*37 + xx0(port)
// A segment of the sequence maps h -> h * m + v with m = 37^len; segments compose associatively (the affine maps
// h -> h m + v form a monoid), so the sequence is cut into gridDim.x contiguous slices: workgroup b reduces slice b to one
// (v, m) pair -- thread t folds a contiguous piece with Horner's rule, an ordered tree combines the pieces -- and
// config_id_final_kernel folds the workgroups' pairs in order.  With one workgroup (small views) the first kernel writes
// the configuration id itself.
__global__ void config_id_kernel(const long long* ids_hi, const long long* ids_lo, int n_ids, const int* ring0,
                                 int n_members, const unsigned long long* hx_host0, const unsigned long long* hx_port0,
                                 long long* out, unsigned long long* partial, volatile unsigned int* seq_out = nullptr,
                                 unsigned int seq = 0u) {
    RAPID_DYNAMIC_LDS(smem_raw);
    unsigned long long* sv = reinterpret_cast<unsigned long long*>(smem_raw);
    unsigned long long* sm = sv + blockDim.x;
    const long long total = 2ll * n_ids + 2ll * n_members;
    const int T = (int)blockDim.x;
    const int t = (int)threadIdx.x;
    const long long G = (long long)gridDim.x;
    const long long per_block = (total + G - 1) / G;
    const long long b0 = per_block * (long long)blockIdx.x, b1 = (b0 + per_block < total) ? b0 + per_block : total;
    const long long span = b1 > b0 ? b1 - b0 : 0;
    const long long per = (span + T - 1) / T;
    const long long beg = b0 + per * t, end = (beg + per < b1) ? beg + per : b1;
    unsigned long long v = 0, m = 1;
    for (long long i = beg; i < end; ++i) {
        unsigned long long x;
        if (i < 2ll * n_ids) {
            const long long j = i >> 1;
            x = xxh64_u64((unsigned long long)((i & 1) ? ids_lo[j] : ids_hi[j]), 0);
        } else {
            const long long j = (i - 2ll * n_ids) >> 1;
            const int node = ring0[j];
            x = ((i - 2ll * n_ids) & 1) ? hx_port0[node] : hx_host0[node];
        }
        v = v * 37ull + x;
        m *= 37ull;
    }
    sv[t] = v;
    sm[t] = m;
    __syncthreads();
    for (int stride = 1; stride < T; stride <<= 1) {  // ordered tree: left segment first
        const int i = 2 * stride * t;
        if (i + stride < T) {
            const unsigned long long lv = sv[i], lm = sm[i], rv = sv[i + stride], rm = sm[i + stride];
            sv[i] = lv * rm + rv;
            sm[i] = lm * rm;
        }
        __syncthreads();
    }
    if (t == 0) {
        if (gridDim.x == 1) {
            out[0] = (long long)(1ull * sm[0] + sv[0]);
            if (seq_out != nullptr) {  // `out` is in host-mapped memory and the host polls seq_out instead of waiting for the stream
                __threadfence_system();
                *seq_out = seq;
            }
        } else {
            partial[2 * blockIdx.x] = sv[0];
            partial[2 * blockIdx.x + 1] = sm[0];
        }
    }
}

// the workgroups' (v, m) pairs, in order, applied to h = 1: one workgroup of 512 threads, the same ordered tree as inside
// config_id_kernel (one thread folding 512 pairs out of memory one after the other took 53 us, a third of the configuration id's time)
__global__ __launch_bounds__(512) void config_id_final_kernel(const unsigned long long* partial, int n_pairs, long long* out, volatile unsigned int* seq_out = nullptr,
                                                              unsigned int seq = 0u) {
    __shared__ unsigned long long sv[512], sm[512];
    const int t = (int)threadIdx.x, T = (int)blockDim.x;
    // thread t folds a contiguous run of pairs (n_pairs <= 512 in the engine: one each), then the tree
    const int per = (n_pairs + T - 1) / T, beg = min(n_pairs, t * per), end = min(n_pairs, beg + per);
    unsigned long long v = 0ull, m = 1ull;
    for (int i = beg; i < end; ++i) {
        v = v * partial[2 * i + 1] + partial[2 * i];
        m *= partial[2 * i + 1];
    }
    sv[t] = v;
    sm[t] = m;
    __syncthreads();
    for (int stride = 1; stride < T; stride <<= 1) {
        const int i = 2 * stride * t;
        if (i + stride < T) {
            const unsigned long long lv = sv[i], lm = sm[i], rv = sv[i + stride], rm = sm[i + stride];
            sv[i] = lv * rm + rv;
            sm[i] = lm * rm;
        }
        __syncthreads();
    }
    if (t == 0) {
        out[0] = (long long)(1ull * sm[0] + sv[0]);
        if (seq_out != nullptr) {  // (see config_id_kernel)
            __threadfence_system();
            *seq_out = seq;
        }
    }
}

}  // namespace rapid
