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

// Removal-only view change: the ring order of the remaining members does not change, so every ring is its old self
// minus the deleted nodes (R/MembershipView.java:167-201 removes the endpoint from each TreeSet, nothing moves).
// One workgroup per ring: stable compaction of the (sortable key, node) pairs by the new member flags.
__global__ void ring_compact_kernel(const int* ring_in, const unsigned long long* skeys_in, int m_old,
                                    const unsigned char* member, int* ring_out, unsigned long long* skeys_out, int m_new) {
    __shared__ int s_count[1024];
    const int k = (int)blockIdx.x, T = (int)blockDim.x, t = (int)threadIdx.x;
    const int per = (m_old + T - 1) / T;
    const int beg = min(m_old, t * per), end = min(m_old, beg + per);
    const int* rin = ring_in + (long long)k * m_old;
    const unsigned long long* kin = skeys_in + (long long)k * m_old;
    int kept = 0;
    for (int i = beg; i < end; ++i) kept += member[rin[i]] ? 1 : 0;
    s_count[t] = kept;
    __syncthreads();
    // inclusive Hillis-Steele scan over the T partial counts
    for (int off = 1; off < T; off <<= 1) {
        const int v = t >= off ? s_count[t - off] : 0;
        __syncthreads();
        s_count[t] += v;
        __syncthreads();
    }
    int w = s_count[t] - kept;  // exclusive prefix
    int* rout = ring_out + (long long)k * m_new;
    unsigned long long* kout = skeys_out + (long long)k * m_new;
    for (int i = beg; i < end; ++i) {
        const int node = rin[i];
        if (member[node] && w < m_new) {
            rout[w] = node;
            kout[w] = kin[i];
            ++w;
        }
    }
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

// ---- configuration id -------------------------------------------------------------------------------------------
// hash = 1; for id in sorted ids: hash = hash*37 + xx0(high); hash = hash*37 + xx0(low);
//           for ep in ring 0:     hash = hash*37 + xx0(hostname); hash = hash*37 + xx0(port)
// A segment of the sequence maps h -> h * m + v with m = 37^len; segments compose associatively, so the block
// reduces (v, m) pairs.  One block; thread t folds a contiguous slice with Horner's rule.
__global__ void config_id_kernel(const long long* ids_hi, const long long* ids_lo, int n_ids, const int* ring0,
                                 int n_members, const unsigned long long* hx_host0, const unsigned long long* hx_port0,
                                 long long* out) {
    RAPID_DYNAMIC_LDS(smem_raw);
    unsigned long long* sv = reinterpret_cast<unsigned long long*>(smem_raw);
    unsigned long long* sm = sv + blockDim.x;
    const long long total = 2ll * n_ids + 2ll * n_members;
    const int T = (int)blockDim.x;
    const int t = (int)threadIdx.x;
    const long long per = (total + T - 1) / T;
    const long long beg = per * t, end = (beg + per < total) ? beg + per : total;
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
    if (t == 0) out[0] = (long long)(1ull * sm[0] + sv[0]);
}

}  // namespace rapid
