// Fast-round vote counting for the whole population (R/FastPaxos.java:125-156): every receiver that announced
// a proposal votes for it; identical proposals (same ordered endpoint list <=> same set, since the list is the
// set sorted by the ring-0 comparator, R/MembershipService.java:346-348) are counted and the winner is tested
// against the fast quorum N - floor((N-1)/4).
//
// Proposals are identified by a 64-bit commutative fingerprint computed by the tally kernel and counted in a
// positional histogram (bucket = hash(fingerprint, salt) mod B) so that ranks can sum their histograms with one
// all-reduce.  Nothing is decided on fingerprints alone: the winning bucket must be pure (min == max fingerprint
// over all ranks) and every voter's element list is compared with the representative's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rapid {

constexpr int kVoteBuckets = 1 << 14;

// the workgroup's dynamic LDS segment (tests/emu/ supplies its own definition: a CPU build has no such thing)
#ifndef RAPID_DYNAMIC_LDS
#define RAPID_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

__device__ __forceinline__ unsigned int vote_bucket(unsigned long long fp, unsigned long long salt) {
    unsigned long long x = fp ^ (salt * 0xD6E8FEB86659FD93ull);
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return (unsigned int)(x & (kVoteBuckets - 1));
}

// hist[b] += number of receivers whose proposal falls into bucket b; hist[kVoteBuckets] += total voters.
// Votes are aggregated per wave before touching global memory (most receivers share one fingerprint).
__global__ void vote_histogram_kernel(const unsigned long long* fp, const int* prop_count, int n_receivers,
                                      unsigned long long salt, unsigned long long* hist) {
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int lane = (int)(threadIdx.x & 63);
    const bool votes = r < n_receivers && prop_count[r] != 0;  // -1 = proposal larger than max_cut: still a vote
    const unsigned int b = votes ? vote_bucket(fp[r], salt) : 0xFFFFFFFFu;
    unsigned long long todo = __ballot(votes);
    const unsigned long long all = todo;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned int lb = (unsigned int)__shfl((int)b, leader, 64);
        const unsigned long long peers = __ballot(votes && b == lb);
        if (lane == leader) atomicAdd(&hist[lb], (unsigned long long)__popcll(peers));
        todo &= ~peers;
    }
    if (all && lane == __ffsll((long long)all) - 1) atomicAdd(&hist[kVoteBuckets], (unsigned long long)__popcll(all));
}

// out[0] = bucket with the most votes, out[1] = its count, out[2] = total voters, out[3] = non-empty buckets.  One block.
__global__ void vote_winner_kernel(const unsigned long long* hist, unsigned long long* out) {
    __shared__ unsigned long long best_cnt[1024];
    __shared__ unsigned int best_idx[1024];
    const int t = (int)threadIdx.x;
    unsigned long long bc = 0;
    unsigned int bi = 0;
    for (int b = t; b < kVoteBuckets; b += (int)blockDim.x) {
        const unsigned long long c = hist[b];
        if (c > bc) {
            bc = c;
            bi = (unsigned int)b;
        }
    }
    best_cnt[t] = bc;
    best_idx[t] = bi;
    __syncthreads();
    for (int s = (int)blockDim.x / 2; s > 0; s >>= 1) {
        if (t < s) {
            if (best_cnt[t + s] > best_cnt[t] || (best_cnt[t + s] == best_cnt[t] && best_idx[t + s] < best_idx[t])) {
                best_cnt[t] = best_cnt[t + s];
                best_idx[t] = best_idx[t + s];
            }
        }
        __syncthreads();
    }
    __shared__ unsigned int nz[1024];
    unsigned int z = 0;
    for (int b = t; b < kVoteBuckets; b += (int)blockDim.x) z += hist[b] != 0ull ? 1u : 0u;
    nz[t] = z;
    __syncthreads();
    for (int s2 = (int)blockDim.x / 2; s2 > 0; s2 >>= 1) {
        if (t < s2) nz[t] += nz[t + s2];
        __syncthreads();
    }
    if (t == 0) {
        out[0] = best_idx[0];
        out[1] = best_cnt[0];
        out[2] = hist[kVoteBuckets];
        out[3] = nz[0];  // non-empty buckets ~ distinct proposals
    }
}

// For the winning bucket (all slots start at 0): mm[0] = max fingerprint, mm[1] = max of ~fingerprint (i.e. ~min),
// mm[3] = my_tag if this rank has a voter in the bucket, mm[4] = max of ~(receiver index) over the local voters.
__global__ void vote_bucket_minmax_kernel(const unsigned long long* fp, const int* prop_count, int n_receivers,
                                          unsigned long long salt, const unsigned long long* winner,
                                          unsigned long long* mm, unsigned long long my_tag) {
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const bool in = r < n_receivers && prop_count[r] != 0 && vote_bucket(fp[r], salt) == (unsigned int)winner[0];
    const unsigned long long f = in ? fp[r] : 0ull;
    // wave-level pre-reduction: most voters share one fingerprint, so one lane per wave touches global memory
    unsigned long long fmx = f, fmn = in ? ~f : 0ull, rr = in ? ~(unsigned long long)r : 0ull;
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long a = __shfl_xor(fmx, off, 64), b = __shfl_xor(fmn, off, 64), c = __shfl_xor(rr, off, 64);
        fmx = a > fmx ? a : fmx;
        fmn = b > fmn ? b : fmn;
        rr = c > rr ? c : rr;
    }
    if ((threadIdx.x & 63u) == 0 && rr != 0ull) {
        atomicMax(&mm[0], fmx);
        atomicMax(&mm[1], fmn);
        atomicMax(&mm[3], my_tag);  // this rank holds a representative
        atomicMax(&mm[4], rr);      // ~(lowest local representative) -- never all-reduced
    }
}

// ref[0] = size (INT_MAX: larger than prop_cap), ref[1..cap] = node list of the proposal every voter must equal.  The rank that owns the lowest
// representative (owner_tag == my_tag, or a single-rank run) copies it from its receiver; every other rank writes
// zeros, so that a max all-reduce of ref[] hands the list to everybody without a host round trip.
// mm[4] = ~(local representative receiver), 0 if this rank has no voter in the winning bucket.
__global__ void vote_prepare_ref_kernel(const unsigned long long* mm, unsigned long long my_tag, int single_rank,
                                        const int* prop_count, const int* props, int prop_cap, int* ref) {
    const bool owner = mm[4] != 0ull && (single_rank || mm[3] == my_tag);
    const int rep = owner ? (int)(~mm[4]) : 0;
    int n = owner ? prop_count[rep] : 0;
    const bool overflow = n < 0;  // the representative's proposal overflowed max_cut: reported by the host
    if (overflow) n = 0;
    for (int i = (int)threadIdx.x; i < prop_cap; i += (int)blockDim.x)
        ref[1 + i] = (owner && i < n) ? props[(long long)rep * prop_cap + i] : 0;
    // INT_MAX, not -1: the other ranks contribute zeros to the max-reduce of ref[]
    if (threadIdx.x == 0) ref[0] = owner ? (overflow ? 0x7FFFFFFF : n) : 0;
}

// Element-wise verification: every local voter whose fingerprint equals mm[0] (== the winning bucket's only
// fingerprint when it is pure) must hold exactly the list ref[1..ref[0]].  mismatch[0] counts offenders, mismatch[1]
// the verified voters.  One wavefront per receiver; the counts are reduced per workgroup first -- thousands of waves
// adding to the same two words would queue up behind each other for longer than the comparison takes.
// publish != nullptr (population held by one rank): the LAST workgroup to finish copies the round's answer --
// res[0 .. res_words) and ref[0 .. 1 + size] -- into host-mapped memory, so the host reads it after synchronising
// without a copy being enqueued; done[0] counts the finished workgroups and is left at zero.
// bits != nullptr (only with rep_in_res: the tally kernel wrote every voter's proposal as a bitmap over the round's hot slots,
// tally_kernel.h: TallyParams::bitmaps): the comparison runs on the bitmaps -- bits_words 64-bit words per receiver, equal
// bitmaps <=> equal node lists (slot -> node is injective within a round), 64 bytes per receiver at C3b instead of a 2 KB list
// (19 MB re-read by 594 workgroups -> 0.6 MB by ten) -- with ONE THREAD per receiver (launch ceil(R / block) workgroups).
__global__ void vote_verify_kernel(const unsigned long long* fp, const int* prop_count, const int* props, int prop_cap,
                                   int n_receivers, const unsigned long long* mm, const int* ref,
                                   unsigned long long* mismatch, const unsigned long long* res, int res_words,
                                   unsigned int* done, volatile unsigned long long* publish, volatile unsigned int* seq_out,
                                   unsigned int seq, int rep_in_res, const unsigned int* tally_errors,
                                   const unsigned long long* bits = nullptr, int bits_words = 0, int bits_wave = 0) {
    // bits_wave != 0 (rounds with thousands of hot slots: bitmaps of kilobytes): a WAVE per receiver walks the bitmap, 512
    // bytes at a time (launch ceil(R * 64 / block) workgroups) -- one thread per receiver put all of a 10^6-node round's 1,024
    // bitmaps of 1.9 KB through one workgroup: 0.2 ms
    __shared__ unsigned int s_bad, s_seen, s_last;
    if (threadIdx.x == 0) {
        s_bad = 0u;
        s_seen = 0u;
        s_last = 0u;
    }
    __syncthreads();
    const bool by_bits = bits != nullptr && rep_in_res != 0;
    const bool thread_per_rx = by_bits && bits_wave == 0;
    const int r = thread_per_rx ? (int)(blockIdx.x * blockDim.x + threadIdx.x) : (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = thread_per_rx ? 0 : (int)(threadIdx.x & 63u);
    // The proposal the voters are compared with.  After a counting kernel: the winning bucket's only fingerprint mm[0] and
    // the representative's list it copied out (ref[0] = size, ref[1..]).  rep_in_res -- no counting kernel ran, res[] came
    // from the tally kernel (tally_kernel.h: vote_res): the CANDIDATE is the proposal of the lowest voter res[0], read in
    // place; mismatch[1] then is the candidate's vote count, and if that is a quorum (N - floor((N-1)/4) > N/2 voters,
    // R/FastPaxos.java:145-150) no other proposal can have one: the round is settled without a histogram.
    int ref_n;
    const int* ref_list;
    unsigned long long cand;
    bool have = true;
    // (this kernel is a chain of memory round trips, not bandwidth: the receiver's own count and fingerprint are requested
    // together with the candidate's index, before anything depends on it, and the lists are compared four steps at a time)
    const bool in_range = r < n_receivers;
    const int my_count = in_range ? prop_count[r] : 0;
    const unsigned long long my_fp = in_range ? fp[r] : 0ull;
    unsigned int rep = 0u;
    if (rep_in_res) {
        rep = (unsigned int)res[0];
        have = rep < (unsigned int)n_receivers;
        cand = have ? fp[rep] : 0ull;
        ref_n = have ? prop_count[rep] : 0;
        ref_list = props + (long long)(have ? rep : 0u) * prop_cap;
    } else {
        cand = mm[0];
        ref_n = ref[0];
        ref_list = ref + 1;
    }
    const bool voter = have && in_range && my_count != 0 && my_fp == cand;
    if (by_bits && bits_wave != 0) {
        bool bad = false;
        if (voter) {  // (the same for all lanes of the wave: they asked about the same receiver)
            const unsigned long long* const mine = bits + (long long)r * bits_words;
            const unsigned long long* const cnd = bits + (long long)rep * bits_words;
            for (int i0 = lane; i0 < bits_words; i0 += 256) {
                unsigned long long a[4], b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = i0 + 64 * j < bits_words ? mine[i0 + 64 * j] : 0ull;
                    b[j] = i0 + 64 * j < bits_words ? cnd[i0 + 64 * j] : 0ull;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) bad |= a[j] != b[j];
            }
        }
        const bool wave_bad = __ballot(voter && bad) != 0ull;
        if (lane == 0 && voter) {
            atomicAdd(&s_seen, 1u);
            if (wave_bad) atomicAdd(&s_bad, 1u);
        }
    } else if (by_bits) {
        bool bad = false;
        if (voter) {
            const unsigned long long* const mine = bits + (long long)r * bits_words;
            const unsigned long long* const cnd = bits + (long long)rep * bits_words;
            for (int i0 = 0; i0 < bits_words; i0 += 8) {
                unsigned long long a[8], b[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a[j] = i0 + j < bits_words ? mine[i0 + j] : 0ull;
                    b[j] = i0 + j < bits_words ? cnd[i0 + j] : 0ull;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) bad |= a[j] != b[j];
            }
        }
        const unsigned long long m_seen = __ballot(voter), m_bad = __ballot(voter && bad);
        if ((threadIdx.x & 63u) == 0u) {
            if (m_bad) atomicAdd(&s_bad, (unsigned int)__popcll(m_bad));
            if (m_seen) atomicAdd(&s_seen, (unsigned int)__popcll(m_seen));
        }
    } else if (voter) {
        bool bad = my_count != ref_n;
        if (!bad) {
            const int* mine = props + (long long)r * prop_cap;
            for (int i0 = 0; i0 < ref_n; i0 += 256) {
                int a[4], b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + 64 * j + lane;
                    a[j] = i < ref_n ? mine[i] : 0;
                    b[j] = i < ref_n ? ref_list[i] : 0;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) bad |= a[j] != b[j];
            }
        }
        const unsigned long long any_bad = __ballot(bad);
        if (lane == 0) {
            if (any_bad) atomicAdd(&s_bad, 1u);
            atomicAdd(&s_seen, 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (publish == nullptr) {
            if (s_bad) atomicAdd(&mismatch[0], (unsigned long long)s_bad);
            if (s_seen) atomicAdd(&mismatch[1], (unsigned long long)s_seen);
        } else {
            // ONE atomic per workgroup: finished workgroups << 44 | offenders << 22 | verified voters (a rank holds at most
            // 262,144 receivers on this path); whoever brings the count to gridDim.x has the totals in its hands -- nothing the
            // other workgroups WROTE is read by it, so no fence is needed either
            unsigned long long* const packed = reinterpret_cast<unsigned long long*>(done);
            const unsigned long long mine = (1ull << 44) | ((unsigned long long)s_bad << 22) | (unsigned long long)s_seen;
            const unsigned long long total = atomicAdd(packed, mine) + mine;
            if ((total >> 44) == (unsigned long long)gridDim.x) {
                s_last = 1u;
                mismatch[0] = (total >> 22) & 0x3FFFFFull;
                mismatch[1] = total & 0x3FFFFFull;
                *packed = 0ull;
            }
        }
    }
    __syncthreads();
    if (publish != nullptr && s_last != 0u) {
        int n = (ref_n < 0 || ref_n > prop_cap) ? 0 : ref_n;
        // (ordinary stores, made visible by the fence below ahead of the one word the host polls: see vote_merge_kernel)
        unsigned long long* const out = const_cast<unsigned long long*>(publish);
        int* const pref = reinterpret_cast<int*>(out + res_words);
        if (threadIdx.x == 0) pref[0] = ref_n;
        for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) pref[1 + i] = ref_list[i];
        for (int i = (int)threadIdx.x; i < res_words; i += (int)blockDim.x) {
            unsigned long long v = __hip_atomic_load(&res[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // mismatch[] was updated by atomics in L2
            if (i == 8 && tally_errors != nullptr) v = (unsigned long long)tally_errors[0];  // the tally kernel's sticky error word, final by now
            if (rep_in_res) {  // complete the answer in the counting kernel's layout: votes of the candidate, its fingerprint
                const unsigned long long votes = __hip_atomic_load(&res[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long voters = __hip_atomic_load(&res[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (i == 1) v = votes;
                if (i == 3) v = voters == 0ull ? 0ull : (votes == voters ? 1ull : 2ull);  // 2 = at least two proposals
                if (i == 4) v = cand;
                if (i == 5) v = ~cand;
            }
            out[i] = v;
        }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0 && seq_out != nullptr) *seq_out = seq;  // the host polls this word instead of waiting for the stream
    }
}

// The whole count for a population held by ONE rank, in one workgroup: histogram of the fingerprints in LDS (64 KiB),
// winner, purity of the winning bucket (min == max fingerprint), the representative = the lowest receiver voting for
// it, its node list.  res[0..3] as vote_winner_kernel's out[], res[4] = max fingerprint of the winning bucket,
// res[5] = max of ~fingerprint (~min), res[6] = mismatch[0] = 0, res[7] = mismatch[1] = 0 (filled by
// vote_verify_kernel, which follows on the stream and reads res[4] as its mm[0]), res[8] = tally_errors[0]; ref[0] = size of the
// representative's list (-1: larger than prop_cap), ref[1..] = the list.  A few tens of thousands of receivers take a
// few microseconds; larger populations use the multi-kernel path.
__global__ __launch_bounds__(1024) void vote_count_local_kernel(const unsigned long long* fp, const int* prop_count, const int* props,
                                                                int prop_cap, int n_receivers, unsigned long long salt,
                                                                const unsigned int* tally_errors, unsigned long long* res, int* ref) {
    RAPID_DYNAMIC_LDS(vote_smem);
    unsigned int* const hist = reinterpret_cast<unsigned int*>(vote_smem);                          // [kVoteBuckets]
    unsigned long long* const red64 = reinterpret_cast<unsigned long long*>(hist + kVoteBuckets);  // [2][16] per-wave partials
    unsigned int* const red32 = reinterpret_cast<unsigned int*>(red64 + 32);                       // [4][16]
    __shared__ unsigned int s_win_bucket, s_win_count, s_total, s_nz, s_rep, s_unanimous;
    __shared__ unsigned long long s_fmax, s_fminc;
    const int t = (int)threadIdx.x, T = (int)blockDim.x, lane = t & 63, wv = t >> 6, nw = T >> 6;

    // max fingerprint, max ~fingerprint and lowest index over the voters `sel` picks; the three land in s_fmax, s_fminc, s_rep
    auto range_of = [&](auto sel, unsigned int* count_out) {
        unsigned long long fmx = 0ull, fmn = 0ull;
        unsigned int rep = 0xFFFFFFFFu, cnt = 0u;
        for (int r = t; r < n_receivers; r += T) {
            if (prop_count[r] != 0) {  // -1 = proposal larger than max_cut: still a vote
                const unsigned long long f = fp[r];
                if (sel(f)) {
                    ++cnt;
                    fmx = f > fmx ? f : fmx;
                    fmn = ~f > fmn ? ~f : fmn;
                    rep = (unsigned int)r < rep ? (unsigned int)r : rep;
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long a = __shfl_xor(fmx, off, 64), b = __shfl_xor(fmn, off, 64);
            const unsigned int c = (unsigned int)__shfl_xor((int)rep, off, 64);
            fmx = a > fmx ? a : fmx;
            fmn = b > fmn ? b : fmn;
            rep = c < rep ? c : rep;
            cnt += (unsigned int)__shfl_xor((int)cnt, off, 64);
        }
        __syncthreads();  // the reduction arrays may still be read from a previous use
        if (lane == 0) {
            red64[wv] = fmx;
            red64[16 + wv] = fmn;
            red32[wv] = rep;
            red32[16 + wv] = cnt;
        }
        __syncthreads();
        if (t == 0) {
            unsigned long long a = 0, b = 0;
            unsigned int c = 0xFFFFFFFFu, m = 0u;
            for (int i = 0; i < nw; ++i) {
                a = red64[i] > a ? red64[i] : a;
                b = red64[16 + i] > b ? red64[16 + i] : b;
                c = red32[i] < c ? red32[i] : c;
                m += red32[16 + i];
            }
            s_fmax = a;
            s_fminc = b;
            s_rep = c;
            *count_out = m;
        }
        __syncthreads();
    };

    // The common round first: every voter holds the same fingerprint (one pass over the fingerprints, no histogram).
    range_of([](unsigned long long) { return true; }, &s_total);
    if (t == 0) {
        s_unanimous = (s_total == 0u || s_fmax == ~s_fminc) ? 1u : 0u;
        if (s_unanimous) {
            s_win_bucket = s_total != 0u ? vote_bucket(s_fmax, salt) : 0u;
            s_win_count = s_total;
            s_nz = s_total != 0u ? 1u : 0u;
        }
    }
    __syncthreads();
    if (s_unanimous == 0u) {
        // several proposals: histogram of the fingerprints over kVoteBuckets positions in LDS
        for (int b = t; b < kVoteBuckets; b += T) hist[b] = 0u;
        __syncthreads();
        for (int r = t; r < n_receivers; r += T)
            if (prop_count[r] != 0) atomicAdd(&hist[vote_bucket(fp[r], salt)], 1u);
        __syncthreads();
        // winner: most votes, lowest bucket among equals; number of non-empty buckets
        unsigned int bc = 0, bi = 0, nz = 0;
        for (int b = t; b < kVoteBuckets; b += T) {
            const unsigned int c = hist[b];
            nz += c != 0u ? 1u : 0u;
            if (c > bc) {
                bc = c;
                bi = (unsigned int)b;
            }
        }
        unsigned long long key = ((unsigned long long)bc << 32) | (unsigned long long)(0xFFFFFFFFu - bi);  // max key = max count, then min bucket
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off, 64);
            key = o > key ? o : key;
            nz += (unsigned int)__shfl_xor((int)nz, off, 64);
        }
        if (lane == 0) {
            red64[wv] = key;
            red32[wv] = nz;
        }
        __syncthreads();
        if (t == 0) {
            unsigned long long k = 0;
            unsigned int z = 0;
            for (int i = 0; i < nw; ++i) {
                k = red64[i] > k ? red64[i] : k;
                z += red32[i];
            }
            s_win_count = (unsigned int)(k >> 32);
            s_win_bucket = 0xFFFFFFFFu - (unsigned int)(k & 0xFFFFFFFFull);
            s_nz = z;
        }
        __syncthreads();
        // the winning bucket: max fingerprint, max ~fingerprint, lowest voter
        const unsigned int wb = s_win_bucket;
        __shared__ unsigned int s_ignored;
        range_of([&](unsigned long long f) { return vote_bucket(f, salt) == wb; }, &s_ignored);
    }
    if (t == 0) {
        res[0] = s_win_bucket;
        res[1] = s_win_count;
        res[2] = s_total;
        res[3] = s_nz;
        res[4] = s_fmax;
        res[5] = s_fminc;
        res[6] = 0ull;
        res[7] = 0ull;
        res[8] = (unsigned long long)tally_errors[0];  // the tally kernel's sticky error word rides along
        res[9] = 0ull;                                  // vote_verify_kernel's count of finished workgroups
    }
    const bool have = s_rep != 0xFFFFFFFFu;
    int n = have ? prop_count[s_rep] : 0;
    if (n < 0) n = -1;  // the representative's proposal overflowed max_cut: reported by the host
    for (int i = t; i < prop_cap; i += T) ref[1 + i] = (have && i < n) ? props[(long long)s_rep * prop_cap + i] : 0;
    if (t == 0) ref[0] = n;
}

// ---- votes accumulated ACROSS LAUNCHES: a population taken tile by tile (rapid_sim_round_tiled; BASELINE configs[4]: 10^6 receivers
// do not fit one launch, and the fast quorum N - floor((N-1)/4) needs three quarters of them) -------------------------------------
// The round's accumulator lives in device memory for the whole round:
//   acc[0] = a candidate proposal has been picked, acc[1] = verified votes for it, acc[2] = voters so far, acc[3] = voters that share
//   the candidate's fingerprint but not its proposal (a collision: reported, never counted), acc[4] = the candidate's fingerprint,
//   acc[5] = its size (-1: larger than prop_cap), acc[6] = the fingerprint the candidate must have (0: the first voter's proposal),
//   acc[7] = the tally's sticky error word; then the candidate as a bitmap over the round's hot slots (acc_bits[bits_words]) and as
//   a node list (acc_list[prop_cap]).
// After every tile's tally: vote_acc_pick_kernel (one workgroup) adds the tile's voters and, while there is no candidate, takes the
// tile's lowest voter (tile_res[0], gathered by the tally kernel itself) -- or, in a counting pass for a given fingerprint, the lowest
// voter that holds it -- as the candidate; vote_acc_count_kernel then compares every voter of the tile that holds the candidate's
// fingerprint with the candidate's BITMAP, word for word (slot -> node is injective within a round: equal bitmaps <=> equal
// proposals; nothing is counted on fingerprints alone), one wave per receiver.  The tile's proposals are gone with the next tile;
// what survives is the count -- which is all R/FastPaxos.java:141-150 keeps per proposal -- and every receiver's fingerprint
// (for the exact plurality when the candidate has no quorum).
constexpr int kVoteAccWords = 8;
__global__ __launch_bounds__(1024) void vote_acc_pick_kernel(const unsigned long long* tile_res, const unsigned long long* fp, const int* prop_count,
                                                             const int* props, int prop_cap, const unsigned long long* bits, int bits_words,
                                                             int n_receivers, const unsigned int* tally_errors, unsigned long long* acc,
                                                             unsigned long long* acc_bits, int* acc_list) {
    __shared__ unsigned int s_rep;
    const int t = (int)threadIdx.x, T = (int)blockDim.x;
    const unsigned long long target = acc[6];
    const bool have = acc[0] != 0ull;
    if (t == 0) s_rep = 0xFFFFFFFFu;
    __syncthreads();
    if (!have) {
        if (target == 0ull) {
            if (t == 0 && tile_res[2] != 0ull) s_rep = (unsigned int)tile_res[0];
        } else {  // the lowest voter of this tile that holds the wanted fingerprint
            unsigned int best = 0xFFFFFFFFu;
            for (int r = t; r < n_receivers; r += T)
                if (prop_count[r] != 0 && fp[r] == target) best = min(best, (unsigned int)r);
            if (best != 0xFFFFFFFFu) atomicMin(&s_rep, best);
        }
    }
    __syncthreads();
    const unsigned int rep = s_rep;
    const bool pick = !have && rep < (unsigned int)n_receivers;
    if (pick) {
        const int n = prop_count[rep];
        for (int i = t; i < bits_words; i += T) acc_bits[i] = bits[(long long)rep * bits_words + i];
        for (int i = t; i < prop_cap; i += T) acc_list[i] = (n > 0 && i < n) ? props[(long long)rep * prop_cap + i] : 0;
        if (t == 0) {
            acc[4] = fp[rep];
            acc[5] = (unsigned long long)(long long)(n < 0 ? -1 : n);
        }
    }
    __syncthreads();
    if (t == 0) {
        acc[2] += tile_res[2];
        acc[7] |= (unsigned long long)tally_errors[0];
        if (pick) {
            __threadfence();
            acc[0] = 1ull;
        }
    }
}

__global__ __launch_bounds__(1024) void vote_acc_count_kernel(const unsigned long long* fp, const int* prop_count, const unsigned long long* bits,
                                                              int bits_words, int n_receivers, unsigned long long* acc,
                                                              const unsigned long long* acc_bits) {
    __shared__ unsigned int s_ok, s_bad;
    if (threadIdx.x == 0) {
        s_ok = 0u;
        s_bad = 0u;
    }
    __syncthreads();
    const int r = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = (int)(threadIdx.x & 63u);
    const bool have = acc[0] != 0ull;
    const unsigned long long cand = acc[4];
    const bool voter = have && r < n_receivers && prop_count[r] != 0 && fp[r] == cand;
    bool bad = false;
    if (voter) {  // (the same for all lanes of the wave)
        const unsigned long long* const mine = bits + (long long)r * bits_words;
        for (int i0 = lane; i0 < bits_words; i0 += 256) {
            unsigned long long a[4], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = i0 + 64 * j < bits_words ? mine[i0 + 64 * j] : 0ull;
                b[j] = i0 + 64 * j < bits_words ? acc_bits[i0 + 64 * j] : 0ull;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) bad |= a[j] != b[j];
        }
    }
    const bool wave_bad = __ballot(voter && bad) != 0ull;
    if (lane == 0 && voter) atomicAdd(wave_bad ? &s_bad : &s_ok, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_ok) atomicAdd(&acc[1], (unsigned long long)s_ok);
        if (s_bad) atomicAdd(&acc[3], (unsigned long long)s_bad);
    }
}

// The round's answer block in the layout a rank contributes to the all-gather (res[res_words] + ref[1 + prop_cap]): what
// vote_count_local_kernel + vote_verify_kernel leave for a population held in one launch.
__global__ void vote_acc_finish_kernel(const unsigned long long* acc, const int* acc_list, int prop_cap, unsigned long long* res, int* ref) {
    const long long n = (long long)acc[5];
    const int len = acc[0] == 0ull ? 0 : (n < 0 ? -1 : (int)n);
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < prop_cap; i += (int)(gridDim.x * blockDim.x)) ref[1 + i] = i < len ? acc_list[i] : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long votes = acc[1], voters = acc[2], fpc = acc[0] != 0ull ? acc[4] : 0ull;
        res[0] = 0ull;
        res[1] = votes;
        res[2] = voters;
        res[3] = voters == 0ull ? 0ull : (votes == voters ? 1ull : 2ull);
        res[4] = fpc;
        res[5] = ~fpc;
        res[6] = acc[3];
        res[7] = votes;
        res[8] = acc[7];
        res[9] = 0ull;
        ref[0] = len;
    }
}

// Populations sharded over ranks: every rank settles its own voters first -- the candidate proposal (the winning bucket's
// after a counting kernel, the lowest voter's when the statistics came from the tally kernel), its verified vote count
// res[1] = res[7], the rank's voters res[2] -- ONE all-gather moves the ranks' answers (res[res_words] + ref[1 + prop_cap],
// `seg_words` 64-bit words per rank) to everybody, and this kernel (one workgroup, on every rank, over identical input)
// merges them.  The merge is exact when every voting rank's candidate is the same proposal (same fingerprint, pure, every
// one of its voters verified element-wise, lists equal element for element across ranks) and either that proposal has a
// quorum -- then no other can (quorum > N / 2, R/FastPaxos.java:145-150), whatever the remaining voters hold -- or every
// voter of every rank holds it: votes = the sum of the ranks' (R/FastPaxos.java:141-150).  The answer is published to
// host-mapped memory in the single-rank format with res[9] = 1.  Anything else -- candidates differ between ranks, no
// quorum among several proposals, an impure bucket -- publishes res[9] = 2 and the host goes through the histogram
// all-reduce path, on every rank alike (they all merged the same data).  A representative larger than prop_cap anywhere
// is reported as ref[0] = -1.
__global__ __launch_bounds__(256) void vote_merge_kernel(const unsigned long long* gathered, int n_ranks, int seg_words,
                                                          int res_words, int prop_cap, long long quorum,
                                                          volatile unsigned long long* publish, volatile unsigned int* seq_out,
                                                          unsigned int seq) {
    __shared__ int s_lead, s_simple, s_overflow;
    __shared__ unsigned long long s_votes, s_voters, s_err;
    if (threadIdx.x == 0) {
        int lead = -1, simple = 1, overflow = 0;
        unsigned long long votes = 0, voters = 0, err = 0, lead_fp = 0;
        for (int k = 0; k < n_ranks; ++k) {
            const unsigned long long* res = gathered + (long long)k * seg_words;
            const int* ref = reinterpret_cast<const int*>(res + res_words);
            err |= res[8];
            voters += res[2];
            if (res[2] == 0ull) continue;  // nobody on this rank proposed
            votes += res[1];
            if (res[1] == 0ull || res[4] != ~res[5] || res[6] != 0ull || res[7] != res[1]) simple = 0;
            if (ref[0] < 0) overflow = 1;
            if (lead < 0) {
                lead = k;
                lead_fp = res[4];
            } else if (res[4] != lead_fp || ref[0] != reinterpret_cast<const int*>(gathered + (long long)lead * seg_words + res_words)[0]) {
                simple = 0;
            }
        }
        if (!((long long)votes >= quorum || votes == voters)) simple = 0;  // several proposals and no quorum: the exact plurality is owed
        s_lead = lead;
        s_simple = simple;
        s_overflow = overflow;
        s_votes = votes;
        s_voters = voters;
        s_err = err;
    }
    __syncthreads();
    const int lead = s_lead;
    const int* const lref = lead < 0 ? nullptr : reinterpret_cast<const int*>(gathered + (long long)lead * seg_words + res_words);
    int n = lead < 0 ? 0 : lref[0];
    n = (n < 0 || n > prop_cap) ? 0 : n;
    int differ = 0;
    if (s_simple != 0 && s_overflow == 0) {
        for (int k = lead + 1; k < n_ranks; ++k) {
            const unsigned long long* res = gathered + (long long)k * seg_words;
            if (res[2] == 0ull) continue;
            const int* ref = reinterpret_cast<const int*>(res + res_words);
            for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) differ |= ref[1 + i] != lref[1 + i];
        }
    }
    differ = __syncthreads_or(differ);
    // (the answer block is written with ordinary stores -- a volatile store to the host-mapped page is waited for before the next
    // one is issued, eleven round trips over the host link from one thread -- and made visible by the fence below, ahead of the
    // one word the host polls)
    unsigned long long* const out = const_cast<unsigned long long*>(publish);
    int* const pref = reinterpret_cast<int*>(out + res_words);
    for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) pref[1 + i] = lref[1 + i];
    if (threadIdx.x == 0) {
        const unsigned long long* lres = lead < 0 ? gathered : gathered + (long long)lead * seg_words;
        const bool ok = s_simple != 0 && differ == 0;
        pref[0] = s_overflow ? -1 : n;
        out[0] = lead < 0 ? 0ull : lres[0];
        out[1] = s_votes;
        out[2] = s_voters;
        out[3] = lead < 0 ? 0ull : (s_votes == s_voters ? 1ull : 2ull);
        out[4] = lead < 0 ? 0ull : lres[4];
        out[5] = lead < 0 ? ~0ull : lres[5];
        out[6] = 0ull;
        out[7] = s_votes;
        out[8] = s_err;
        out[9] = ok ? 1ull : 2ull;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && seq_out != nullptr) *seq_out = seq;
}

}  // namespace rapid
