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
    const bool votes = r < n_receivers && prop_count[r] > 0;
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

// out[0] = bucket with the most votes, out[1] = its count, out[2] = total voters.  One block.
__global__ void vote_winner_kernel(const unsigned long long* hist, unsigned long long* out) {
    __shared__ unsigned long long best_cnt[256];
    __shared__ unsigned int best_idx[256];
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
    if (t == 0) {
        out[0] = best_idx[0];
        out[1] = best_cnt[0];
        out[2] = hist[kVoteBuckets];
    }
}

// For the winning bucket: mm[0] = max fingerprint, mm[1] = max of ~fingerprint (i.e. ~min), mm[2] = max of
// ~(receiver index) over local voters in the bucket (i.e. ~lowest local representative); all start at 0.
__global__ void vote_bucket_minmax_kernel(const unsigned long long* fp, const int* prop_count, int n_receivers,
                                          unsigned long long salt, const unsigned long long* winner,
                                          unsigned long long* mm) {
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n_receivers || prop_count[r] <= 0) return;
    const unsigned long long f = fp[r];
    if (vote_bucket(f, salt) != (unsigned int)winner[0]) return;
    atomicMax(&mm[0], f);
    atomicMax(&mm[1], ~f);
    atomicMax(&mm[2], ~(unsigned long long)r);
}

// Element-wise verification: every local voter whose fingerprint equals `want` must hold exactly the list
// ref[0..ref_n).  mismatch[0] counts offenders; mismatch[1] counts the verified voters.
__global__ void vote_verify_kernel(const unsigned long long* fp, const int* prop_count, const int* props, int prop_cap,
                                   int n_receivers, unsigned long long want, const int* ref, int ref_n,
                                   unsigned long long* mismatch) {
    const int r = (int)blockIdx.x;
    if (r >= n_receivers || prop_count[r] <= 0 || fp[r] != want) return;
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    if (prop_count[r] != ref_n) {
        if (threadIdx.x == 0) bad = 1;
    } else {
        const int* mine = props + (long long)r * prop_cap;
        for (int i = (int)threadIdx.x; i < ref_n; i += (int)blockDim.x)
            if (mine[i] != ref[i]) bad = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (bad) atomicAdd(&mismatch[0], 1ull);
        atomicAdd(&mismatch[1], 1ull);
    }
}

}  // namespace rapid
