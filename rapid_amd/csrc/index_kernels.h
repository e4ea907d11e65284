// Per-round index over the loaded alert streams (built once per rapid_sim_load_streams / view change, NOT part of
// the timed tally): which subjects does this round's alert set name at all, which of them can ever reach the L
// watermark ("hot"), a dense slot numbering (hot subjects first, ascending node index), and the adjacency among
// hot subjects along the K-ring monitoring graph -- the only (observer, subject, ring) triples for which
// MultiNodeCutDetector.invalidateFailingEdges (R/MultiNodeCutDetector.java:137-164) can ever apply an implicit
// report at any receiver, because both ends need >= L explicit reports (R/MultiNodeCutDetector.java:104-107, 153).
//
// The index is a SUPERSET construction: it ignores the per-alert filter (configuration id, UP/DOWN vs membership),
// so it can only make more subjects "touched"/"hot" than a receiver will see, never fewer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tally_kernel.h"  // the LDS budget formulas: this kernel and the host must agree on where the dictionary will live

namespace rapid {

// gmask[dst] |= ring_mask over every record given (the round's distinct alert set if the host declared one, else
// every delivered record).  After the first few thousand records nearly every bit is already set, so the (possibly
// stale, L1-cached) pre-test avoids almost all atomics.  Also validates the records once: vflags bit0 is set if ANY
// record fails the filter of R/MembershipService.java:644-675 under the current view (or names a node out of range
// or no ring), bit1 if any record is an UP alert.
// Two record sources: the round's declared alert set as it crossed the boundary (20-byte records), or -- when nothing was
// declared -- every delivered record of the resident streams (subject array + core word).
template <bool kSplit>
__global__ void index_touch_kernel(const unsigned char* records, const unsigned int* dstv, long long n_records, int n_nodes,
                                   unsigned int kmask, long long cfg_id, const unsigned char* member, unsigned int* gmask,
                                   unsigned int* vflags) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const unsigned int cfg_lo = (unsigned int)(unsigned long long)cfg_id, cfg_hi = (unsigned int)((unsigned long long)cfg_id >> 32);
    unsigned int f = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_records; i += stride) {
        unsigned int dst, cw;  // cw: the resident core word (tally_kernel.h: core_word)
        bool current;          // the record carries the engine's configuration id
        if (kSplit) {  // the id was compared when the record became resident (kCoreStale); the subject from its own array
            const unsigned int d = dstv[i];  // (the record's first dword may hold the subject's resolved entry instead)
            dst = d & ~kCoreStale, cw = reinterpret_cast<const uint2*>(records)[i].y, current = (d & kCoreStale) == 0u;
        } else {
            const unsigned int* w = reinterpret_cast<const unsigned int*>(records + i * 20);
            dst = w[3], cw = core_word(w[4]), current = w[0] == cfg_lo && w[1] == cfg_hi;
        }
        const unsigned int bits = cw & kmask;
        const bool down = (cw & kCoreDown) != 0u;
        if (dst < (unsigned)n_nodes && bits != 0u && (bits & ~gmask[dst]) != 0u) atomicOr(&gmask[dst], bits);
        const bool ok = current && dst < (unsigned)n_nodes && bits != 0u &&
                        ((member[dst < (unsigned)n_nodes ? dst : 0u] != 0) == down);
        f |= (ok ? 0u : 1u) | (down ? 0u : 2u);
    }
    if (f) atomicOr(vflags, f);
}

// The boundary hands over 20-byte records (include/rapid_mi355x.h); resident they are split: core[i] = {dst, core word
// (tally_kernel.h: core_word -- ring mask, the status as two bits, the batch end in the sign bit)}, cfg[i] = configuration
// id.  One pass at load time; 16-byte granules of the source are not aligned with records, so each thread reads its
// record's five dwords.
//
// Every configuration id passes through this kernel anyway, so it is compared with the view's current one right here
// (R/MembershipService.java:653-657 drops an alert of another configuration): load_flags bit0 = some delivered record
// carries another id, bit1 = some record names a subject >= n_nodes.  The engine selects the tally instantiation that skips
// the per-delivery id check only for a load whose flags stayed clear (engine.hip: launch_tally) -- the caller's promise
// costs no traffic to verify.
__global__ void split_records_kernel(const unsigned char* records, long long n_records, uint2* core, uint2* cfg, unsigned int* dstv,
                                     long long cfg_id, unsigned int n_nodes, unsigned int* load_flags) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const unsigned int cfg_lo = (unsigned int)(unsigned long long)cfg_id, cfg_hi = (unsigned int)((unsigned long long)cfg_id >> 32);
    unsigned int other = 0u, range = 0u;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_records; i += stride) {
        const unsigned int* w = reinterpret_cast<const unsigned int*>(records + i * 20);
        const unsigned int c0 = w[0], c1 = w[1];
        other |= (c0 ^ cfg_lo) | (c1 ^ cfg_hi);
        range |= w[3] >= n_nodes ? 1u : 0u;
        cfg[i] = make_uint2(c0, c1);
        // (a subject index that would collide with the mark is no node of any view: kept out of range for good)
        const unsigned int d = (w[3] >= kCoreStale ? kCoreStale - 1u : w[3]) | (((c0 ^ cfg_lo) | (c1 ^ cfg_hi)) != 0u ? kCoreStale : 0u);
        dstv[i] = d;  // kept beside the record: its first dword is overwritten when the subjects are resolved to their entries
        core[i] = make_uint2(d, core_word(w[4]));
    }
    const unsigned int f = (__ballot(other != 0u) != 0ull ? 1u : 0u) | (__ballot(range != 0u) != 0ull ? 2u : 0u);
    if (f != 0u && (threadIdx.x & 63u) == 0u) atomicOr(load_flags, f);
}

// node -> dict_entry for rounds whose tables stay in memory (tally_kernel.h: RoundIndex::entries): what the tally's direct mode
// assembles while it stages its tables in LDS, written out once per round; entries[n_nodes] = the poison entry.
__global__ void dict_entries_kernel(const unsigned short* dict, const unsigned short* decl, int n_nodes, int n_hot, unsigned int* entries) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i > n_nodes) return;
    unsigned int e = kEntryPoison | ((unsigned int)n_hot << 17);
    if (i < n_nodes) {
        unsigned int sl = (unsigned int)dict[i] & kSlotMask;
        if (sl == kNoSlot) sl = (unsigned int)n_hot + ((unsigned int)i & (unsigned int)(kDummySlots - 1));
        e = dict_entry((unsigned int)decl[i], sl);
    }
    entries[i] = e;
}

// The first dword of every resident record <- the dict_entry of its subject (entries != nullptr: kDictResolved; a stale or
// unknown subject gets the poison entry at entries[n_nodes]) or the subject itself again (entries == nullptr: the modes that
// look the subject up in the tally).  4 B read + 4 B written per record + a gather that hits the caches (most records name
// one of the round's hot subjects); once per (stream set, round index), not per launch of the tally.
__global__ void resolve_records_kernel(long long n_records, uint2* core, const unsigned int* dstv, const unsigned int* entries, unsigned int n_nodes) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_records; i += stride) {
        const unsigned int d = dstv[i];
        uint2 r = core[i];  // (whole records are rewritten: a store of every other dword leaves half-written lines behind)
        r.x = entries == nullptr ? d : entries[d < n_nodes ? d : n_nodes];  // (the stale mark makes d >= n_nodes)
        core[i] = r;
    }
}

// The view changed while streams stayed loaded: the same comparison against the new configuration id, over the retained ids
// (8 B per record read, 4 B rewritten; once per view change, and only if the streams are tallied again at all).
__global__ void remark_records_kernel(long long n_records, unsigned int* dstv, const uint2* cfg, long long cfg_id, unsigned int* load_flags) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const unsigned int cfg_lo = (unsigned int)(unsigned long long)cfg_id, cfg_hi = (unsigned int)((unsigned long long)cfg_id >> 32);
    unsigned int other = 0u;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_records; i += stride) {
        const uint2 c = cfg[i];
        const unsigned int stale = ((c.x ^ cfg_lo) | (c.y ^ cfg_hi)) != 0u ? kCoreStale : 0u;
        other |= stale;
        const unsigned int old = dstv[i];
        if (((old ^ stale) & kCoreStale) != 0u) dstv[i] = (old & ~kCoreStale) | stale;
    }
    if (__ballot(other != 0u) != 0ull && (threadIdx.x & 63u) == 0u) atomicOr(load_flags, 1u);
}

// Dictionary format (built by index_build_block_kernel below).  Slot numbering: the hot subjects (>= L distinct rings
// named by the round's alert set), ascending by node index.  Every node gets a dictionary entry
// member << 15 | has_adjacency << 14 | slot  with slot = 0x3FFF for subjects that are not hot.
// decl[node] = the rings the round's alert set names for the node (all rings for a hot one), bit 15 = member: what the
// tally kernel checks every delivered report about a subject WITHOUT a slot against, so that a declared alert set that
// does not cover the delivered streams is reported instead of silently under-counting.

// Exclusive scan of one value per thread over a workgroup of up to 1024 threads (wave shuffles + one LDS hop);
// *total receives the sum.  s_wave: 16 ints of LDS scratch.
__device__ inline int block_exclusive_scan(int v, int* s_wave, int* total) {
    const int t = (int)threadIdx.x, lane = t & 63, wv = t >> 6, nw = ((int)blockDim.x + 63) >> 6;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    __syncthreads();  // s_wave may still be read from a previous call
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int base = 0, sum = 0;
    for (int i = 0; i < nw; ++i) {
        const int w = s_wave[i];
        if (i < wv) base += w;
        sum += w;
    }
    *total = sum;
    return base + incl - v;
}

// ---- large populations (N >= kIndexChunkedMin): the walk over the nodes in several workgroups -------------------------
// One workgroup walking 10^5 .. 10^6 nodes lives on memory latency (0.13 ms at 10^5).  Here workgroup b owns the nodes
// [b * kIndexChunk, (b + 1) * kIndexChunk): index_count_kernel counts its hot and its touched nodes, index_assign_kernel
// turns the counts of the workgroups before it into its first slot / first rank and writes everything that is per node --
// dict, decl, node_of_slot, and the compressed tables tbits / trank / tent (a population of this size never has direct
// tables) -- and index_build_block_kernel then only does what needs the whole dictionary: the hot adjacency, the counts
// for the host, the zeroing.  Slot numbering is the same (hot subjects ascending by node index).
constexpr int kIndexChunk = 8192;        // nodes per workgroup: 16 waves x 8 steps of 64
constexpr int kIndexChunkedMin = 40000;  // below: one workgroup does it all (direct tables may still fit there)

__global__ __launch_bounds__(1024) void index_count_kernel(const unsigned int* gmask, int n_nodes, int L, int* blk_counts) {
    __shared__ int s_hot, s_touched;
    if (threadIdx.x == 0) {
        s_hot = 0;
        s_touched = 0;
    }
    __syncthreads();
    const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6);
    const int n0 = (int)blockIdx.x * kIndexChunk + wv * 512;
    unsigned int g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = n0 + 64 * j + lane;
        g[j] = n < n_nodes ? gmask[n] : 0u;
    }
    int hot = 0, touched = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hot += __popcll(__ballot(__popc(g[j]) >= L));
        touched += __popcll(__ballot(g[j] != 0u));
    }
    if (lane == 0) {
        atomicAdd(&s_hot, hot);
        atomicAdd(&s_touched, touched);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        blk_counts[2 * blockIdx.x] = s_hot;
        blk_counts[2 * blockIdx.x + 1] = s_touched;
    }
}

__global__ __launch_bounds__(1024) void index_assign_kernel(const unsigned int* gmask, const unsigned char* member, int n_nodes, int L,
                                                            const int* blk_counts, unsigned short* dict, unsigned short* decl,
                                                            int* node_of_slot, unsigned int* tbits, unsigned short* trank,
                                                            unsigned int* tent, int tent_cap) {
    __shared__ int s_base_hot, s_base_touched, s_wave_hot[16], s_wave_touched[16];
    const int t = (int)threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) {
        s_base_hot = 0;
        s_base_touched = 0;
    }
    __syncthreads();
    // hot / touched nodes of the workgroups before this one
    for (int b = t; b < (int)blockIdx.x; b += (int)blockDim.x) {
        atomicAdd(&s_base_hot, blk_counts[2 * b]);
        atomicAdd(&s_base_touched, blk_counts[2 * b + 1]);
    }
    const int n0 = (int)blockIdx.x * kIndexChunk + wv * 512;
    unsigned int g[8], mem[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = n0 + 64 * j + lane;
        g[j] = n < n_nodes ? gmask[n] : 0u;
        mem[j] = (n < n_nodes && member[n]) ? 0x8000u : 0u;
    }
    int hot = 0, touched = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hot += __popcll(__ballot(__popc(g[j]) >= L));
        touched += __popcll(__ballot(g[j] != 0u));
    }
    if (lane == 0) {
        s_wave_hot[wv] = hot;
        s_wave_touched[wv] = touched;
    }
    __syncthreads();
    int ph = s_base_hot, pt = s_base_touched;  // first slot / first rank of this wave's nodes
    for (int i = 0; i < wv; ++i) {
        ph += s_wave_hot[i];
        pt += s_wave_touched[i];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = n0 + 64 * j + lane;
        const bool in = n < n_nodes;
        const bool is_hot = __popc(g[j]) >= L, is_touched = g[j] != 0u;  // (out-of-range lanes loaded 0: neither)
        const unsigned long long hots = __ballot(is_hot), tch = __ballot(is_touched);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int my_slot = ph + __popcll(hots & below), my_rank = pt + __popcll(tch & below);
        unsigned int slot = 0x3FFFu;
        if (is_hot && my_slot < 16319) {
            slot = (unsigned int)my_slot;
            node_of_slot[my_slot] = n;
        }
        const unsigned int dcl = (slot != 0x3FFFu ? 0x3FFFu : (g[j] & 0x3FFFu)) | mem[j];
        if (in) {
            dict[n] = (unsigned short)(slot | mem[j]);
            decl[n] = (unsigned short)dcl;
        }
        // compressed tables: the two 32-node words of this step, the touched nodes before each, one entry per touched node
        if (in && (lane & 31) == 0) {
            const int w = n >> 5;
            tbits[w] = lane == 0 ? (unsigned int)tch : (unsigned int)(tch >> 32);
            const int r = lane == 0 ? pt : pt + __popcll(tch & 0xFFFFFFFFull);
            trank[w] = (unsigned short)(r > 65535 ? 65535 : r);
        }
        if (is_touched && my_rank < tent_cap) tent[my_rank] = (dcl << 16) | slot;
        ph += __popcll(hots);
        pt += __popcll(tch);
    }
}


// The whole index after the touch pass in ONE workgroup (the round's hot set is a few hundred to a few thousand
// subjects): slot numbering + dictionary + declared masks, then the hot adjacency.  One launch and one read-back of
// info[] instead of six launches and two synchronisations, and no atomics: the list of a hot slot e is built by the
// thread that owns e from its row of the observer table -- one triple (subject slot | observer slot << 14 | ring << 28)
// per ring on which a hot node observes e (for a joiner: its expected observers, R/MembershipView.java:292-322): the
// potential implicit reports of R/MultiNodeCutDetector.java:137-164; smask[e] = the rings of e's triples.
// pairs has room for adj_cap entries (info[2] |= 2 if more are needed: nothing is written past it).  info_out: a second
// copy of info[0..7] (host-mapped memory: the host reads it after synchronising, no copy is enqueued).  info[5] = touched
// nodes, info[6] = 1 if the compressed tables are complete (at most 65535 touched nodes and tent_cap entries).
__global__ __launch_bounds__(1024) void index_build_block_kernel(unsigned int* gmask, const unsigned char* member, const int* obs,
                                                                 int n_nodes, int K, int L, unsigned short* dict,
                                                                 unsigned short* decl, int* node_of_slot, unsigned short* smask,
                                                                 unsigned int* pairs, int adj_cap, unsigned int* tbits,
                                                                 unsigned short* trank, unsigned int* tent, int tent_cap, int* info,
                                                                 volatile int* info_out, int direct_budget,
                                                                 unsigned long long* zero_words, int n_zero_words,
                                                                 unsigned int* zero_flags, int seq, const int* blk_counts,
                                                                 int n_chunks) {
    __shared__ int s_wave[16];
    __shared__ int s_pre_hot, s_pre_touched;
    const int T = (int)blockDim.x, t = (int)threadIdx.x;
    const bool prebuilt = blk_counts != nullptr;  // index_count_kernel + index_assign_kernel did everything that is per node
    if (t == 0) {
        s_pre_hot = 0;
        s_pre_touched = 0;
    }
    __syncthreads();  // (more than 64 chunks: the adds below come from several waves)
    // the round's launch statistics / pool words and the sticky error flags start at zero: cleared here instead of by
    // memsets of their own between this kernel and the tally (each costs a launch gap)
    for (int i = t; i < n_zero_words; i += T) zero_words[i] = 0ull;
    if (t < 2 && zero_flags != nullptr) zero_flags[t] = 0u;
    // ---- slots (ascending node order), dictionary, declared ring masks ----
    // Wave w owns a contiguous range of nodes and walks it 64 at a time (coalesced; this kernel is one workgroup and lives on
    // memory latency, not bandwidth): a node's slot = hot nodes of earlier waves + of earlier steps + of lower lanes.
    const int lane = t & 63, wv = t >> 6, nw = T >> 6;
    const int per_wave_nodes = ((n_nodes + nw * 64 - 1) / (nw * 64)) * 64;
    const int beg = min(n_nodes, wv * per_wave_nodes), end = min(n_nodes, beg + per_wave_nodes);
    // (eight steps' loads are issued before the first ballot: left to itself the compiler keeps every load next to the
    // ballot that consumes it, and a pass over 10^5 nodes then costs 98 memory round trips per wave instead of 13)
    int n_hot_all = 0;
    if (!prebuilt) {
        constexpr int kBatch = 8;
        int nh = 0;
        for (int n0 = beg; n0 < end; n0 += 64 * kBatch) {
            unsigned int g[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int n = n0 + 64 * j + lane;
                g[j] = n < end ? gmask[n] : 0u;
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) nh += __popcll(__ballot(__popc(g[j]) >= L));
        }
        int ph = block_exclusive_scan(lane == 0 ? nh : 0, s_wave, &n_hot_all);
        ph = __shfl(ph, 0, 64);  // hot nodes of the waves before this one
        for (int n0 = beg; n0 < end; n0 += 64 * kBatch) {
            unsigned int g[kBatch], mem[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int n = n0 + 64 * j + lane;
                g[j] = n < end ? gmask[n] : 0u;
                mem[j] = (n < end && member[n]) ? 0x8000u : 0u;
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int n = n0 + 64 * j + lane;
                const bool in = n < end;
                const bool hot = in && __popc(g[j]) >= L;
                const unsigned long long hots = __ballot(hot);
                const int mine_slot = ph + __popcll(hots & ((1ull << lane) - 1ull));
                ph += __popcll(hots);
                unsigned int slot = 0x3FFFu;
                if (hot && mine_slot < 16319) {
                    slot = (unsigned int)mine_slot;
                    node_of_slot[mine_slot] = n;
                }
                if (in) {
                    dict[n] = (unsigned short)(slot | mem[j]);
                    decl[n] = (unsigned short)((slot != 0x3FFFu ? 0x3FFFu : (g[j] & 0x3FFFu)) | mem[j]);
                }
            }
        }
    } else {
        for (int b = t; b < n_chunks; b += T) {
            atomicAdd(&s_pre_hot, blk_counts[2 * b]);
            atomicAdd(&s_pre_touched, blk_counts[2 * b + 1]);
        }
        __syncthreads();
        n_hot_all = s_pre_hot;
    }
    __threadfence_block();
    __syncthreads();
    const int n_hot = min(n_hot_all, 16318);
    // ---- the hot adjacency as a flat list: thread t owns a contiguous chunk of the hot slots ----
    const int per2 = (n_hot + T - 1) / T;
    const int b2 = min(n_hot, t * per2), e2 = min(n_hot, b2 + per2);
    int mine = 0;
    for (int e = b2; e < e2; ++e) {
        const int node = node_of_slot[e];
        for (int k = 0; k < K; ++k) {
            const int o = obs[node * K + k];
            if (o >= 0 && (int)(dict[o] & 0x3FFF) < n_hot) ++mine;
        }
    }
    int total = 0;
    int at = block_exclusive_scan(mine, s_wave, &total);
    const bool fits = total <= 65535 && total <= adj_cap;
    for (int e = b2; e < e2; ++e) {
        const int node = node_of_slot[e];
        unsigned int am = 0u;
        for (int k = 0; k < K; ++k) {
            const int o = obs[node * K + k];
            const int eo = o >= 0 ? (int)(dict[o] & 0x3FFF) : 0x3FFF;
            if (eo < n_hot) {  // a hot node observes e on ring k (for a joiner: one of its expected observers)
                if (fits) pairs[at] = (unsigned int)e | ((unsigned int)eo << 14) | ((unsigned int)k << 28);
                ++at;
                am |= 1u << k;
            }
        }
        smask[e] = (unsigned short)am;
        if (am != 0u) dict[node] |= (unsigned short)0x4000;  // only this thread touches dict[node] after the barrier above
    }
    __threadfence_block();
    __syncthreads();  // dict[] is final (adjacency flags included)
    // ---- the compressed form of dict[] / decl[] for populations whose direct tables do not fit the LDS: one bit per node
    // (touched = named by the alert set at all), touched nodes before each 32-node word, one entry per touched node ----
    // Skipped when the direct tables will be used anyway -- the same test as the host's (engine.hip: build_round_index), on
    // the same numbers; direct_budget < 0: always build them.  info[7] tells the host which way it went.
    const bool direct_fits = direct_budget >= 0 &&
                             tally_shared_bytes(kDictDirect, n_nodes, 0, n_hot, total) + 8 * tally_wave_bytes(n_hot) <= direct_budget;
    const int n_words = (direct_fits || prebuilt) ? 0 : (n_nodes + 31) / 32;
    const int perw = (n_words + T - 1) / T;
    const int w0 = min(n_words, t * perw), w1 = min(n_words, w0 + perw);
    int cnt = 0;
    for (int w = w0; w < w1; ++w) {
        // the 32 entries of a word are 64 contiguous, 64-byte aligned bytes of decl[]: four 16-byte loads instead of 32
        // two-byte ones (this single workgroup lives on memory latency; decl[] is allocated with 40 entries of slack)
        const uint4* const v = reinterpret_cast<const uint4*>(decl + (size_t)w * 32);
        unsigned int bits = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 e = v[q];
            const unsigned int ws[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if ((ws[j] & 0x3FFFu) != 0u) bits |= 1u << (q * 8 + 2 * j);
                if (((ws[j] >> 16) & 0x3FFFu) != 0u) bits |= 1u << (q * 8 + 2 * j + 1);
            }
        }
        const int valid = n_nodes - w * 32;  // entries of this word that are nodes
        if (valid < 32) bits &= (1u << valid) - 1u;
        tbits[w] = bits;
        cnt += __popc(bits);
    }
    int n_touched = 0;
    int rk = block_exclusive_scan(cnt, s_wave, &n_touched);
    if (prebuilt) n_touched = s_pre_touched;  // tbits / trank / tent were written by index_assign_kernel
    const bool tfits = n_touched <= 65535 && n_touched <= tent_cap;
    for (int w = w0; w < w1; ++w) {
        trank[w] = (unsigned short)(rk > 65535 ? 65535 : rk);
        unsigned int bits = tbits[w];
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            const int n = w * 32 + b;
            if (tfits) tent[rk] = ((unsigned int)decl[n] << 16) | ((unsigned int)dict[n] & 0x3FFFu);
            ++rk;
        }
    }
    if (t == 0) {
        info[5] = n_touched;
        info[6] = tfits && !direct_fits ? 1 : 0;
        info[7] = direct_fits ? 1 : 0;
    }
    __syncthreads();
    if (t == 0) {
        info[0] = n_hot_all;
        info[1] = n_hot_all;
        info[2] = (n_hot_all > 16318 ? 1 : 0) | (fits ? 0 : 2);  // 64 slot numbers are kept for the tally kernel's dummy slots
        info[3] = total;
        for (int i = 0; i < 8; ++i) info_out[i] = info[i];  // info[4] was written by the touch pass
        // the host does not wait for the stream: it polls this word of the mapped page (what it reads afterwards was written
        // before the fence; the zeroing above is ordered before the tally kernel by the stream)
        __threadfence_system();
        info_out[15] = seq;
    }
    // leave the work area as the next round's touch pass needs it (all zero): nobody reads gmask[] / info[] after this point
    __syncthreads();
    for (int n = t; n < n_nodes; n += T) gmask[n] = 0u;
    if (t < 8) info[t] = 0;
}

// ---- rapid_sim_generate: the delivered streams made on the device ----------------------------------------------------
// Every receiver gets every BatchedAlertMessage of the round exactly once, in a receiver-specific seeded order (the
// reference's fan-out: UnicastToAllBroadcaster.java:46-63 sends each batch to all members; arrival order differs per
// receiver -- paper Fig.11 methodology).  The order of receiver r = the batches sorted by
//     key(r, b) = mix64(mix64(seed + node index of r) + b)            (ties, never seen, by b: the sort is stable)
// (rapid_amd/scenarios.py: deliver_hashed states the same on the host).  gen_keys_kernel writes the keys, one segmented radix
// sort orders every receiver's batches, and gen_streams_kernel -- one workgroup per receiver -- lays the batches down back to
// back directly in the RESIDENT layout (core = {entry or subject, core word}, subject array, configuration ids): the 20-byte
// records of a round's deliveries never exist, neither on the host nor on the device.
__device__ inline unsigned long long gen_mix64(unsigned long long x) {  // splitmix64 finaliser (== mix64 of tally_kernel.h)
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void gen_keys_kernel(const int* receivers, int n_receivers, int n_batches, unsigned long long seed, unsigned long long* keys,
                                unsigned int* vals) {
    const long long total = (long long)n_receivers * n_batches;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int r = (int)(t / n_batches);
        const unsigned int b = (unsigned int)(t - (long long)r * n_batches);
        keys[t] = gen_mix64(gen_mix64(seed + (unsigned long long)(unsigned int)receivers[r]) + (unsigned long long)b);
        vals[t] = b;
    }
}
// grid = receivers, block = 256.  alerts = the round's distinct alerts (20-byte records) in batch order, boff[b] .. boff[b + 1]
// = batch b; perm[r][j] = the j-th batch receiver r gets; every receiver gets all n_alerts records: rec_off[r] = r * n_alerts.
// entries != nullptr: the first dword of a record is its subject's dict_entry (kDictResolved), else the subject itself.
__global__ __launch_bounds__(256) void gen_streams_kernel(const unsigned char* alerts, const long long* boff, int n_batches, const unsigned int* perm,
                                                          long long n_alerts, uint2* core, uint2* cfg, unsigned int* dstv, long long cfg_id,
                                                          unsigned int n_nodes, const unsigned int* entries, unsigned int* load_flags) {
    __shared__ int s_wave[16];
    const int r = (int)blockIdx.x, t = (int)threadIdx.x;
    const unsigned int cfg_lo = (unsigned int)(unsigned long long)cfg_id, cfg_hi = (unsigned int)((unsigned long long)cfg_id >> 32);
    const long long base = (long long)r * n_alerts;
    long long carry = 0;
    unsigned int other = 0u, range = 0u;
    for (int j0 = 0; j0 < n_batches; j0 += (int)blockDim.x) {
        const int j = j0 + t;
        long long b0 = 0;
        int len = 0;
        if (j < n_batches) {
            const unsigned int b = perm[(long long)r * n_batches + j];
            b0 = boff[b];
            len = (int)(boff[b + 1] - b0);
        }
        int total = 0;
        const long long at = base + carry + block_exclusive_scan(len, s_wave, &total);
        carry += total;
        for (int k = 0; k < len; ++k) {
            const unsigned int* w = reinterpret_cast<const unsigned int*>(alerts + (b0 + k) * 20);
            const unsigned int c0 = w[0], c1 = w[1];
            other |= (c0 ^ cfg_lo) | (c1 ^ cfg_hi);
            range |= w[3] >= n_nodes ? 1u : 0u;
            const unsigned int d = (w[3] >= kCoreStale ? kCoreStale - 1u : w[3]) | (((c0 ^ cfg_lo) | (c1 ^ cfg_hi)) != 0u ? kCoreStale : 0u);
            // (a batch ends with its last alert, whatever the flags of the set say)
            const unsigned int word = (core_word(w[4]) & ~kCoreEob) | (k == len - 1 ? kCoreEob : 0u);
            dstv[at + k] = d;
            cfg[at + k] = make_uint2(c0, c1);
            core[at + k] = make_uint2(entries == nullptr ? d : entries[d < n_nodes ? d : n_nodes], word);
        }
    }
    const unsigned int f = (__ballot(other != 0u) != 0ull ? 1u : 0u) | (__ballot(range != 0u) != 0ull ? 2u : 0u);
    if (f != 0u && (threadIdx.x & 63u) == 0u) atomicOr(load_flags, f);
}
__global__ void gen_offsets_kernel(long long* rec_off, int n_receivers, long long n_alerts) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i <= n_receivers) rec_off[i] = (long long)i * n_alerts;
}


}  // namespace rapid
