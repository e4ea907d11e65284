// Per-round index over the round's alert set (rebuilt every round: part of every timed step of bench.py): which subjects does
// this round's alert set name at all, which of them can ever reach the L watermark ("hot"), a dense slot numbering (hot
// subjects first, ascending node index), and the adjacency among hot subjects along the K-ring monitoring graph -- the only
// (observer, subject, ring) triples for which MultiNodeCutDetector.invalidateFailingEdges (R/MultiNodeCutDetector.java:137-164)
// can ever apply an implicit report at any receiver, because both ends need >= L explicit reports
// (R/MultiNodeCutDetector.java:104-107, 153).  Also here: the kernels of rapid_sim_generate.
//
// The index is a SUPERSET construction: it ignores the per-alert filter (configuration id, UP/DOWN vs membership),
// so it can only make more subjects "touched"/"hot" than a receiver will see, never fewer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tally_kernel.h"  // the LDS budget formulas: this kernel and the host must agree on where the dictionary will live

namespace rapid {

// gmask[dst] |= ring_mask over every record given (the round's distinct alert set if the host declared one, else
// every delivered record -- 20-byte boundary records either way).  After the first few thousand records nearly every bit is
// already set, so the (possibly stale, L1-cached) pre-test avoids almost all atomics.  Also validates the records once:
// vflags bit0 is set if ANY record of the CURRENT configuration fails the rest of the filter of R/MembershipService.java:644-675
// under the current view (or names a node out of range or no ring), bit1 if any record is an UP alert.
__global__ void index_touch_kernel(const unsigned char* records, long long n_records, int n_nodes, unsigned int kmask, long long cfg_id,
                                   const unsigned char* member, unsigned int* gmask, unsigned int* vflags) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const unsigned int cfg_lo = (unsigned int)(unsigned long long)cfg_id, cfg_hi = (unsigned int)((unsigned long long)cfg_id >> 32);
    unsigned int f = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_records; i += stride) {
        const unsigned int* w = reinterpret_cast<const unsigned int*>(records + i * 20);
        const unsigned int dst = w[3], cw = core_word(w[4]);  // (tally_kernel.h: core_word)
        const bool current = w[0] == cfg_lo && w[1] == cfg_hi;  // the record carries the engine's configuration id
        const unsigned int bits = cw & kmask;
        const bool down = (cw & kCoreDown) != 0u;
        if (dst < (unsigned)n_nodes && bits != 0u && (bits & ~gmask[dst]) != 0u) atomicOr(&gmask[dst], bits);
        // (an alert of another configuration is not validated: whatever it says, every delivery that is a copy of it is dropped
        // whole by the tally's configuration-id compare, R/MembershipService.java:653-657 -- late deliveries among a round's
        // batches do not cost the round its pre-validated instantiation)
        const bool ok = !current || (dst < (unsigned)n_nodes && bits != 0u && ((member[dst < (unsigned)n_nodes ? dst : 0u] != 0) == down));
        f |= (ok ? 0u : 1u) | (down ? 0u : 2u) | (current ? 0u : 4u);  // (bit 2: an alert of another configuration is among them)
    }
    if (f) atomicOr(vflags, f);
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
        if (is_touched && my_rank < tent_cap) tent[my_rank] = dict_entry(dcl, slot);
        ph += __popcll(hots);
        pt += __popcll(tch);
    }
}


// The hot adjacency of a LARGE population (the form of index_count_kernel / index_assign_kernel), slot by slot and in parallel: one
// thread per hot slot works out which of its K observers are hot (today's row or the memo of quirk Q4 -> the observers'
// dictionary entries: three dependent, scattered reads per slot into tables of tens of megabytes) and leaves the K observer
// slots and their ring mask where index_build_block_kernel only has to count and lay them down.  As the single workgroup's own
// work -- fifteen slots per thread, one after the other, twice -- this chain was 0.89 ms of a 2.7 ms round at 10^6 nodes and
// 15,000 hot subjects.  The launch also clears the touch pass's work area (nobody reads it after index_assign_kernel), 4 MB
// that the single workgroup used to clear at its end.
constexpr int kIndexEdgeStride = 14;  // RAPID_MAX_K observer slots per hot slot
__global__ __launch_bounds__(256) void index_edges_kernel(const int* blk_counts, int n_chunks, const int* node_of_slot, const unsigned char* member,
                                                          const int* obs, int n_nodes, int K, const unsigned short* dict, int* q4_rows,
                                                          unsigned char* q4_valid, unsigned short* edges, unsigned short* edge_mask, int* info,
                                                          unsigned int* gmask) {
    __shared__ int s_hot;
    if (threadIdx.x == 0) s_hot = 0;
    __syncthreads();
    for (int b = (int)threadIdx.x; b < n_chunks; b += (int)blockDim.x) atomicAdd(&s_hot, blk_counts[2 * b]);
    __syncthreads();
    const int n_hot = min(s_hot, 16318);
    const long long gsz = (long long)gridDim.x * blockDim.x, gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long n = gid; n < n_nodes; n += gsz) gmask[n] = 0u;
    const int e = (int)gid;
    if (e >= n_hot) return;
    const int node = node_of_slot[e];
    const bool use_memo = q4_valid != nullptr && member[node] != 0;
    const bool have = use_memo && q4_valid[node] != 0;
    const int* const today = obs + (long long)node * K;
    int* const memo = q4_rows + (long long)node * K;
    int o[kIndexEdgeStride], td[kIndexEdgeStride];
    bool stale = false;
#pragma unroll
    for (int k = 0; k < kIndexEdgeStride; ++k) td[k] = k < K ? today[k] : -1;
    if (have) {
#pragma unroll
        for (int k = 0; k < kIndexEdgeStride; ++k) {
            o[k] = k < K ? memo[k] : -1;
            stale = stale || o[k] != td[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < kIndexEdgeStride; ++k) o[k] = td[k];
        if (use_memo) {  // the first time this member is hot since its entry was dropped (one slot, one thread)
#pragma unroll
            for (int k = 0; k < kIndexEdgeStride; ++k)
                if (k < K) memo[k] = td[k];
            q4_valid[node] = 1;
        }
    }
    unsigned int am = 0u;
#pragma unroll
    for (int k = 0; k < kIndexEdgeStride; ++k) {
        const unsigned int eo = (o[k] >= 0 && o[k] < n_nodes) ? ((unsigned int)dict[o[k]] & 0x3FFFu) : 0x3FFFu;
        edges[(long long)e * kIndexEdgeStride + k] = (unsigned short)eo;
        am |= (int)eo < n_hot ? 1u << k : 0u;
    }
    edge_mask[e] = (unsigned short)am;
    if (stale) atomicOr(reinterpret_cast<unsigned int*>(&info[2]), 4u);  // (see index_build_block_kernel)
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
                                                                 int n_chunks, int* q4_rows, unsigned char* q4_valid,
                                                                 const unsigned short* edges = nullptr, const unsigned short* edge_mask = nullptr) {
    // edges != nullptr (with blk_counts): index_edges_kernel has worked out every hot slot's observer slots and ring mask (and
    // cleared the work area): what is left here is to count them and lay the triples down
    __shared__ int s_wave[16];
    __shared__ int s_pre_hot, s_pre_touched;
    __shared__ int s_pub[8];  // the answer on its way to the host-mapped page
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
    // Quirk Q4 of the reference, reproduced (q4_valid != nullptr): invalidateFailingEdges asks getObserversOf(subject) for the
    // MEMBERS in preProposal (R/MultiNodeCutDetector.java:147-149), and that call is memoised per node
    // (R/MembershipView.java:210-224) with an invalidation that misses a change of the ring minimum (:143-152, 181-195).  A hot
    // member's row is therefore read from the memo q4_rows -- written from today's table the first time the member is hot since
    // its entry was last dropped (engine.hip: rebuild_view drops what ringAdd / ringDelete drop) -- and a stale row stays
    // stale exactly as long as the Java's does.  Joiners go through getExpectedObserversOf, which is not memoised: fresh.
    const int per2 = (n_hot + T - 1) / T;
    const int b2 = min(n_hot, t * per2), e2 = min(n_hot, b2 + per2);
    int mine = 0;
    bool stale = false;
    // One slot's K observers -> their slots, as K INDEPENDENT loads twice over (rows, then dictionary entries): this kernel is one
    // workgroup living on memory latency, and a loop over the rings with a dependent pair of loads per ring costs 2 K round trips
    // per slot and pass (two thirds of the kernel's 30 us at C3b) where two suffice.
    constexpr int kKMax = 14;  // RAPID_MAX_K
    auto slot_edges = [&](int e, unsigned int (&eo)[kKMax]) -> unsigned int {
        const int node = node_of_slot[e];
        const bool use_memo = q4_valid != nullptr && member[node] != 0;
        const bool have = use_memo && q4_valid[node] != 0;
        const int* const today = obs + (long long)node * K;
        int* const memo = q4_rows + (long long)node * K;
        int o[kKMax], td[kKMax];
#pragma unroll
        for (int k = 0; k < kKMax; ++k) td[k] = k < K ? today[k] : -1;
        if (have) {
#pragma unroll
            for (int k = 0; k < kKMax; ++k) {
                o[k] = k < K ? memo[k] : -1;
                stale = stale || o[k] != td[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < kKMax; ++k) o[k] = td[k];
            if (use_memo) {  // the first time this member is hot since its entry was dropped (this thread owns the node: one slot, one thread)
#pragma unroll
                for (int k = 0; k < kKMax; ++k)
                    if (k < K) memo[k] = td[k];
                q4_valid[node] = 1;
            }
        }
        unsigned int am = 0u;
#pragma unroll
        for (int k = 0; k < kKMax; ++k) eo[k] = (o[k] >= 0 && o[k] < n_nodes) ? ((unsigned int)dict[o[k]] & 0x3FFFu) : 0x3FFFu;
#pragma unroll
        for (int k = 0; k < kKMax; ++k) am |= (int)eo[k] < n_hot ? 1u << k : 0u;  // a hot node observes e on ring k (for a joiner: one of its expected observers)
        return am;
    };
    for (int e = b2; e < e2; ++e) {
        if (edges != nullptr) {
            mine += __popc((unsigned int)edge_mask[e]);
            continue;
        }
        unsigned int eo[kKMax];
        mine += __popc(slot_edges(e, eo));
    }
    if (stale) atomicOr(reinterpret_cast<unsigned int*>(&info[2]), 4u);  // (info[2] bit 2: some hot member's memoised observers are not today's -- the quirk is live in this round)
    int total = 0;
    int at = block_exclusive_scan(mine, s_wave, &total);
    const bool fits = total <= 65535 && total <= adj_cap;
    for (int e = b2; e < e2; ++e) {
        unsigned int eo[kKMax];
        unsigned int am;
        if (edges != nullptr) {
            am = (unsigned int)edge_mask[e];
#pragma unroll
            for (int k = 0; k < kKMax; ++k) eo[k] = (unsigned int)edges[(long long)e * kIndexEdgeStride + k];
        } else {
            am = slot_edges(e, eo);
        }
#pragma unroll
        for (int k = 0; k < kKMax; ++k) {
            if ((am >> k) & 1u) {
                if (fits) pairs[at] = (unsigned int)e | (eo[k] << 14) | ((unsigned int)k << 28);
                ++at;
            }
        }
        smask[e] = (unsigned short)am;
        if (am != 0u) dict[node_of_slot[e]] |= (unsigned short)0x4000;  // only this thread touches dict[node] after the barrier above
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
            if (tfits) tent[rk] = dict_entry((unsigned int)decl[n], (unsigned int)dict[n] & 0x3FFFu);
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
        info[2] = (info[2] & 4) | (n_hot_all > 16318 ? 1 : 0) | (fits ? 0 : 2);  // 64 slot numbers are kept for the tally kernel's dummy slots
        info[3] = total;
        for (int i = 0; i < 8; ++i) s_pub[i] = info[i];  // info[4] was written by the touch pass
    }
    __syncthreads();
    // the host does not wait for the stream: it polls the last word of the mapped page (what it reads afterwards was written
    // before the sequence word; the zeroing below is ordered before the tally kernel by the stream).  Eight lanes, one store
    // instruction -- see index_fused_kernel, also for why every storing lane fences and a barrier stands before the sequence word.
    if (t < 8) {
        info_out[t] = s_pub[t];
        __threadfence_system();
    }
    __syncthreads();
    if (t == 0) {
        __threadfence_system();
        info_out[15] = seq;
    }
    // leave the work area as the next round's touch pass needs it (all zero): nobody reads gmask[] / info[] after this point
    // (index_edges_kernel has cleared gmask[] already where it ran: 4 MB at 10^6 nodes are not one workgroup's job)
    __syncthreads();
    if (edges == nullptr)
        for (int n = t; n < n_nodes; n += T) gmask[n] = 0u;
    if (t < 8) info[t] = 0;
}

// ---- the hashed dictionary of rounds with thousands of hot subjects (tally_kernel.h: kDictHashed) ---------------------------------
// One workgroup, behind the index build of a round whose every named subject is hot.  The hot subjects are RENUMBERED: slot =
// position in the order of (bucket, remainder) of the key x = (node * mul) mod 2^bits -- the order the dictionary's lookup yields
// by itself (bucket's first slot + index of the matching remainder), so that no position -> slot table is needed.  Everything that
// carries slot numbers follows: node_of_slot and smask (written in the new order into their second buffers), the triples (both
// slot fields mapped), and the per-node tables in memory (entries[], dict[]: patched for the hot nodes, so that a launch that looks
// subjects up in memory, or records resolved against entries[], agree with the renumbered tables).  The multiplier is the first
// of kHashMultipliers under which no bucket holds more than kHashBucketCap keys (consecutive node indices -- joiners are registered
// in a block -- spread evenly under a multiplicative hash; random ones fill buckets like balls into bins: 4,096 buckets, 15,000 keys,
// a bucket of 17 once in 10^3 rounds); the answer is 1 + the multiplier's index, or 0: the host falls back
// to the dictionary in memory.  mail: the host-mapped page, word 12 (0: no multiplier fits; 1 + the multiplier's index), sequence word 11.
constexpr int kHashTries = 4;
__host__ __device__ inline unsigned int hash_multiplier(int i) {  // odd 24-bit constants (v_mul_u32_u24 territory)
    return i == 0 ? 0x9E3779u : i == 1 ? 0x85EBCBu : i == 2 ? 0xC2B2AFu : 0x27D4EBu;
}
__global__ __launch_bounds__(1024) void index_hash_kernel(const int* node_of_slot, const unsigned short* smask, const unsigned int* pairs, int n_hot, int n_adj,
                                                          int n_nodes, const unsigned char* member, unsigned short* hoff, unsigned char* hrem,
                                                          unsigned int* hmem, int* node_of_slot_new, unsigned short* smask_new, unsigned int* pairs_new,
                                                          unsigned short* new_of_old, unsigned int* entries, unsigned short* dict,
                                                          volatile int* mail, int seq) {
    unsigned char* const hash_lds = dynamic_lds();
    __shared__ int s_wave[16];
    __shared__ int s_max, s_choice;
    const int T = (int)blockDim.x, t = (int)threadIdx.x;
    const int bits = hash_key_bits(n_nodes), nb = 1 << (bits - kHashRemBits);
    const unsigned int kmask = (1u << bits) - 1u;
    unsigned int* const cnt = reinterpret_cast<unsigned int*>(hash_lds);              // [nb] keys per bucket, then the running cursor
    unsigned int* const off = cnt + nb;                                                // [nb + 1] first slot per bucket
    unsigned short* const byb = reinterpret_cast<unsigned short*>(off + nb + 1);      // [n_hot] old slots grouped by bucket (unordered inside)
    if (t == 0) s_choice = -1;
    __syncthreads();
    for (int tr = 0; tr < kHashTries; ++tr) {
        const unsigned int mul = hash_multiplier(tr);
        for (int b = t; b < nb; b += T) cnt[b] = 0u;
        if (t == 0) s_max = 0;
        __syncthreads();
        for (int i = t; i < n_hot; i += T) atomicAdd(&cnt[(((unsigned int)node_of_slot[i] * mul) & kmask) >> kHashRemBits], 1u);
        __syncthreads();
        int mx = 0;
        for (int b = t; b < nb; b += T) mx = max(mx, (int)cnt[b]);
        if (mx > 0) atomicMax(&s_max, mx);
        __syncthreads();
        if (s_max <= kHashBucketCap) {
            if (t == 0) s_choice = tr;
            __syncthreads();
            break;
        }
        __syncthreads();
    }
    const int choice = s_choice;
    if (choice >= 0) {
        const unsigned int mul = hash_multiplier(choice);
        // first slot of every bucket: an exclusive scan of the counts (each thread owns a contiguous run of buckets)
        const int per = (nb + T - 1) / T, b0 = min(nb, t * per), b1 = min(nb, b0 + per);
        int mine = 0;
        for (int b = b0; b < b1; ++b) mine += (int)cnt[b];
        int total = 0;
        int at = block_exclusive_scan(mine, s_wave, &total);
        for (int b = b0; b < b1; ++b) {
            off[b] = (unsigned int)at;
            at += (int)cnt[b];
        }
        if (t == 0) off[nb] = (unsigned int)total;
        __syncthreads();
        for (int b = t; b < nb; b += T) cnt[b] = 0u;  // (now the cursor inside the bucket)
        __syncthreads();
        for (int i = t; i < n_hot; i += T) {
            const unsigned int b = (((unsigned int)node_of_slot[i] * mul) & kmask) >> kHashRemBits;
            byb[off[b] + atomicAdd(&cnt[b], 1u)] = (unsigned short)i;
        }
        __syncthreads();
        // a key's slot = its bucket's first slot + the keys of the bucket with a smaller remainder (distinct within a bucket)
        for (int i = t; i < n_hot; i += T) {
            const int node = node_of_slot[i];
            const unsigned int x = ((unsigned int)node * mul) & kmask, b = x >> kHashRemBits, r = x & 255u;
            int below = 0;
            for (unsigned int j = off[b]; j < off[b + 1]; ++j) {
                const unsigned int xo = ((unsigned int)node_of_slot[byb[j]] * mul) & kmask;
                below += (xo & 255u) < r ? 1 : 0;
            }
            const unsigned int pos = off[b] + (unsigned int)below;
            new_of_old[i] = (unsigned short)pos;
            hrem[pos] = (unsigned char)r;
            node_of_slot_new[pos] = node;
            smask_new[pos] = smask[i];
            entries[node] = (entries[node] & 0x1FFFFu) | (pos << 17);
            dict[node] = (unsigned short)((dict[node] & 0xC000u) | pos);
        }
        for (int b = t; b <= nb; b += T) hoff[b] = (unsigned short)off[b];
        for (int i = n_hot + t; i < n_hot + kHashPad; i += T) hrem[i] = 0;
        for (int w = t; w < (n_hot + 31) / 32; w += T) hmem[w] = 0u;
        __threadfence_block();
        __syncthreads();
        for (int i = t; i < n_hot; i += T)
            if (member[node_of_slot[i]] != 0) atomicOr(&hmem[new_of_old[i] >> 5], 1u << (new_of_old[i] & 31u));
        for (int a = t; a < n_adj; a += T) {
            const unsigned int pr = pairs[a];
            pairs_new[a] = (unsigned int)new_of_old[pr & 0x3FFFu] | ((unsigned int)new_of_old[(pr >> 14) & 0x3FFFu] << 14) | (pr & 0xF0000000u);
        }
    }
    __syncthreads();
    if (t == 0) {
        mail[12] = choice + 1;
        __threadfence_system();
        mail[11] = seq;
    }
}
__host__ __device__ inline int index_hash_lds_bytes(int n_nodes, int n_hot) {
    const int nb = hash_buckets(n_nodes);
    return (2 * nb + 1) * 4 + ((n_hot + 1) & ~1) * 2 + 16;
}

// ---- a declared alert set over a population of up to kIndexFusedMaxNodes nodes: the whole index in ONE launch ----------------
// The round index is on every round's path, ahead of the tally, and as two kernels (touch, then one workgroup) it lived on
// dependent memory round trips: the per-node ring masks written by atomics of the touch pass and read back by the build, the
// dictionary written to memory and read back entry by entry for the adjacency, every table walked twice -- ~45 us at N = 10^4
// against a tally of 380 us.  Here ONE workgroup keeps everything it computes in LDS (per node: the rings the alert set names
// -> the declared mask, the dictionary entry, the member flag: 5 bytes) and touches memory three times: the alert set and
// the member flags (independent loads, one round trip), the observer rows of the hot subjects (one round trip), and the
// finished tables written out (stores).  Same outputs, bit for bit, as index_touch_kernel + index_build_block_kernel
// (tests/test_kernel_emulated.py compares them); the global work area of the two-kernel form is not touched.
constexpr int kIndexFusedMaxNodes = 24576;
constexpr int kIndexFusedNosLds = 4096;  // slot -> node for the first slots in LDS (the rest is read back from memory)
__host__ __device__ inline int index_fused_lds_bytes(int n_nodes) {
    const int n2 = (n_nodes + 1) & ~1;
    return 2 * align16(n2 * 2 + 64) + align16(n_nodes + 1) + kIndexFusedNosLds * 4;
}

__global__ __launch_bounds__(1024) void index_fused_kernel(const unsigned char* alerts, long long n_alerts, long long cfg_id,
                                                           const unsigned char* member, const int* obs, int n_nodes, int K, int L,
                                                           unsigned short* dict, unsigned short* decl, int* node_of_slot,
                                                           unsigned short* smask, unsigned int* pairs, int adj_cap, unsigned int* tbits,
                                                           unsigned short* trank, unsigned int* tent, int tent_cap, volatile int* info_out,
                                                           int direct_budget, unsigned long long* zero_words, int n_zero_words,
                                                           unsigned int* zero_flags, int seq, int* q4_rows, unsigned char* q4_valid,
                                                           unsigned int* entries) {
    __shared__ int s_wave[16];
    __shared__ unsigned int s_flags, s_stale;
    unsigned char* const lds = dynamic_lds();
    const int T = (int)blockDim.x, t = (int)threadIdx.x;
    const int n2 = (n_nodes + 1) & ~1;
    const int tab_bytes = align16(n2 * 2 + 64);  // (+ 64: the compressed tables read a word's 32 entries as four 16-byte loads)
    unsigned short* const l_decl = reinterpret_cast<unsigned short*>(lds);  // first the rings the alert set names, then the declared mask | member << 15
    unsigned int* const l_decl32 = reinterpret_cast<unsigned int*>(lds);
    unsigned short* const l_dict = reinterpret_cast<unsigned short*>(lds + tab_bytes);
    unsigned char* const l_mem = lds + 2 * tab_bytes;
    int* const l_nos = reinterpret_cast<int*>(lds + 2 * tab_bytes + align16(n_nodes + 1));
    const unsigned int cfg_lo = (unsigned int)(unsigned long long)cfg_id, cfg_hi = (unsigned int)((unsigned long long)cfg_id >> 32);
    const unsigned int kmask = (1u << K) - 1u;

    // ---- the alert set and the member flags: all loads of the first batch are in flight before anything waits ----
    constexpr int kAB = 8;
    unsigned int a_c0[kAB], a_c1[kAB], a_dst[kAB], a_w4[kAB];
    auto load_alerts = [&](long long i0) {
#pragma unroll
        for (int j = 0; j < kAB; ++j) {
            const long long i = i0 + (long long)j * T + t;
            a_c0[j] = a_c1[j] = a_w4[j] = 0u;
            a_dst[j] = 0xFFFFFFFFu;
            if (i < n_alerts) {
                const unsigned int* w = reinterpret_cast<const unsigned int*>(alerts + i * 20);
                a_c0[j] = w[0];
                a_c1[j] = w[1];
                a_dst[j] = w[3];
                a_w4[j] = w[4];
            }
        }
    };
    load_alerts(0);
    for (int n = t; n < n_nodes; n += T) l_mem[n] = member[n];
    for (int i = t; i < tab_bytes / 4; i += T) l_decl32[i] = 0u;
    for (int i = t; i < n_zero_words; i += T) zero_words[i] = 0ull;  // (see index_build_block_kernel)
    if (t < 2 && zero_flags != nullptr) zero_flags[t] = 0u;
    if (t == 0) {
        s_flags = 0u;
        s_stale = 0u;
    }
    __syncthreads();
    unsigned int f = 0u;
    for (long long i0 = 0; i0 < n_alerts; i0 += (long long)kAB * T) {
        if (i0 > 0) load_alerts(i0);
#pragma unroll
        for (int j = 0; j < kAB; ++j) {
            if (i0 + (long long)j * T + t >= n_alerts) continue;
            const unsigned int dst = a_dst[j], cw = core_word(a_w4[j]);
            const bool current = a_c0[j] == cfg_lo && a_c1[j] == cfg_hi, inr = dst < (unsigned int)n_nodes;
            const unsigned int bits = cw & kmask;
            const bool down = (cw & kCoreDown) != 0u;
            if (inr && bits != 0u) atomicOr(&l_decl32[dst >> 1], bits << ((dst & 1u) * 16u));  // (whatever the filter says: a superset, as in index_touch_kernel)
            const bool ok = !current || (inr && bits != 0u && ((l_mem[inr ? dst : 0u] != 0) == down));  // (see index_touch_kernel)
            f |= (ok ? 0u : 1u) | (down ? 0u : 2u) | (current ? 0u : 4u);
        }
    }
    if (f) atomicOr(&s_flags, f);
    __syncthreads();

    // ---- slots (ascending node order), dictionary, declared ring masks: wave w owns a contiguous range of nodes ----
    const int lane = t & 63, wv = t >> 6, nw = T >> 6;
    const int per_wave_nodes = ((n_nodes + nw * 64 - 1) / (nw * 64)) * 64;
    const int beg = min(n_nodes, wv * per_wave_nodes), end = min(n_nodes, beg + per_wave_nodes);
    int nh = 0;
    for (int n0 = beg; n0 < end; n0 += 64) {
        const int n = n0 + lane;
        const unsigned int g = n < end ? (unsigned int)l_decl[n] : 0u;
        nh += __popcll(__ballot(__popc(g) >= L));
    }
    int n_hot_all = 0;
    int ph = block_exclusive_scan(lane == 0 ? nh : 0, s_wave, &n_hot_all);
    ph = __shfl(ph, 0, 64);
    for (int n0 = beg; n0 < end; n0 += 64) {
        const int n = n0 + lane;
        const bool in = n < end;
        const unsigned int g = in ? (unsigned int)l_decl[n] : 0u;
        const unsigned int mem = (in && l_mem[n] != 0) ? 0x8000u : 0u;
        const bool hot = in && __popc(g) >= L;
        const unsigned long long hots = __ballot(hot);
        const int mine_slot = ph + __popcll(hots & ((1ull << lane) - 1ull));
        ph += __popcll(hots);
        unsigned int slot = 0x3FFFu;
        if (hot && mine_slot < 16319) {
            slot = (unsigned int)mine_slot;
            node_of_slot[mine_slot] = n;
            if (mine_slot < kIndexFusedNosLds) l_nos[mine_slot] = n;
        }
        if (in) {
            l_dict[n] = (unsigned short)(slot | mem);
            l_decl[n] = (unsigned short)((slot != 0x3FFFu ? 0x3FFFu : (g & 0x3FFFu)) | mem);
        }
    }
    __threadfence_block();
    __syncthreads();
    const int n_hot = min(n_hot_all, 16318);

    // ---- the hot adjacency (see index_build_block_kernel; quirk Q4: a hot member's row comes from the memo) ----
    const int per2 = (n_hot + T - 1) / T;
    const int b2 = min(n_hot, t * per2), e2 = min(n_hot, b2 + per2);
    int mine = 0;
    bool stale = false;
    constexpr int kKMax = 14;
    auto node_of = [&](int e) -> int { return e < kIndexFusedNosLds ? l_nos[e] : node_of_slot[e]; };
    auto slot_edges = [&](int e, unsigned int (&eo)[kKMax]) -> unsigned int {
        const int node = node_of(e);
        const bool use_memo = q4_valid != nullptr && l_mem[node] != 0;
        const int* const today = obs + (long long)node * K;
        int* const memo = q4_rows + (long long)node * K;
        int o[kKMax], td[kKMax], mm[kKMax];
        // (today's row, the memo flag and the memoised row are requested together: one round trip, not three)
        const bool have = use_memo && q4_valid[node] != 0;
#pragma unroll
        for (int k = 0; k < kKMax; ++k) td[k] = k < K ? today[k] : -1;
#pragma unroll
        for (int k = 0; k < kKMax; ++k) mm[k] = (use_memo && k < K) ? memo[k] : -1;
        if (have) {
#pragma unroll
            for (int k = 0; k < kKMax; ++k) {
                o[k] = mm[k];
                stale = stale || o[k] != td[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < kKMax; ++k) o[k] = td[k];
        }
        unsigned int am = 0u;
#pragma unroll
        for (int k = 0; k < kKMax; ++k) eo[k] = (o[k] >= 0 && o[k] < n_nodes) ? ((unsigned int)l_dict[o[k]] & 0x3FFFu) : 0x3FFFu;
#pragma unroll
        for (int k = 0; k < kKMax; ++k) am |= (int)eo[k] < n_hot ? 1u << k : 0u;
        return am;
    };
    // (a thread owns at most a few slots: their edges are kept between the counting and the writing pass when it owns one)
    unsigned int eo1[kKMax];
    unsigned int am1 = 0u;
    for (int e = b2; e < e2; ++e) {
        unsigned int eo[kKMax];
        const unsigned int am = slot_edges(e, eo);
        mine += __popc(am);
        if (e == b2) {
            am1 = am;
#pragma unroll
            for (int k = 0; k < kKMax; ++k) eo1[k] = eo[k];
        }
    }
    if (stale) atomicOr(&s_stale, 1u);
    int total = 0;
    int at = block_exclusive_scan(mine, s_wave, &total);
    const bool fits = total <= 65535 && total <= adj_cap;
    for (int e = b2; e < e2; ++e) {
        unsigned int eo[kKMax];
        unsigned int am = am1;
        if (e == b2) {
#pragma unroll
            for (int k = 0; k < kKMax; ++k) eo[k] = eo1[k];
        } else {
            am = slot_edges(e, eo);
        }
        const int node = node_of(e);
        if (q4_valid != nullptr && l_mem[node] != 0 && q4_valid[node] == 0) {  // the first time this member is hot since its entry was dropped
            const int* const today = obs + (long long)node * K;
            for (int k = 0; k < K; ++k) q4_rows[(long long)node * K + k] = today[k];
            q4_valid[node] = 1;
        }
#pragma unroll
        for (int k = 0; k < kKMax; ++k) {
            if ((am >> k) & 1u) {
                if (fits) pairs[at] = (unsigned int)e | (eo[k] << 14) | ((unsigned int)k << 28);
                ++at;
            }
        }
        smask[e] = (unsigned short)am;
        if (am != 0u) l_dict[node] |= (unsigned short)0x4000;  // (only this thread touches the node's entry after the barrier above)
    }
    __syncthreads();  // l_dict is final (adjacency flags included)

    // ---- the compressed form of the tables, when the direct ones will not fit the tally's LDS (see index_build_block_kernel) ----
    const bool direct_fits = direct_budget >= 0 &&
                             tally_shared_bytes(kDictDirect, n_nodes, 0, n_hot, total) + 8 * tally_wave_bytes(n_hot) <= direct_budget;
    const int n_words = direct_fits ? 0 : (n_nodes + 31) / 32;
    const int perw = (n_words + T - 1) / T;
    const int w0 = min(n_words, t * perw), w1 = min(n_words, w0 + perw);
    int cnt = 0;
    for (int w = w0; w < w1; ++w) {
        unsigned int bits = 0u;
        const int valid = min(32, n_nodes - w * 32);
        for (int b = 0; b < valid; ++b)
            if (((unsigned int)l_decl[w * 32 + b] & 0x3FFFu) != 0u) bits |= 1u << b;
        tbits[w] = bits;
        cnt += __popc(bits);
    }
    int n_touched = 0;
    int rk = block_exclusive_scan(cnt, s_wave, &n_touched);
    const bool tfits = n_touched <= 65535 && n_touched <= tent_cap;
    for (int w = w0; w < w1; ++w) {
        trank[w] = (unsigned short)(rk > 65535 ? 65535 : rk);
        const int valid = min(32, n_nodes - w * 32);
        for (int b = 0; b < valid; ++b) {
            const int n = w * 32 + b;
            if (((unsigned int)l_decl[n] & 0x3FFFu) == 0u) continue;
            if (tfits) tent[rk] = dict_entry((unsigned int)l_decl[n], (unsigned int)l_dict[n] & 0x3FFFu);
            ++rk;
        }
    }
    // ---- the finished per-node tables, written out two entries at a time ----
    if (t == 0 && (n_nodes & 1) != 0) l_dict[n_nodes] = 0;  // (the odd entry behind the last node: written, read by nobody)
    __syncthreads();
    for (int i = t; i < n2 / 2; i += T) {
        reinterpret_cast<unsigned int*>(dict)[i] = reinterpret_cast<const unsigned int*>(l_dict)[i];
        reinterpret_cast<unsigned int*>(decl)[i] = l_decl32[i];
    }
    // ... and the 32-bit entries the tally stages (or gathers from): what dict_entries_kernel writes behind the other forms
    for (int i = t; i <= n_nodes; i += T) {
        unsigned int e = kEntryPoison | ((unsigned int)n_hot_all << 17);
        if (i < n_nodes) {
            unsigned int sl = (unsigned int)l_dict[i] & kSlotMask;
            if (sl == kNoSlot) sl = (unsigned int)n_hot_all + ((unsigned int)i & (unsigned int)(kDummySlots - 1));
            e = dict_entry((unsigned int)l_decl[i], sl);
        }
        entries[i] = e;
    }
    __syncthreads();
    // The answer goes into the host-mapped page the host polls.  Eight lanes store the eight words with ONE instruction: a volatile
    // store to that page is waited for before the next one is issued -- nine stores in a row by one thread were nine round trips
    // over the host link at the tail of a kernel the whole round waits for.
    // Ordering: a thread's fence orders that thread's own stores only, so each of the eight storing lanes fences its word, the
    // workgroup barrier orders all of that before lane 0, and lane 0 publishes the sequence word behind a fence of its own -- the
    // release pattern vote_verify_kernel uses, not an assumption about the lanes of a wave moving in lock step.
    if (t < 8) {
        const int v = t <= 1 ? n_hot_all
                    : t == 2 ? ((s_stale != 0u ? 4 : 0) | (n_hot_all > 16318 ? 1 : 0) | (fits ? 0 : 2))
                    : t == 3 ? total
                    : t == 4 ? (int)s_flags
                    : t == 5 ? n_touched
                    : t == 6 ? (tfits && !direct_fits ? 1 : 0)
                             : (direct_fits ? 1 : 0);
        info_out[t] = v;
        __threadfence_system();
    }
    __syncthreads();
    if (t == 0) {
        __threadfence_system();
        info_out[15] = seq;
    }
}

// ---- rapid_sim_generate: the delivered streams made on the device ----------------------------------------------------
// Every receiver gets every BatchedAlertMessage of the round exactly once, in a receiver-specific seeded order (the
// reference's fan-out: UnicastToAllBroadcaster.java:46-63 sends each batch to all members; arrival order differs per
// receiver -- paper Fig.11 methodology).  The order is a seeded PERMUTATION evaluated in place, not a sort (gen_perm_at below: two
// levels, Feistel networks on mixed-radix domains, cycle walking -- a bijection of a domain restricted to a subset is a bijection of
// the subset), keyed by mix64(seed + node index of r).  Any position is computed in O(1) by itself, so one WAVE per receiver lays
// the stream down 64 deliveries at a time with nothing but a prefix sum of the batch lengths over its lanes in between -- no keys
// in memory, no sort, no barrier, no LDS, no limit on receivers x batches.  (Round 4 ran a four-round network on the next power of
// two above n_batches: up to half of the positions walked, and a wave walks as long as its unluckiest lane -- five to six network
// evaluations per delivery at 1.4 x 10^5 batches.)  rapid_amd/scenarios.py: hashed_order / deliver_hashed state the same on the host.
// keep != nullptr: batch b reaches receiver r only if (uint32)(mix64(keepk_r + b) >> 32) <= keep[b] -- late deliveries of an
// earlier configuration (R/MembershipService.java:653-657 drops them) and lossy links reach SOME receivers; the places of an
// undelivered batch hold empty records (no ring, no batch end: what the zeros behind a stream's end are), so that every
// stream keeps its fixed length and no second pass has to count.
__host__ __device__ inline unsigned long long gen_mix64(unsigned long long x) {  // splitmix64 finaliser (== mix64 of tally_kernel.h)
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// The delivery order of a receiver is a permutation with TWO levels: the batch list is cut into lines of kGenLine consecutive
// batches (their table entries fill one 128-byte cache line), the LINES are delivered in a receiver-specific pseudo-random order,
// the batches of a line one after the other in a line-and-receiver-specific order.  Why: lane l of a step fetches its batch's table
// entry; with every position on a line of its own that is one cache-line fill per delivery -- 4 x 10^8 per 4,096 receivers at
// N = 10^6, 1.9 of the generator's 3.1 ms, at the rate the GPU turns scattered addresses into lines (2.65 x 10^11 per second,
// profiles/r05_gather_rate.txt; probe builds without the gather: profiles/r05_generator_probes.txt) -- and with eight positions per
// line it is one fill per eight deliveries.  A receiver still gets every batch exactly once in an order of its own; what the two
// levels give up is entropy nobody uses: which eight list neighbours arrive back to back (their subjects are unrelated: the list
// is in sender order).
constexpr unsigned int kGenLineBits = 3, kGenLine = 1u << kGenLineBits;
struct GenPerm {
    unsigned int rk[3];        // round keys: two for the order of the lines, one for the order inside a line
    unsigned long long keepk;  // key of the per-batch delivery draw
    unsigned int n;            // batches
    unsigned int lines;        // ceil(n / kGenLine)
    unsigned int s;            // the right half of a line number has s bits: a = 1 << s
    unsigned int mask_r;       // a - 1
    unsigned int b;            // radix of the left half: ceil(lines / a) <= 2^16
};
__host__ __device__ inline GenPerm gen_perm_make(unsigned long long seed, unsigned int receiver_node, unsigned int n_batches) {
    GenPerm g;
    const unsigned long long key = gen_mix64(seed + (unsigned long long)receiver_node);
    g.rk[0] = (unsigned int)(gen_mix64(key + 1ull) >> 32);  // (spelled out: a loop over rk[] left the round keys in scratch memory)
    g.rk[1] = (unsigned int)(gen_mix64(key + 2ull) >> 32);
    g.rk[2] = (unsigned int)(gen_mix64(key + 3ull) >> 32);
    g.keepk = gen_mix64(key ^ 0xD1B54A32D192ED03ull);
    g.n = n_batches;
    g.lines = (n_batches + kGenLine - 1u) >> kGenLineBits;
    unsigned int s = 1;
    while (s < 16u && (1ull << (2 * s)) < (unsigned long long)g.lines) ++s;  // a * a >= lines
    g.s = s;
    g.mask_r = (1u << s) - 1u;
    g.b = (unsigned int)(((unsigned long long)g.lines + g.mask_r) >> s);
    if (g.b == 0u) g.b = 1u;
    return g;
}
// The round function: sixteen pseudo-random bits of a half line number x < 2^16 under a round key.  Both multiplies take 24-bit
// operands -- one full-rate v_mul_u32_u24 each where a 32-bit multiply costs four issue slots.
__host__ __device__ inline unsigned int gen_f16(unsigned int x, unsigned int k) {
    unsigned int h = ((x ^ k) & 0xFFFFFFu) * 0x9E3779u;
    h ^= h >> 15;
    h = (h & 0xFFFFFFu) * 0x85EBCBu;
    return h >> 16;
}
// q < lines -> the line delivered q-th: a two-round alternating Feistel network on [0, b) x [0, a) -- the left half moved by a
// function of the right one, then the right half by a function of the new left one; each step is a bijection whatever the function
// -- walked until it lands below `lines` (the domain a * b exceeds it by less than a).
__host__ __device__ inline unsigned int gen_line_at(const GenPerm& g, unsigned int q) {
    unsigned int x = q;
    do {
        unsigned int r = x & g.mask_r, l = x >> g.s;
        l += (gen_f16(r, g.rk[0]) * (g.b & 0xFFFFFFu)) >> 16;  // (16 bits x at most 16 bits) -> [0, b); the mask says so to the compiler: v_mul_u32_u24
        l -= l >= g.b ? g.b : 0u;
        r = (r + gen_f16(l, g.rk[1])) & g.mask_r;
        x = (l << g.s) | r;
    } while (x >= g.lines);
    return x;
}
// j < n -> the batch delivered j-th: position (q, t) = (j / 8, j % 8) holds batch 8 Q + T, Q = gen_line_at(q) and T = t under an
// affine map of the line's eight places (an odd multiplier, an offset and a mask, all drawn from the line and the receiver); the map
// (q, t) -> (Q, T) is a bijection of [0, 8 lines), walked until it lands below n (only places of the last line can lie beyond it).
__host__ __device__ inline unsigned int gen_perm_at(const GenPerm& g, unsigned int j) {
    if (g.n <= 1u) return 0u;
    unsigned int x = j;
    do {
        const unsigned int Q = gen_line_at(g, x >> kGenLineBits);
        const unsigned int h = gen_f16(Q & 0xFFFFu, g.rk[2] ^ (Q >> 16));
        const unsigned int T = ((((x & (kGenLine - 1u)) ^ (h >> 8)) * ((h & 6u) | 1u)) + (h >> 3)) & (kGenLine - 1u);
        x = (Q << kGenLineBits) | T;
    } while (x >= g.n);
    return x;
}
__host__ __device__ inline bool gen_delivered(const GenPerm& g, const unsigned int* keep, unsigned int b) {
    return keep == nullptr || (unsigned int)(gen_mix64(g.keepk + (unsigned long long)b) >> 32) <= keep[b];
}

// The round's distinct alerts, resolved once per generation: res[a] = {dict_entry of alert a's subject, its core word without
// the batch end}; an alert of another configuration id or about an unknown node gets the poison entry entries[n_nodes]
// (dropped by the per-delivery filter; engine.hip: gen_clean).
__global__ void gen_resolve_alerts_kernel(const unsigned char* alerts, long long n_alerts, long long cfg_id, unsigned int n_nodes,
                                          const unsigned int* entries, uint2* res) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_alerts) return;
    const unsigned int cfg_lo = (unsigned int)(unsigned long long)cfg_id, cfg_hi = (unsigned int)((unsigned long long)cfg_id >> 32);
    const unsigned int* w = reinterpret_cast<const unsigned int*>(alerts + i * 20);
    const bool current = w[0] == cfg_lo && w[1] == cfg_hi;
    const unsigned int d = current && w[3] < n_nodes ? w[3] : n_nodes;
    res[i] = make_uint2(entries[d], core_word(w[4]) & ~kCoreEob);
}

// bat[b] = {first alert of batch b, its length, the first alert's resolved record}: what a delivery needs to know about its batch
// in ONE 16-byte gather -- and, for the nine batches in ten that hold a single alert, the record itself (res == nullptr, boundary
// records: zeros there).  The generator is bound by the rate at which a CU turns scattered addresses into cache lines (2.65 x 10^11
// scattered reads per second over the whole GPU, profiles/r05_gather_rate.txt): a second gather per delivery halves it.
// Bit 31 of the length: the batch reaches EVERY receiver (no per-batch draw, or one that cannot fail) -- true of all but the late
// batches of a round; the generator then neither evaluates the draw (a 64-bit mix: two quarter-rate multiplies) nor gathers keep[b].
constexpr unsigned int kGenAlways = 0x80000000u;
__global__ void gen_pack_batches_kernel(const long long* boff, int n_batches, const uint2* res, const unsigned int* keep, uint4* bat) {
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= n_batches) return;
    const unsigned int f = (unsigned int)boff[b];
    const uint2 r0 = res != nullptr ? res[f] : make_uint2(0u, 0u);
    const unsigned int always = (keep == nullptr || keep[b] == 0xFFFFFFFFu) ? kGenAlways : 0u;
    bat[b] = make_uint4(f, (unsigned int)(boff[b + 1] - boff[b]) | always, r0.x, r0.y);
}

// ONE WAVE PER RECEIVER (a workgroup = kGenWavesPerBlock receivers; no barrier, no LDS): the wave walks its receiver's delivery
// order kGenChunks x 64 deliveries at a time -- lane l of chunk u evaluates position j0 + 64 u + l of the permutation and gathers
// its batch's {first alert, length} (all kGenChunks gathers in flight together), an inclusive scan over the lanes turns the
// lengths into output positions, and the records go out: a batch of up to kGenLaneLoop alerts is written by its own lane, alert by
// alert (consecutive lanes hold consecutive deliveries, so a step's stores fall into one contiguous stretch of the stream); a
// chunk holding a longer batch is written record by record instead, every output record finding its delivery by a binary search
// over the lanes' positions (six shuffles), whatever the lengths are.  alerts = the round's distinct alerts (20-byte records) in
// batch order; every receiver's stream has n_alerts records: rec_off[r] = r * n_alerts.  boundary == 0: 8-byte resident records
// {entry, core word} from res[]; else the 20-byte boundary records themselves (flags bit 0 = the batch end).  A batch ends with
// its last alert, whatever the flags of the set say.  (Round 4: a workgroup per receiver, 1,024 deliveries per step between two
// barriers, every output record found by a ten-step binary search in LDS: 1.8 - 2.5 ms per 1.5 x 10^8 records at 10^6 nodes.)
#ifndef RAPID_GEN_PRIO
#define RAPID_GEN_PRIO 0  // (measurement knob: the issue priority of the generator's waves beside a tally launch, s_setprio)
#endif
constexpr int kGenWavesPerBlock = 4;
constexpr int kGenChunks = 4;
constexpr int kGenLaneLoop = 4;
__global__ __launch_bounds__(kGenWavesPerBlock * 64) void gen_streams_kernel(const uint2* res, const unsigned char* alerts, const uint4* bat, int n_batches,
                                                                             const unsigned int* keep, const int* receivers, int n_receivers,
                                                                             long long n_alerts, unsigned long long seed, unsigned char* out,
                                                                             int boundary) {
    const int lane = (int)(threadIdx.x & 63u);
    const int r = (int)blockIdx.x * kGenWavesPerBlock + (int)(threadIdx.x >> 6);
    if (r >= n_receivers) return;
#if RAPID_GEN_PRIO
    __builtin_amdgcn_s_setprio(RAPID_GEN_PRIO);
#endif
    const GenPerm g = gen_perm_make(seed, (unsigned int)receivers[r], (unsigned int)n_batches);
    long long at = (long long)r * n_alerts;  // the stream's next record
    // record k of the batch whose first alert is f, `last`: it closes the batch; delivered == false: an empty record
    // (r0 = the batch's first record as it came with the batch's table entry: resolved records only)
    auto put = [&](long long i_out, unsigned int f, unsigned int k, bool last, bool delivered, uint2 r0) {
        if (boundary == 0) {
            uint2 v = make_uint2(0u, 0u);
            if (delivered) {
                v = r0;  // (not `k == 0 ? r0 : res[f + k]`: a select between a register pair and memory is compiled as a select of ADDRESSES --
                if (k != 0u) v = res[f + k];  //  r0 spilled to scratch memory so that it has one, and a flat load through the chosen pointer)
                v.y |= last ? kCoreEob : 0u;
            }
            reinterpret_cast<uint2*>(out)[i_out] = v;  // (stored non-temporally: 139 instead of 99 ms per C5 shard round -- L2 merges these)
        } else {
            unsigned int w[5] = {0u, 0u, 0u, 0u, 0u};
            if (delivered) {
                const unsigned int* src = reinterpret_cast<const unsigned int*>(alerts + (long long)(f + k) * 20);
#pragma unroll
                for (int q = 0; q < 5; ++q) w[q] = src[q];
                w[4] = (w[4] & ~0x01000000u) | (last ? 0x01000000u : 0u);
            }
            unsigned int* dst = reinterpret_cast<unsigned int*>(out + i_out * 20);
#pragma unroll
            for (int q = 0; q < 5; ++q) dst[q] = w[q];
        }
    };
    static_assert(64 * kGenChunks / (int)kGenLine <= 64, "the lines of one step are evaluated by the lanes of one wave");
    // A STEP = 64 kGenChunks positions of the order.  locate(): which batches they hold, and the gathers of those batches' table
    // entries REQUESTED; lay(): the entries used -- lengths summed over the lanes, records written.  The wave requests step s + 1
    // before it lays step s down: the gather (a scattered read out of L2, a microsecond or two with a tally launch running beside
    // it) is the longest wait of a step, and with four or five waves per SIMD -- one receiver per wave, a tile of 4,096 receivers --
    // nothing else was there to cover it.  The loads are unconditional (positions beyond the order ask for batch 0): a load under a
    // branch costs the compiler's wait-count pass its count, and it then waits for everything in flight.
    // The step's 256 positions lie on 32 lines of the order: lane l evaluates line (j0 / 8) + (l mod 32) ONCE -- the Feistel network
    // and the line's hash, two thirds of gen_perm_at's instructions -- and the eight positions of a line fetch the result from that
    // lane (gen_perm_at evaluates it in each of the eight lanes).  Same permutation: gen_perm_at's first turn, spelled out; a
    // position that lands beyond n (the last line only) walks on through gen_perm_at itself.
    auto locate = [&](int j0, uint4 (&bt)[kGenChunks], unsigned int (&bsel)[kGenChunks]) {
        unsigned int Ql = 0u, hl = 0u;
        if (g.n > 1u) {
            const unsigned int ql = min(((unsigned int)j0 >> kGenLineBits) + (unsigned int)(lane & 31), g.lines - 1u);
            Ql = gen_line_at(g, ql);
            hl = gen_f16(Ql & 0xFFFFu, g.rk[2] ^ (Ql >> 16));
        }
#pragma unroll
        for (int u = 0; u < kGenChunks; ++u) {
            const int j = j0 + 64 * u + lane;
            const int src = u * (64 / (int)kGenLine) + (lane >> kGenLineBits);
            const unsigned int Q = (unsigned int)__shfl((int)Ql, src, 64), h = (unsigned int)__shfl((int)hl, src, 64);
            unsigned int b = 0u;
            if (j < n_batches && g.n > 1u) {
                const unsigned int T = (((((unsigned int)j & (kGenLine - 1u)) ^ (h >> 8)) * ((h & 6u) | 1u)) + (h >> 3)) & (kGenLine - 1u);
                b = (Q << kGenLineBits) | T;
                if (b >= g.n) b = gen_perm_at(g, b);
            }
            bsel[u] = b;
            bt[u] = bat[b];
        }
    };
    auto lay = [&](int j0, const uint4 (&bt)[kGenChunks], const unsigned int (&bsel)[kGenChunks]) {
        unsigned int first[kGenChunks], len[kGenChunks];
        uint2 rec0[kGenChunks];
        bool del[kGenChunks];
#pragma unroll
        for (int u = 0; u < kGenChunks; ++u) {
            const bool in = j0 + 64 * u + lane < n_batches;
            first[u] = in ? bt[u].x : 0u;
            len[u] = in ? bt[u].y & ~kGenAlways : 0u;
            rec0[u] = in ? make_uint2(bt[u].z, bt[u].w) : make_uint2(0u, 0u);
            del[u] = in && ((bt[u].y & kGenAlways) != 0u || gen_delivered(g, keep, bsel[u]));
        }
#pragma unroll
        for (int u = 0; u < kGenChunks; ++u) {
            if (j0 + 64 * u >= n_batches) continue;  // (wave-uniform; `continue`, so that the loop unrolls and first[] / len[] stay in registers)
            const int incl = wave_inclusive_sum((int)len[u]);  // (seven adds on the DPP paths: stream_load.h)
            const int start = incl - (int)len[u];
            const int total = __shfl(incl, 63, 64);
            const bool any_long = __ballot(len[u] > (unsigned int)kGenLaneLoop) != 0ull;
            if (!any_long) {
                for (unsigned int k = 0; __ballot(len[u] > k) != 0ull; ++k)
                    if (len[u] > k) put(at + start + (long long)k, first[u], k, k + 1u == len[u], del[u], rec0[u]);
            } else {
                for (int o0 = 0; o0 < total; o0 += 64) {
                    const int o = o0 + lane;
                    int lo = 0;  // the last lane whose batch starts at or before output record o (lengths are >= 1 for real deliveries)
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1) {
                        const int mid = lo + step;
                        const int s_mid = __shfl(start, mid & 63, 64);
                        const int l_mid = __shfl((int)len[u], mid & 63, 64);
                        if (mid < 64 && l_mid > 0 && s_mid <= o) lo = mid;
                    }
                    const int s_lo = __shfl(start, lo, 64);
                    const unsigned int f_lo = (unsigned int)__shfl((int)first[u], lo, 64), n_lo = (unsigned int)__shfl((int)len[u], lo, 64);
                    const int d_lo = __shfl(del[u] ? 1 : 0, lo, 64);
                    const uint2 r_lo = make_uint2((unsigned int)__shfl((int)rec0[u].x, lo, 64), (unsigned int)__shfl((int)rec0[u].y, lo, 64));
                    if (o < total) {
                        const unsigned int k = (unsigned int)(o - s_lo);
                        put(at + o, f_lo, k, k + 1u == n_lo, d_lo != 0, r_lo);
                    }
                }
            }
            at += total;
        }
    };
    constexpr int kStepPos = 64 * kGenChunks;
    uint4 ta[kGenChunks], tb[kGenChunks];
    unsigned int ba[kGenChunks], bb[kGenChunks];
    if (n_batches > 0) locate(0, ta, ba);
    for (int j0 = 0; j0 < n_batches; j0 += 2 * kStepPos) {
        locate(j0 + kStepPos, tb, bb);
        lay(j0, ta, ba);
        locate(j0 + 2 * kStepPos, ta, ba);
        if (j0 + kStepPos < n_batches) lay(j0 + kStepPos, tb, bb);
    }
}
__global__ void gen_offsets_kernel(long long* rec_off, int n_receivers, long long n_alerts) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i <= n_receivers) rec_off[i] = (long long)i * n_alerts;
}


}  // namespace rapid
