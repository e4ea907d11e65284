// TEST INFRASTRUCTURE ONLY: a tiny SIMT emulator that shadows <hip/hip_runtime.h> (via -I tests/emu) so the
// *unmodified* kernel sources under rapid_amd/csrc/ can be compiled with g++ and run on the CPU under
// sanitizers.  One ucontext fiber per lane; every wave-collective (__ballot, __shfl*, __syncthreads) is a
// rendezvous of all live lanes; between rendezvous points the lanes run one at a time in a RANDOM order, so
// any dependence on the serialisation order of LDS atomics shows up as a test failure.  A collective reached
// by only part of the live lanes (divergent wave op) aborts.  This is how the tally kernel's logic is checked
// against the oracle without a GPU; it says nothing about performance and is never used by the product.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__
#define __launch_bounds__(...)

struct uint4 {
    unsigned int x, y, z, w;
};
inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
struct uint2 {
    unsigned int x, y;
};
inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }

namespace emu {
struct Dim {
    unsigned x = 0, y = 0, z = 0;
};
constexpr int W = 64;          // lanes per wave
constexpr int MAXT = 16 * 64;  // threads per block
struct Block {
    ucontext_t sched;
    std::vector<ucontext_t> ctx;
    std::vector<std::vector<char>> stacks;
    std::vector<char> done;
    std::vector<int> op;         // collective the fiber is parked at (0 = none)
    std::vector<uint64_t> arg;   // deposited operand
    std::vector<uint64_t> aux;   // second operand (shuffle source lane)
    std::vector<uint64_t> res;   // result handed back
    int nthreads = 0;
    int cur = -1;
    uint64_t rng = 0x1234567;
    std::function<void()> body;
};
extern Block* g_block;
extern Dim g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
enum { OP_NONE = 0, OP_BALLOT = 1, OP_SHFL = 2, OP_WAVE_SYNC = 3, OP_BLOCK_SYNC = 4, OP_FIRST = 5 };

inline uint64_t park(int op, uint64_t a, uint64_t b) {
    Block* w = g_block;
    const int t = w->cur;
    w->op[t] = op;
    w->arg[t] = a;
    w->aux[t] = b;
    swapcontext(&w->ctx[t], &w->sched);
    return w->res[t];
}
inline int cur_lane() { return g_block->cur & 63; }
void run_block(unsigned block, unsigned grid, unsigned nthreads, const std::function<void()>& body, uint64_t seed);
}  // namespace emu

#define threadIdx (emu::g_threadIdx)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

inline unsigned long long __ballot(int p) { return emu::park(emu::OP_BALLOT, p ? 1 : 0, 0); }
inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return emu::park(emu::OP_BALLOT, p ? 1 : 0, 0); }
inline int __shfl(int v, int src, int = 64) { return (int)(uint32_t)emu::park(emu::OP_SHFL, (uint32_t)v, (unsigned)src & 63u); }
inline unsigned __shfl(unsigned v, int src, int = 64) { return (unsigned)emu::park(emu::OP_SHFL, v, (unsigned)src & 63u); }
inline int __shfl_up(int v, unsigned delta, int = 64) {  // lanes below `delta` keep their own value
    const unsigned l = (unsigned)emu::cur_lane();
    return (int)(uint32_t)emu::park(emu::OP_SHFL, (uint32_t)v, l >= delta ? l - delta : l);
}
inline int __shfl_xor(int v, int m, int = 64) {
    return (int)(uint32_t)emu::park(emu::OP_SHFL, (uint32_t)v, (unsigned)(emu::cur_lane() ^ m) & 63u);
}
inline unsigned __shfl_xor(unsigned v, int m, int = 64) {
    return (unsigned)emu::park(emu::OP_SHFL, v, (unsigned)(emu::cur_lane() ^ m) & 63u);
}
inline unsigned long long __shfl_xor(unsigned long long v, int m, int = 64) {
    const unsigned lo = __shfl_xor((unsigned)v, m), hi = __shfl_xor((unsigned)(v >> 32), m);
    return ((unsigned long long)hi << 32) | lo;
}
// readlane / readfirstlane: uniform reads (all live lanes of a wave execute them together)
inline int __builtin_amdgcn_readlane(int v, int src) { return (int)(uint32_t)emu::park(emu::OP_SHFL, (uint32_t)v, (unsigned)src & 63u); }
inline int __builtin_amdgcn_readfirstlane(int v) { return (int)(uint32_t)emu::park(emu::OP_FIRST, (uint32_t)v, 0); }
inline void __syncthreads() { (void)emu::park(emu::OP_BLOCK_SYNC, 0, 0); }
inline int __syncthreads_or(int p) {  // (fibers run one at a time: a plain accumulator between two barriers)
    static int acc;
    __syncthreads();
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    if (p) acc = 1;
    __syncthreads();
    return acc;
}
// wave-local LDS ordering point of the kernels (compiler-only on the device): a rendezvous here, so that the
// emulator keeps shuffling the lane order around every point where lanes exchange data through LDS
inline unsigned int __builtin_amdgcn_mbcnt_lo(unsigned int m, unsigned int add) {
    const unsigned int l = threadIdx.x & 63u;
    return add + (unsigned int)__builtin_popcount(m & (l >= 32u ? 0xFFFFFFFFu : (1u << l) - 1u));
}
inline unsigned int __builtin_amdgcn_mbcnt_hi(unsigned int m, unsigned int add) {
    const unsigned int l = threadIdx.x & 63u;
    return add + (l > 32u ? (unsigned int)__builtin_popcount(m & ((1u << (l - 32u)) - 1u)) : 0u);
}
inline void __builtin_amdgcn_wave_barrier() { (void)emu::park(emu::OP_WAVE_SYNC, 0, 0); }
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __builtin_amdgcn_fence(order, scope) ((void)0)

inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long)v); }
using std::min;
using std::max;

// Lanes are cooperative fibers: nothing else runs between two rendezvous points, so plain RMW is atomic.
inline unsigned atomicOr(unsigned* p, unsigned v) {
    unsigned o = *p;
    *p = o | v;
    return o;
}
inline unsigned atomicAnd(unsigned* p, unsigned v) {
    unsigned o = *p;
    *p = o & v;
    return o;
}
inline unsigned atomicCAS(unsigned* p, unsigned expected, unsigned desired) {
    const unsigned o = *p;
    if (o == expected) *p = desired;
    return o;
}
inline void __threadfence() {}
inline void __threadfence_block() {}
inline void __threadfence_system() {}
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
    const unsigned long long o = *p;
    if (v > o) *p = v;
    return o;
}
inline int atomicMax(int* p, int v) {
    const int o = *p;
    if (v > o) *p = v;
    return o;
}
inline unsigned atomicMin(unsigned* p, unsigned v) {
    const unsigned o = *p;
    if (v < o) *p = v;
    return o;
}
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
    unsigned long long o = *p;
    *p = o + v;
    return o;
}
inline int atomicAdd(int* p, int v) {
    int o = *p;
    *p = o + v;
    return o;
}
inline unsigned atomicAdd(unsigned* p, unsigned v) {
    unsigned o = *p;
    *p = o + v;
    return o;
}
