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

namespace emu {
struct Dim {
    unsigned x = 0, y = 0, z = 0;
};
struct Wave {
    static constexpr int W = 64;
    ucontext_t sched;
    ucontext_t ctx[W];
    std::vector<char> stacks[W];
    bool done[W];
    int op[W];           // collective id the lane is parked at (0 = none)
    uint64_t arg[W];     // deposited operand
    uint64_t aux[W];     // second operand (shuffle source lane)
    uint64_t res[W];     // result handed back
    int cur = -1;
    uint64_t rng = 0x1234567;
    std::function<void()> body;
};
extern Wave* g_wave;
extern Dim g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
enum { OP_NONE = 0, OP_BALLOT = 1, OP_SHFL = 2, OP_SYNC = 3 };

inline uint64_t park(int op, uint64_t a, uint64_t b) {
    Wave* w = g_wave;
    const int l = w->cur;
    w->op[l] = op;
    w->arg[l] = a;
    w->aux[l] = b;
    swapcontext(&w->ctx[l], &w->sched);
    return w->res[l];
}
void run_block(unsigned block, unsigned grid, const std::function<void()>& body, uint64_t seed);
}  // namespace emu

#define threadIdx (emu::g_threadIdx)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

inline unsigned long long __ballot(int p) { return emu::park(emu::OP_BALLOT, p ? 1 : 0, 0); }
inline int __shfl(int v, int src, int = 64) { return (int)(uint32_t)emu::park(emu::OP_SHFL, (uint32_t)v, (unsigned)src & 63u); }
inline unsigned __shfl(unsigned v, int src, int = 64) { return (unsigned)emu::park(emu::OP_SHFL, v, (unsigned)src & 63u); }
inline int __shfl_xor(int v, int m, int = 64) {
    return (int)(uint32_t)emu::park(emu::OP_SHFL, (uint32_t)v, (unsigned)(emu::g_wave->cur ^ m) & 63u);
}
inline unsigned __shfl_xor(unsigned v, int m, int = 64) {
    return (unsigned)emu::park(emu::OP_SHFL, v, (unsigned)(emu::g_wave->cur ^ m) & 63u);
}
inline void __syncthreads() { (void)emu::park(emu::OP_SYNC, 0, 0); }

inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
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
