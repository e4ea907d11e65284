// Asynchronous global -> LDS copies (gfx950 LDS-DMA: buffer_load_dwordx4 ... lds) for the tally kernel's record
// stream.  The loads are issued from inline assembly, so the compiler's s_waitcnt bookkeeping does not know them: it
// never drains them on its own, and completion is counted by hand with wait_dma<N>() -- vector-memory loads complete
// in issue order, so "at most N younger loads outstanding" means everything older has landed.  Any additional
// compiler-visible vector-memory traffic only makes a wait_dma<N>() wait for more than it needs, never less.
//
// tests/emu/ shadows this header with a CPU model of the same four entry points (the copy is delayed until the
// covering wait, the destination is poisoned in between).
#pragma once
#include <hip/hip_runtime.h>

namespace rapid {

typedef int dma_rsrc_t __attribute__((ext_vector_type(4)));  // buffer resource descriptor (V#), wave-uniform
typedef unsigned int lds_addr_t;                             // byte address inside the workgroup's LDS allocation

// `bytes` readable bytes at `base`; reads past the end return zeros without touching memory.
__device__ __forceinline__ dma_rsrc_t dma_make_rsrc(const void* base, unsigned int bytes) {
    const unsigned long long b = (unsigned long long)base;
    dma_rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned int)b);
    r.y = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xFFFFull));  // stride 0: raw buffer
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;  // DATA_FORMAT = 32-bit, no swizzle (same word the raw-buffer builtins are given in this file set)
    return r;
}

// Declares a descriptor wave-uniform again after it travelled through something the compiler does not prove uniform.
__device__ __forceinline__ dma_rsrc_t dma_uniform(dma_rsrc_t r) {
    dma_rsrc_t u;
    u.x = __builtin_amdgcn_readfirstlane(r.x);
    u.y = __builtin_amdgcn_readfirstlane(r.y);
    u.z = __builtin_amdgcn_readfirstlane(r.z);
    u.w = __builtin_amdgcn_readfirstlane(r.w);
    return u;
}

__device__ __forceinline__ lds_addr_t lds_address(const void* p) {
    return (lds_addr_t)(unsigned long long)(__attribute__((address_space(3))) const char*)p;
}

// Declares an LDS address wave-uniform (it is: every lane computes it from the wave index), so that it lives in an SGPR.
__device__ __forceinline__ lds_addr_t lds_uniform(lds_addr_t a) { return (lds_addr_t)__builtin_amdgcn_readfirstlane((int)a); }

// One wave instruction (non-temporal: every byte of the stream is read once): lane i copies the 16 bytes at (rsrc base + lane_off + soff) to LDS address lds_dst + 16 * i.
// lds_dst and soff must be wave-uniform (SGPRs).  M0 carries the LDS destination and is left modified: hipcc has no
// use for M0 in these kernels (no LDS-direct, GWS, sendmsg or movrel code) and rewrites it before any use of its own.  s_mov + s_nop 3 give the five wait states that cover both the
// M0-write -> LDS-DMA hazard and a VALU-written SGPR operand (an SGPR reloaded from a spill lane) -> VMEM hazard.
__device__ __forceinline__ void lds_dma16(dma_rsrc_t rsrc, unsigned int lane_off, unsigned int soff, lds_addr_t lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %3\n\t"
        "s_nop 3\n\t"
        "buffer_load_dwordx4 %0, %1, %2 offen nt lds"
        :
        : "v"(lane_off), "s"(rsrc), "s"(soff), "s"(lds_dst)
        : "memory");
}

// Returns once at most N of this wave's vector-memory loads are still outstanding.
template <int N>
__device__ __forceinline__ void wait_dma() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace rapid
