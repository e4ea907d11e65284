// The tally kernel's record stream: bounds-checked buffer loads straight into registers.
//
// A receiver's delivered stream is a raw buffer (base = its first record, size = 20 B x its record count); lane l of
// "quarter" q of a window reads the dwords it needs of record 64 q + l with ONE wave instruction per quarter
// (buffer_load_dwordx2 ... offen nt, lane stride 20 B: the 64 lanes cover 1,280 contiguous bytes, so every cache line
// the instruction touches is used completely by it and its neighbours).  Reads past the end of the stream return
// zeros without touching memory -- a zero record names no ring, closes no batch and is not a DOWN report, so the
// tail of the last window needs no special case.
//
// The loads are ordinary compiler-visible loads, so the register allocation around them is the compiler's business and
// correct by construction; the next window is requested before the current one is tallied, and the compiler's
// wait-count pass inserts s_waitcnt vmcnt(N) with N = the loads issued since (loads return in issue order).  That
// count is only precise while no OTHER vector-memory result is pending in a long-lived register around the loop: a
// vector load whose destination stays live across window iterations (round 2 first had the next receiver's stream
// bounds prefetched that way) makes the pass wait with vmcnt(0) -- i.e. for the window it has just requested --
// before every write of a register that may still be that load's target, and tallying and streaming then do not
// overlap at all (measured: idle time between receivers was fully additive).  Hence stream_scalar_load below: the
// bounds come through the scalar cache (lgkmcnt), and what remains in the loop is loads in issue order.
//
// The per-receiver results are stored from inline assembly (stream_store*): fire-and-forget instructions the pass does
// not see.  This was the first suspect for the vmcnt(0) waits and turned out not to be the cause; it is kept because it
// is harmless -- the stores can only make the pass's waits longer than necessary (the hardware counter includes them),
// never shorter: "at most N operations outstanding" still means every load older than the N youngest has landed.
//
// tests/emu/ shadows this header with plain bounds-checked reads and plain stores.
#pragma once
#include <hip/hip_runtime.h>

namespace rapid {

typedef __amdgpu_buffer_rsrc_t stream_rsrc_t;
typedef unsigned int stream_u2 __attribute__((ext_vector_type(2)));

// `bytes` readable bytes at `base` (wave-uniform).
__device__ __forceinline__ stream_rsrc_t stream_make_rsrc(const void* base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}

// Two dwords at byte offset lane_off + imm of the stream (imm: wave-uniform, a compile-time constant at every call
// site); non-temporal (every byte of a stream is read once).
#ifndef RAPID_STREAM_AUX
#define RAPID_STREAM_AUX 2
#endif
template <int kAux = RAPID_STREAM_AUX>
__device__ __forceinline__ void stream_load2(stream_rsrc_t rsrc, unsigned int lane_off, unsigned int imm, unsigned int& a,
                                             unsigned int& b) {
    const stream_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)lane_off, (int)imm, kAux);
    a = v.x;
    b = v.y;
}

// Inclusive prefix sum over the wave's 64 lanes: seven adds whose second operand arrives over the data-parallel-primitive paths
// (shifts inside the rows of 16 lanes, then the row totals broadcast) -- no trip through the LDS crossbar, where the six
// ds_bpermute steps of a shuffle scan each cost a round trip the adds behind them wait for.
__device__ __forceinline__ int wave_inclusive_sum(int x) {
    int s = x;
    s += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);  // row_shr:1
    s += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);  // row_shr:2
    s += __builtin_amdgcn_update_dpp(0, x, 0x113, 0xf, 0xf, true);  // row_shr:3: s = x[i-3 .. i] within the row
    s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xe, true);  // row_shr:4 into lanes 4 .. 15 of every row
    s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xc, true);  // row_shr:8 into lanes 8 .. 15: every row holds its own prefix sums
    s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, true);  // row_bcast:15 into rows 1 and 3: the total of the row before
    s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, true);  // row_bcast:31 into rows 2 and 3: the total of the first two rows
    return s;
}

// Four dwords / one dword, same addressing (measurement variants of the boundary-record loads).
typedef unsigned int stream_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_load4(stream_rsrc_t rsrc, unsigned int lane_off, unsigned int imm, unsigned int& a, unsigned int& b,
                                             unsigned int& c, unsigned int& d) {
    const stream_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)imm, RAPID_STREAM_AUX);
    a = v.x;
    b = v.y;
    c = v.z;
    d = v.w;
}
__device__ __forceinline__ void stream_load1(stream_rsrc_t rsrc, unsigned int lane_off, unsigned int imm, unsigned int& a) {
    a = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane_off, (int)imm, RAPID_STREAM_AUX);
}

// A 64-bit word of a read-only table at a wave-uniform address, through the scalar cache (s_load, counted by lgkmcnt):
// a vector load here would put a long-lived destination register into the window loop, and the wait-count pass then
// waits for everything before each write of a register that MAY still be the target of that load.
__device__ __forceinline__ long long stream_scalar_load(const long long* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) long long*>(reinterpret_cast<unsigned long long>(p));
}

__device__ __forceinline__ unsigned int stream_scalar_load32(const unsigned int* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) unsigned int*>(reinterpret_cast<unsigned long long>(p));
}

// Stores the wait-count pass does not see (see above).  The address is per lane; inactive lanes store nothing.
__device__ __forceinline__ void stream_store(int* p, int v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void stream_store(unsigned long long* p, unsigned long long v) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// Waits for every memory operation of this wave, the assembly stores above included (the compiler's wait-count pass does not know of
// them, so a barrier's own s_waitcnt does not cover them): behind a workgroup barrier after this, the workgroup's results have left the CU.
__device__ __forceinline__ void stream_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Sets bits in a word of global memory (no value returned); rare, issued from assembly like the stores.
__device__ __forceinline__ void stream_flag_or(unsigned int* p, unsigned int bits) {
    asm volatile("global_atomic_or %0, %1, off" ::"v"(p), "v"(bits) : "memory");
}

// Everything computed from a window's registers must be FINISHED before those registers are requested again: if the
// scheduler lets a use of the old contents sink below the reload (it likes to issue loads early), the reload needs other
// destination registers, and the loop then closes with copies of a window that is still in flight -- i.e. with
// s_waitcnt vmcnt(0) in every turn.  This is a point no memory operation crosses, at which the three values that outlive a
// window (coverage flags, the carried word and its slot) exist in registers.
__device__ __forceinline__ void stream_settle(unsigned int& a, unsigned int& b, unsigned int& c) {
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c)::"memory");
}

}  // namespace rapid
