// The tally kernel's record stream: bounds-checked buffer loads straight into registers.
//
// A receiver's delivered stream is a raw buffer (base = its first record, size = 20 B x its record count); lane l of
// "quarter" q of a window reads the dwords it needs of record 64 q + l with ONE wave instruction per quarter
// (buffer_load_dwordx2 ... offen nt, lane stride 20 B: the 64 lanes cover 1,280 contiguous bytes, so every cache line
// the instruction touches is used completely by it and its neighbours).  Reads past the end of the stream return
// zeros without touching memory -- a zero record names no ring, closes no batch and is not a DOWN report, so the
// tail of the last window needs no special case.  The loads are ordinary compiler-visible loads: the compiler's own
// s_waitcnt bookkeeping lets a whole window stay in flight while the previous one is tallied.
//
// tests/emu/ shadows this header with plain bounds-checked reads.
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
__device__ __forceinline__ void stream_load2(stream_rsrc_t rsrc, unsigned int lane_off, unsigned int imm, unsigned int& a,
                                             unsigned int& b) {
    const stream_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)lane_off, (int)imm, 2);
    a = v.x;
    b = v.y;
}

}  // namespace rapid
