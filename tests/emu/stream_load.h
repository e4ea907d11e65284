// TEST INFRASTRUCTURE ONLY: CPU model of rapid_amd/csrc/stream_load.h for the SIMT emulator -- a bounds-checked read
// (out of range: zeros, as the buffer resource of the device gives) and plain stores.
#pragma once
#include <hip/hip_runtime.h>

namespace rapid {

struct stream_rsrc_t {
    const unsigned char* base;
    unsigned int bytes;
};

inline stream_rsrc_t stream_make_rsrc(const void* base, unsigned int bytes) {
    return stream_rsrc_t{static_cast<const unsigned char*>(base), bytes};
}

template <int kAux = 2>
inline void stream_load2(stream_rsrc_t rsrc, unsigned int lane_off, unsigned int imm, unsigned int& a, unsigned int& b) {
    const unsigned long long off = (unsigned long long)lane_off + (unsigned long long)imm;
    a = 0u;
    b = 0u;
    // the hardware checks every dword on its own
    if (rsrc.base != nullptr && off + 4ull <= rsrc.bytes) std::memcpy(&a, rsrc.base + off, 4);
    if (rsrc.base != nullptr && off + 8ull <= rsrc.bytes) std::memcpy(&b, rsrc.base + off + 4, 4);
}

inline void stream_load1(stream_rsrc_t rsrc, unsigned int lane_off, unsigned int imm, unsigned int& a) {
    const unsigned long long off = (unsigned long long)lane_off + (unsigned long long)imm;
    a = 0u;
    if (rsrc.base != nullptr && off + 4ull <= rsrc.bytes) std::memcpy(&a, rsrc.base + off, 4);
}
inline void stream_load4(stream_rsrc_t rsrc, unsigned int lane_off, unsigned int imm, unsigned int& a, unsigned int& b, unsigned int& c,
                         unsigned int& d) {
    stream_load1(rsrc, lane_off, imm, a);
    stream_load1(rsrc, lane_off, imm + 4u, b);
    stream_load1(rsrc, lane_off, imm + 8u, c);
    stream_load1(rsrc, lane_off, imm + 12u, d);
}

// One data-parallel-primitive move as the hardware defines it (the controls wave_inclusive_sum uses: row_shr:n, row_bcast:15,
// row_bcast:31): a disabled lane (its row / bank not in the masks) keeps `old`, a lane whose source lies outside gets 0 (bound_ctrl).
inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask) {
    const int lane = (int)(threadIdx.x & 63u), row = lane >> 4, in_row = lane & 15, bank = in_row >> 2;
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11f) from = in_row >= (ctrl & 15) ? lane - (ctrl & 15) : -1;
    else if (ctrl == 0x142) from = row >= 1 ? row * 16 - 1 : -1;
    else if (ctrl == 0x143) from = row >= 2 ? 31 : -1;
    const int v = __shfl(src, from < 0 ? lane : from, 64);  // (every lane takes part in the exchange)
    const bool enabled = ((row_mask >> row) & 1) != 0 && ((bank_mask >> bank) & 1) != 0;
    return !enabled ? old : from < 0 ? 0 : v;
}
inline int wave_inclusive_sum(int x) {  // the same seven steps as csrc/stream_load.h
    int s = x;
    s += emu_update_dpp(0, x, 0x111, 0xf, 0xf);
    s += emu_update_dpp(0, x, 0x112, 0xf, 0xf);
    s += emu_update_dpp(0, x, 0x113, 0xf, 0xf);
    s += emu_update_dpp(0, s, 0x114, 0xf, 0xe);
    s += emu_update_dpp(0, s, 0x118, 0xf, 0xc);
    s += emu_update_dpp(0, s, 0x142, 0xa, 0xf);
    s += emu_update_dpp(0, s, 0x143, 0xc, 0xf);
    return s;
}

inline long long stream_scalar_load(const long long* p) { return *p; }
inline unsigned int stream_scalar_load32(const unsigned int* p) { return *p; }
inline void stream_store(int* p, int v) { *p = v; }
inline void stream_store(unsigned long long* p, unsigned long long v) { *p = v; }

inline void stream_drain() {}
inline void stream_flag_or(unsigned int* p, unsigned int bits) { *p |= bits; }
inline void stream_settle(unsigned int&, unsigned int&, unsigned int&) {}  // a scheduling fence on the device, nothing here

}  // namespace rapid
