// TEST INFRASTRUCTURE ONLY: CPU model of rapid_amd/csrc/lds_dma.h for the SIMT emulator.  A copy is queued at issue
// (its destination is POISONED at once, as the hardware may overwrite it at any time from then on) and performed
// when a wait_dma<N>() leaves at most N younger copies outstanding -- the latest moment the hardware allows -- so a
// kernel that reads a slot before the covering wait, or recycles a slot that is still being read, fails the parity
// tests here instead of passing by luck.  Every lane is a fibre with its own queue (a lane copies its own 16 bytes).
#pragma once
#include <hip/hip_runtime.h>

#include <deque>

namespace rapid {

struct dma_rsrc_t {
    const unsigned char* base;
    unsigned int bytes;
};
typedef unsigned char* lds_addr_t;

inline dma_rsrc_t dma_make_rsrc(const void* base, unsigned int bytes) {
    return dma_rsrc_t{static_cast<const unsigned char*>(base), bytes};
}
inline lds_addr_t lds_uniform(lds_addr_t a) { return a; }
inline dma_rsrc_t dma_uniform(dma_rsrc_t r) { return r; }
inline lds_addr_t lds_address(const void* p) { return const_cast<unsigned char*>(static_cast<const unsigned char*>(p)); }

struct EmuDmaCopy {
    unsigned char* dst;
    const unsigned char* src;  // nullptr: out of range -> zeros
    int n = 16;
};
inline std::deque<EmuDmaCopy>& emu_dma_queue() {
    static std::vector<std::deque<EmuDmaCopy>> q(emu::MAXT);
    return q[emu::g_block->cur];
}
inline void lds_dma16(dma_rsrc_t rsrc, unsigned int lane_off, unsigned int soff, lds_addr_t lds_dst) {
    unsigned char* dst = lds_dst + 16 * emu::cur_lane();
    const unsigned long long off = (unsigned long long)lane_off + soff;
    const bool in = rsrc.base != nullptr && off + 16ull <= rsrc.bytes;
    std::memset(dst, 0xEE, 16);
    emu_dma_queue().push_back(EmuDmaCopy{dst, in ? rsrc.base + off : nullptr, 16});
}
template <int N>
inline void wait_dma() {
    std::deque<EmuDmaCopy>& q = emu_dma_queue();
    while ((int)q.size() > N) {
        const EmuDmaCopy c = q.front();
        q.pop_front();
        if (c.src != nullptr)
            std::memcpy(c.dst, c.src, c.n);
        else
            std::memset(c.dst, 0, c.n);
    }
}

}  // namespace rapid
