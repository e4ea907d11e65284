// Measurement aid: where does `buffer_load_dwordx3 ... lds` put each lane's 12 bytes?  Source = dwords 0,1,2,...
// build: hipcc --offload-arch=gfx950 -O2 dma12_layout.hip -o dma12_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int rsrc4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned bytes, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 512; i += 64) reinterpret_cast<unsigned*>(smem)[i] = 0xDEADBEEFu;
    __syncthreads();
    unsigned long long b = (unsigned long long)src;
    rsrc4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xFFFF));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    unsigned lane_off = threadIdx.x * 12, soff = 0;
    unsigned ldsd = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    ldsd = __builtin_amdgcn_readfirstlane(ldsd);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 3\n\tbuffer_load_dwordx3 %0, %1, %2 offen lds\n\ts_waitcnt vmcnt(0)" :: "v"(lane_off), "s"(r), "s"(soff), "s"(ldsd) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = reinterpret_cast<unsigned*>(smem)[i];
}
int main() {
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = i;
    unsigned *d, *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, 4096u, o);
    std::vector<unsigned> r(512);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    for (int i = 0; i < 272; ++i) { if (r[i] == 0xDEADBEEFu) printf("   . "); else printf("%4u ", r[i]); if (i % 16 == 15) printf("\n"); }
    printf("\n");
    return 0;
}
