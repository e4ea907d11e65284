// Micro-benchmark (measurement aid, not part of the product): how do SALU and VALU instructions of the co-resident
// waves of one SIMD share its issue slots on gfx950?  Each wave runs ITER x (NS scalar + NV vector) independent
// instructions; the host varies waves per SIMD.  build: hipcc --offload-arch=gfx950 -O2 issue_mix.hip -o issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void mix_kernel(int iters, unsigned long long* out) {
    unsigned int s0 = blockIdx.x, v0 = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)  // 64 SALU
            asm volatile(".rept 64\n s_add_u32 %0, %0, 1\n .endr" : "+s"(s0) : : "scc");
        if (MODE == 1)  // 64 VALU
            asm volatile(".rept 64\n v_add_u32 %0, %0, 1\n .endr" : "+v"(v0));
        if (MODE == 2)  // 64 SALU + 64 VALU interleaved 1:1
            asm volatile(".rept 64\n s_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n .endr" : "+s"(s0), "+v"(v0) : : "scc");
        if (MODE == 3)  // 128 SALU + 64 VALU interleaved 2:1
            asm volatile(".rept 64\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n .endr" : "+s"(s0), "+v"(v0) : : "scc");
        if (MODE == 5)  // 32 x (s_cmp; TAKEN s_cbranch over one s_nop): cost of a taken branch
            asm volatile(".rept 32\n s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1\n s_nop 0\n .endr" : "+s"(s0) : : "scc");
        if (MODE == 6)  // 32 x (s_cmp; NOT-taken s_cbranch; s_nop)
            asm volatile(".rept 32\n s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 1\n s_nop 0\n .endr" : "+s"(s0) : : "scc");
        if (MODE == 4)  // 64 x (v_cmp -> s_and on its result -> s_bcnt1): VALU->SALU dependency chain
            asm volatile(".rept 64\n v_cmp_ne_u32 vcc, 0, %1\n s_and_b64 vcc, vcc, exec\n s_bcnt1_i32_b64 %0, vcc\n .endr" : "+s"(s0), "+v"(v0) : : "vcc", "scc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) + ((unsigned long long)(s0 + v0) & 0);
}

template <int MODE>
static void run(const char* name, int instr_per_iter) {
    unsigned long long* d;
    hipMalloc(&d, 256 * 16 * 8);
    const int iters = 2000;
    for (int waves : {1, 2, 4, 8, 12, 16}) {  // waves per CU (block = waves x 64; one block per CU)
        hipLaunchKernelGGL(mix_kernel<MODE>, dim3(256), dim3(waves * 64), 0, 0, iters, d);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * waves);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto x : h) avg += (double)x;
        avg /= h.size();
        const double per_simd = (double)waves / 4.0;
        printf("%-28s waves/CU %2d  cycles/instr/wave %6.2f   SIMD cycles per instr %5.2f\n", name, waves,
               avg / ((double)iters * instr_per_iter), avg / ((double)iters * instr_per_iter) / (per_simd < 1 ? 1 : per_simd));
        fflush(stdout);
    }
    hipFree(d);
}

int main() {
    run<0>("64 SALU", 64);
    run<1>("64 VALU", 64);
    run<2>("64 SALU + 64 VALU (1:1)", 128);
    run<3>("128 SALU + 64 VALU (2:1)", 192);
    run<4>("v_cmp -> s_and -> s_bcnt1", 192);
    run<5>("32 x taken branch (+cmp)", 32);
    run<6>("32 x not-taken branch (+cmp,nop)", 32);
    return 0;
}
