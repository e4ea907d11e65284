// Micro-benchmark (measurement aid, round 5): what does a per-record table lookup cost on gfx950, by where the table lives?
//   (a) scattered 4-byte reads from a table in memory (L2 / MALL / HBM resident by size), every lane its own address -- the
//       tally's kDictMemory lookup at 10^6 nodes (one dict_entry per record out of a 4 MB table);
//   (b) scattered 4-byte reads + 4-byte atomic ORs in LDS -- the direct-table lookup and the detector's ds_or;
//   (c) the remainder-bucket lookup of the hashed dictionary: two 16-bit reads (bucket bounds) + one 8-byte read + compare.
// Rates are per GPU: lookups per second with `waves` waves per CU and `ilp` independent lookups in flight per lane.
//   build: hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate ; run: ./gather_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                     \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__device__ __forceinline__ unsigned int mix32(unsigned int x) {
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    return x ^ (x >> 16);
}

// (a) ILP independent scattered loads per lane and step
template <int ILP>
__global__ __launch_bounds__(256) void mem_gather(const unsigned int* table, unsigned int mask, int steps, unsigned int* sink) {
    unsigned int x = mix32(blockIdx.x * 256u + threadIdx.x + 1u), acc = 0u;
    for (int s = 0; s < steps; ++s) {
        unsigned int v[ILP];
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            x = x * 1664525u + 1013904223u;
            v[i] = table[(x >> 8) & mask];
        }
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc ^= v[i];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// (b) LDS: `reads` scattered ds_read_b32 from a table of `tab_words` words + `ors` scattered ds_or_b32 into a private region of
// `state_words` words per wave, per lane and step
template <int READS, int ORS>
__global__ __launch_bounds__(1024) void lds_gather(int tab_words, int state_words, int steps, unsigned int* sink) {
    extern __shared__ unsigned int lds[];
    const int wave = threadIdx.x >> 6;
    unsigned int* const tab = lds;
    unsigned int* const st = lds + tab_words + wave * state_words;
    for (int i = threadIdx.x; i < tab_words + (int)(blockDim.x >> 6) * state_words; i += blockDim.x) lds[i] = mix32((unsigned)i);
    __syncthreads();
    unsigned int x = mix32(blockIdx.x * 1024u + threadIdx.x + 1u), acc = 0u;
    for (int s = 0; s < steps; ++s) {
        unsigned int v[READS > 0 ? READS : 1];
#pragma unroll
        for (int i = 0; i < READS; ++i) {
            x = x * 1664525u + 1013904223u;
            v[i] = tab[(x >> 8) % (unsigned)tab_words];
        }
#pragma unroll
        for (int i = 0; i < ORS; ++i) {
            x = x * 1664525u + 1013904223u;
            const unsigned int slot = READS > 0 ? (v[i % (READS > 0 ? READS : 1)] ^ x) : x;
            atomicOr(&st[(slot >> 8) % (unsigned)state_words], 1u << (x & 15u));
        }
#pragma unroll
        for (int i = 0; i < READS; ++i) acc ^= v[i];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// (c) remainder buckets: off[bucket] / off[bucket + 1] (16 bit) -> up to 8 one-byte remainders at rem[off] -> position of the match
__global__ __launch_bounds__(1024) void lds_bucket_lookup(int n_buckets, int n_keys, int steps, unsigned int* sink) {
    extern __shared__ unsigned int lds[];
    unsigned short* const off = reinterpret_cast<unsigned short*>(lds);
    unsigned char* const rem = reinterpret_cast<unsigned char*>(lds) + ((n_buckets + 1) * 2 + 15) / 16 * 16;
    for (int i = threadIdx.x; i <= n_buckets; i += blockDim.x) off[i] = (unsigned short)((long long)i * n_keys / n_buckets);
    for (int i = threadIdx.x; i < n_keys + 16; i += blockDim.x) rem[i] = (unsigned char)mix32((unsigned)i);
    __syncthreads();
    unsigned int x = mix32(blockIdx.x * 1024u + threadIdx.x + 1u), acc = 0u;
    for (int s = 0; s < steps; ++s) {
        x = x * 1664525u + 1013904223u;
        const unsigned int b = (x >> 8) % (unsigned)n_buckets, r = x & 255u;
        const unsigned int o0 = off[b], o1 = off[b + 1];
        // eight remainders starting at o0 (byte-aligned): two dwords each side of the 4-byte boundary
        const unsigned int* w = reinterpret_cast<const unsigned int*>(rem + (o0 & ~3u));
        const unsigned long long lo = w[0], mid = w[1], hi = w[2];
        const unsigned int sh = (o0 & 3u) * 8u;
        unsigned long long eight = ((lo | (mid << 32)) >> sh) | (sh ? (hi << (64u - sh)) : 0ull);
        const unsigned long long pat = 0x0101010101010101ull * r;
        unsigned long long z = eight ^ pat;  // a zero byte where the remainder matches
        z = (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull;
        const unsigned int n = o1 - o0;
        z &= n >= 8u ? ~0ull : ((1ull << (8u * n)) - 1ull);
        const unsigned int pos = z ? o0 + (unsigned)(__ffsll((long long)z) >> 3) : 0xFFFFu;
        acc ^= pos;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <class F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    launch();
    CHK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned int* sink;
    CHK(hipMalloc(&sink, 64));
    printf("device %s, %d CUs\n", prop.gcnArchName, cus);
    printf("(a) scattered 4-byte reads from memory: table size, waves per CU, loads in flight per lane -> lookups/s\n");
    const size_t sizes[] = {64u << 10, 1u << 20, 4u << 20, 16u << 20, 256u << 20};
    for (size_t bytes : sizes) {
        unsigned int* tab;
        CHK(hipMalloc(&tab, bytes));
        CHK(hipMemset(tab, 1, bytes));
        const unsigned int mask = (unsigned int)(bytes / 4 - 1);
        for (int wpc : {4, 8, 16}) {
            const int blocks = cus * wpc / 4, steps = 2048;
            const float m1 = time_ms([&] { hipLaunchKernelGGL(mem_gather<1>, dim3(blocks), dim3(256), 0, 0, tab, mask, steps, sink); }, 3);
            const float m4 = time_ms([&] { hipLaunchKernelGGL(mem_gather<4>, dim3(blocks), dim3(256), 0, 0, tab, mask, steps / 4, sink); }, 3);
            const float m8 = time_ms([&] { hipLaunchKernelGGL(mem_gather<8>, dim3(blocks), dim3(256), 0, 0, tab, mask, steps / 8, sink); }, 3);
            const double n = (double)blocks * 256 * steps;
            printf("  table %7zu KB  %2d waves/CU  ilp1 %8.3e  ilp4 %8.3e  ilp8 %8.3e lookups/s\n", bytes >> 10, wpc, n / m1 * 1e3, n / m4 * 1e3, n / m8 * 1e3);
        }
        CHK(hipFree(tab));
    }
    printf("(b) LDS: scattered reads from a shared table + scattered atomic ORs into a per-wave region, per CU workgroup of W waves\n");
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_gather<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_gather<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_gather<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_bucket_lookup), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    struct Cfg { int tab_words, state_words, waves; const char* what; };
    const Cfg cfgs[] = {{10001, 572, 15, "C3b: 40 KB table, 572-word state, 15 waves"},
                        {10001, 572, 8, "C3b shape, 8 waves"},
                        {4096, 5120, 6, "C5 (3 slots per word): 20 KB state, 6 waves"},
                        {4096, 7600, 4, "C5 (2 slots per word): 30 KB state, 4 waves"}};
    for (const Cfg& c : cfgs) {
        const size_t lds = (size_t)(c.tab_words + c.waves * c.state_words) * 4;
        const int steps = 4096;
        const float mr = time_ms([&] { hipLaunchKernelGGL((lds_gather<1, 0>), dim3(cus), dim3(c.waves * 64), lds, 0, c.tab_words, c.state_words, steps, sink); }, 3);
        const float mo = time_ms([&] { hipLaunchKernelGGL((lds_gather<0, 1>), dim3(cus), dim3(c.waves * 64), lds, 0, c.tab_words, c.state_words, steps, sink); }, 3);
        const float mb = time_ms([&] { hipLaunchKernelGGL((lds_gather<1, 1>), dim3(cus), dim3(c.waves * 64), lds, 0, c.tab_words, c.state_words, steps, sink); }, 3);
        const double n = (double)cus * c.waves * 64 * steps;
        printf("  %-48s read %8.3e  or %8.3e  read+or %8.3e records/s\n", c.what, n / mr * 1e3, n / mo * 1e3, n / mb * 1e3);
    }
    printf("(c) LDS remainder buckets (4,096 buckets, 15,000 keys: two 16-bit reads + three dwords + byte compare), W waves per CU\n");
    for (int waves : {4, 6, 8, 16}) {
        const int n_buckets = 4096, n_keys = 15000, steps = 4096;
        const size_t lds = ((n_buckets + 1) * 2 + 15) / 16 * 16 + n_keys + 64;
        const float m = time_ms([&] { hipLaunchKernelGGL(lds_bucket_lookup, dim3(cus), dim3(waves * 64), lds, 0, n_buckets, n_keys, steps, sink); }, 3);
        printf("  %2d waves/CU: %8.3e lookups/s\n", waves, (double)cus * waves * 64 * steps / m * 1e3);
    }
    return 0;
}
