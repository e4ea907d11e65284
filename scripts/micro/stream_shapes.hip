// Micro-benchmark (measurement aid, not part of the product): what does the memory system of an MI355X deliver for the tally
// kernel's access pattern -- one wavefront per receiver stream, each stream a contiguous run of 8-byte records -- and which
// load SHAPE (bytes per lane, bytes per burst, windows in flight, waves per CU, cache policy) gets closest to a plain
// grid-stride read of the same bytes?  Streams are what C3b has: 9,492 streams of 9,870 records x 8 B.
//   build: hipcc --offload-arch=gfx950 -O3 stream_shapes.hip -o stream_shapes ; run: ./stream_shapes [n_streams] [records]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
template <int VEC, int AUX>
struct Ld;
template <int AUX>
struct Ld<2, AUX> {
    typedef u2 T;
    static __device__ __forceinline__ T ld(rsrc_t r, unsigned int voff, unsigned int soff) { return __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, AUX); }
    static __device__ __forceinline__ unsigned int fold(T v) { return v.x ^ v.y; }
    static __device__ __forceinline__ void eat(unsigned int& acc, T v) {  // 
        acc = (acc ^ v.x) * 0x9E3779B1u;  // a dependent chain the optimiser cannot re-associate across the reload
        acc = (acc ^ v.y) * 0x9E3779B1u;
    }
};
template <int AUX>
struct Ld<4, AUX> {
    typedef u4 T;
    static __device__ __forceinline__ T ld(rsrc_t r, unsigned int voff, unsigned int soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX); }
    static __device__ __forceinline__ unsigned int fold(T v) { return v.x ^ v.y ^ v.z ^ v.w; }
    static __device__ __forceinline__ void eat(unsigned int& acc, T v) {
        acc = (acc ^ v.x) * 0x9E3779B1u;
        acc = (acc ^ v.y) * 0x9E3779B1u;
        acc = (acc ^ v.z) * 0x9E3779B1u;
        acc = (acc ^ v.w) * 0x9E3779B1u;
    }
};

// One wave per stream.  A window = NI wave instructions of 64 x VEC x 4 contiguous bytes; D windows in flight behind the one
// being consumed (ring of D + 1, statically unrolled: the compiler's vmcnt waits stay exact).  WORK: dependent VALU
// instructions per loaded dword, to see how much processing the stream hides.
template <int VEC, int NI, int D, int AUX, int WORK>
__global__ __launch_bounds__(1024) void per_wave_kernel(const unsigned char* base, const long long* off, int n_streams, unsigned int* sink) {
    typedef Ld<VEC, AUX> L;
    constexpr unsigned int IB = 64u * VEC * 4u, WB = IB * NI;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int G = gridDim.x * wpb;
    unsigned int acc = 0;
    for (int s = wave * gridDim.x + blockIdx.x; s < n_streams; s += G) {  // consecutive streams go to different CUs, like the kernel
        const long long b0 = __builtin_amdgcn_readfirstlane((int)off[s]) | ((long long)__builtin_amdgcn_readfirstlane((int)(off[s] >> 32)) << 32);
        const long long b1 = __builtin_amdgcn_readfirstlane((int)off[s + 1]) | ((long long)__builtin_amdgcn_readfirstlane((int)(off[s + 1] >> 32)) << 32);
        const unsigned int bytes = (unsigned int)(b1 - b0);
        const rsrc_t r = make_rsrc(base + b0, bytes);
        const int nwin = (int)((bytes + WB - 1) / WB);
        typename L::T R[D + 1][NI];
        unsigned int voff = (unsigned int)lane * VEC * 4u;
#pragma unroll
        for (int p = 0; p <= D; ++p) {
#pragma unroll
            for (int i = 0; i < NI; ++i) R[p][i] = L::ld(r, voff, (unsigned int)i * IB);
            voff += WB;
            __builtin_amdgcn_sched_barrier(0);  // (issue order = consumption order: the waits at the loop head stay exact)
        }
        for (int w0 = 0; w0 < nwin; w0 += D + 1) {
#pragma unroll
            for (int p = 0; p <= D; ++p) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    if (WORK == 0) {
                        L::eat(acc, R[p][i]);
                    } else {
                        unsigned int v = L::fold(R[p][i]);
#pragma unroll
                        for (int k = 0; k < WORK; ++k) v = v * 1664525u + acc;
                        acc = (acc ^ v) * 0x9E3779B1u;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // (the window is dead before its registers are loaded again: no copies, exact waits)
#pragma unroll
                for (int i = 0; i < NI; ++i) R[p][i] = L::ld(r, voff, (unsigned int)i * IB);
                voff += WB;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// The same bytes as a plain grid-stride read: at any moment the whole machine reads one contiguous region.
template <int VEC, int NI, int D, int AUX>
__global__ __launch_bounds__(1024) void ideal_kernel(const unsigned char* base, unsigned long long total_bytes, unsigned int* sink) {
    typedef Ld<VEC, AUX> L;
    constexpr unsigned int IB = 64u * VEC * 4u, WB = IB * NI;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const unsigned int G = gridDim.x * wpb, gw = blockIdx.x * wpb + wave;
    const rsrc_t r = make_rsrc(base, (unsigned int)total_bytes);
    const unsigned int nchunk = (unsigned int)((total_bytes + WB - 1) / WB);
    unsigned int acc = 0;
    typename L::T R[D + 1][NI];
    unsigned int c = gw;
#pragma unroll
    for (int p = 0; p <= D; ++p) {
#pragma unroll
        for (int i = 0; i < NI; ++i) R[p][i] = L::ld(r, c * WB + (unsigned int)lane * VEC * 4u, (unsigned int)i * IB);
        c += G;
        __builtin_amdgcn_sched_barrier(0);  // (issue order = consumption order: the waits at the loop head stay exact)
    }
    for (unsigned int c0 = gw; c0 < nchunk; c0 += G * (D + 1)) {
#pragma unroll
        for (int p = 0; p <= D; ++p) {
#pragma unroll
            for (int i = 0; i < NI; ++i) L::eat(acc, R[p][i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NI; ++i) R[p][i] = L::ld(r, (c < nchunk ? c : nchunk) * WB + (unsigned int)lane * VEC * 4u, (unsigned int)i * IB);
            c += G;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// Streams interleaved at window granularity inside groups of `wpb` streams: window w of the group's streams is one contiguous
// run of wpb x WB bytes (what a resident layout written by the load pass could look like): wave m of workgroup g reads
// [(g * nwin + w) * wpb + m] * WB.
template <int VEC, int NI, int D, int AUX>
__global__ __launch_bounds__(1024) void grouped_kernel(const unsigned char* base, unsigned long long total_bytes, int n_groups, int nwin, unsigned int* sink) {
    typedef Ld<VEC, AUX> L;
    constexpr unsigned int IB = 64u * VEC * 4u, WB = IB * NI;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const rsrc_t r = make_rsrc(base, (unsigned int)total_bytes);
    unsigned int acc = 0;
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        typename L::T R[D + 1][NI];
        const unsigned int first = ((unsigned int)g * (unsigned int)nwin * (unsigned int)wpb + (unsigned int)wave) * WB + (unsigned int)lane * VEC * 4u;
        const unsigned int step = (unsigned int)wpb * WB;
        unsigned int voff = first;
#pragma unroll
        for (int p = 0; p <= D; ++p) {
#pragma unroll
            for (int i = 0; i < NI; ++i) R[p][i] = L::ld(r, voff, (unsigned int)i * IB);
            voff += step;
            __builtin_amdgcn_sched_barrier(0);  // (issue order = consumption order: the waits at the loop head stay exact)
        }
        const unsigned int end = first + (unsigned int)nwin * step;
        for (int w0 = 0; w0 < nwin; w0 += D + 1) {
#pragma unroll
            for (int p = 0; p <= D; ++p) {
#pragma unroll
                for (int i = 0; i < NI; ++i) L::eat(acc, R[p][i]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NI; ++i) R[p][i] = L::ld(r, voff < end ? voff : 0xFFFFFF00u, (unsigned int)i * IB);
                voff += step;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static unsigned char* d_data;
static long long* d_off;
static unsigned int* d_sink;
static unsigned long long total_bytes;
static int n_streams, n_rec;

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms / reps;
}

template <int VEC, int NI, int D, int AUX, int WORK = 0>
static void run_per_wave(int waves, int blocks_per_cu = 1) {
    const double ms = time_ms([&] { hipLaunchKernelGGL((per_wave_kernel<VEC, NI, D, AUX, WORK>), dim3(256 * blocks_per_cu), dim3(waves * 64), 0, 0, d_data, d_off, n_streams, d_sink); });
    printf("per-wave   %2d B/lane x %2d instr = %5d B window, %d in flight, aux %d, work %2d, %2d waves x %d wg/CU : %.4f ms  %6.0f GB/s\n", VEC * 4, NI,
           VEC * 4 * 64 * NI, D, AUX, WORK, waves, blocks_per_cu, ms, total_bytes / ms / 1e6);
    fflush(stdout);
}
template <int VEC, int NI, int D, int AUX>
static void run_ideal(int waves, int blocks_per_cu = 1) {
    const double ms = time_ms([&] { hipLaunchKernelGGL((ideal_kernel<VEC, NI, D, AUX>), dim3(256 * blocks_per_cu), dim3(waves * 64), 0, 0, d_data, total_bytes, d_sink); });
    printf("grid-stride %2d B/lane x %2d instr = %5d B chunk, %d in flight, aux %d, %2d waves x %d wg/CU : %.4f ms  %6.0f GB/s\n", VEC * 4, NI, VEC * 4 * 64 * NI, D, AUX,
           waves, blocks_per_cu, ms, total_bytes / ms / 1e6);
    fflush(stdout);
}
template <int VEC, int NI, int D, int AUX>
static void run_grouped(int waves) {
    const unsigned int WB = 64u * VEC * 4u * NI;
    const int nwin = (int)(((unsigned long long)n_rec * 8 + WB - 1) / WB);
    const int n_groups = (int)std::min<unsigned long long>((unsigned long long)(n_streams + waves - 1) / waves, total_bytes / ((unsigned long long)nwin * waves * WB));
    const double ms = time_ms([&] { hipLaunchKernelGGL((grouped_kernel<VEC, NI, D, AUX>), dim3(256), dim3(waves * 64), 0, 0, d_data, total_bytes, n_groups, nwin, d_sink); });
    const double bytes = (double)n_groups * nwin * waves * WB;
    printf("grouped    %2d B/lane x %2d instr = %5d B window, %d in flight, aux %d, %2d waves/CU (%d groups) : %.4f ms  %6.0f GB/s\n", VEC * 4, NI, (int)WB, D, AUX, waves, n_groups, ms,
           bytes / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    n_streams = argc > 1 ? atoi(argv[1]) : 9492;
    n_rec = argc > 2 ? atoi(argv[2]) : 9870;
    total_bytes = (unsigned long long)n_streams * n_rec * 8ull;
    hipMalloc(&d_data, total_bytes + 65536);
    hipMemset(d_data, 1, total_bytes + 65536);
    std::vector<long long> off(n_streams + 1);
    for (int i = 0; i <= n_streams; ++i) off[i] = (long long)i * n_rec * 8;
    hipMalloc(&d_off, off.size() * 8);
    hipMemcpy(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&d_sink, 64);
    printf("%d streams x %d records x 8 B = %.1f MB\n", n_streams, n_rec, total_bytes / 1e6);

    // the ceiling: plain grid-stride reads of the same bytes
    run_ideal<4, 2, 1, 0>(16);
    run_ideal<4, 2, 3, 0>(16);
    run_ideal<4, 2, 3, 2>(16);
    run_ideal<4, 4, 1, 2>(16);
    run_ideal<2, 4, 1, 2>(16);
    run_ideal<2, 4, 3, 2>(16);
    run_ideal<4, 2, 3, 2>(8, 2);
    // today's shape: 8 B/lane, 4 instructions = 2 KiB window, one in flight, nt
    for (int waves : {8, 12, 15, 16}) run_per_wave<2, 4, 1, 2>(waves);
    run_per_wave<2, 4, 1, 0>(15);
    run_per_wave<2, 4, 1, 1>(15);
    run_per_wave<2, 4, 1, 3>(15);
    // deeper
    for (int waves : {8, 12, 15, 16}) run_per_wave<2, 4, 2, 2>(waves);
    for (int waves : {8, 12, 15, 16}) run_per_wave<2, 4, 3, 2>(waves);
    // bigger windows (bigger bursts)
    for (int waves : {8, 15}) run_per_wave<2, 8, 1, 2>(waves);
    for (int waves : {8, 15}) run_per_wave<2, 8, 2, 2>(waves);
    for (int waves : {8, 15}) run_per_wave<2, 16, 1, 2>(waves);
    // 16 B per lane
    for (int waves : {8, 12, 15, 16}) run_per_wave<4, 2, 1, 2>(waves);
    for (int waves : {8, 12, 15, 16}) run_per_wave<4, 2, 2, 2>(waves);
    for (int waves : {8, 15, 16}) run_per_wave<4, 2, 3, 2>(waves);
    for (int waves : {8, 15, 16}) run_per_wave<4, 4, 1, 2>(waves);
    for (int waves : {8, 15, 16}) run_per_wave<4, 4, 2, 2>(waves);
    for (int waves : {8, 15}) run_per_wave<4, 8, 1, 2>(waves);
    run_per_wave<4, 2, 2, 0>(15);
    run_per_wave<4, 4, 1, 0>(15);
    // more waves per CU than the tally kernel's LDS allows (what would occupancy buy?)
    run_per_wave<2, 4, 1, 2>(16, 2);
    run_per_wave<4, 2, 2, 2>(16, 2);
    run_per_wave<2, 4, 1, 2>(8, 2);
    run_per_wave<4, 2, 2, 2>(8, 2);
    // with work per loaded dword
    for (int work : {0}) (void)work;
    run_per_wave<2, 4, 1, 2, 4>(15);
    run_per_wave<2, 4, 1, 2, 8>(15);
    run_per_wave<2, 4, 1, 2, 16>(15);
    run_per_wave<2, 4, 2, 2, 8>(15);
    run_per_wave<2, 4, 2, 2, 16>(15);
    run_per_wave<4, 2, 2, 2, 8>(15);
    run_per_wave<4, 2, 2, 2, 16>(15);
    // streams interleaved per window inside groups of waves (another resident layout)
    for (int waves : {8, 15, 16}) run_grouped<2, 4, 1, 2>(waves);
    for (int waves : {8, 15, 16}) run_grouped<2, 4, 2, 2>(waves);
    for (int waves : {15, 16}) run_grouped<4, 2, 2, 2>(waves);
    for (int waves : {15, 16}) run_grouped<4, 2, 2, 0>(waves);
    return 0;
}
