// Micro-benchmark (measurement aid, not part of the product): what does the memory system deliver for ONE PASS over 20-byte
// boundary records, one wavefront per receiver stream -- by load shape?  A window = 256 records = 5,120 contiguous bytes.
//   shape 0: per quarter of 64 records, lane l loads {dwords 0,1} and {dwords 3,4} of record l: two dwordx2 at a lane stride of 20 B
//   shape 1: dwordx4 {0..3} + dword {4} at a lane stride of 20 B
//   shape 2: only {dwords 3,4} (no configuration id): one dwordx2 at a lane stride of 20 B
//   shape 3: five dwordx4 at a lane stride of 16 B (every cache line requested once, by one instruction; records arrive transposed)
//   shape 4: like 3, grid-stride over the whole array (the ceiling: no streams)
// AUXA / AUXB: cache-policy bits of the first / second load of a quarter (0 = default, 2 = nt).
//   build: hipcc --offload-arch=gfx950 -O3 boundary_shapes.hip -o boundary_shapes ; run: ./boundary_shapes [n_streams] [records]
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
__device__ __forceinline__ void eat(unsigned int& acc, unsigned int v) { acc = (acc ^ v) * 0x9E3779B1u; }

template <int SHAPE>
struct Win;
template <>
struct Win<0> { u2 a[4], b[4]; };
template <>
struct Win<1> { u4 a[4]; unsigned int b[4]; };
template <>
struct Win<2> { u2 b[4]; };
template <>
struct Win<3> { u4 a[5]; };

template <int SHAPE, int AUXA, int AUXB>
__device__ __forceinline__ void load(rsrc_t r, unsigned int wbase, int lane, Win<SHAPE>& w) {
    if constexpr (SHAPE == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w.a[q] = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(wbase + lane * 20u), q * 1280, AUXA);
            w.b[q] = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(wbase + lane * 20u), q * 1280 + 12, AUXB);
        }
    } else if constexpr (SHAPE == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w.a[q] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(wbase + lane * 20u), q * 1280, AUXA);
            w.b[q] = __builtin_amdgcn_raw_buffer_load_b32(r, (int)(wbase + lane * 20u), q * 1280 + 16, AUXB);
        }
    } else if constexpr (SHAPE == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) w.b[q] = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(wbase + lane * 20u), q * 1280 + 12, AUXB);
    } else {
#pragma unroll
        for (int i = 0; i < 5; ++i) w.a[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(wbase + lane * 16u), i * 1024, AUXA);
    }
}
template <int SHAPE>
__device__ __forceinline__ void consume(unsigned int& acc, const Win<SHAPE>& w) {
    if constexpr (SHAPE == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { eat(acc, w.a[q].x); eat(acc, w.a[q].y); eat(acc, w.b[q].x); eat(acc, w.b[q].y); }
    } else if constexpr (SHAPE == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { eat(acc, w.a[q].x); eat(acc, w.a[q].y); eat(acc, w.a[q].w); eat(acc, w.b[q]); }
    } else if constexpr (SHAPE == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { eat(acc, w.b[q].x); eat(acc, w.b[q].y); }
    } else {
#pragma unroll
        for (int i = 0; i < 5; ++i) { eat(acc, w.a[i].x); eat(acc, w.a[i].y); eat(acc, w.a[i].z); eat(acc, w.a[i].w); }
    }
}

// one wave per stream, D windows in flight behind the one being consumed
template <int SHAPE, int D, int AUXA, int AUXB>
__global__ __launch_bounds__(1024) void per_wave_kernel(const unsigned char* base, int n_streams, int n_rec, unsigned int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int G = gridDim.x * wpb;
    unsigned int acc = 0;
    const unsigned int bytes = (unsigned int)n_rec * 20u;
    const int nwin = (n_rec + 255) / 256;
    for (int s = wave * gridDim.x + blockIdx.x; s < n_streams; s += G) {
        const rsrc_t r = make_rsrc(base + (unsigned long long)s * bytes, bytes);
        Win<SHAPE> R[D + 1];
        unsigned int wb = 0;
#pragma unroll
        for (int p = 0; p <= D; ++p) {
            load<SHAPE, AUXA, AUXB>(r, wb, lane, R[p]);
            wb += 5120u;
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int w0 = 0; w0 < nwin; w0 += D + 1) {
#pragma unroll
            for (int p = 0; p <= D; ++p) {
                consume<SHAPE>(acc, R[p]);
                __builtin_amdgcn_sched_barrier(0);
                load<SHAPE, AUXA, AUXB>(r, wb, lane, R[p]);
                wb += 5120u;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// the ceiling: the same bytes as a plain grid-stride read, five dwordx4 per 5 KiB chunk (buffer offsets are 32 bit: per-chunk base)
template <int D, int AUX>
__global__ __launch_bounds__(1024) void ideal_kernel(const unsigned char* base, unsigned long long total_bytes, unsigned int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const unsigned long long G = (unsigned long long)gridDim.x * wpb, gw = (unsigned long long)blockIdx.x * wpb + wave;
    const unsigned long long nchunk = total_bytes / 5120ull;
    unsigned int acc = 0;
    for (unsigned long long c = gw; c < nchunk; c += G) {
        const u4* p = reinterpret_cast<const u4*>(base + c * 5120ull) + lane;
        u4 v[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) v[i] = __builtin_nontemporal_load(p + 64 * i);
#pragma unroll
        for (int i = 0; i < 5; ++i) { eat(acc, v[i].x); eat(acc, v[i].y); eat(acc, v[i].z); eat(acc, v[i].w); }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static unsigned char* d_data;
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
    return ms / reps;
}
template <int SHAPE, int D, int AUXA, int AUXB>
static void run(int waves) {
    const double ms = time_ms([&] { hipLaunchKernelGGL((per_wave_kernel<SHAPE, D, AUXA, AUXB>), dim3(256), dim3(waves * 64), 0, 0, d_data, n_streams, n_rec, d_sink); });
    printf("shape %d, %d windows in flight behind, aux %d/%d, %2d waves/CU : %.4f ms  %6.0f GB/s of the 20 B per record\n", SHAPE, D, AUXA, AUXB, waves, ms,
           total_bytes / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    n_streams = argc > 1 ? atoi(argv[1]) : 9492;
    n_rec = argc > 2 ? atoi(argv[2]) : 9870;
    total_bytes = (unsigned long long)n_streams * n_rec * 20ull;
    hipMalloc(&d_data, total_bytes + 65536);
    hipMemset(d_data, 1, total_bytes + 65536);
    hipMalloc(&d_sink, 64);
    printf("%d streams x %d records x 20 B = %.1f MB\n", n_streams, n_rec, total_bytes / 1e6);
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL((ideal_kernel<1, 2>), dim3(256), dim3(1024), 0, 0, d_data, total_bytes, d_sink); });
        printf("grid-stride 5 x dwordx4 nt, 16 waves/CU : %.4f ms  %6.0f GB/s\n", ms, total_bytes / ms / 1e6);
    }
    for (int waves : {12, 15}) {
        run<0, 1, 2, 2>(waves);
        run<0, 1, 0, 0>(waves);
        run<0, 1, 0, 2>(waves);
        run<0, 2, 0, 0>(waves);
        run<1, 1, 2, 2>(waves);
        run<1, 1, 0, 0>(waves);
        run<2, 1, 2, 2>(waves);
        run<2, 1, 0, 0>(waves);
        run<3, 1, 2, 2>(waves);
        run<3, 1, 0, 0>(waves);
        run<3, 2, 2, 2>(waves);
    }
    run<0, 1, 0, 0>(8);
    run<3, 1, 2, 2>(8);
    run<3, 1, 2, 2>(16);
    return 0;
}
