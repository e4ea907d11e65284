// Micro-benchmark (measurement aid): how much does instruction issue cost a streaming kernel?  One wave per stream of 20-byte
// records (two dwordx2 per record, two windows of 256 records in flight, 15 waves per CU -- the tally kernel's load shape), plus
// NALU vector instructions and NSALU scalar instructions of dummy work per window (four independent chains).
//   build: hipcc --offload-arch=gfx950 -O3 issue_load.hip -o issue_load ; run: ./issue_load [n_streams] [records]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
struct Win { u2 a[4], b[4]; };
__device__ __forceinline__ void load(rsrc_t r, unsigned int wbase, int lane, Win& w) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        w.a[q] = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(wbase + lane * 20u), q * 1280, 0);
        w.b[q] = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(wbase + lane * 20u), q * 1280 + 12, 0);
    }
}

template <int NALU, int NSALU>
__global__ __launch_bounds__(1024) void stream_kernel(const unsigned char* base, int n_streams, int n_rec, unsigned int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int G = gridDim.x * wpb;
    unsigned int acc[4] = {1u, 2u, 3u, 4u};
    unsigned int sacc[4] = {5u, 6u, 7u, 8u};
    const unsigned int bytes = (unsigned int)n_rec * 20u;
    const int nwin = (n_rec + 255) / 256;
    for (int s = wave * gridDim.x + blockIdx.x; s < n_streams; s += G) {
        const rsrc_t r = make_rsrc(base + (unsigned long long)s * bytes, bytes);
        Win R[2];
        unsigned int wb = 0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            load(r, wb, lane, R[p]);
            wb += 5120u;
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int w0 = 0; w0 < nwin; w0 += 2) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] ^= R[p].a[q].x ^ R[p].a[q].y ^ R[p].b[q].x ^ R[p].b[q].y;
#pragma unroll
                for (int k = 0; k < NALU / 8; ++k) {  // two vector instructions per chain and round (mul + xor)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = (acc[c] * 0x9E3779B1u) ^ (unsigned int)(k + c);
                }
                if constexpr (NSALU > 0) {
                    unsigned int u[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) u[c] = __builtin_amdgcn_readfirstlane(sacc[c]);
#pragma unroll
                    for (int k = 0; k < NSALU / 8; ++k) {  // two scalar instructions per chain and round
#pragma unroll
                        for (int c = 0; c < 4; ++c) u[c] = (u[c] * 0x85EBCA6Bu) ^ (unsigned int)(k + c);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) sacc[c] = u[c];
                }
                __builtin_amdgcn_sched_barrier(0);
                load(r, wb, lane, R[p]);
                wb += 5120u;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ sacc[0] ^ sacc[1] ^ sacc[2] ^ sacc[3]) == 0x12345678u) sink[0] = acc[0];
}

static unsigned char* d_data;
static unsigned int* d_sink;
static unsigned long long total_bytes;
static int n_streams, n_rec;

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    const int reps = 5;
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
template <int NALU, int NSALU>
static void run(int waves) {
    const double ms = time_ms([&] { hipLaunchKernelGGL((stream_kernel<NALU, NSALU>), dim3(256), dim3(waves * 64), 0, 0, d_data, n_streams, n_rec, d_sink); });
    printf("%2d waves/CU, %4d vector + %4d scalar dummy instructions per window : %.4f ms  %6.0f GB/s\n", waves, NALU, NSALU, ms, total_bytes / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    n_streams = argc > 1 ? atoi(argv[1]) : 9492;
    n_rec = argc > 2 ? atoi(argv[2]) : 9870;
    total_bytes = (unsigned long long)n_streams * n_rec * 20ull;
    (void)hipMalloc(&d_data, total_bytes + 65536);
    (void)hipMemset(d_data, 1, total_bytes + 65536);
    (void)hipMalloc(&d_sink, 64);
    printf("%d streams x %d records x 20 B = %.1f MB\n", n_streams, n_rec, total_bytes / 1e6);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>(15);
        run<104, 0>(15);
        run<200, 0>(15);
        run<400, 0>(15);
        run<800, 0>(15);
        run<0, 104>(15);
        run<0, 200>(15);
        run<0, 400>(15);
        run<200, 200>(15);
        run<400, 400>(15);
        run<400, 400>(12);
    }
    return 0;
}
