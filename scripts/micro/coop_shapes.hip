// Micro-benchmark (measurement aid, not part of the product): does the memory system deliver more for 20-byte boundary records
// when a WORKGROUP takes one receiver's stream at a time (all of it in flight at once, its windows dealt to the waves) than when
// every wave follows a stream of its own (boundary_shapes.hip)?  A window = 256 records = 5,120 contiguous bytes.
//   coop, shape 0: lane l of quarter q loads {dwords 0,1} and {dwords 3,4} of its record: two dwordx2 at a lane stride of 20 B
//   coop, shape 3: five dwordx4 at a lane stride of 16 B per window (every cache line requested once; records arrive transposed)
//   DB = 1: the next stream's windows are requested before the current one's are consumed (two register sets)
//   deal 0: workgroup b takes streams b, b + G, ...; deal 1: contiguous runs of streams per workgroup
//   per-wave kernels of boundary_shapes.hip for reference, with the streams of a workgroup's waves adjacent (deal 1) or G apart (0)
//   build: hipcc --offload-arch=gfx950 -O3 coop_shapes.hip -o coop_shapes ; run: ./coop_shapes [n_streams] [records]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

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
struct Win<3> { u4 a[5]; };

template <int SHAPE, int AUX>
__device__ __forceinline__ void load(rsrc_t r, unsigned int wbase, int lane, Win<SHAPE>& w) {
    if constexpr (SHAPE == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w.a[q] = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(wbase + lane * 20u), q * 1280, AUX);
            w.b[q] = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(wbase + lane * 20u), q * 1280 + 12, AUX);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 5; ++i) w.a[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(wbase + lane * 16u), i * 1024, AUX);
    }
}
template <int SHAPE>
__device__ __forceinline__ void consume(unsigned int& acc, const Win<SHAPE>& w) {
    if constexpr (SHAPE == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { eat(acc, w.a[q].x); eat(acc, w.a[q].y); eat(acc, w.b[q].x); eat(acc, w.b[q].y); }
    } else {
#pragma unroll
        for (int i = 0; i < 5; ++i) { eat(acc, w.a[i].x); eat(acc, w.a[i].y); eat(acc, w.a[i].z); eat(acc, w.a[i].w); }
    }
}

// one workgroup per stream at a time; wave w takes windows w, w + W, w + 2 W (W = waves per workgroup; <= 3 W windows per stream)
template <int SHAPE, int AUX, int DB, int SYNC>
__global__ __launch_bounds__(1024) void coop_kernel(const unsigned char* base, int n_streams, int n_rec, int deal, unsigned int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int G = gridDim.x;
    unsigned int acc = 0;
    const unsigned int bytes = (unsigned int)n_rec * 20u;
    const int per = (n_streams + G - 1) / G;
    const int s_begin = deal ? blockIdx.x * per : blockIdx.x, s_step = deal ? 1 : G;
    const int s_end = deal ? min(n_streams, s_begin + per) : n_streams;
    Win<SHAPE> R[DB + 1][3];
    auto request = [&](int s, Win<SHAPE>(&dst)[3]) {
        const rsrc_t r = make_rsrc(base + (unsigned long long)s * bytes, bytes);
#pragma unroll
        for (int j = 0; j < 3; ++j) load<SHAPE, AUX>(r, (unsigned int)(wave + j * W) * 5120u, lane, dst[j]);  // (past the end: zeros, no traffic)
    };
    if (s_begin < s_end) request(s_begin, R[0]);
    for (int s = s_begin; s < s_end; s += s_step * (DB + 1)) {
#pragma unroll
        for (int p = 0; p <= DB; ++p) {
            const int cur = s + p * s_step;
            if (cur >= s_end) break;
            if constexpr (DB == 1) {
                if (cur + s_step < s_end) request(cur + s_step, R[p ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) consume<SHAPE>(acc, R[p][j]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SYNC != 0) __syncthreads();  // (what a tally of the receiver by the whole workgroup would need at least once)
            if constexpr (DB == 0) {
                if (cur + s_step < s_end) request(cur + s_step, R[0]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// one wave per stream, 1 window in flight behind the one being consumed (boundary_shapes.hip), two deals
template <int SHAPE, int AUX>
__global__ __launch_bounds__(1024) void per_wave_kernel(const unsigned char* base, int n_streams, int n_rec, int deal, unsigned int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int G = gridDim.x * wpb;
    unsigned int acc = 0;
    const unsigned int bytes = (unsigned int)n_rec * 20u;
    const int nwin = (n_rec + 255) / 256;
    const int per = (n_streams + G - 1) / G;
    const int gw = blockIdx.x * wpb + wave;
    const int s_begin = deal == 1 ? gw * per : deal == 2 ? gw : wave * gridDim.x + blockIdx.x;
    const int s_step = deal == 1 ? 1 : G;
    const int s_end = deal == 1 ? min(n_streams, s_begin + per) : n_streams;
    for (int s = s_begin; s < s_end; s += s_step) {
        const rsrc_t r = make_rsrc(base + (unsigned long long)s * bytes, bytes);
        Win<SHAPE> R[2];
        unsigned int wb = 0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            load<SHAPE, AUX>(r, wb, lane, R[p]);
            wb += 5120u;
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int w0 = 0; w0 < nwin; w0 += 2) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                consume<SHAPE>(acc, R[p]);
                __builtin_amdgcn_sched_barrier(0);
                load<SHAPE, AUX>(r, wb, lane, R[p]);
                wb += 5120u;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
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
template <int SHAPE, int AUX, int DB, int SYNC>
static void run_coop(int waves, int deal, int grid = 256) {
    const double ms = time_ms([&] { hipLaunchKernelGGL((coop_kernel<SHAPE, AUX, DB, SYNC>), dim3(grid), dim3(waves * 64), 0, 0, d_data, n_streams, n_rec, deal, d_sink); });
    printf("workgroup per stream, shape %d aux %d, %s, barrier %d, deal %d, %2d waves x %d workgroups : %.4f ms  %6.0f GB/s\n", SHAPE, AUX,
           DB ? "next stream requested ahead" : "one stream at a time", SYNC, deal, waves, grid, ms, total_bytes / ms / 1e6);
    fflush(stdout);
}
template <int SHAPE, int AUX>
static void run_wave(int waves, int deal) {
    const double ms = time_ms([&] { hipLaunchKernelGGL((per_wave_kernel<SHAPE, AUX>), dim3(256), dim3(waves * 64), 0, 0, d_data, n_streams, n_rec, deal, d_sink); });
    printf("wave per stream, shape %d aux %d, deal %d, %2d waves/CU : %.4f ms  %6.0f GB/s\n", SHAPE, AUX, deal, waves, ms, total_bytes / ms / 1e6);
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
    for (int rep = 0; rep < 2; ++rep) {
        run_wave<0, 0>(15, 0);
        run_wave<0, 0>(15, 1);
        run_wave<0, 0>(15, 2);
        run_wave<3, 2>(15, 0);
        run_coop<0, 0, 1, 0>(16, 0);
        run_coop<0, 0, 1, 1>(16, 0);
        run_coop<0, 0, 1, 0>(16, 1);
        run_coop<0, 2, 1, 0>(16, 0);
        run_coop<0, 0, 0, 0>(16, 0);
        run_coop<3, 2, 1, 0>(16, 0);
        run_coop<3, 2, 1, 1>(16, 0);
        run_coop<3, 0, 1, 0>(16, 0);
        run_coop<0, 0, 1, 0>(13, 0);
        run_coop<0, 0, 1, 1>(8, 0, 512);
        run_coop<3, 2, 1, 1>(8, 0, 512);
    }
    return 0;
}
