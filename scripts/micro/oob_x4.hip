// Micro-test (measurement aid): what does a raw buffer_load_dwordx4 return when only its first dwords lie inside the descriptor's
// range -- zeros for the dwords beyond it (checked per dword) or zeros for all four (checked per access)?
//   build: hipcc --offload-arch=gfx950 -O3 oob_x4.hip -o oob_x4
#include <hip/hip_runtime.h>

#include <cstdio>

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned int* base, unsigned int bytes, unsigned int* out) {
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(base), (short)0, (int)bytes, 0x00020000);
    const int lane = threadIdx.x;
    // lane l reads 16 bytes at byte offset 20 l + 12 (the last lanes straddle or pass the end)
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 20, 12, 0);
    out[4 * lane + 0] = v.x;
    out[4 * lane + 1] = v.y;
    out[4 * lane + 2] = v.z;
    out[4 * lane + 3] = v.w;
    const u4 n = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 20, 12, 2);
    out[256 + 4 * lane + 0] = n.x;
    out[256 + 4 * lane + 1] = n.y;
    out[256 + 4 * lane + 2] = n.z;
    out[256 + 4 * lane + 3] = n.w;
}

int main() {
    unsigned int *d, *o, h[512];
    hipMalloc(&d, 4096);
    hipMalloc(&o, 2048);
    for (int i = 0; i < 512; ++i) h[i] = 1000u + i;
    hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
    for (unsigned int nrec : {60u, 61u}) {
        const unsigned int bytes = nrec * 20u;  // the stream holds nrec records: dwords 0 .. 5 nrec - 1
        hipMemset(o, 0xFF, 2048);
        probe<<<1, 64>>>(d, bytes, o);
        unsigned int r[512];
        hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
        printf("stream of %u records (%u bytes, %u dwords)\n", nrec, bytes, bytes / 4);
        for (int l = (int)nrec - 3; l < (int)nrec + 1 && l < 64; ++l)
            printf("  lane %2d wants dwords %3d..%3d : default %u %u %u %u | nt %u %u %u %u\n", l, 5 * l + 3, 5 * l + 6, r[4 * l], r[4 * l + 1], r[4 * l + 2],
                   r[4 * l + 3], r[256 + 4 * l], r[256 + 4 * l + 1], r[256 + 4 * l + 2], r[256 + 4 * l + 3]);
    }
    return 0;
}
