// What does hipHostFree say about a host-mapped allocation a kernel has written to?  (round 2: it said "invalid argument")
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void w(volatile int* p) { p[threadIdx.x] = (int)threadIdx.x; }
int main() {
    for (unsigned flags : {0u, (unsigned)hipHostMallocMapped, (unsigned)hipHostMallocDefault}) {
        unsigned char* h = nullptr;
        unsigned char* d = nullptr;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&h), 4096, flags);
        printf("flags %u: malloc %s", flags, hipGetErrorString(e));
        e = hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0);
        printf(" getdev %s (same %d)", hipGetErrorString(e), (int)(d == h));
        memset(h, 0, 4096);
        hipLaunchKernelGGL(w, dim3(1), dim3(64), 0, 0, reinterpret_cast<volatile int*>(d));
        e = hipDeviceSynchronize();
        printf(" sync %s v=%d", hipGetErrorString(e), reinterpret_cast<int*>(h)[5]);
        e = hipHostFree(h);
        printf(" free %s\n", hipGetErrorString(e));
    }
    return 0;
}
