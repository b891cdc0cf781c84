// does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on this GPU?  (hip_ext.h says: not supported on gfx9xx)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (out) atomicAdd(out, 1u);
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    unsigned* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int flags = 0; flags < 2; ++flags) {
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, 20000ull, d);
        hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, 20000ull, d);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("flags %d: two 200-us kernels on one stream took %.1f us\n", flags, ms * 1e3f);
    }
    return 0;
}
