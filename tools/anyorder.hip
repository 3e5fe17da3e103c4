// Does hipExtAnyOrderLaunch let a kernel start before its predecessor on the same stream has finished (gfx950)?
// Kernel A spins 20 us and stamps entry/exit; kernel B stamps entry.  No hand-off between them: nothing can hang.
// Build: hipcc --offload-arch=gfx950 -O2 tools/anyorder.hip -o tools/anyorder.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void kA(long long* t) { long long a = wall_clock64(); while (wall_clock64() - a < 2000) {} if (threadIdx.x == 0) { t[0] = a; t[1] = wall_clock64(); } }
__global__ void kB(long long* t) { if (threadIdx.x == 0) t[2] = wall_clock64(); }
int main() {
    long long* t; hipMalloc(&t, 64); hipStream_t s; hipStreamCreate(&s);
    for (int flags = 0; flags < 2; flags++) for (int rep = 0; rep < 3; rep++) {
        hipMemsetAsync(t, 0, 64, s);
        hipExtLaunchKernelGGL(kA, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, t);
        hipExtLaunchKernelGGL(kB, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags, t); hipError_t e = hipGetLastError();
        hipStreamSynchronize(s);
        long long h[3]; hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
        printf("flags %d: err %d  A ran %.2f us, B entered %.2f us after A's exit\n", flags, (int)e, (h[1] - h[0]) / 100.0, (h[2] - h[1]) / 100.0);
    }
    return 0;
}
