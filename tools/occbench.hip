// occbench.hip - development tool: how many 512-thread workgroups of a given LDS size and register count does a CU of this GPU hold at once?
// Every workgroup stamps its entry, spins for ~30 us and stamps its exit; "resident at once" = workgroups that entered before the first one left.
// build: hipcc --offload-arch=gfx950 -O3 tools/occbench.hip -o tools/occbench.bin     run: tools/occbench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int NV>
__global__ void __launch_bounds__(512) k_occ(unsigned long long* out, unsigned long long ticks) {
    extern __shared__ unsigned int s_dyn[];
    if (NV >= 128) asm volatile("" ::: "v125");
    if (NV >= 168) asm volatile("" ::: "v165");
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) s_dyn[0] = (unsigned int)t0;
    __syncthreads();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = wall_clock64() + (s_dyn[0] & 0u); }
}

template <int NV> static void run(int threads, size_t lds, int wgs) {
    unsigned long long* d; (void)hipMalloc(&d, sizeof(unsigned long long) * 2 * wgs);
    (void)hipFuncSetAttribute((const void*)k_occ<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; rep++) { k_occ<NV><<<wgs, threads, lds>>>(d, 3000); (void)hipDeviceSynchronize(); }
    hipError_t e = hipGetLastError();
    std::vector<unsigned long long> h(2 * wgs);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long first_exit = ~0ull, t0 = ~0ull, tend = 0;
    for (int i = 0; i < wgs; i++) { first_exit = std::min(first_exit, h[2 * i + 1]); t0 = std::min(t0, h[2 * i]); tend = std::max(tend, h[2 * i + 1]); }
    int at_once = 0;
    for (int i = 0; i < wgs; i++) at_once += h[2 * i] < first_exit;
    printf("vgpr %3d threads %4d lds %6zu B: %4d of %d workgroups resident at once (%.2f per CU), kernel %.1f us%s\n", NV, threads, lds, at_once, wgs, at_once / 256.0, (tend - t0) * 0.01, e == hipSuccess ? "" : "  LAUNCH ERROR");
    (void)hipFree(d);
}

int main() {
    const size_t sizes[] = {0, 16384, 32768, 40960, 49152, 57344, 65536, 66560, 73728, 80896, 81920};
    for (size_t s : sizes) run<64>(512, s, 1024);
    for (size_t s : sizes) run<128>(512, s, 1024);
    for (size_t s : {(size_t)0, (size_t)32768, (size_t)73728}) run<128>(256, s, 2048);
    for (size_t s : {(size_t)0, (size_t)32768}) run<168>(512, s, 1024);
    return 0;
}
