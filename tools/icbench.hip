// icbench.hip - development aid: what executing N KiB of straight-line code ONCE costs one workgroup on MI355X
// (instruction fetch), back to back and with another kernel / a big stream in between.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
// one v_add_u32 = 4 bytes (VOP2) .. use 8-byte VOP3 form to be safe: v_add3_u32 8 bytes
#define INS "v_add3_u32 %0, %0, %0, 1\n"
template <int KB> __global__ void __launch_bounds__(1024) code(unsigned long long* out, unsigned* sink) {
    unsigned v = threadIdx.x;
    unsigned long long t0 = wall_clock64();
    // 256 instructions x 8 B = 2 KiB per block
#pragma unroll
    for (int i = 0; i < KB / 2; i++) asm volatile(R256(INS) : "+v"(v));
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (v == 0x12345678u) sink[0] = v;
}
__global__ void stream(const float4* a, size_t n, float* out) {
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) out[0] = s;
}
template <int KB> int run(hipStream_t st, unsigned long long* out, unsigned* sink, const float4* big, float* fo, int threads) {
    for (int mode = 0; mode < 3; mode++)
        for (int r = 0; r < 3; r++) {
            if (mode == 1) code<2><<<256, 256, 0, st>>>(out + 8, sink);          // another kernel on every CU in between
            if (mode == 2) stream<<<1024, 256, 0, st>>>(big, (512ull << 20) / 16, fo);
            code<KB><<<1, threads, 0, st>>>(out, sink);
            CK(hipStreamSynchronize(st));
            unsigned long long v; CK(hipMemcpy(&v, out, 8, hipMemcpyDeviceToHost));
            printf("%3d KiB code, %4d threads, %-28s rep %d: %7.2f us (%5.1f ns per 64-B line)\n", KB, threads,
                   mode == 0 ? "back to back" : mode == 1 ? "small kernel in between" : "512 MiB stream in between", r, v / 100.0, v * 10.0 / (KB * 16));
        }
    return 0;
}
int main() {
    unsigned long long* out; CK(hipMalloc(&out, 256)); unsigned* sink; CK(hipMalloc(&sink, 64));
    float4* big; CK(hipMalloc(&big, 512ull << 20)); float* fo; CK(hipMalloc(&fo, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    run<8>(st, out, sink, big, fo, 1024);
    run<32>(st, out, sink, big, fo, 1024);
    run<64>(st, out, sink, big, fo, 1024);
    run<32>(st, out, sink, big, fo, 64);
    return 0;
}
