// latbench.hip - development aid: what a DEPENDENT global load costs one workgroup on MI355X, cold and warm,
// within one allocation and across many (TLB reach), with and without a streaming kernel in between.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/latbench.bin tools/latbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// chain[i] holds the address of the next element; thread 0 chases n hops, reports ticks (100 MHz)
__global__ void chase(unsigned long long** start, int n, unsigned long long* out) {
    if (threadIdx.x != 0) return;
    unsigned long long t0 = wall_clock64();
    unsigned long long** p = start;
    for (int i = 0; i < n; i++) p = (unsigned long long**)__hip_atomic_load((unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long t1 = wall_clock64();
    out[0] = t1 - t0; out[1] = (unsigned long long)p;
}
__global__ void stream(const float4* a, size_t n, float* out) {
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) out[0] = s;
}
__global__ void empty(int* p) { if (p && threadIdx.x == 12345) p[0] = 1; }

int main() {
    const int NB = 48, HOPS = 48;
    std::vector<unsigned long long*> bufs(NB);
    for (int i = 0; i < NB; i++) CK(hipMalloc(&bufs[i], 4 << 20));
    unsigned long long* big; CK(hipMalloc(&big, 512ull << 20));
    unsigned long long* out; CK(hipMalloc(&out, 64));
    float* fo; CK(hipMalloc(&fo, 64));
    // chain A: HOPS hops across different allocations (one per buffer, offset varies)
    std::vector<unsigned long long> h(1);
    for (int i = 0; i < HOPS; i++) {
        unsigned long long next = (unsigned long long)(bufs[(i + 1) % NB] + 128 * ((i + 1) % 7));
        CK(hipMemcpy(bufs[i % NB] + 128 * (i % 7), &next, 8, hipMemcpyHostToDevice));
    }
    // chain B: HOPS hops inside ONE allocation at 4 KiB strides (same 2 MiB region)
    unsigned long long* one; CK(hipMalloc(&one, 4 << 20));
    for (int i = 0; i < HOPS; i++) {
        unsigned long long next = (unsigned long long)(one + 512 * ((i + 1) % HOPS));
        CK(hipMemcpy(one + 512 * i, &next, 8, hipMemcpyHostToDevice));
    }
    // chain C: inside the big allocation at 8 MiB strides
    for (int i = 0; i < HOPS; i++) {
        unsigned long long next = (unsigned long long)(big + (1u << 20) * ((i + 1) % HOPS));
        CK(hipMemcpy(big + (size_t)(1u << 20) * i, &next, 8, hipMemcpyHostToDevice));
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    auto run = [&](const char* name, unsigned long long** start, int reps, bool flush) -> int {
        for (int r = 0; r < reps; r++) {
            if (flush) { stream<<<1024, 256, 0, st>>>((const float4*)big, (512ull << 20) / 16, fo); }
            chase<<<1, 64, 0, st>>>(start, HOPS, out);
            CK(hipStreamSynchronize(st));
            unsigned long long v[2]; CK(hipMemcpy(v, out, 16, hipMemcpyDeviceToHost));
            printf("%-44s rep %d: %6.2f us total, %5.0f ns per dependent load\n", name, r, v[0] / 100.0, v[0] * 10.0 / HOPS);
        }
        return 0;
    };
    run("48 allocations, back to back", (unsigned long long**)bufs[0], 3, false);
    run("48 allocations, 512 MiB stream before each", (unsigned long long**)bufs[0], 3, true);
    run("one allocation 4 KiB strides, back to back", (unsigned long long**)one, 3, false);
    run("one allocation 4 KiB strides, stream before", (unsigned long long**)one, 3, true);
    run("big allocation 8 MiB strides, back to back", (unsigned long long**)big, 3, false);
    run("big allocation 8 MiB strides, stream before", (unsigned long long**)big, 3, true);
    // launch + sync cost of an empty kernel, and of a chain of 10
    for (int r = 0; r < 3; r++) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a, st));
        for (int i = 0; i < 20; i++) empty<<<1, 64, 0, st>>>(nullptr);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("20 dependent empty kernels: %.2f us each\n", ms * 1000 / 20);
    }
    // the same chain as kernel nodes of a captured hipGraph (does a graph shorten the dependent-dispatch gap?)
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; i++) empty<<<1, 64, 0, st>>>(nullptr);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 4; r++) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("graph of 200 dependent empty kernels: %.2f us each\n", ms * 1000 / 200);
        }
        for (int r = 0; r < 3; r++) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, st));
            for (int i = 0; i < 200; i++) empty<<<1, 64, 0, st>>>(nullptr);
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("200 dependent empty kernels, plain launches: %.2f us each\n", ms * 1000 / 200);
        }
    }
    return 0;
}
