// xcdbench.hip - measurements and litmus tests behind the persistent band kernel:
//   1. where a CU-masked stream puts workgroups (HW_REG_XCC_ID per workgroup, for several candidate masks)
//   2. the price of a grid barrier among N workgroups (one counter, relaxed agent-scope add, sc1-load poll, no fences),
//      alone and beside a kernel that saturates HBM from the other XCDs
//   3. LITMUS: a word handed from workgroup to workgroup across such a barrier - payload stored plain (or sc1), drained
//      with s_waitcnt vmcnt(0) before the arrival; read with L1-bypassing loads (nt / sc1) - checked every iteration;
//      and the NEGATIVE variant (payload re-read with plain loads: stale lines in the reader's L1), which is EXPECTED to
//      show errors - so that the protocol is pinned by a test that can fail.
//   4. LITMUS: the ticket pattern of the product's last-workgroup reductions (k_ticket), with its negative variant
// Build: hipcc --offload-arch=gfx950 -O3 tools/xcdbench.hip -o tools/xcdbench.bin    Run: tools/xcdbench.bin [iters [ticket rounds]]
// (tests/test_litmus.py runs it on the GPU box)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }   // HW_REG_XCC_ID[3:0]

__global__ void k_where(uint32_t* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// a load with no cache-policy bits at all (a `volatile` access compiles to sc0 sc1: that is not the hazard under test)
__device__ __forceinline__ uint32_t plain_load(const uint32_t* p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

constexpr unsigned long long SPIN_LIMIT = 200000000ull;   // 2 s of the 100 MHz wall clock

// arrive + wait: generation g (1, 2, ...) is complete when the counter reaches g * n
__device__ __forceinline__ bool grid_barrier(uint32_t* ctr, uint32_t n, uint32_t gen, uint32_t* tmo) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's stores have reached L2 / memory
    __syncthreads();                                       // ... and every other wave's of the workgroup
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * n) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > SPIN_LIMIT) { *tmo = 1; ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

// MODE 0: barriers only.  1: plain store + nt load.  2: sc1 store + sc1 load.  3 (negative): plain store + PLAIN load.
// 4: plain store + nt load with NO drain before the arrival (negative: the flag may overtake the payload)
template <int MODE>
__global__ void __launch_bounds__(256) k_chain(uint32_t* ctr, uint32_t* payload, uint32_t iters, uint32_t* errs, uint32_t* tmo, unsigned long long* ticks, uint32_t* xcc) {
    const uint32_t n = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    if (t == 0) xcc[b] = xcc_id();
    uint32_t gen = 0, bad = 0;
    const unsigned long long t0 = wall_clock64();
    for (uint32_t it = 1; it <= iters; it++) {
        if (MODE != 0) {
            const uint32_t v = it * 4096u + b;
            uint32_t* p = payload + (size_t)b * 256 + t;
            if (MODE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *p = v;
        }
        if (MODE == 4) {       // no drain: arrive at once
            __syncthreads();
            if (t == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long s0 = wall_clock64();
                ++gen;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * n) { if (wall_clock64() - s0 > SPIN_LIMIT) { *tmo = 1; break; } }
            }
            __syncthreads();
        } else if (!grid_barrier(ctr, n, ++gen, tmo)) break;
        if (MODE != 0) {
            const uint32_t src = (b + 1u) % n;
            const uint32_t* p = payload + (size_t)src * 256 + t;
            uint32_t v;
            if (MODE == 1 || MODE == 4) v = __builtin_nontemporal_load(p);
            else if (MODE == 2) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = plain_load(p);
            if (v != it * 4096u + src) bad++;
        }
        if (!grid_barrier(ctr, n, ++gen, tmo)) break;       // (the payload is rewritten next iteration)
    }
    if (t == 0 && b == 0) *ticks = wall_clock64() - t0;
    if (bad) atomicAdd(errs, bad);
}

// The same chain among the workgroups that find themselves on ONE XCD, without any CU mask: 8 x the workgroups are
// launched; the first to arrive elects its XCD (CAS on a word), every workgroup adds itself to `started` and - when it
// sits on the elected XCD - to `members`; the others leave at once.  The first barrier also waits for every launched
// workgroup to have started, so that `members` is final.  Same-XCD is then a hardware fact each member checked for
// itself (HW_REG_XCC_ID), not an assumption about the dispatcher: plain stores + L1-bypassing loads meet in that L2.
// ctl: [0] barrier counter, [16] elected XCD + 1, [32] started, [48] members   (words on cache lines of their own)
template <int MODE>
__global__ void __launch_bounds__(256) k_chain_elect(uint32_t* ctl, uint32_t* payload, uint32_t iters, uint32_t* errs, uint32_t* tmo, unsigned long long* ticks, uint32_t* xcc) {
    __shared__ uint32_t s_me, s_n, s_go;
    const uint32_t t = threadIdx.x;
    if (t == 0) {
        const uint32_t mine = xcc_id() + 1u;
        uint32_t expected = 0u;
        __hip_atomic_compare_exchange_strong(ctl + 16, &expected, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t elected = expected ? expected : mine;
        s_go = elected == mine;
        s_me = s_go ? __hip_atomic_fetch_add(ctl + 48, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // (the member count before the start count)
        __hip_atomic_fetch_add(ctl + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_go) {
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(ctl + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) { __builtin_amdgcn_s_sleep(4); if (wall_clock64() - t0 > SPIN_LIMIT) { *tmo = 2; s_go = 0; break; } }
            s_n = __hip_atomic_load(ctl + 48, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!s_go) return;
    const uint32_t n = s_n, b = s_me;
    if (t == 0) { xcc[b] = xcc_id(); if (b == 0) xcc[1023] = n; }
    uint32_t gen = 0, bad = 0;
    const unsigned long long t0 = wall_clock64();
    for (uint32_t it = 1; it <= iters; it++) {
        if (MODE != 0) {
            const uint32_t v = it * 4096u + b;
            uint32_t* p = payload + (size_t)b * 256 + t;
            if (MODE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
        }
        if (MODE == 6) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // (every wave: the textbook form, L2 write-back)
        if (!grid_barrier(ctl, n, ++gen, tmo)) break;
        if (MODE >= 5) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // L1 invalidate, every wave
        if (MODE != 0) {
            const uint32_t src = (b + 1u) % n;
            const uint32_t* p = payload + (size_t)src * 256 + t;
            uint32_t v;
            if (MODE == 1) v = __builtin_nontemporal_load(p);
            else if (MODE == 2) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = plain_load(p);
            if (v != it * 4096u + src) bad++;
        }
        if (!grid_barrier(ctl, n, ++gen, tmo)) break;
    }
    if (t == 0 && b == 0) *ticks = wall_clock64() - t0;
    if (bad) atomicAdd(errs, bad);
}

// TICKET pattern of the product (sweep_finish of k_recount_bits, k_close): every workgroup stores its four partial values
// write-through (sc1), drains them, takes a ticket (relaxed agent-scope add); the workgroup whose add came LAST reads every
// slot with sc1 loads behind a workgroup barrier and checks the total.  PLAIN = 1: the negative variant - the slots are
// stored with plain stores, which stay in the storing XCD's L2: the last workgroup (on another XCD) must see stale slots.
// Rounds are separated by the grid barrier (the product: by a kernel boundary).
template <int PLAIN>
__global__ void __launch_bounds__(256) k_ticket(uint32_t* ctr, uint32_t* ticket, unsigned long long* slots, uint32_t iters, uint32_t* errs, uint32_t* tmo, uint32_t* lasts) {
    __shared__ int s_last;
    const uint32_t n = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    uint32_t gen = 0;
    for (uint32_t it = 1; it <= iters; it++) {
        if (t < 4) {
            const unsigned long long v = (unsigned long long)it * 1000003ull + b * 4u + t;
            if (PLAIN) slots[(size_t)b * 4 + t] = v; else __hip_atomic_store(&slots[(size_t)b * 4 + t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            const uint32_t k = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (k == it * n - 1u);
        }
        __syncthreads();
        if (s_last) {
            unsigned long long sum = 0;
            for (uint32_t i = t; i < n * 4u; i += 256u) sum += __hip_atomic_load(&slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            __shared__ unsigned long long s_sum[4];
            if ((t & 63) == 0) s_sum[t >> 6] = sum;
            __syncthreads();
            if (t == 0) {
                const unsigned long long got = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
                unsigned long long want = 0;
                for (uint32_t i = 0; i < n * 4u; i++) want += (unsigned long long)it * 1000003ull + i;
                if (got != want) atomicAdd(errs, 1u);
                atomicAdd(lasts, 1u);
            }
        }
        if (!grid_barrier(ctr, n, ++gen, tmo)) break;
    }
}

// background load: streams `n` float4 from src with non-temporal loads, `reps` times
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_stream(const float4* src_, size_t n, int reps, float* sink) {
    const f4v* src = reinterpret_cast<const f4v*>(src_);
    float acc = 0;
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            f4v v = __builtin_nontemporal_load(src + i);
            acc += v.x + v.y + v.z + v.w;
        }
    if (acc == 12345.678f) *sink = acc;
}

static void mask_every(uint32_t* m, int stride, int offset) { std::memset(m, 0, 32); for (int i = offset; i < 256; i += stride) m[i / 32] |= 1u << (i % 32); }
static void mask_range(uint32_t* m, int lo, int hi) { std::memset(m, 0, 32); for (int i = lo; i < hi; i++) m[i / 32] |= 1u << (i % 32); }
static void mask_not(uint32_t* m) { for (int i = 0; i < 8; i++) m[i] = ~m[i]; }

int main(int argc, char** argv) {
    const uint32_t iters = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 20000;
    CK(hipSetDevice(0));
    uint32_t *d_where, *d_ctr, *d_payload, *d_errs, *d_tmo, *d_xcc; unsigned long long* d_ticks; float* d_sink;
    CK(hipMalloc(&d_where, 4096 * 4)); CK(hipMalloc(&d_ctr, 256)); CK(hipMalloc(&d_payload, 256 * 256 * 4)); CK(hipMalloc(&d_errs, 4));
    CK(hipMalloc(&d_tmo, 4)); CK(hipMalloc(&d_xcc, 1024 * 4)); CK(hipMalloc(&d_ticks, 8)); CK(hipMalloc(&d_sink, 4));
    const size_t big = (size_t)1 << 30;                    // 1 GiB to stream
    float4* d_big; CK(hipMalloc(&d_big, big)); CK(hipMemset(d_big, 0, big));

    // ---- 1. placement under candidate CU masks
    struct { const char* name; uint32_t m[8]; } masks[4];
    masks[0].name = "bits i%8==0 (32 CUs)"; mask_every(masks[0].m, 8, 0);
    masks[1].name = "bits 0..31 (32 CUs)"; mask_range(masks[1].m, 0, 32);
    masks[2].name = "bits i%8==3 (32 CUs)"; mask_every(masks[2].m, 8, 3);
    masks[3].name = "all"; mask_range(masks[3].m, 0, 256);
    int one_xcd_mask = -1;
    for (int k = 0; k < 4; k++) {
        hipStream_t st; CK(hipExtStreamCreateWithCUMask(&st, 8, masks[k].m));
        CK(hipMemsetAsync(d_where, 0xff, 4096 * 4, st));
        k_where<<<512, 64, 0, st>>>(d_where);
        CK(hipStreamSynchronize(st));
        std::vector<uint32_t> h(512); CK(hipMemcpy(h.data(), d_where, 512 * 4, hipMemcpyDeviceToHost));
        int hist[16] = {0}; for (auto v : h) hist[v & 15]++;
        std::printf("placement, mask '%s': workgroups per XCC_ID:", masks[k].name);
        int used = 0; for (int i = 0; i < 16; i++) if (hist[i]) { std::printf(" [%d]=%d", i, hist[i]); used++; }
        std::printf("\n");
        if (used == 1 && one_xcd_mask < 0) one_xcd_mask = k;
        CK(hipStreamDestroy(st));
    }
    std::printf("one-XCD mask: %s\n", one_xcd_mask >= 0 ? masks[one_xcd_mask].name : "NONE FOUND");

    // ---- 2./3. barrier price and hand-off litmus
    hipStream_t s_one = nullptr, s_rest = nullptr, s_all = nullptr, s_bg = nullptr;
    CK(hipStreamCreateWithFlags(&s_all, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_bg, hipStreamNonBlocking));
    if (one_xcd_mask >= 0) {
        CK(hipExtStreamCreateWithCUMask(&s_one, 8, masks[one_xcd_mask].m));
        uint32_t rest[8]; std::memcpy(rest, masks[one_xcd_mask].m, 32); mask_not(rest);
        CK(hipExtStreamCreateWithCUMask(&s_rest, 8, rest));
    }
    auto run = [&](const char* what, int mode, hipStream_t st, uint32_t nwg, bool loaded, hipStream_t bg) {
        CK(hipMemset(d_ctr, 0, 256)); CK(hipMemset(d_errs, 0, 4)); CK(hipMemset(d_tmo, 0, 4)); CK(hipMemset(d_payload, 0, 256 * 256 * 4));
        CK(hipDeviceSynchronize());
        if (loaded) k_stream<<<1792, 256, 0, bg>>>(d_big, big / 16, 300, d_sink);      // ~0.2 ms per GiB: outlasts the chain kernel
        switch (mode) {
            case 0: k_chain<0><<<nwg, 256, 0, st>>>(d_ctr, d_payload, iters, d_errs, d_tmo, d_ticks, d_xcc); break;
            case 1: k_chain<1><<<nwg, 256, 0, st>>>(d_ctr, d_payload, iters, d_errs, d_tmo, d_ticks, d_xcc); break;
            case 2: k_chain<2><<<nwg, 256, 0, st>>>(d_ctr, d_payload, iters, d_errs, d_tmo, d_ticks, d_xcc); break;
            case 3: k_chain<3><<<nwg, 256, 0, st>>>(d_ctr, d_payload, iters, d_errs, d_tmo, d_ticks, d_xcc); break;
            default: k_chain<4><<<nwg, 256, 0, st>>>(d_ctr, d_payload, iters, d_errs, d_tmo, d_ticks, d_xcc); break;
        }
        CK(hipDeviceSynchronize());
        uint32_t errs, tmo; unsigned long long ticks; std::vector<uint32_t> x(nwg);
        CK(hipMemcpy(&errs, d_errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&tmo, d_tmo, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(x.data(), d_xcc, nwg * 4, hipMemcpyDeviceToHost));
        int hist[16] = {0}, used = 0; for (auto v : x) hist[v & 15]++; for (int i = 0; i < 16; i++) used += hist[i] != 0;
        std::printf("%-58s nwg %3u on %d XCD(s) %s: %.3f us per iteration (2 barriers%s), errors %u of %llu%s\n", what, nwg, used, loaded ? "beside an HBM stream" : "alone",
                    ticks * 0.01 / iters, mode ? " + store + dependent load" : "", errs, (unsigned long long)iters * nwg * 256, tmo ? "  TIMEOUT" : "");
        return errs;
    };
    auto run_elect = [&](const char* what, int mode, uint32_t launched, bool loaded) {
        CK(hipMemset(d_ctr, 0, 256)); CK(hipMemset(d_errs, 0, 4)); CK(hipMemset(d_tmo, 0, 4)); CK(hipMemset(d_payload, 0, 256 * 256 * 4)); CK(hipMemset(d_xcc, 0, 1024 * 4));
        CK(hipDeviceSynchronize());
        if (loaded) k_stream<<<1792, 256, 0, s_bg>>>(d_big, big / 16, 300, d_sink);
        switch (mode) {
            case 0: k_chain_elect<0><<<launched, 256, 0, s_all>>>(d_ctr, d_payload, iters, d_errs, d_tmo, d_ticks, d_xcc); break;
            case 1: k_chain_elect<1><<<launched, 256, 0, s_all>>>(d_ctr, d_payload, iters, d_errs, d_tmo, d_ticks, d_xcc); break;
            case 2: k_chain_elect<2><<<launched, 256, 0, s_all>>>(d_ctr, d_payload, iters, d_errs, d_tmo, d_ticks, d_xcc); break;
            case 3: k_chain_elect<3><<<launched, 256, 0, s_all>>>(d_ctr, d_payload, iters, d_errs, d_tmo, d_ticks, d_xcc); break;
            case 5: k_chain_elect<5><<<launched, 256, 0, s_all>>>(d_ctr, d_payload, iters, d_errs, d_tmo, d_ticks, d_xcc); break;
            default: k_chain_elect<6><<<launched, 256, 0, s_all>>>(d_ctr, d_payload, iters, d_errs, d_tmo, d_ticks, d_xcc); break;
        }
        CK(hipDeviceSynchronize());
        uint32_t errs, tmo, n; unsigned long long ticks;
        CK(hipMemcpy(&errs, d_errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&tmo, d_tmo, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&n, d_xcc + 1023, 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> x(n ? n : 1); CK(hipMemcpy(x.data(), d_xcc, x.size() * 4, hipMemcpyDeviceToHost));
        int used = 0, hist[16] = {0}; for (auto v : x) hist[v & 15]++; for (int i = 0; i < 16; i++) used += hist[i] != 0;
        std::printf("%-58s %3u members of %u launched, on %d XCD(s) %s: %.3f us per iteration (2 barriers%s), errors %u%s\n", what, n, launched, used, loaded ? "beside an HBM stream" : "alone",
                    ticks * 0.01 / iters, mode ? " + store + dependent load" : "", errs, tmo ? "  TIMEOUT" : "");
        return errs;
    };
    int rc = 0;
    for (int loaded = 0; loaded < 2; loaded++) {
        for (uint32_t launched : {64u, 256u, 512u}) run_elect("ELECTED XCD: barriers only", 0, launched, loaded);
        for (uint32_t launched : {256u, 512u}) if (run_elect("ELECTED XCD LITMUS plain store -> drain -> barrier -> nt load", 1, launched, loaded)) rc = 1;
        if (run_elect("ELECTED XCD LITMUS sc1 store -> drain -> barrier -> sc1 load", 2, 256, loaded)) rc = 1;
        if (run_elect("ELECTED XCD plain store -> drain -> barrier -> ACQUIRE fence (L1 inv) -> plain load", 5, 256, loaded)) rc = 1;
        if (run_elect("ELECTED XCD plain store -> RELEASE fence -> barrier -> ACQUIRE fence -> plain load", 6, 256, loaded)) rc = 1;
        { const uint32_t neg = run_elect("ELECTED XCD NEGATIVE plain store -> barrier -> PLAIN load", 3, 256, loaded);
          std::printf("   negative variant %s\n", neg ? "failed as expected: the litmus can see a broken protocol" : "showed NO error (litmus blind here?)"); }
        if (s_one) {
            for (uint32_t nwg : {8u, 32u, 64u}) run("barriers only, one XCD", 0, s_one, nwg, loaded, s_rest);
            if (run("LITMUS plain store -> drain -> barrier -> nt load, one XCD", 1, s_one, 32, loaded, s_rest)) rc = 1;
            if (run("LITMUS plain store -> drain -> barrier -> nt load, one XCD", 1, s_one, 64, loaded, s_rest)) rc = 1;
            if (run("LITMUS sc1 store -> drain -> barrier -> sc1 load, one XCD", 2, s_one, 32, loaded, s_rest)) rc = 1;
            const uint32_t neg = run("NEGATIVE plain store -> barrier -> PLAIN load (expect errors)", 3, s_one, 32, loaded, s_rest);
            std::printf("   negative variant %s\n", neg ? "failed as expected: the litmus can see a broken protocol" : "showed NO error (litmus blind here?)");
            run("NEGATIVE no drain before the arrival (may show errors)", 4, s_one, 32, loaded, s_rest);
        }
        for (uint32_t nwg : {32u, 256u}) run("barriers only, all XCDs", 0, s_all, nwg, loaded, s_bg);
        if (run("LITMUS sc1 store -> drain -> barrier -> sc1 load, all XCDs", 2, s_all, 32, loaded, s_bg)) rc = 1;
        if (run("LITMUS sc1 store -> drain -> barrier -> sc1 load, all XCDs", 2, s_all, 256, loaded, s_bg)) rc = 1;
        const uint32_t e1 = run("plain store -> drain -> barrier -> nt load, ALL XCDs (cross-XCD: expect errors)", 1, s_all, 32, loaded, s_bg);
        std::printf("   plain stores across XCDs: %s\n", e1 ? "stale, as the guide says (plain stores are not write-through)" : "no error seen");
    }
    // ---- ticket pattern of the product's last-workgroup reductions
    {
        unsigned long long* d_slots; uint32_t* d_ticket; uint32_t* d_lasts;
        CK(hipMalloc(&d_slots, 1024 * 4 * 8)); CK(hipMalloc(&d_ticket, 256)); CK(hipMalloc(&d_lasts, 4));
        for (int plain = 0; plain < 2; plain++)
            for (int loaded = 0; loaded < 2; loaded++) {
                const uint32_t nwg = 256, its = argc > 2 ? (uint32_t)std::atoi(argv[2]) : iters * 5;
                CK(hipMemset(d_ctr, 0, 256)); CK(hipMemset(d_ticket, 0, 256)); CK(hipMemset(d_errs, 0, 4)); CK(hipMemset(d_tmo, 0, 4)); CK(hipMemset(d_lasts, 0, 4)); CK(hipMemset(d_slots, 0, 1024 * 4 * 8));
                CK(hipDeviceSynchronize());
                if (loaded) k_stream<<<1792, 256, 0, s_bg>>>(d_big, big / 16, 300, d_sink);
                if (plain) k_ticket<1><<<nwg, 256, 0, s_all>>>(d_ctr, d_ticket, d_slots, its, d_errs, d_tmo, d_lasts);
                else k_ticket<0><<<nwg, 256, 0, s_all>>>(d_ctr, d_ticket, d_slots, its, d_errs, d_tmo, d_lasts);
                CK(hipDeviceSynchronize());
                uint32_t errs, tmo, lasts;
                CK(hipMemcpy(&errs, d_errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&tmo, d_tmo, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&lasts, d_lasts, 4, hipMemcpyDeviceToHost));
                std::printf("TICKET %s slots -> drain -> ticket -> last workgroup reads sc1, 256 workgroups on all XCDs %s: %u rounds checked, %u wrong totals%s\n",
                            plain ? "PLAIN-stored (negative)" : "sc1-stored", loaded ? "beside an HBM stream" : "alone", lasts, errs, tmo ? "  TIMEOUT" : "");
                if (!plain && (errs || lasts != its)) rc = 1;
                if (plain) std::printf("   negative variant %s\n", errs ? "failed as expected: a plain store in the hand-off set is caught" : "showed NO error");
                if (plain && !errs) rc = 1;
            }
    }
    std::printf(rc ? "LITMUS FAILED\n" : "LITMUS OK\n");
    return rc;
}
