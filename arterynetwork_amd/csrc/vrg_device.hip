// vrg_device.hip - the product backend: HIP kernels for MI355X (gfx950, wave64).
//
// One while-loop trip of variationalRegionGrowing.py:58-117 (be_sweep_once), two HIP streams:
//   stream A, band kernels (O(band) work, grid-stride over device-resident counts, no host round trip):
//     k_decide_exact (decide + listing; its second half: exact densities of the entries the previous sweep
//        added, which it then decides) -> k_marks_prepass (stop tests, marks, skip-rule prepass)
//     -> k_relabel (skip-rule fix-point; 3x3x3 / 5x5x5 label stencil on the marked voxels, old labels only)
//     -> k_apply_entry_post (writes the new label bytes, keeps the region sizes and the class bits in step; its
//        second half: per-entry survivor test and flip bookkeeping, fed by the relabel's per-entry result)
//     -> k_levels_tab_scan (level-delta compaction, correction memo, first pass of the rebuild scan)
//     -> k_scan_down -> k_scatter -> k_finalize closes the trip                                  (8 launches);
//   stream B, the dense pass, forked after k_levels_tab_scan:
//     k_recount_bits : the dense kernel (every voxel, HBM-bound, read-only: 4 B intensity + 2 class bits per
//        voxel): region sizes and intensity sums (:113-116, :249-250), reduced by its last workgroup, checked
//        against the sizes the band side keeps by increments
//     -> on several GPUs: slab all-reduce -> k_dense_fin (the same check on the totals, trace sums).
//   Stream A does not join: it runs up to two sweeps ahead of the dense pass (two copies of the class bits; see
//   be_sweep_once).
// Labels are updated IN PLACE: measured on MI355X, streaming I + labels read-only runs at 5.8-6.0 TB/s
// while the same stream with a 1 B/voxel label write-back drops to 4.8 TB/s, so unchanged labels are
// never rewritten.  (The full-stencil check variant relabels every voxel through lab[1].)
// Every kernel starts by reading the device-resident VrgState and returns at once when the stop
// flag is set, so the host can enqueue batches of sweeps without synchronising.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>
#include <rccl/rccl.h>

#include "vrg_backend.h"
#include "vrg_items.h"

// a failed HIP call is remembered (first one wins) and reported by be_last_error(); the engine turns it into
// VRG_E_INTERNAL at its next synchronisation point
static char g_hip_error[256] = "";
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess && !g_hip_error[0]) { \
    std::snprintf(g_hip_error, sizeof(g_hip_error), "HIP error '%s' in %s (%s:%d)", hipGetErrorString(e_), #x, __FILE__, __LINE__); \
    std::fprintf(stderr, "%s\n", g_hip_error); } } while (0)

namespace {

constexpr int TPB = 256;            // 4 waves of 64
constexpr int ITEM_BLOCKS = 256;    // band kernels: 64 Ki threads, grid-stride
constexpr int SCAN_BLOCKS = 256;
constexpr int SWEEP_BLOCKS = 256;   // 1 workgroup (4 waves) per CU, each wave with 3 KiB of labels + 12 KiB of intensities in
                                    // flight: measured best for the HBM-bound recount while stream B's band kernels run beside
                                    // it (880x880x640: 256 -> 0.38 ms, 192/384 -> 0.42-0.43, 320 -> 0.49, 512 -> 0.40, 1024 -> 0.44)

hipStream_t g_stream = nullptr;      // stream A: the band kernels of every trip in program order, copies
hipStream_t g_stream_b = nullptr;    // stream B: the dense pass (recount, slab all-reduce, k_dense_fin); trails stream A by up to one sweep
hipEvent_t g_ev_a = nullptr, g_ev_d[2] = {nullptr, nullptr};   // labels applied (A -> B) / class copy read by the dense pass (B -> A)
hipEvent_t g_read[2] = {nullptr, nullptr};       // the "class copy read" events of the last two dense passes, by trip parity
unsigned long long g_trip = 0;
int g_sweep_blocks = 0;              // 0 = auto (dense_blocks)
int g_prio_mode = 2;                 // the dense stream gets the higher priority (measured: -1..2 % step time)
int g_use_graph_req = 0;
ncclComm_t g_comm = nullptr;          // per-sweep all-reduce of the slab statistics (multi-GPU)

// ---- wave / block primitives (wave = 64 lanes) -------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);   // fixed butterfly: deterministic
    return v;
}
__device__ __forceinline__ long long wave_sum(long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    return v;
}
// exclusive scan of one value per thread over a 256-thread block; returns the block total in `total`
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t& total, uint32_t* sh /*4+*/) {
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int i = 0; i < w; i++) base += sh[i];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + inc - v;
}

// ---- item kernels ---------------------------------------------------------------------------------
#define ITEM_LOOP(n) for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n_ = (n); i < n_; i += gridDim.x * blockDim.x)
// the same over the first `g` workgroups of a grid whose other workgroups do something else
#define ITEM_LOOP_G(n, g) for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n_ = (n); i < n_; i += (g) * blockDim.x)
// same with a 64-bit item index: (listed flips) x (positions) can exceed 2^32 on adversarial volumes
#define ITEM_LOOP64(n) for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n_ = (n); i < n_; i += (uint64_t)gridDim.x * blockDim.x)

__device__ void exact_wave(const VrgCtx& c, int par, uint32_t nfresh, uint32_t wid, uint32_t nw, bool then_decide);
// decide (:79-88) + listing of the flips - and, in the second half of the grid, the exact densities of the entries
// that (re-)entered the band in the sweep just closed (:252-255), one wave per entry, each of which is decided by
// that wave as soon as its densities exist (the first half skips entries whose densities are still pending).
// The closed sweep's k_exact launch is gone from the chain.
__global__ void k_decide_exact(VrgCtx c) {
    if (c.st->done) return;
    if (blockIdx.x < ITEM_BLOCKS) { ITEM_LOOP_G(c.st->ni + c.st->no, ITEM_BLOCKS) vrg_item_decide(c, i, c.st->nfx != 0); return; }
    const uint32_t nfx = c.st->nfx;
    const uint32_t wid = ((blockIdx.x - ITEM_BLOCKS) * blockDim.x + threadIdx.x) >> 6, nw = (ITEM_BLOCKS * blockDim.x) >> 6;
    exact_wave(c, c.st->iter & 1, nfx, wid, nw, true);
}
// stop tests (:91-104) once all entries have decided, then per listed flip: 125 mark positions + prepass
__global__ void k_marks_prepass(VrgCtx c) {
    if (c.st->done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) c.st->nfx = 0;     // k_decide_exact has computed them
    int32_t stop = vrg_stop_test(c);
    if (stop || c.st->error) {
        if (blockIdx.x == 0 && threadIdx.x == 0) c.st->done = stop ? stop : -1;
        return;
    }
    ITEM_LOOP64((uint64_t)c.st->nf * 128u) {
        uint32_t r = (uint32_t)(i >> 7), p = (uint32_t)(i & 127u);
        if (p < 125u) vrg_item_scatter_marks(c, r, p);
        else if (p == 125u) vrg_item_prepass(c, r);
    }
}
// skip-rule fix-point as a kernel of its own (one workgroup): only the full-stencil check variant launches it
__global__ void k_fix(VrgCtx c) {
    if (c.st->done) return;
    __shared__ int changed;
    uint32_t np = c.st->npend;
    if (np == 0) return;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < np; j += blockDim.x)
            if (vrg_item_fix(c, j) == 2) changed = 1;
        __threadfence();
        __syncthreads();
        if (!changed) break;
    }
}
// Skip-rule fix-point first (rare: only when a flip-in dropped to 3 in phase A, npend > 0).  It is a monotone
// closure (P bits are only ever set) over facts the previous kernel left behind, so EVERY workgroup computes all
// of it by itself - the same bits, set with atomic ORs - instead of one workgroup in a kernel of its own that the
// common case (npend == 0) would pay a launch for.  A workgroup is done after a pass in which it neither applied
// anything NOR saw a bit that it had not seen the pass before (another workgroup may set an entry between this
// one's look at a dependent entry and its look at the entry itself; the per-thread count of set entries catches
// that).  Then the relabel stencil of the marked voxels.
__global__ void k_relabel(VrgCtx c) {
    if (c.st->done) return;
    const uint32_t np = c.st->npend;
    if (np) {
        __shared__ int changed;
        uint32_t seen_before = 0;
        for (;;) {
            __syncthreads();
            if (threadIdx.x == 0) changed = 0;
            __syncthreads();
            uint32_t seen = 0; bool mine = false;
            for (uint32_t j = threadIdx.x; j < np; j += blockDim.x) {
                const int r = vrg_item_fix(c, j);
                seen += r != 0; mine |= r == 2;
            }
            if (mine || seen != seen_before) changed = 1;
            seen_before = seen;
            __threadfence();
            __syncthreads();
            if (!changed) break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // the stencil below reads the P bits with plain loads
    }
    ITEM_LOOP(min(c.st->nmk, c.mcap)) vrg_item_relabel(c, i);
}
// k_apply + k_entry_post in one launch: the survivor test of an entry takes its voxel's new byte from the relabel
// (e_new) or, for a voxel the relabel did not visit, from the label it keeps - so it does not wait for the bytes
// being written.  First half of the grid: write the new label bytes (+ the class changes of the sweep before into
// this sweep's class copy, see VrgCtx::clsb); second half: per old band entry, survivor test and flip bookkeeping.
__global__ void k_apply_entry_post(VrgCtx c) {
    if (c.st->done) return;
    if (blockIdx.x < ITEM_BLOCKS) {
        if (blockIdx.x == 0 && threadIdx.x == 0) vrg_request_dense(c);
        const uint32_t nm = min(c.st->nmk, c.mcap);
        ITEM_LOOP_G(nm + vrg_catchup_count(c), ITEM_BLOCKS) { if (i < nm) vrg_item_apply(c, i); else vrg_item_catchup(c, i - nm); }
        return;
    }
    for (uint32_t i = (blockIdx.x - ITEM_BLOCKS) * blockDim.x + threadIdx.x, n = c.st->ni + c.st->no; i < n; i += ITEM_BLOCKS * blockDim.x)
        vrg_item_entry_post(c, i);
}
__global__ void k_dense_fin(VrgCtx c) { vrg_dense_fin(c); }
__global__ void k_entry_post(VrgCtx c) {               // full-stencil check variant: after k_copy_back
    if (c.st->done) return;
    ITEM_LOOP(c.st->ni + c.st->no) vrg_item_entry_post(c, i);
}
__global__ void k_scatter(VrgCtx c) {                  // items: every old entry, then (listed flip, neighbour k)
    if (c.st->done) return;
    const uint32_t n = c.st->ni + c.st->no;
    ITEM_LOOP64((uint64_t)n + (uint64_t)c.st->nf * 32u) {
        if (i < n) vrg_item_scatter_entry(c, (uint32_t)i);
        else { uint64_t j = i - n; vrg_item_scatter_promo(c, (uint32_t)(j >> 5), (uint32_t)(j & 31u)); }
    }
}

// level-delta compaction (:232-235 regrouped by distinct intensity value)
__global__ void k_delta_flag(VrgCtx c) {
    if (c.st->done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) vrg_post_apply(c);
    const uint32_t off = vrg_delta_off(c);
    ITEM_LOOP(c.L) c.lscan[i] = (c.dIn[off + i] | c.dOut[off + i] | c.dConv[off + i]) ? 1u : 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) c.st->nscan = c.L;
}
__global__ void k_delta_scatter(VrgCtx c) {
    if (c.st->done) return;
    const uint32_t off = vrg_delta_off(c);
    ITEM_LOOP(c.L) {
        uint32_t a = c.dIn[off + i], b = c.dOut[off + i], d = c.dConv[off + i];
        if (a | b | d) {
            uint32_t j = c.lscan[i];
            c.nz_lev[j] = i; c.nz_val[j] = c.lev[i]; c.nz_cin[j] = a; c.nz_cout[j] = b; c.nz_cconv[j] = d;
            c.hout[i] += (int32_t)d;                 // included voxels join the outer region
            c.dIn[off + i] = 0; c.dOut[off + i] = 0; c.dConv[off + i] = 0;
        }
    }
}
__global__ void k_post_prep(VrgCtx c) {               // after the level scan, before the rebuild scan
    if (c.st->done) return;
    VrgState& s = *c.st;
    uint32_t n = s.ni + s.no;
    s.nnz = s.scan_total;
    s.use_tab = c.L <= n;
    s.ncnt = 3 * n;
    s.nscan = s.ncnt;
}
// the four kernels above in ONE workgroup when the level table is small (the common, quantised case):
// ordered compaction of the touched levels by tiles of 1024 + the k_post_prep bookkeeping
constexpr uint32_t LEVELS_ONEBLOCK = 32768;
__global__ void __launch_bounds__(1024) k_levels_small(VrgCtx c) {
    if (c.st->done) return;
    if (threadIdx.x == 0) vrg_post_apply(c);
    __shared__ uint32_t sh[16];
    __shared__ uint32_t sh_run;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t off = vrg_delta_off(c);
    if (threadIdx.x == 0) sh_run = 0;
    __syncthreads();
    for (uint32_t base = 0; base < c.L; base += 1024) {
        uint32_t l = base + threadIdx.x;
        uint32_t a = 0, b = 0, d = 0;
        if (l < c.L) { a = c.dIn[off + l]; b = c.dOut[off + l]; d = c.dConv[off + l]; }
        uint32_t f = (a | b | d) ? 1u : 0u;
        uint32_t inc = wave_incl_scan(f);
        if (lane == 63) sh[w] = inc;
        __syncthreads();
        uint32_t pos = sh_run, tot = 0;
        for (int i = 0; i < 16; i++) { if (i < w) pos += sh[i]; tot += sh[i]; }
        if (f) {
            uint32_t j = pos + inc - 1;
            c.nz_lev[j] = l; c.nz_val[j] = c.lev[l]; c.nz_cin[j] = a; c.nz_cout[j] = b; c.nz_cconv[j] = d;
            c.hout[l] += (int32_t)d;                 // included voxels join the outer region
            c.dIn[off + l] = 0; c.dOut[off + l] = 0; c.dConv[off + l] = 0;
        }
        __syncthreads();
        if (threadIdx.x == 0) sh_run += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        VrgState& s = *c.st;
        uint32_t n = s.ni + s.no;
        s.nnz = sh_run;
        s.use_tab = c.L <= n;
        s.ncnt = 3 * n;
        s.nscan = s.ncnt;
    }
}
// per-level memo of the three density corrections: one wave per level
__global__ void k_tab(VrgCtx c) {
    if (c.st->done || !c.st->use_tab) return;
    uint32_t nnz = c.st->nnz;
    int lane = threadIdx.x & 63;
    uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t l = wid; l < c.L; l += nw) {
        double v = c.lev[l], a = 0, b = 0, d = 0;
        for (uint32_t i = lane; i < nnz; i += 64) {
            double k = vrg_kern(c, c.nz_val[i] - v);
            a += (double)c.nz_cin[i] * k; b += (double)c.nz_cout[i] * k; d += (double)c.nz_cconv[i] * k;
        }
        a = wave_sum(a); b = wave_sum(b); d = wave_sum(d);
        if (lane == 0) { c.tabC[3 * (size_t)l] = a; c.tabC[3 * (size_t)l + 1] = b; c.tabC[3 * (size_t)l + 2] = d; }
    }
}
// exact densities (:152-155, :252-255): one wave per fresh entry, lanes stride over the levels.  The level table
// (the same for every entry) is fetched first, four levels per lane at a time, so that it travels together with
// the entry's own look-ups instead of behind them.
__device__ void exact_wave(const VrgCtx& c, int par, uint32_t nfresh, uint32_t wid, uint32_t nw, bool then_decide) {
    const int lane = threadIdx.x & 63;
    if (wid >= nfresh) return;
    int32_t ha[4], hb[4]; double lv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t l = lane + 64u * q;
        bool in = l < c.L;
        ha[q] = in ? c.hin[l] : 0; hb[q] = in ? c.hout[l] : 0; lv[q] = in ? c.lev[l] : 0.0;
    }
    for (uint32_t f = wid; f < nfresh; f += nw) {
        uint32_t pos = c.fresh[f];
        double v = c.lev[c.b_lev[par][pos]], si = 0, so = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (!(ha[q] | hb[q])) continue;
            double k = vrg_kern(c, lv[q] - v);
            si += (double)ha[q] * k; so += (double)hb[q] * k;
        }
        for (uint32_t l = lane + 256u; l < c.L; l += 64) {
            int32_t a = c.hin[l], b = c.hout[l];
            if (!(a | b)) continue;
            double k = vrg_kern(c, c.lev[l] - v);
            si += (double)a * k; so += (double)b * k;
        }
        si = wave_sum(si); so = wave_sum(so);
        if (lane == 0) {
            c.b_ip[par][pos] = si; c.b_op[par][pos] = so;
            if (then_decide) {
                const VrgState s = *c.st;
                if (s.iter < s.iterMax) vrg_decide_core(c, s, pos, si, so);   // while iterNum <= iterMax (:58)
            }
        }
    }
}
__global__ void k_exact(VrgCtx c) {                   // init mode (:152-155): every band entry
    const uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    exact_wave(c, 0, c.st->nfresh, wid, nw, false);
}
__global__ void k_finalize(VrgCtx c) {                // iterNum += 1 (:117) + trace record
    VrgState s = *c.st;                               // one round trip for the whole state, one to write it back
    if (s.done) return;
    const int64_t n_in = c.inc[VC_NIN], n_out = c.inc[VC_NOUT];
    s.ni = s.ni_new; s.no = s.nb_new - s.ni_new; s.iter++;
    if ((uint32_t)s.iter < c.trace_cap) {
        VrgTrace& t = c.trace[s.iter];                // the intensity sums are filed by the dense pass (vrg_dense_fin)
        t.nflip = s.nf; t.nseg = n_in; t.n_in = n_in; t.n_out = n_out; t.ni = s.ni; t.no = s.no;
    }
    s.nf = 0; s.npend = 0; s.nmk = 0;
    s.nfx = s.nfresh; s.nfresh = 0;                   // exact densities of the new entries: first thing next trip
    if (s.error) s.done = -1;
    *c.st = s;
}

// ---- device-wide exclusive scan of c-array `a` (length st->nscan), total -> st->scan_total ----------
__device__ __forceinline__ void scan_range(uint32_t n, uint32_t& lo, uint32_t& hi) {
    uint32_t chunk = (n + SCAN_BLOCKS - 1) / SCAN_BLOCKS;
    chunk = (chunk + TPB - 1) / TPB * TPB;
    lo = min(n, blockIdx.x * chunk); hi = min(n, lo + chunk);
}
__global__ void k_scan_reduce(VrgCtx c, uint32_t* a) {
    if (c.st->done) return;
    __shared__ uint32_t sh[4];
    uint32_t lo, hi; scan_range(c.st->nscan, lo, hi);
    uint32_t s = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += TPB) s += a[i];
    uint32_t tot; block_excl_scan(s, tot, sh);
    if (threadIdx.x == 0) c.bsum[blockIdx.x] = tot;
}
// second pass: every workgroup adds up the partials before its own (256 values: one per thread), scans its chunk;
// fin != 0: this is the rebuild scan - also derive the new list lengths (ni_new = scan value at the start of
// segment B0 = number of entries of the new inner list)
__global__ void k_scan_down(VrgCtx c, uint32_t* a, int fin) {
    if (c.st->done) return;
    __shared__ uint32_t sh[4];
    __shared__ uint32_t sh_off;
    VrgState& s = *c.st;
    const uint32_t n = s.nscan;
    uint32_t lo, hi; scan_range(n, lo, hi);
    uint32_t gtot, gex = block_excl_scan(c.bsum[threadIdx.x], gtot, sh);
    if (threadIdx.x == blockIdx.x) sh_off = gex;
    __syncthreads();
    uint32_t run = sh_off;
    const uint32_t b0 = fin ? vrg_slot_B0(s, 0) : 0xffffffffu;
    for (uint32_t base = lo; base < hi; base += TPB) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < hi ? a[i] : 0, tot;
        uint32_t ex = block_excl_scan(v, tot, sh);
        if (i < hi) {
            a[i] = run + ex;
            if (i == b0) s.ni_new = run + ex;
        }
        run += tot;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        s.scan_total = gtot;
        if (fin) {
            if (b0 >= n) s.ni_new = gtot;
            s.nb_new = gtot;
            if (gtot > c.bcap) { s.error = 1; s.done = -1; }
        }
    }
}
void device_scan(const VrgCtx& c, uint32_t* a, hipStream_t st, int fin) {
    static_assert(SCAN_BLOCKS == TPB, "k_scan_down adds up one partial per thread");
    k_scan_reduce<<<SCAN_BLOCKS, TPB, 0, st>>>(c, a);
    k_scan_down<<<SCAN_BLOCKS, TPB, 0, st>>>(c, a, fin);
}

// k_levels_small + k_tab + the first pass of the rebuild scan in ONE launch, for level tables up to LT_MAX:
//  * workgroups [0, ITEM_BLOCKS): each compacts the touched levels into LDS for itself (same ordered compaction,
//    a few hundred levels), then its waves fill their share of the per-level correction memo from LDS; workgroup 0
//    also publishes the compacted list, folds the included voxels into the outer histogram, sets the bookkeeping
//    scalars and clears the delta counters of the OTHER parity (this sweep's are still being read by the rest);
//  * workgroups [ITEM_BLOCKS, ITEM_BLOCKS + SCAN_BLOCKS): k_scan_reduce of the rebuild count array (3 n entries).
// Two dependent launches fewer on the band chain.
constexpr uint32_t LT_MAX = 2048;
__global__ void __launch_bounds__(TPB) k_levels_tab_scan(VrgCtx c) {
    if (c.st->done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) vrg_post_apply(c);     // first kernel after the labels are applied
    const uint32_t n = c.st->ni + c.st->no;
    __shared__ uint32_t sh[4];
    if (blockIdx.x >= ITEM_BLOCKS) {
        const uint32_t b = blockIdx.x - ITEM_BLOCKS, n3 = 3u * n;
        uint32_t chunk = (n3 + SCAN_BLOCKS - 1) / SCAN_BLOCKS;
        chunk = (chunk + TPB - 1) / TPB * TPB;
        const uint32_t lo = min(n3, b * chunk), hi = min(n3, lo + chunk);
        uint32_t sum = 0;
        for (uint32_t i = lo + threadIdx.x; i < hi; i += TPB) sum += c.scan[i];
        uint32_t tot; block_excl_scan(sum, tot, sh);
        if (threadIdx.x == 0) c.bsum[b] = tot;
        return;
    }
    __shared__ double s_val[LT_MAX];
    __shared__ uint32_t s_lev[LT_MAX], s_cin[LT_MAX], s_cout[LT_MAX], s_cconv[LT_MAX];
    __shared__ uint32_t sh_run;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t off = vrg_delta_off(c);
    if (threadIdx.x == 0) sh_run = 0;
    __syncthreads();
    for (uint32_t base = 0; base < c.L; base += TPB) {
        const uint32_t l = base + threadIdx.x;
        uint32_t a = 0, bb = 0, d = 0;
        if (l < c.L) { a = c.dIn[off + l]; bb = c.dOut[off + l]; d = c.dConv[off + l]; }
        const uint32_t f = (a | bb | d) ? 1u : 0u;
        const uint32_t inc = wave_incl_scan(f);
        if (lane == 63) sh[w] = inc;
        __syncthreads();
        uint32_t pos = sh_run, tot = 0;
        for (int i = 0; i < 4; i++) { if (i < w) pos += sh[i]; tot += sh[i]; }
        if (f) {
            const uint32_t j = pos + inc - 1;
            s_lev[j] = l; s_val[j] = c.lev[l]; s_cin[j] = a; s_cout[j] = bb; s_cconv[j] = d;
        }
        __syncthreads();
        if (threadIdx.x == 0) sh_run += tot;
        __syncthreads();
    }
    const uint32_t nnz = sh_run;
    const bool use_tab = c.L <= n;
    if (blockIdx.x == 0) {
        for (uint32_t j = threadIdx.x; j < nnz; j += TPB) {
            const uint32_t l = s_lev[j];
            c.nz_lev[j] = l; c.nz_val[j] = s_val[j]; c.nz_cin[j] = s_cin[j]; c.nz_cout[j] = s_cout[j]; c.nz_cconv[j] = s_cconv[j];
            c.hout[l] += (int32_t)s_cconv[j];        // included voxels join the outer region
        }
        const uint32_t other = off ? 0u : c.L;       // next sweep's counters: nobody touches them during this kernel
        for (uint32_t l = threadIdx.x; l < c.L; l += TPB) { c.dIn[other + l] = 0; c.dOut[other + l] = 0; c.dConv[other + l] = 0; }
        if (threadIdx.x == 0) {
            VrgState& s = *c.st;
            s.nnz = nnz; s.use_tab = use_tab; s.ncnt = 3 * n; s.nscan = 3 * n;
        }
    }
    if (!use_tab) return;
    const uint32_t wid = (blockIdx.x * TPB + threadIdx.x) >> 6, nw = (ITEM_BLOCKS * TPB) >> 6;
    for (uint32_t l = wid; l < c.L; l += nw) {
        double v = c.lev[l], a = 0, b = 0, d = 0;
        for (uint32_t i = lane; i < nnz; i += 64) {
            double k = vrg_kern(c, s_val[i] - v);
            a += (double)s_cin[i] * k; b += (double)s_cout[i] * k; d += (double)s_cconv[i] * k;
        }
        a = wave_sum(a); b = wave_sum(b); d = wave_sum(d);
        if (lane == 0) { c.tabC[3 * (size_t)l] = a; c.tabC[3 * (size_t)l + 1] = b; c.tabC[3 * (size_t)l + 2] = d; }
    }
}

// ---- the dense pass ----------------------------------------------------------------------------------
// Region recount (:113-116 innerSize/outerSize, :249-250 dataArray[mask]) over every voxel, every sweep.
// Streams the interior planes of the padded volume in units of 1024 voxels (k_recount_bits below).  Sums are
// reduced lane -> wave butterfly -> LDS -> one slot per workgroup, added in fixed slot order by the last workgroup
// to finish: bit-reproducible.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));

struct SweepAcc { long long nin, nout; double sin_, sout; };

// per-workgroup slot, then the LAST workgroup to arrive adds all slots in slot order and publishes the
// totals (agent-scope release before the ticket, acquire after it: cdna guide, Guideline 16)
__device__ __forceinline__ void sweep_finish(const VrgCtx& c, SweepAcc a, int fin) {
    __shared__ long long sh_n[2][4];
    __shared__ double sh_s[2][4];
    __shared__ int is_last;
    a.nin = wave_sum(a.nin); a.nout = wave_sum(a.nout); a.sin_ = wave_sum(a.sin_); a.sout = wave_sum(a.sout);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh_n[0][wv] = a.nin; sh_n[1][wv] = a.nout; sh_s[0][wv] = a.sin_; sh_s[1][wv] = a.sout; }
    __syncthreads();
    if (threadIdx.x == 0) {
        c.st_nin[blockIdx.x] = sh_n[0][0] + sh_n[0][1] + sh_n[0][2] + sh_n[0][3];
        c.st_nout[blockIdx.x] = sh_n[1][0] + sh_n[1][1] + sh_n[1][2] + sh_n[1][3];
        c.st_sin[blockIdx.x] = ((sh_s[0][0] + sh_s[0][1]) + sh_s[0][2]) + sh_s[0][3];
        c.st_sout[blockIdx.x] = ((sh_s[1][0] + sh_s[1][1]) + sh_s[1][2]) + sh_s[1][3];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t t = __hip_atomic_fetch_add(&c.counters[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == gridDim.x - 1);
        if (is_last) {
            c.counters[0] = 0;                       // every workgroup has arrived: reset for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!is_last) return;
    long long x = 0, y = 0; double sx = 0, sy = 0;
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += TPB) { x += c.st_nin[i]; y += c.st_nout[i]; sx += c.st_sin[i]; sy += c.st_sout[i]; }
    x = wave_sum(x); y = wave_sum(y); sx = wave_sum(sx); sy = wave_sum(sy);
    __syncthreads();
    if (lane == 0) { sh_n[0][wv] = x; sh_n[1][wv] = y; sh_s[0][wv] = sx; sh_s[1][wv] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        VrgDense d;
        d.n_in = (double)(sh_n[0][0] + sh_n[0][1] + sh_n[0][2] + sh_n[0][3]);
        d.n_out = (double)(sh_n[1][0] + sh_n[1][1] + sh_n[1][2] + sh_n[1][3]);
        d.sum_in = ((sh_s[0][0] + sh_s[0][1]) + sh_s[0][2]) + sh_s[0][3];
        d.sum_out = ((sh_s[1][0] + sh_s[1][1]) + sh_s[1][2]) + sh_s[1][3];
        *c.dn_part = d;                              // slab partials: input of the all-reduce
        if (c.world == 1) *c.dn = d;
        if (fin == 2) vrg_dense_fin(c);              // nothing to sum over ranks: close the pass here
    }
}

constexpr uint32_t LEV16_MAX = 16384;   // 16-bit storage: the level values sit in LDS (<= 16384 x f32)

// The recount needs two facts per voxel - inner / outer - so it streams the
// 2-bit class volume (VrgCtx::clsb, 0.25 B/voxel) instead of the label bytes: 4.25 B (fp32 storage) or 2.25 B
// (16-bit storage) per voxel.  Units are 1024-voxel aligned in the absolute voxel index; lane l owns class dword l
// of a unit and the 4 x 4 intensities at 256*j + 4*l, i.e. one 256-B + four 1-KiB (or 512-B) requests per wave
// and unit.  A slab edge that cuts a unit is handled by masking (first / last wave); padding planes are class 0.
// UNITS = units a wave loads per trip (bytes in flight); NT = non-temporal loads: the volume is read once per
// sweep and is far larger than the 256-MiB Infinity Cache, so nothing is worth keeping.
// Dense pass number seq (= passes closed + 1) reads copy seq & 1 of the class bits.
__device__ __forceinline__ void stats_bits(SweepAcc& a, uint32_t w, const f4v* f) {
    a.nin += __popc(w & 0x55555555u); a.nout += __popc(w & 0xAAAAAAAAu);
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint32_t t = w >> (2 * (4 * j + b));
            double x = (double)f[j][b];
            a.sin_ += (t & 1u) ? x : 0.0;
            a.sout += (t & 2u) ? x : 0.0;
        }
}
template <bool L16, bool NT>
__device__ __forceinline__ void load_unit(const VrgCtx& c, const uint32_t* cls, const float* s_val, uint32_t u, uint32_t lane, uint32_t& w, f4v* f) {
    const uint32_t* pc = cls + ((size_t)u << 6) + lane;
    w = NT ? __builtin_nontemporal_load(pc) : *pc;
    const uint32_t base = (u << 10) + (lane << 2);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (L16) {
            const u2v* pq = reinterpret_cast<const u2v*>(c.lev16 + base + (j << 8));
            u2v q = NT ? __builtin_nontemporal_load(pq) : *pq;
            f[j] = f4v{s_val[q.x & 0xffffu], s_val[q.x >> 16], s_val[q.y & 0xffffu], s_val[q.y >> 16]};
        } else {
            const f4v* pi = reinterpret_cast<const f4v*>(c.I + base + (j << 8));
            f[j] = NT ? __builtin_nontemporal_load(pi) : *pi;
        }
    }
}
template <int UNITS, bool NT, bool L16>
__global__ void __launch_bounds__(TPB) k_recount_bits(VrgCtx c, int check_done) {
    if (check_done && !vrg_dense_due(c)) return;     // no sweep was applied since the last pass (stop flag)
    __shared__ float s_val[L16 ? LEV16_MAX : 1];
    if (L16) {
        for (uint32_t i = threadIdx.x; i < c.L; i += TPB) s_val[i] = (float)c.lev[i];
        __syncthreads();
    }
    const uint32_t* __restrict__ cls = c.clsb[(c.dctl[VD_SEQ] + 1) & 1];
    const uint32_t plane = (uint32_t)c.PY * (uint32_t)c.PX;
    const uint32_t lo = (2u + (uint32_t)c.z0) * plane;         // this device's Z-slab [z0, z1) as a voxel range
    const uint32_t hi = (2u + (uint32_t)c.z1) * plane;
    uint32_t f_lo = (lo + 1023u) >> 10, f_hi = hi >> 10;       // units wholly inside it
    if (f_hi < f_lo) f_hi = f_lo;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    SweepAcc acc = {0, 0, 0.0, 0.0};
    uint32_t u = f_lo + wave * UNITS;
    for (; u + UNITS <= f_hi; u += nwaves * UNITS) {
        uint32_t w[UNITS]; f4v f[UNITS][4];
#pragma unroll
        for (int q = 0; q < UNITS; q++) load_unit<L16, NT>(c, cls, s_val, u + q, lane, w[q], f[q]);
#pragma unroll
        for (int q = 0; q < UNITS; q++) stats_bits(acc, w[q], f[q]);
    }
    for (; u < f_hi; u++) {                                    // whole units left over by the UNITS-stride
        uint32_t w; f4v f[4];
        load_unit<L16, false>(c, cls, s_val, u, lane, w, f);
        stats_bits(acc, w, f);
    }
    // units the slab edges cut: the first and the last unit touching [lo, hi), masked to the slab
    const uint32_t e0 = lo >> 10, e1 = (hi - 1u) >> 10;
    const uint32_t edge = wave == 0 ? e0 : (wave == nwaves - 1 && e1 != e0 ? e1 : 0xffffffffu);
    if (edge != 0xffffffffu && !(edge >= f_lo && edge < f_hi)) {
        uint32_t w; f4v f[4];
        load_unit<L16, false>(c, cls, s_val, edge, lane, w, f);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t v = (edge << 10) + (j << 8) + (lane << 2);        // groups of 4 voxels never straddle a plane
            if (v < lo || v >= hi) w &= ~(0xffu << (8 * j));
        }
        stats_bits(acc, w, f);
    }
    sweep_finish(c, acc, check_done);
}
__global__ void k_cls_build(VrgCtx c) {
    const uint32_t nd = ((c.PV + 1023u) >> 10) << 6;
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < nd; d += gridDim.x * blockDim.x) vrg_item_cls_build(c, d);
}

// full-stencil check variant: every voxel runs the relabel stencil (no marks); new bytes go to lab[1]
// and are copied back, so stencil reads only ever see pre-sweep labels.
__global__ void __launch_bounds__(TPB) k_full_relabel(VrgCtx c) {
    if (c.st->done) return;
    const uint8_t* __restrict__ in = c.lab[0];
    uint8_t* __restrict__ out = c.lab[1];
    const uint32_t plane = (uint32_t)c.PY * (uint32_t)c.PX;
    const uint32_t first = 2u * plane;
    const uint32_t ndw = (uint32_t)(((uint64_t)c.nz * plane) >> 2);
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < ndw; d += gridDim.x * blockDim.x) {
        const uint32_t base = first + (d << 2);
        uint32_t v = *reinterpret_cast<const uint32_t*>(in + base);
        if ((v & 0x20202020u) != 0x20202020u)
            for (int b = 0; b < 4; b++) {
                uint8_t cb = (uint8_t)(v >> (8 * b));
                if (!(cb & VB_OOB)) {
                    uint8_t nb = vrg_sweep_core(c, in, base + b, cb);
                    if (cb & VB_B) c.e_new[c.vent[base + b]] = (uint8_t)(nb | VE_VALID);
                    v = (v & ~(0xffu << (8 * b))) | ((uint32_t)nb << (8 * b));
                }
            }
        *reinterpret_cast<uint32_t*>(out + base) = v;
    }
}
__global__ void __launch_bounds__(TPB) k_copy_back(VrgCtx c) {
    if (c.st->done) return;
    const uint32_t plane = (uint32_t)c.PY * (uint32_t)c.PX;
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(c.lab[1] + 2u * plane);
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(c.lab[0] + 2u * plane);
    const uint32_t n16 = (uint32_t)(((uint64_t)c.nz * plane) >> 4);
    if (blockIdx.x == 0 && threadIdx.x == 0) vrg_request_dense(c);
    ITEM_LOOP(vrg_catchup_count(c)) vrg_item_catchup(c, i);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) {
        uint4 a = src[i], b = dst[i];
        if (a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w) {
            const uint32_t nw[4] = {a.x, a.y, a.z, a.w}, od[4] = {b.x, b.y, b.z, b.w};
            for (int k = 0; k < 16; k++) vrg_count_change(c, 2u * plane + 16u * i + (uint32_t)k, (uint8_t)(od[k >> 2] >> (8 * (k & 3))), (uint8_t)(nw[k >> 2] >> (8 * (k & 3))));
            a.x &= ~0x40404040u; a.y &= ~0x40404040u; a.z &= ~0x40404040u; a.w &= ~0x40404040u;   // F: see vrg_item_apply
            dst[i] = a;
        }
    }
}

// ---- dense helpers over the real voxels -------------------------------------------------------------
__device__ __forceinline__ uint32_t real_idx(const VrgCtx& c, uint64_t t, int& x, int& y, int& z) {
    x = (int)(t % (uint64_t)c.nx); uint64_t r = t / (uint64_t)c.nx;
    y = (int)(r % (uint64_t)c.ny); z = (int)(r / (uint64_t)c.ny);
    return vrg_idx(c, x, y, z);
}
#define VOXEL_LOOP(c) \
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nV_ = (uint64_t)(c).nx * (c).ny * (c).nz; \
         t < nV_; t += (uint64_t)gridDim.x * blockDim.x)

__global__ void k_init_voxel(VrgCtx c) {
    VOXEL_LOOP(c) { int x, y, z; vrg_item_init_voxel(c, real_idx(c, t, x, y, z)); }
}
__global__ void k_hist_voxel(VrgCtx c) {
    VOXEL_LOOP(c) { int x, y, z; vrg_item_hist_voxel(c, real_idx(c, t, x, y, z)); }
}
// same, for level tables that fit LDS: per-workgroup private histograms (the level values too), streamed
// over the padded interior 16 bytes per lane, flushed with one global atomic per non-zero bin.
constexpr uint32_t HIST_LDS_LEVELS = 4096;
__global__ void __launch_bounds__(TPB) k_hist_lds(VrgCtx c) {
    __shared__ float s_lev[HIST_LDS_LEVELS];
    __shared__ uint32_t s_h[2][HIST_LDS_LEVELS];
    const uint32_t L = c.L;
    for (uint32_t i = threadIdx.x; i < L; i += TPB) { s_lev[i] = (float)c.lev[i]; s_h[0][i] = 0; s_h[1][i] = 0; }
    __syncthreads();
    const uint8_t* __restrict__ in = c.lab[0];
    const uint32_t plane = (uint32_t)c.PY * (uint32_t)c.PX, first = 2u * plane;
    const uint32_t ndw = (uint32_t)(((uint64_t)c.nz * plane) >> 2);
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < ndw; d += gridDim.x * blockDim.x) {
        const uint32_t base = first + (d << 2);
        uint32_t v = *reinterpret_cast<const uint32_t*>(in + base);
        if ((v & 0x24242424u) == 0x24242424u) continue;          // all four excluded or padding
        const float4 f = *reinterpret_cast<const float4*>(c.I + base);
        const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint8_t cb = (uint8_t)(v >> (8 * b));
            if (cb & (VB_OOB | VB_X)) continue;
            uint32_t lo = 0, hi = L - 1;
            if (c.lev16) lo = c.lev16[base + b];
            else while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (s_lev[m] < fv[b]) lo = m + 1; else hi = m; }
            atomicAdd(&s_h[(cb & VB_S) ? 0 : 1][lo], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < L; i += TPB) {
        if (s_h[0][i]) atomicAdd(&c.hin[i], (int32_t)s_h[0][i]);
        if (s_h[1][i]) atomicAdd(&c.hout[i], (int32_t)s_h[1][i]);
    }
}
__global__ void k_init_entry(VrgCtx c) {
    ITEM_LOOP(c.st->ni + c.st->no) vrg_item_init_entry(c, i);
}
__global__ void k_fin_init(VrgCtx c) {
    VrgState& s = *c.st;
    s.nfresh = 0; s.nf = 0; s.npend = 0; s.nmk = 0;
    vrg_init_counts(c);
    const VrgDense& d = *c.dn;
    VrgTrace& t = c.trace[0];
    t.nflip = 0; t.nseg = (int64_t)d.n_in; t.n_in = (int64_t)d.n_in; t.n_out = (int64_t)d.n_out; t.ni = s.ni; t.no = s.no;
    t.sum_in = d.sum_in; t.sum_out = d.sum_out;
}
__global__ void k_recount_hist(VrgCtx c, int32_t* rin, int32_t* rout) {
    VOXEL_LOOP(c) {
        int x, y, z; uint32_t idx = real_idx(c, t, x, y, z);
        uint8_t b = c.lab[0][idx];
        if (b & VB_X) continue;
        uint32_t lev = vrg_level_of(c, (double)c.I[idx]);
        atomicAdd((b & VB_S) ? &rin[lev] : &rout[lev], 1);
    }
}
__global__ void k_collect_seg(VrgCtx c, uint64_t* stamps, uint32_t* idxs, uint32_t cap, uint32_t* count) {
    VOXEL_LOOP(c) {
        int x, y, z; uint32_t idx = real_idx(c, t, x, y, z);
        if (c.lab[0][idx] & VB_S) {
            uint32_t p = atomicAdd(count, 1u);
            if (p < cap) { stamps[p] = c.stamp[idx]; idxs[p] = idx; }
        }
    }
}
__global__ void k_gather_I(VrgCtx c, float* dst) {
    VOXEL_LOOP(c) { int x, y, z; dst[t] = c.I[real_idx(c, t, x, y, z)]; }
}
__global__ void k_f2d(const float* a, double* b, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) b[i] = (double)a[i];
}

// ---- repacking caller arrays ------------------------------------------------------------------------
__device__ __forceinline__ double load_as_double(const void* p, int dtype, int64_t i) {
    switch (dtype) {
        case 0: return ((const uint8_t*)p)[i];
        case 1: return ((const int16_t*)p)[i];
        case 2: return ((const uint16_t*)p)[i];
        case 3: return ((const int32_t*)p)[i];
        case 4: return (double)((const int64_t*)p)[i];
        case 5: return ((const float*)p)[i];
        default: return ((const double*)p)[i];
    }
}
__device__ __forceinline__ void store_int(void* p, int dtype, int64_t i, int v) {
    switch (dtype) {
        case 0: ((uint8_t*)p)[i] = (uint8_t)v; break;
        case 1: ((int16_t*)p)[i] = (int16_t)v; break;
        case 2: ((uint16_t*)p)[i] = (uint16_t)v; break;
        case 3: ((int32_t*)p)[i] = v; break;
        case 4: ((int64_t*)p)[i] = v; break;
        case 5: ((float*)p)[i] = (float)v; break;
        default: ((double*)p)[i] = v; break;
    }
}
__global__ void k_pack_volume(VrgCtx c, float* dst, const void* src, int dtype, int64_t s0, int64_t s1, int64_t s2, int* flag) {
    VOXEL_LOOP(c) {
        int x, y, z; uint32_t idx = real_idx(c, t, x, y, z);
        double v = load_as_double(src, dtype, x * s0 + y * s1 + z * s2);
        float f = (float)v;
        if ((double)f != v) *flag = 1;
        dst[idx] = f;
    }
}
__global__ void k_pack_labels(VrgCtx c, uint8_t* dst, const void* src, int dtype, int64_t s0, int64_t s1, int64_t s2, int* flag) {
    VOXEL_LOOP(c) {
        int x, y, z; uint32_t idx = real_idx(c, t, x, y, z);
        double v = load_as_double(src, dtype, x * s0 + y * s1 + z * s2);
        uint8_t b = 0;
        if (v == 0) b = VB_S; else if (v == 3) b = 0; else if (v == 4) b = VB_X; else *flag = 1;
        dst[idx] = b;
    }
}
__global__ void k_unpack_labels(VrgCtx c, const uint8_t* lab, void* dst, int dtype, int64_t s0, int64_t s1, int64_t s2) {
    VOXEL_LOOP(c) {
        int x, y, z; uint32_t idx = real_idx(c, t, x, y, z);
        store_int(dst, dtype, x * s0 + y * s1 + z * s2, vrg_dec(lab[idx]));
    }
}

const size_t kElem[7] = {1, 2, 2, 4, 8, 4, 8};

// strides must describe a dense permutation of the three axes (numpy C or F order)
bool dense_strides(const VrgCtx& c, const int64_t st[3]) {
    int64_t dim[3] = {c.nx, c.ny, c.nz};
    int o[3] = {0, 1, 2};
    for (int i = 0; i < 3; i++) for (int j = i + 1; j < 3; j++) if (st[o[j]] < st[o[i]]) { int t = o[i]; o[i] = o[j]; o[j] = t; }
    int64_t expect = 1;
    for (int i = 0; i < 3; i++) {
        if (dim[o[i]] == 1) continue;                 // stride of a length-1 axis is irrelevant
        if (st[o[i]] != expect) return false;
        expect *= dim[o[i]];
    }
    return true;
}
bool is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}
int voxel_blocks(const VrgCtx& c) {
    uint64_t V = (uint64_t)c.nx * c.ny * c.nz;
    return (int)std::min<uint64_t>(4096, (V + TPB - 1) / TPB);
}

// workgroups of the dense recount: >= 32 one-KiB units per wave, at most 1 workgroup per CU
int dense_blocks(const VrgCtx& c) {
    if (g_sweep_blocks > 0) return g_sweep_blocks;
    uint64_t units = ((uint64_t)(c.z1 - c.z0) * c.PY * c.PX) >> 10;
    // the 16-bit variant spends issue slots on LDS table look-ups and wants twice the waves (512: 0.275 ms, 256: 0.335)
    return (int)std::min<uint64_t>(c.lev16 ? 2 * SWEEP_BLOCKS : SWEEP_BLOCKS, std::max<uint64_t>(64, units / 128));
}

struct EvPair { hipEvent_t a, b; };
std::vector<EvPair> g_ev_pool;
size_t g_ev_used = 0;

}  // namespace

// ---- backend interface ---------------------------------------------------------------------------------
static void make_streams() {
    int lo = 0, hi = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));      // hi = numerically lowest = highest priority
    if (g_stream) { HIP_CHECK(hipStreamSynchronize(g_stream)); HIP_CHECK(hipStreamDestroy(g_stream)); }
    if (g_stream_b) { HIP_CHECK(hipStreamSynchronize(g_stream_b)); HIP_CHECK(hipStreamDestroy(g_stream_b)); }
    // prio_mode 0: equal; 1: critical stream A high; 2: bookkeeping stream B high
    HIP_CHECK(hipStreamCreateWithPriority(&g_stream, hipStreamNonBlocking, g_prio_mode == 1 ? hi : (g_prio_mode == 2 ? lo : 0)));
    HIP_CHECK(hipStreamCreateWithPriority(&g_stream_b, hipStreamNonBlocking, g_prio_mode == 2 ? hi : (g_prio_mode == 1 ? lo : 0)));
}
void be_set_tuning(const char* name, long long v) {
    if (std::strcmp(name, "sweep_blocks") == 0 && v >= 0 && v <= 4096) g_sweep_blocks = (int)v;
    if (std::strcmp(name, "graph") == 0) g_use_graph_req = v != 0;
    if (std::strcmp(name, "prio_mode") == 0 && v >= 0 && v <= 2 && v != g_prio_mode) { g_prio_mode = (int)v; make_streams(); }
}

int be_set_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) { (void)hipGetLastError(); return -1; }
    if (hipSetDevice(device) != hipSuccess) return -1;
    if (!g_stream) {
        make_streams();
        HIP_CHECK(hipEventCreateWithFlags(&g_ev_a, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&g_ev_d[0], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&g_ev_d[1], hipEventDisableTiming));
    }
    return 0;
}
void* be_alloc(size_t bytes) { void* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; } return p; }
void be_free(void* p) { HIP_CHECK(hipFree(p)); }
void be_fill(void* p, int byte, size_t bytes) { HIP_CHECK(hipMemsetAsync(p, byte, bytes, g_stream)); }
void be_upload(void* dst, const void* src, size_t bytes) { HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, g_stream)); HIP_CHECK(hipStreamSynchronize(g_stream)); }
void be_download(void* dst, const void* src, size_t bytes) { HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, g_stream)); HIP_CHECK(hipStreamSynchronize(g_stream)); }
const char* be_last_error() {
    if (!g_hip_error[0]) { hipError_t e = hipGetLastError(); if (e != hipSuccess) std::snprintf(g_hip_error, sizeof(g_hip_error), "HIP error '%s' (asynchronous)", hipGetErrorString(e)); }
    return g_hip_error[0] ? g_hip_error : nullptr;
}
void be_sync() { HIP_CHECK(hipStreamSynchronize(g_stream)); HIP_CHECK(hipStreamSynchronize(g_stream_b)); }

static const void* stage_in(const VrgCtx& c, const void* src, int dtype, void** tmp) {
    *tmp = nullptr;
    if (is_device_ptr(src)) return src;
    size_t bytes = (size_t)c.nx * c.ny * c.nz * kElem[dtype];
    if (hipMalloc(tmp, bytes) != hipSuccess) return nullptr;
    HIP_CHECK(hipMemcpyAsync(*tmp, src, bytes, hipMemcpyHostToDevice, g_stream));
    return *tmp;
}

int be_pack_volume(const VrgCtx& c, float* dst, const void* src, int dtype, const int64_t st[3], int* inexact) {
    if (!dense_strides(c, st)) return -1;
    void* tmp; const void* d = stage_in(c, src, dtype, &tmp);
    if (!d) return -1;
    int* flag; HIP_CHECK(hipMalloc(&flag, sizeof(int))); HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), g_stream));
    k_pack_volume<<<voxel_blocks(c), TPB, 0, g_stream>>>(c, dst, d, dtype, st[0], st[1], st[2], flag);
    HIP_CHECK(hipMemcpyAsync(inexact, flag, sizeof(int), hipMemcpyDeviceToHost, g_stream));
    HIP_CHECK(hipStreamSynchronize(g_stream));
    HIP_CHECK(hipFree(flag)); if (tmp) HIP_CHECK(hipFree(tmp));
    return 0;
}
int be_pack_labels(const VrgCtx& c, uint8_t* dst, const void* src, int dtype, const int64_t st[3], int* bad) {
    if (!dense_strides(c, st)) return -1;
    void* tmp; const void* d = stage_in(c, src, dtype, &tmp);
    if (!d) return -1;
    int* flag; HIP_CHECK(hipMalloc(&flag, sizeof(int))); HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), g_stream));
    k_pack_labels<<<voxel_blocks(c), TPB, 0, g_stream>>>(c, dst, d, dtype, st[0], st[1], st[2], flag);
    HIP_CHECK(hipMemcpyAsync(bad, flag, sizeof(int), hipMemcpyDeviceToHost, g_stream));
    HIP_CHECK(hipStreamSynchronize(g_stream));
    HIP_CHECK(hipFree(flag)); if (tmp) HIP_CHECK(hipFree(tmp));
    return 0;
}
int be_unpack_labels(const VrgCtx& c, const uint8_t* lab, void* dst, int dtype, const int64_t st[3]) {
    if (!dense_strides(c, st)) return -1;
    bool dev = is_device_ptr(dst);
    size_t bytes = (size_t)c.nx * c.ny * c.nz * kElem[dtype];
    void* d = dst;
    if (!dev && hipMalloc(&d, bytes) != hipSuccess) return -1;
    k_unpack_labels<<<voxel_blocks(c), TPB, 0, g_stream>>>(c, lab, d, dtype, st[0], st[1], st[2]);
    if (!dev) { HIP_CHECK(hipMemcpyAsync(dst, d, bytes, hipMemcpyDeviceToHost, g_stream)); }
    HIP_CHECK(hipStreamSynchronize(g_stream));
    if (!dev) HIP_CHECK(hipFree(d));
    return 0;
}

int be_build_levels(const VrgCtx& c, double** lev, uint32_t* L) {
    size_t V = (size_t)c.nx * c.ny * c.nz;
    float *a = nullptr, *b = nullptr; uint32_t* cnt = nullptr; void* tmp = nullptr; size_t tb = 0, tb2 = 0;
    if (hipMalloc(&a, V * 4) != hipSuccess || hipMalloc(&b, V * 4) != hipSuccess || hipMalloc(&cnt, 4) != hipSuccess) return -1;
    k_gather_I<<<voxel_blocks(c), TPB, 0, g_stream>>>(c, a);
    HIP_CHECK(rocprim::radix_sort_keys(nullptr, tb, a, b, V, 0, 32, g_stream));
    HIP_CHECK(rocprim::unique(nullptr, tb2, b, a, cnt, V, rocprim::equal_to<float>(), g_stream));
    tb = std::max(tb, tb2);
    if (hipMalloc(&tmp, tb) != hipSuccess) return -1;
    HIP_CHECK(rocprim::radix_sort_keys(tmp, tb, a, b, V, 0, 32, g_stream));
    HIP_CHECK(rocprim::unique(tmp, tb, b, a, cnt, V, rocprim::equal_to<float>(), g_stream));
    uint32_t n = 0;
    HIP_CHECK(hipMemcpyAsync(&n, cnt, 4, hipMemcpyDeviceToHost, g_stream));
    HIP_CHECK(hipStreamSynchronize(g_stream));
    double* out = nullptr;
    if (hipMalloc(&out, (size_t)n * 8) != hipSuccess) return -1;
    k_f2d<<<256, TPB, 0, g_stream>>>(a, out, n);
    HIP_CHECK(hipStreamSynchronize(g_stream));
    HIP_CHECK(hipFree(a)); HIP_CHECK(hipFree(b)); HIP_CHECK(hipFree(cnt)); HIP_CHECK(hipFree(tmp));
    *lev = out; *L = n;
    return 0;
}

__global__ void k_build_lev16(VrgCtx c, uint16_t* dst) {
    VOXEL_LOOP(c) { int x, y, z; uint32_t idx = real_idx(c, t, x, y, z); dst[idx] = (uint16_t)vrg_level_of(c, (double)c.I[idx]); }
}
void be_build_lev16(const VrgCtx& c, uint16_t* dst) {
    HIP_CHECK(hipMemsetAsync(dst, 0, ((size_t)c.PV + 1023) / 1024 * 1024 * 2, g_stream));
    k_build_lev16<<<voxel_blocks(c), TPB, 0, g_stream>>>(c, dst);
}

void be_init_band(const VrgCtx& c) {
    k_init_voxel<<<voxel_blocks(c), TPB, 0, g_stream>>>(c);
}

void be_init_sort(const VrgCtx& c, uint32_t n_in, uint32_t n_out) {
    uint32_t nmax = std::max(n_in, n_out);
    if (nmax == 0) return;
    uint64_t* kout = nullptr; void* tmp = nullptr; size_t tb = 0;
    HIP_CHECK(hipMalloc(&kout, (size_t)nmax * 8));
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tb, c.init_key, kout, c.init_idx, c.b_idx[0], nmax, 0, 64, g_stream));
    HIP_CHECK(hipMalloc(&tmp, tb));
    if (n_in) HIP_CHECK(rocprim::radix_sort_pairs(tmp, tb, c.init_key, kout, c.init_idx, c.b_idx[0], n_in, 0, 64, g_stream));
    if (n_out) HIP_CHECK(rocprim::radix_sort_pairs(tmp, tb, c.init_key + (c.bcap - n_out), kout, c.init_idx + (c.bcap - n_out),
                                                   c.b_idx[0] + n_in, n_out, 0, 64, g_stream));
    HIP_CHECK(hipStreamSynchronize(g_stream));
    HIP_CHECK(hipFree(kout)); HIP_CHECK(hipFree(tmp));
}

// sum the slab statistics over the ranks: RCCL on the stream, or the host callback (synchronises)
static void reduce_dense(const VrgCtx& c, be_reduce_fn cb, void* user, hipStream_t g_stream) {
    if (g_comm) {
        ncclResult_t r = ncclAllReduce(c.dn_part, c.dn, 4, ncclDouble, ncclSum, g_comm, g_stream);
        if (r != ncclSuccess && !g_hip_error[0]) {   // sticky: the engine turns it into VRG_E_INTERNAL at its next synchronisation point
            std::snprintf(g_hip_error, sizeof(g_hip_error), "RCCL all-reduce of the slab statistics failed: %s", ncclGetErrorString(r));
            std::fprintf(stderr, "%s\n", g_hip_error);
        }
    } else if (cb) {
        double v[4];
        HIP_CHECK(hipMemcpyAsync(v, c.dn_part, sizeof(v), hipMemcpyDeviceToHost, g_stream));
        HIP_CHECK(hipStreamSynchronize(g_stream));
        cb(v, user);
        HIP_CHECK(hipMemcpyAsync(c.dn, v, sizeof(v), hipMemcpyHostToDevice, g_stream));
        HIP_CHECK(hipStreamSynchronize(g_stream));
    }
}

int be_comm_unique_id(void* id128) {
    static_assert(sizeof(ncclUniqueId) == 128, "id size");
    return ncclGetUniqueId((ncclUniqueId*)id128) == ncclSuccess ? 0 : -1;
}
int be_comm_init(int nranks, int rank, const void* id128) {
    if (g_comm) { ncclCommDestroy(g_comm); g_comm = nullptr; }
    ncclUniqueId id; std::memcpy(&id, id128, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&g_comm, nranks, id, rank);
    if (r != ncclSuccess) { std::fprintf(stderr, "ncclCommInitRank failed: %s\n", ncclGetErrorString(r)); g_comm = nullptr; return -1; }
    return 0;
}

// The start / stop events ride on the dispatch itself (hipExtLaunchKernel): no separate event packets in the stream,
// which cost ~4 us each between two back-to-back recounts.
static void launch_recount(const VrgCtx& c, int blocks, int check, hipStream_t st, hipEvent_t e_start = nullptr, hipEvent_t e_stop = nullptr) {
    if (c.lev16) hipExtLaunchKernelGGL((k_recount_bits<3, true, true>), dim3(blocks), dim3(TPB), 0, st, e_start, e_stop, 0, c, check);
    else hipExtLaunchKernelGGL((k_recount_bits<3, true, false>), dim3(blocks), dim3(TPB), 0, st, e_start, e_stop, 0, c, check);
}

void be_init_finish(const VrgCtx& c, be_reduce_fn cb, void* user) {
    HIP_CHECK(hipStreamSynchronize(g_stream_b));     // both class copies are rebuilt: no dense pass may be in flight
    k_init_entry<<<ITEM_BLOCKS, TPB, 0, g_stream>>>(c);
    if (c.L <= HIST_LDS_LEVELS) k_hist_lds<<<1024, TPB, 0, g_stream>>>(c);
    else k_hist_voxel<<<voxel_blocks(c), TPB, 0, g_stream>>>(c);
    k_exact<<<1024, TPB, 0, g_stream>>>(c);
    k_cls_build<<<2048, TPB, 0, g_stream>>>(c);
    launch_recount(c, dense_blocks(c), 0, g_stream);
    reduce_dense(c, cb, user, g_stream);
    k_fin_init<<<1, 1, 0, g_stream>>>(c);
}

// ---- one sweep ------------------------------------------------------------------------------------------
// Stream A ("band") carries every sparse kernel in program order; stream B ("dense") carries the recounts (+ the
// slab all-reduce and k_dense_fin on several GPUs).  The only edges between them:
//   recount(k) waits for apply(k) + entry_post(k)    (labels of sweep k in class copy k & 1, expected sizes filed)
//   apply(k)   waits for recount(k-2)                (it rewrites class copy k & 1, which pass k-2 was reading)
// Neither wait blocks in steady state: on one big volume stream A is a sweep ahead and stream B runs its recounts
// back to back; on small slabs stream B is idle most of the time and stream A never finds pass k-2 unfinished.
// The band kernels read and write the label BYTES only; the dense pass reads the class bits only.
static void enqueue_pre(const VrgCtx& c, int variant) {         // decide + flip list, marks + prepass, fix-point, relabel
    k_decide_exact<<<2 * ITEM_BLOCKS, TPB, 0, g_stream>>>(c);
    k_marks_prepass<<<ITEM_BLOCKS, TPB, 0, g_stream>>>(c);
    if (!(variant & 1)) k_relabel<<<ITEM_BLOCKS, TPB, 0, g_stream>>>(c);     // + the skip-rule fix-point
    else { k_fix<<<1, 1024, 0, g_stream>>>(c); k_full_relabel<<<2048, TPB, 0, g_stream>>>(c); }
}
// first kernel(s) after the labels are applied (level deltas; files the expected sizes): launched eagerly, the event
// that releases the dense pass rides on the dispatch (small level tables) or follows it
static void launch_levels(const VrgCtx& c, hipEvent_t ev) {
    if (c.L <= LT_MAX) {
        hipExtLaunchKernelGGL(k_levels_tab_scan, dim3(ITEM_BLOCKS + SCAN_BLOCKS), dim3(TPB), 0, g_stream, nullptr, ev, 0, c);
        return;
    }
    if (c.L <= LEVELS_ONEBLOCK) k_levels_small<<<1, 1024, 0, g_stream>>>(c);
    else k_delta_flag<<<ITEM_BLOCKS, TPB, 0, g_stream>>>(c);
    HIP_CHECK(hipEventRecord(ev, g_stream));
}
static void enqueue_post(const VrgCtx& c) {                     // rest of the band bookkeeping (new lists, densities), iterNum += 1
    if (c.L <= LT_MAX) {
        k_scan_down<<<SCAN_BLOCKS, TPB, 0, g_stream>>>(c, c.scan, 1);
    } else {
        if (c.L > LEVELS_ONEBLOCK) {
            device_scan(c, c.lscan, g_stream, 0);
            k_post_prep<<<1, 1, 0, g_stream>>>(c);
            k_delta_scatter<<<ITEM_BLOCKS, TPB, 0, g_stream>>>(c);
        }
        k_tab<<<ITEM_BLOCKS, TPB, 0, g_stream>>>(c);
        device_scan(c, c.scan, g_stream, 1);
    }
    k_scatter<<<ITEM_BLOCKS, TPB, 0, g_stream>>>(c);
    k_finalize<<<1, 1, 0, g_stream>>>(c);             // the new entries' exact densities: k_decide_exact of the next trip
}

// With option "graph" the two runs of band kernels are replayed from captured hipGraphs (one host call each);
// apply, entry_post, the event edges, the recount and the collective stay eager, so nothing depends on RCCL
// supporting stream capture and the per-launch HIP-event timing of the recount keeps working.
struct GraphCache { hipGraphExec_t exec = nullptr; VrgCtx key; int variant = -1; bool valid = false; };
static GraphCache g_graph_pre, g_graph_post;
#define g_use_graph g_use_graph_req

template <class F> static void run_band(GraphCache& g, const VrgCtx& c, int variant, F&& enqueue) {
    if (g_use_graph) {
        if (!g.valid || g.variant != variant || std::memcmp(&g.key, &c, sizeof(VrgCtx)) != 0) {
            if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
            g.valid = false;
            hipGraph_t graph = nullptr;
            bool ok = hipStreamBeginCapture(g_stream, hipStreamCaptureModeRelaxed) == hipSuccess;
            if (ok) {
                enqueue();
                ok = hipStreamEndCapture(g_stream, &graph) == hipSuccess && graph != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
            if (ok) { std::memcpy(&g.key, &c, sizeof(VrgCtx)); g.variant = variant; g.valid = true; }
            else {                                   // capture not possible here: stay with eager launches
                (void)hipGetLastError();
                g_use_graph = 0;
                std::fprintf(stderr, "vrg: hipGraph capture failed, falling back to eager launches\n");
            }
        }
        if (g.valid) { HIP_CHECK(hipGraphLaunch(g.exec, g_stream)); return; }
    }
    enqueue();
}

void be_sweep_once(const VrgCtx& c, int variant, VrgEvents* ev, be_reduce_fn cb, void* user) {
    // every event record / wait is a barrier packet of a few us on its stream: the end-of-recount timing event
    // doubles as the edge "class copy read"
    hipEvent_t e_start = nullptr, e_read = g_ev_d[g_trip & 1];
    if (ev && ev->enabled) {
        if (g_ev_used == g_ev_pool.size()) { EvPair n; HIP_CHECK(hipEventCreate(&n.a)); HIP_CHECK(hipEventCreate(&n.b)); g_ev_pool.push_back(n); }
        EvPair& p = g_ev_pool[g_ev_used++];
        e_start = p.a; e_read = p.b;
    }
    run_band(g_graph_pre, c, variant & 3, [&] { enqueue_pre(c, variant & 3); });
    // the labels change now, in the class copy the dense pass of two sweeps ago was reading
    if (g_read[g_trip & 1]) HIP_CHECK(hipStreamWaitEvent(g_stream, g_read[g_trip & 1], 0));
    if (!(variant & 1)) k_apply_entry_post<<<2 * ITEM_BLOCKS, TPB, 0, g_stream>>>(c);
    else { k_copy_back<<<2048, TPB, 0, g_stream>>>(c); k_entry_post<<<ITEM_BLOCKS, TPB, 0, g_stream>>>(c); }
    launch_levels(c, g_ev_a);                        // files the sizes the dense pass must reproduce; g_ev_a on its dispatch
    // dense stream: every voxel once, read-only.  Enqueued before the rest of the bookkeeping so that its dispatch
    // never waits for the host to issue those launches.
    if (!(variant & 4)) {                            // (variant & 4: measurement aid, the band chain alone)
    HIP_CHECK(hipStreamWaitEvent(g_stream_b, g_ev_a, 0));
    const bool ranks = c.world > 1 || g_comm || cb;
    launch_recount(c, dense_blocks(c), ranks ? 1 : 2, g_stream_b, e_start, e_read);
    g_read[g_trip & 1] = e_read;
    g_trip++;
    if (ranks) {                                     // one GPU: the last workgroup of the recount closes the pass itself
        reduce_dense(c, cb, user, g_stream_b);       // sum over the Z-slabs (RCCL on the stream / host callback)
        k_dense_fin<<<1, 1, 0, g_stream_b>>>(c);
    }
    }
    run_band(g_graph_post, c, 0, [&] { enqueue_post(c); });
}

void be_events_collect(VrgEvents* ev, long long n_valid) {
    if (!ev) return;
    if (g_ev_used) HIP_CHECK(hipStreamSynchronize(g_stream_b));   // the dense stream may trail the band stream by one pass
    for (size_t i = 0; i < g_ev_used; i++) {
        if ((long long)i < n_valid) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, g_ev_pool[i].a, g_ev_pool[i].b) == hipSuccess) { ev->ms_total += ms; ev->launches++; }
            else (void)hipGetLastError();
        }
    }
    g_ev_used = 0;
}

void be_recount_hist(const VrgCtx& c, int par, int32_t* rin, int32_t* rout) {
    k_recount_hist<<<voxel_blocks(c), TPB, 0, g_stream>>>(c, rin, rout);
    HIP_CHECK(hipStreamSynchronize(g_stream));
}

uint32_t be_collect_segmented(const VrgCtx& c, int par, uint64_t* stamps, uint32_t* idxs, uint32_t cap) {
    uint64_t* ds = nullptr; uint32_t* di = nullptr; uint32_t* dc = nullptr;
    HIP_CHECK(hipMalloc(&ds, (size_t)(cap + 1) * 8)); HIP_CHECK(hipMalloc(&di, (size_t)(cap + 1) * 4)); HIP_CHECK(hipMalloc(&dc, 4));
    HIP_CHECK(hipMemsetAsync(dc, 0, 4, g_stream));
    k_collect_seg<<<voxel_blocks(c), TPB, 0, g_stream>>>(c, ds, di, cap, dc);
    uint32_t n = 0;
    HIP_CHECK(hipMemcpyAsync(&n, dc, 4, hipMemcpyDeviceToHost, g_stream));
    HIP_CHECK(hipStreamSynchronize(g_stream));
    uint32_t m = std::min(n, cap);
    HIP_CHECK(hipMemcpy(stamps, ds, (size_t)m * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(idxs, di, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipFree(ds)); HIP_CHECK(hipFree(di)); HIP_CHECK(hipFree(dc));
    return n;
}
