// vrg_device.hip - the product backend: HIP kernels for MI355X (gfx950, wave64).
//
// One while-loop trip of variationalRegionGrowing.py:58-117 (be_sweep_once) is k_band + update() on stream A, plus the dense
// pass on stream B.  update() (:156-259) comes in three kinds, all running the item functions of vrg_items.h on the same data:
//   k_band  (many workgroups, one thread - or 16 / 8 / 4 lanes - per band-pool slot): adds the density corrections of the
//           sweep before (:236-247) to the surviving entries, decides every entry (:79-88) and appends the flips to an
//           unordered list; extra workgroups compute the exact densities (:252-255) of the entries the sweep before added
//           and decide those; behind a fused sweep 32 more write its label bytes, class bits and free list.
//   FUSED   k_sweep: ONE launch, one flip per workgroup, for sweeps with at most 128 flips (65 on a large level table) -
//           every workgroup ranks the flips and resolves the skip rule itself; nothing is applied inside the sweep
//           (-> k_memo on large bands).  A sweep with more flips hands itself back untouched (VBAIL_FUSE).
//   CHAIN   k_order (-> k_rank_wide -> k_list_wide -> k_prepass_wide -> k_fix_wide above 512 flips) -> k_mark_relabel<1> (up to 256 flips) or
//           k_mark_compact (a flip per half-wave) -> k_mark_relabel<4> over the flips it left -> k_close: up to 65 536 flips without a host
//           synchronisation; the relabel kernels' workgroups reserve their stretches of the sweep's lists through VrgCtx::rsv (cache lines of their
//           own: same-address atomics execute one after the other).  A sweep with more flips (or one that needs larger arrays) is handed back
//           untouched (VrgState::bail) and
//   HOST-DRIVEN (be_sweep_once with VRG_SWEEP_SYNC): the host reads the flip count - above 65 536 flips a radix sort ranks them and the chain's
//           chip-wide kernels do the rest; the full-stencil check variant (and a lowered "small_flips") runs the item functions as device-wide kernels.
//   stream B, the dense pass (enqueued behind the kernel that raises its request: k_close, or the k_band after a fused sweep):
//     k_recount_pipe / k_recount_bits : the dense kernel (every listed 1024-voxel unit, HBM-bound, read-only: 4 B intensity of included
//        voxels + 2 class bits per voxel): region sizes and intensity sums (:113-116, :249-250), reduced by its last workgroup, checked
//        against the sizes the band side keeps by increments
//     -> on several GPUs: slab all-reduce -> k_dense_fin (the same check on the totals, trace sums).
//   Stream A does not join: it runs up to two sweeps ahead of the dense pass (two copies of the class bits).
// Labels are updated IN PLACE: measured on MI355X, streaming I + labels read-only runs at 5.8-6.0 TB/s
// while the same stream with a 1 B/voxel label write-back drops to 4.8 TB/s, so unchanged labels are
// never rewritten.  (The full-stencil check variant relabels every voxel through lab[1].)
// Every kernel starts by reading the device-resident VrgState and returns at once when the stop
// flag is set, so the host can enqueue batches of sweeps without synchronising.
// All state of the backend (device, streams, events, communicator, first error) lives in VrgBackend: one per handle.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>
#include <rccl/rccl.h>

#include "vrg_backend.h"
#include "vrg_items.h"

struct EvPair { hipEvent_t a, b; long long trip; int kind; int ntrips; };   // kind 0: a dense launch, 1: the band chain of ntrips trips (the last of them: trip)

struct VrgBackend {
    int device = 0;
    hipStream_t sa = nullptr;            // stream A: the band kernels of every trip in program order, copies
    hipStream_t sb = nullptr;            // stream B: the dense pass (recount, slab all-reduce, k_dense_fin); trails stream A by up to one sweep
    hipStream_t sc = nullptr;            // stream C: the change log's transport (leader / follower replication: RCCL broadcasts), created on first use
    int repl = 0;                        // this handle is a rank of a leader / follower group: the communicator carries the log, not slab sums
    hipStream_t sd = nullptr;            // stream D: a follower's label bytes and stamps (beside its dense passes, which read the class bits only)
    hipEvent_t mark[4] = {nullptr, nullptr, nullptr, nullptr};   // a follower's staging buffers: the kernels that read buffer j have been enqueued up to here (per stream)
    int sweep_blocks = 0;                // 0 = auto (dense_blocks)
    int prio_mode = 2;                   // the dense stream gets the higher priority (measured: -1..2 % step time)
    uint32_t small_flips = 65536;        // flips per sweep the device-resident four-launch chain takes on (option "small_flips", at most NF_WIDE); a sweep with more is driven from the host
    ncclComm_t comm = nullptr;           // per-sweep all-reduce of the slab statistics (multi-GPU)
    char err[256] = "";                  // first HIP / RCCL failure; the engine turns it into VRG_E_INTERNAL
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
    long long ev_trip = 0;               // trips enqueued since the last be_events_collect
    void* tmp = nullptr; size_t tmp_bytes = 0;        // scratch of the host-driven sorts
    uint64_t* keys2 = nullptr; size_t keys2_n = 0;
    int dense_pending = 0;                            // Z-slabs: recounts enqueued since the last staged all-reduce
    int serial = 0;                                   // option "serial_streams": see be_sweep_once
    int skip = 1;                                     // option "skip_excluded": the dense pass does not fetch the intensities of excluded voxels
    int nt_loads = -1;                                // option "nt_loads": -1 = by the size of the pass, 0 / 1 = ordinary / non-temporal loads
    int verify_every = 1;                             // option "verify_every": the dense pass on every n-th sweep only (0: never)
    int dense_pipe = 1;                               // option "dense_pipe": fp32 storage + skip_excluded run the two-trips-deep recount (k_recount_pipe)
    uint64_t pass_bytes = 0;                          // bytes a dense pass fetches, counted at the end of init (0: not known yet)
    uint32_t memo_above = 32768;                      // option "memo_above": band entries above which a fused trip keeps the per-level memo (k_memo)
    long long memo_trips = 0;                         // fused trips that did
    bool fused_memo = false;                          // ... and it kept the per-level memo (k_memo)
    bool fused_prev = false;                          // the trip enqueued last was a fused one: the dense pass of the sweep it applied is not enqueued yet
                                                      // (its request comes from THIS trip's k_band; if that trip stopped or handed itself back, the stop word makes the gate leave)
    bool prev_open = false;                           // ... and its sweep was open-ended: this trip's k_band derives the closed state (and lists the touched levels itself)
    int open_par = 0;                                 // ... the set of per-level counters it filled
    long long follow_counts = 0;                      // a follower's dense passes so far (which of them are timed: option "events")
    uint32_t band_blocks_max = 2048;                  // option "band_blocks_max": most workgroups k_band uses for the pool (BAND_BLOCKS)
    uint64_t* rsv = nullptr;                          // VrgCtx::rsv of this handle's four-launch trips (64 words, zero between sweeps)
    int mark_compact = 1;                             // option "mark_compact": four-launch trips of thousands of flips relabel with k_mark_compact (+ k_mark_relabel for what it leaves)
    int open_sweeps = 1;                              // option "open_sweeps": fused sweeps inside a batch end at their commit, without a closing workgroup
    int iter_hint = 0;                                // sweeps applied when the engine last read the state + trips enqueued since
    uint32_t band_hint = 0;                           // pool slots in use when the engine last read the state (0: unknown)
    void* xfer[2] = {nullptr, nullptr}; size_t xfer_bytes = 0;      // two page-locked buffers for host arrays on their way in / out
    uint32_t flip_hint_min = 0;                       // (a trip came back with this many flips: the launches are sized for at least that until the engine reads a state again)
    uint32_t flip_hint = 0;                           // ... and the flips of the sweep applied last (sizes the chip-wide launches of a four-launch trip)
    int direct_hint = 1;                              // ... and whether corrections are then evaluated entry by entry (8 lanes per slot)
};

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess && !b->err[0]) { \
    std::snprintf(b->err, sizeof(b->err), "HIP error '%s' in %s (%s:%d)", hipGetErrorString(e_), #x, __FILE__, __LINE__); \
    std::fprintf(stderr, "%s\n", b->err); } } while (0)

namespace {

constexpr int TPB = 256;            // 4 waves of 64
constexpr int ITEM_BLOCKS = 256;    // item kernels: 64 Ki threads, grid-stride
constexpr int EXACT_BLOCKS = 512;   // k_band: workgroups for the exact densities (one pending slot per workgroup at a time)
constexpr int SWEEP_BLOCKS = 256;   // 1 workgroup (4 waves) per CU, each wave with 3 KiB of labels + 12 KiB of intensities in
                                    // flight: measured best for the HBM-bound recount while stream B's band kernels run beside
                                    // it (880x880x640: 256 -> 0.38 ms, 192/384 -> 0.42-0.43, 320 -> 0.49, 512 -> 0.40, 1024 -> 0.44)
constexpr uint32_t NF_SMALL = 4096; // flips one workgroup sorts in LDS
constexpr uint32_t NF_WIDE = 65536; // flips the device-resident chain can take (option small_flips); more: host-driven trips
constexpr uint32_t NF_ORDER = 512;  // ... above this many the ordering step runs chip-wide (k_rank_wide, k_prepass_wide, k_fix_wide) instead of in k_order's one workgroup (86 us at 1600 flips)
constexpr uint32_t NZ_LDS = 1024;   // touched levels k_band keeps in LDS
constexpr int KS_THREADS = 1024;    // k_fix (host-driven trips): one big workgroup
constexpr int KC_THREADS = 256;     // k_close: one wave per SIMD, so that its workgroups fit on a CU beside the three recount waves
                                    // per SIMD (16-wave workgroups had to wait for the recount to end: 0.1 ms per sweep)

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


// in-kernel time stamps of the band chain (diagnostic build -DVRG_STAMPS only; in the product build no stamp executes)
#if defined(VRG_STAMPS)
#define VRG_STAMP(c, k) do { c.dbg[k] = wall_clock64(); } while (0)
#define VRG_STAMP_NOW() wall_clock64()
#define VRG_STAMP_PUT(c, k, v) do { c.dbg[k] = (v); } while (0)
#define VRG_STAMP_MAX(c, k) do { atomicMax(&c.dbg[k], (unsigned long long)wall_clock64()); } while (0)   // the last workgroup's exit
// per-workgroup stamps (thread 0 of every workgroup; word k of the workgroup's VRG_DBG_PER)
#define VRG_STAMP_WG(c, k) do { if (threadIdx.x == 0 && blockIdx.x < (uint32_t)VRG_DBG_WG && (k) < (uint32_t)VRG_DBG_PER) c.dbg[64 + blockIdx.x * VRG_DBG_PER + (k)] = wall_clock64(); } while (0)
#define VRG_STAMP_WG_PUT(c, k, v) do { if (threadIdx.x == 0 && blockIdx.x < (uint32_t)VRG_DBG_WG) c.dbg[64 + blockIdx.x * VRG_DBG_PER + (k)] = (v); } while (0)
#else
#define VRG_STAMP_WG(c, k) do { } while (0)
#define VRG_STAMP_WG_PUT(c, k, v) do { (void)(v); } while (0)
#define VRG_STAMP(c, k) do { } while (0)
#define VRG_STAMP_NOW() 0ull
#define VRG_STAMP_PUT(c, k, v) do { (void)(v); } while (0)
#define VRG_STAMP_MAX(c, k) do { } while (0)
#endif
// random delays at the entry of every concurrent kernel and in front of every hand-off (diagnostic build -DVRG_CHAOS only,
// tools/build_chaos.sh; in the product build nothing executes): one wave in four sleeps for up to ~110 us, so workgroups,
// kernels and the two streams meet in orders a quiet machine never produces - the results must not change
// (tools/gpu.sh <tag> chaos)
#if defined(VRG_CHAOS)
__device__ __forceinline__ void vrg_chaos_delay(uint32_t salt) {
    uint32_t h = ((uint32_t)wall_clock64() * 2654435761u) ^ (blockIdx.x * 40503u + (threadIdx.x >> 6) * 9973u + salt * 7919u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
    if ((h & 3u) == 0u) { const uint32_t n = (h >> 2) & 63u; for (uint32_t i = 0; i < n; i++) __builtin_amdgcn_s_sleep(64); }
}
#define VRG_CHAOS_POINT(salt) vrg_chaos_delay(salt)
#else
#define VRG_CHAOS_POINT(salt) do { } while (0)
#endif
#define ITEM_LOOP(n) for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n_ = (n); i < n_; i += gridDim.x * blockDim.x)
// same with a 64-bit item index: (listed flips) x (positions) can exceed 2^32 on adversarial volumes
#define ITEM_LOOP64(n) for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n_ = (n); i < n_; i += (uint64_t)gridDim.x * blockDim.x)

// ---- k_band -----------------------------------------------------------------------------------------------
// exact densities (:152-155, :252-255): one wave per pending slot, lanes stride over the levels.  The level table
// (the same for every entry) is fetched first, four levels per lane at a time, so that it travels together with
// the entry's own look-ups instead of behind them.  The wave that computed an entry's densities decides it.
// Every lane sums its levels (lane, lane + 64, ...) in ascending order whatever the batching - the order of the
// additions, and so the result, does not depend on EXQ.
constexpr int EXQ = 8;      // levels per lane fetched together: 512 per wave and batch
// (with bins - large level tables, vrg_items.h "binned exact densities" - the lanes stride over the bins within reach of the entry)
__device__ void exact_wave_binned(const VrgCtx& c, const VrgState& s, uint32_t nfresh, uint32_t wid, uint32_t nw, bool then_decide, int64_t n_in, int64_t n_out) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t f = wid; f < nfresh; f += nw) {
        const uint32_t slot = c.fresh[f];
        const double v = c.lev[c.p_lev[slot]];
        uint32_t b0, b1; vrg_bin_range(c, v, b0, b1);
        double si = 0, so = 0;
        for (uint32_t b = b0 + lane; b <= b1; b += 64u) { double ti, to; vrg_bin_terms(c, v, b, ti, to); si += ti; so += to; }
        si = wave_sum(si); so = wave_sum(so);
        if (lane == 0) {
            const float err = vrg_exact_err(c, si, so);
            c.p_ip[slot] = si; c.p_op[slot] = so; c.p_err[slot] = err;
            if (then_decide && s.iter < s.iterMax)
                vrg_decide_core(c, s, n_in, n_out, slot, c.p_flag[slot] & PF_INNER, si, so, c.p_key[slot], c.p_idx[slot], c.p_lev[slot], (double)err);
        }
    }
}
__device__ void exact_wave(const VrgCtx& c, const VrgState& s, uint32_t nfresh, uint32_t wid, uint32_t nw, bool then_decide) {
    const int lane = threadIdx.x & 63;
    if (wid >= nfresh) return;
    if (c.nb) { exact_wave_binned(c, s, nfresh, wid, nw, then_decide, c.inc[VC_NIN], c.inc[VC_NOUT]); return; }
    int32_t ha[EXQ], hb[EXQ]; double lv[EXQ];      // the first batch stays in registers for every entry of this wave
#pragma unroll
    for (int q = 0; q < EXQ; q++) {
        uint32_t l = lane + 64u * q;
        bool in = l < c.L;
        ha[q] = in ? c.hin[l] : 0; hb[q] = in ? c.hout[l] : 0; lv[q] = in ? c.lev[l] : 0.0;
    }
    for (uint32_t f = wid; f < nfresh; f += nw) {
        const uint32_t slot = c.fresh[f];
        double v = c.lev[c.p_lev[slot]], si = 0, so = 0;
#pragma unroll
        for (int q = 0; q < EXQ; q++) {
            if (!(ha[q] | hb[q])) continue;
            double k = vrg_kern(c, lv[q] - v);
            si += (double)ha[q] * k; so += (double)hb[q] * k;
        }
        for (uint32_t l0 = 64u * EXQ; l0 < c.L; l0 += 64u * EXQ) {     // (one round trip per batch, not per level)
            int32_t a[EXQ], bb[EXQ]; double x[EXQ];
#pragma unroll
            for (int q = 0; q < EXQ; q++) {
                const uint32_t l = l0 + lane + 64u * q;
                const bool in = l < c.L;
                a[q] = in ? c.hin[l] : 0; bb[q] = in ? c.hout[l] : 0; x[q] = in ? c.lev[l] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < EXQ; q++) {
                if (!(a[q] | bb[q])) continue;
                double k = vrg_kern(c, x[q] - v);
                si += (double)a[q] * k; so += (double)bb[q] * k;
            }
        }
        si = wave_sum(si); so = wave_sum(so);
        if (lane == 0) {
            c.p_ip[slot] = si; c.p_op[slot] = so; c.p_err[slot] = 0.0f;   // (the pending flag is cleared by the slot's own thread in the other half; sums over the levels: no binning error)
            if (then_decide && s.iter < s.iterMax)       // while iterNum <= iterMax (:58)
                vrg_decide_core(c, s, c.inc[VC_NIN], c.inc[VC_NOUT], slot, c.p_flag[slot] & PF_INNER, si, so, c.p_key[slot], c.p_idx[slot], c.p_lev[slot]);
        }
    }
}
// The same for the few hundred slots a sweep adds: one WORKGROUP per pending slot, its four waves taking every fourth
// batch of 512 levels (with a wave per slot most of the chip idles while each wave walks the whole table: 38 us of
// k_band at 6111 levels).  Wave partial sums are added in the order 0..3; a table of <= 512 levels is wave 0's alone,
// which then makes exactly exact_wave's additions.
// (n_in / n_out: the region sizes the decisions read - the caller's, which may have derived them from an open-ended sweep)
// (sink: where the many-slot branch lists the slots that flip - most of a large sweep's flips are entries the sweep before added)
__device__ void exact_wg(const VrgCtx& c, const VrgState& s, uint32_t nfresh, uint32_t wg, uint32_t nwg, int64_t n_in, int64_t n_out, VrgFlipSink* sink = nullptr) {
    __shared__ double sh_i[TPB / 64], sh_o[TPB / 64];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr uint32_t NWV = TPB / 64, BATCH = 64u * EXQ;
    if (wg >= nfresh) return;
    if (c.nb) {                                           // (with bins: a wave per entry does it - at most 2983 bins, 47 per lane)
        exact_wave_binned(c, s, nfresh, wg * NWV + wv, nwg * NWV, true, n_in, n_out);
        return;
    }
    if (c.L <= BATCH && nfresh > nwg) {
        // Tens of thousands of pending slots (a sweep of thousands of flips) and a table that is one batch - wave 0's alone below, the other
        // three waves idle while each workgroup walks ~100 slots one dependent chain after the other (0.2 ms at 12 900 flips): every WAVE
        // takes slots of its own, two per turn so that their look-ups travel together.  The additions are wave 0's, in its order.
        int32_t ha[EXQ], hb[EXQ]; double lv[EXQ];
#pragma unroll
        for (int q = 0; q < EXQ; q++) {
            const uint32_t l = lane + 64u * q;
            const bool in = l < c.L;
            ha[q] = in ? c.hin[l] : 0; hb[q] = in ? c.hout[l] : 0; lv[q] = in ? c.lev[l] : 0.0;
        }
        const uint32_t W = nwg * NWV;
        for (uint32_t f = wg * NWV + wv; f < nfresh; f += 2u * W) {
            const bool two = f + W < nfresh;
            const uint32_t slotA = c.fresh[f], slotB = c.fresh[two ? f + W : f];
            const uint32_t levA = c.p_lev[slotA], levB = c.p_lev[slotB];
            const double vA = c.lev[levA], vB = c.lev[levB];
            double siA = 0, soA = 0, siB = 0, soB = 0;
#pragma unroll
            for (int q = 0; q < EXQ; q++) {
                if (!(ha[q] | hb[q])) continue;
                const double kA = vrg_kern(c, lv[q] - vA), kB = vrg_kern(c, lv[q] - vB);
                siA += (double)ha[q] * kA; soA += (double)hb[q] * kA;
                siB += (double)ha[q] * kB; soB += (double)hb[q] * kB;
            }
            siA = wave_sum(siA); soA = wave_sum(soA); siB = wave_sum(siB); soB = wave_sum(soB);
            if (lane == 0 || (lane == 1 && two)) {
                const uint32_t slot = lane ? slotB : slotA;
                const double si = lane ? siB : siA, so = lane ? soB : soA;
                c.p_ip[slot] = si; c.p_op[slot] = so; c.p_err[slot] = 0.0f;
                if (s.iter < s.iterMax)
                    vrg_decide_core(c, s, n_in, n_out, slot, c.p_flag[slot] & PF_INNER, si, so, c.p_key[slot], c.p_idx[slot], lane ? levB : levA, 0.0, sink);
            }
        }
        return;
    }
    int32_t ha[EXQ], hb[EXQ]; double lv[EXQ];      // this wave's first batch stays in registers for every slot
#pragma unroll
    for (int q = 0; q < EXQ; q++) {
        uint32_t l = BATCH * wv + lane + 64u * q;
        bool in = l < c.L;
        ha[q] = in ? c.hin[l] : 0; hb[q] = in ? c.hout[l] : 0; lv[q] = in ? c.lev[l] : 0.0;
    }
    for (uint32_t f = wg; f < nfresh; f += nwg) {
        const uint32_t slot = c.fresh[f];
        double v = c.lev[c.p_lev[slot]], si = 0, so = 0;
#pragma unroll
        for (int q = 0; q < EXQ; q++) {
            if (!(ha[q] | hb[q])) continue;
            double k = vrg_kern(c, lv[q] - v);
            si += (double)ha[q] * k; so += (double)hb[q] * k;
        }
        for (uint32_t l0 = BATCH * (wv + NWV); l0 < c.L; l0 += BATCH * NWV) {
            int32_t a[EXQ], bb[EXQ]; double x[EXQ];
#pragma unroll
            for (int q = 0; q < EXQ; q++) {
                const uint32_t l = l0 + lane + 64u * q;
                const bool in = l < c.L;
                a[q] = in ? c.hin[l] : 0; bb[q] = in ? c.hout[l] : 0; x[q] = in ? c.lev[l] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < EXQ; q++) {
                if (!(a[q] | bb[q])) continue;
                double k = vrg_kern(c, x[q] - v);
                si += (double)a[q] * k; so += (double)bb[q] * k;
            }
        }
        si = wave_sum(si); so = wave_sum(so);
        if (lane == 0) { sh_i[wv] = si; sh_o[wv] = so; }
        __syncthreads();
        if (threadIdx.x == 0) {
            si = sh_i[0]; so = sh_o[0];
            for (uint32_t w = 1; w < NWV; w++) { si += sh_i[w]; so += sh_o[w]; }
            c.p_ip[slot] = si; c.p_op[slot] = so; c.p_err[slot] = 0.0f;   // (the pending flag is cleared by the slot's own thread in the other half; sums over the levels: no binning error)
            if (s.iter < s.iterMax)                      // while iterNum <= iterMax (:58)
                vrg_decide_core(c, s, n_in, n_out, slot, c.p_flag[slot] & PF_INNER, si, so, c.p_key[slot], c.p_idx[slot], c.p_lev[slot]);
        }
        __syncthreads();
    }
}
// band side, before the labels of sweep k are written into class copy k & 1: recount number `need` (= k - 2) has read that copy.
// Spins are bounded: a wait that does not end within SPIN_LIMIT raises an error instead of hanging the queue.
constexpr unsigned long long SPIN_LIMIT = 300000000ull;    // wall_clock64 ticks (100 MHz): 3 s
__device__ __forceinline__ void wait_dense_read_for(const VrgCtx& c, int64_t need) {
    if (need <= 0 || vrg_load_i64(&c.dctl[VD_RSEQ]) >= need) return;
    const unsigned long long t0 = wall_clock64();
    while (vrg_load_i64(&c.dctl[VD_RSEQ]) < need) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > SPIN_LIMIT) { vrg_store_i32(&c.stg->error, 9); return; }
    }
}
// the deferred work of this workgroup has reached memory; the LAST of the `n` workgroups to say so closes it (vrg_deferred_done)
__device__ __forceinline__ void band_deferred_done(const VrgCtx& c, int k, uint32_t n) {
    vrg_drain();
    __syncthreads();
    if (threadIdx.x == 0) {
        VRG_CHAOS_POINT(10);
        const uint32_t q = __hip_atomic_fetch_add(&c.counters[32], 1u, VRG_MO_TICKET, __HIP_MEMORY_SCOPE_AGENT);
#if defined(VRG_MUTANT)      // (tools/mutant_check.py: a deliberately broken hand-off - the FIRST workgroup to arrive asks for the dense pass - that the campaigns must catch)
        if (q == n - 1u) c.counters[32] = 0;
        if (q == 0u) vrg_deferred_done(c, k);
#else
        if (q == n - 1u) { c.counters[32] = 0; vrg_deferred_done(c, k); }
#endif
    }
}
// First kernel of a trip.  Workgroups [0, BAND_BLOCKS): the pool slots - correction of the sweep before, then the sign
// test; a flip is appended to the unordered flip list.  When the correction is evaluated entry by entry from the
// touched-level list (staged in LDS when it fits), LPE lanes share one slot: each sums every LPE-th level (nnz f64
// exp per entry is what this kernel costs), a fixed butterfly adds the partial sums.  With the per-level memo (or
// nothing to correct) it is one thread per slot.  Workgroups [BAND_BLOCKS, +EXACT_BLOCKS): the exact densities of the
// slots that (re-)entered the band in the sweep before, then their sign tests (exact_wg).
constexpr int BAND_BLOCKS = 2048;     // most workgroups k_band uses for the pool (above 2048 x 256 slots a thread takes several turns: every workgroup files its flips with one bump of the
                                      // flip counter, and those bumps run one after the other); fewer when the engine knows the pool is small (band_blocks())
// Lanes that share a slot when its correction is summed entry by entry (LPE): 16, 8 or 4 by the size of the pool, so that the
// pool's workgroups stay within one round of the chip (two workgroups per CU) - band_lanes().  Their partial sums are added
// by a DPP butterfly inside the group (xor 1, xor 2, mirror of 8, mirror of 16: no LDS traffic, where a shuffle is two
// bpermutes per step); every lane ends up with the total, lane 0's order of additions is the one used (deterministic).
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int LPE> __device__ __forceinline__ double group_sum(double v) {
    v += dpp_mov_f64<0xB1>(v);                            // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);                            // quad_perm [2,3,0,1]
    if constexpr (LPE >= 8) v += dpp_mov_f64<0x141>(v);   // row_half_mirror
    if constexpr (LPE >= 16) v += dpp_mov_f64<0x140>(v);  // row_mirror
    return v;
}
constexpr uint32_t TAB_LDS = 832;     // levels whose memo entries k_band stages in LDS (the room of the entry-by-entry path's arrays)
constexpr uint32_t DEFER_WGS = 32;    // pool workgroups of k_band that carry out what a fused sweep deferred (label bytes, class bits, free list)
constexpr uint32_t SINK_ABOVE = 1u << 17;   // pool slots above which a workgroup of k_band lists its flips together (VrgFlipSink)
// all threads of the workgroup, once its decisions are made: the sink's records into the flip list
__device__ void band_sink_file(const VrgCtx& c, VrgFlipSink& sk, uint32_t tid) {
    __syncthreads();
    const uint32_t n = min(sk.n, VRG_SINK_CAP);
    if (tid == 0 && n) sk.base = vrg_atomic_add(&c.stg->nf, n);
    __syncthreads();
    for (uint32_t i = tid; i < n; i += TPB) {
        const uint32_t q = sk.base + i;
        if (q >= c.fcap) { vrg_store_i32(&c.stg->error, 2); continue; }
        c.flist[q] = sk.slot[i]; c.f_key[q] = sk.key[i]; c.fr_idx[q] = sk.idx[i]; c.fr_lev[q] = sk.lev[i];
    }
}
template <int LPE>
__global__ void __launch_bounds__(TPB) k_band(VrgCtx c, uint32_t band_blocks, int dense_on, int direct_hint) {
    VRG_CHAOS_POINT(1);
    // One LDS block, two uses: the touched-level list (entry-by-entry corrections) or the head of the per-level memo.
    __shared__ double s_raw[NZ_LDS + NZ_LDS * 3 / 2];
    double* s_val = s_raw;
    uint32_t* s_nzl = reinterpret_cast<uint32_t*>(s_raw);   // (with the kernel table: the touched levels' indices instead of their values)
    uint32_t* s_cin = reinterpret_cast<uint32_t*>(s_raw + NZ_LDS); uint32_t* s_cout = s_cin + NZ_LDS; uint32_t* s_cconv = s_cout + NZ_LDS;
    static_assert(3 * TAB_LDS <= NZ_LDS + NZ_LDS * 3 / 2, "memo head must fit the block");
    // The kernel is a chain of dependent round trips (state -> slot fields -> memo entry / touched levels -> flip counter),
    // each of which takes 2-3 x longer beside a recount.  Every workgroup therefore fetches, TOGETHER WITH THE STATE, what
    // its threads' first items will need (any index below an array's capacity is readable, whatever the state then says):
    // the first slot's fields; the head of the memo, or - where the engine expects corrections entry by entry
    // (direct_hint: after a fused sweep, which keeps no memo) - the touched-level list and the level table; the places of
    // the marked list a fused sweep left to be applied.  A wrong hint costs round trips, never correctness.
    const bool pool_wg = blockIdx.x < band_blocks, defer_wg = blockIdx.x >= band_blocks + EXACT_BLOCKS;
    const bool st0 = blockIdx.x == 0 && threadIdx.x == 0, stx = blockIdx.x == band_blocks && threadIdx.x == 0;
    const unsigned long long t_entry = (st0 || stx) ? VRG_STAMP_NOW() : 0ull;     // (written below, and only by a trip that applies a sweep)
    const uint32_t tid = threadIdx.x, gtid = blockIdx.x * TPB + tid;
    const uint32_t dtid = (blockIdx.x - (band_blocks + EXACT_BLOCKS)) * TPB + tid;      // (deferred workgroups: their thread number)
    const bool use_ktab = c.ktab != nullptr;           // (uniform) the kernel between two levels is a table look-up
    const uint32_t slot0 = direct_hint ? gtid / LPE : gtid;
    uint8_t fl0 = 0; double ip0 = 0, op0 = 0; float err0 = 0; uint32_t lev0 = 0, idx0 = 0; uint64_t key0 = 0; int64_t nin0 = 0, nout0 = 0;
    const uint32_t tab_n = c.L < TAB_LDS ? c.L : TAB_LDS;
    constexpr uint32_t NZQ = 1;                           // touched levels per thread fetched with the state (256 per workgroup; a longer list: the rest once its length is known)
    double zv[NZQ]; uint32_t zi[NZQ], zo[NZQ], zc[NZQ], zl[NZQ];
    constexpr uint32_t defer_wgs = DEFER_WGS, G = DEFER_WGS * TPB;
    uint32_t mxa = VRG_NONE, mxb = VRG_NONE, cda0 = VRG_NOCHG, cxa0 = 0, cda1 = VRG_NOCHG, cxa1 = 0; uint8_t moa = 0, mna = 0, mob = 0, mnb = 0; int64_t rseq0 = 0;
    // (an open-ended sweep before this trip - c.lvl_par says which counter set it filled: this thread's stretch of the per-level counters,
    // from which the workgroup lists the touched levels itself; at most OPEN_PER levels per thread, i.e. OPEN_LEVELS in all)
    constexpr uint32_t OPEN_PER = 4;
    const int lpar = c.lvl_par >= 0 ? (c.lvl_par & 1) : 0;
    const uint32_t lper = (c.L + TPB - 1) / TPB;
    uint32_t lci[OPEN_PER], lco[OPEN_PER], lcc[OPEN_PER];
#pragma unroll
    for (uint32_t k = 0; k < OPEN_PER; k++) lci[k] = lco[k] = lcc[k] = 0;
    nin0 = c.inc_in[VC_NIN]; nout0 = c.inc_in[VC_NOUT];   // (the sizes that go with the state this kernel READS)
    if (pool_wg) {
        // (every load of this batch is unconditional with its index clamped into the array: a load under a divergent branch makes
        // the compiler wait for all loads in flight before the next one)
        { const uint32_t q = slot0 < c.bcap ? slot0 : c.bcap - 1u; fl0 = c.p_flag[q]; ip0 = c.p_ip[q]; op0 = c.p_op[q]; err0 = c.p_err[q]; lev0 = c.p_lev[q]; idx0 = c.p_idx[q]; key0 = c.p_key[q]; }
        if (!direct_hint) { for (uint32_t j = tid; j < 3 * tab_n; j += TPB) s_raw[j] = c.tabC[j]; }
        else if (c.lvl_par >= 0) {
#pragma unroll
            for (uint32_t k = 0; k < OPEN_PER; k++) { const uint32_t l0 = tid * lper + k, l = l0 < c.L ? l0 : c.L - 1u; lci[k] = c.dInS[lpar][l]; lco[k] = c.dOutS[lpar][l]; lcc[k] = c.dConvS[lpar][l]; }
        } else {
#pragma unroll
            for (uint32_t k = 0; k < NZQ; k++) { const uint32_t j0 = tid + k * TPB, j = j0 < c.zcap ? j0 : c.zcap - 1u; zv[k] = c.nz_val[j]; zl[k] = (uint32_t)c.nz_key[j]; zi[k] = c.nz_cin[j]; zo[k] = c.nz_cout[j]; zc[k] = c.nz_cconv[j]; }
        }
    }
    if (defer_wg) {                                       // (a fused sweep's marked list: this thread's first two places, its first class change of the sweep before - both parities)
        const uint32_t qa = dtid < c.mcap ? dtid : c.mcap - 1u, qb = dtid + G < c.mcap ? dtid + G : c.mcap - 1u;
        mxa = c.mk_idx[qa]; moa = c.mk_old[qa]; mna = c.mk_new[qa]; cda0 = c.chg_dw[0][qa]; cxa0 = c.chg_x[0][qa]; cda1 = c.chg_dw[1][qa]; cxa1 = c.chg_x[1][qa];
        mxb = c.mk_idx[qb]; mob = c.mk_old[qb]; mnb = c.mk_new[qb];
        if (dense_on) rseq0 = vrg_load_i64(&c.dctl[VD_RSEQ]);
    }
    VrgState s_ = *c.st;                                  // a copy (nf is only ever bumped atomically)
    if (s_.done || s_.bail) {
        if (st0 && c.st != c.stg) { vrg_state_store(c.stg, s_); c.inc[VC_NIN] = nin0; c.inc[VC_NOUT] = nout0; }     // (a fused trip swaps the state buffers whether it does anything or not)
        return;
    }
    // An OPEN-ENDED sweep ran on this state (vrg_items.h "open-ended sweeps"): what it ran on + what its workgroups added up.  Every
    // workgroup derives the closed state for itself - arithmetic on what it has just loaded; the pool's workgroups also list the levels
    // the sweep touched, each from the counters into its own LDS.
    const bool was_open = s_.open != 0;
    VrgFuseClosed fcl;
    uint32_t open_nnz = 0;
    if (was_open) {
        if (pool_wg) {
            __shared__ uint32_t s_oscan[TPB / 64];
            uint32_t cnt = 0;
            const int apar = (s_.iter + 1) & 1;           // (the set the sweep really filled: a wrong hint costs a round trip, never correctness)
            if (c.lvl_par < 0 || !direct_hint || lpar != apar) {
#pragma unroll
                for (uint32_t k = 0; k < OPEN_PER; k++) { const uint32_t l0 = tid * lper + k, l = l0 < c.L ? l0 : c.L - 1u; lci[k] = c.dInS[apar][l]; lco[k] = c.dOutS[apar][l]; lcc[k] = c.dConvS[apar][l]; }
            }
#pragma unroll
            for (uint32_t k = 0; k < OPEN_PER; k++) { if (k >= lper || tid * lper + k >= c.L) lci[k] = lco[k] = lcc[k] = 0; cnt += (lci[k] | lco[k] | lcc[k]) ? 1u : 0u; }
            uint32_t q = block_excl_scan(cnt, open_nnz, s_oscan);
#pragma unroll
            for (uint32_t k = 0; k < OPEN_PER; k++)
                if (lci[k] | lco[k] | lcc[k]) { s_nzl[q] = tid * lper + k; s_cin[q] = lci[k]; s_cout[q] = lco[k]; s_cconv[q] = lcc[k]; q++; }
        }
        vrg_fuse_close_core(c, s_, nin0, nout0, open_nnz, false, fcl);
        nin0 = fcl.n_in; nout0 = fcl.n_out;
    }
    const VrgState& s = s_;
    const bool live = s.iter < s.iterMax;
    if (st0 && live) { VRG_STAMP_PUT(c, 6, c.dbg[0]); VRG_STAMP_PUT(c, 0, t_entry); VRG_STAMP(c, 1); }    // (6: the sweep before this one)
    // What the fused sweep before this trip (k_sweep) left to do - nothing in this kernel reads a label: its label bytes in
    // place (+ the class bits the dense pass reads, the class changes of the sweep before that), its dead slots onto the
    // free list - by workgroups of their own (the last DEFER_WGS of the grid), beside the ones that decide the slots.
    // Whichever of them finishes last (ticket) asks for the sweep's dense pass.  Their first thread files the state this trip
    // works on (vrg_fuse_persist) - before its workgroup's ticket: the sizes the dense pass has to reproduce are filed with it.
    if (defer_wg) {
        if (dtid == 0) vrg_fuse_persist(c, s, fcl, was_open, nin0, nout0);
        if (!s.apply_pending) return;
        const int k = s.iter;
        if (tid == 0 && dense_on && (int64_t)k - 2 > rseq0) wait_dense_read_for(c, (int64_t)k - 2);   // (the pass of two sweeps ago has read the class copy this sweep rewrites)
        __syncthreads();
        if (dtid < s.ap_n) vrg_deferred_apply_vals(c, dtid, k, mxa, moa, mna);
        if (dtid + G < s.ap_n) vrg_deferred_apply_vals(c, dtid + G, k, mxb, mob, mnb);
        for (uint32_t i = dtid + 2u * G; i < s.ap_n; i += G) vrg_deferred_apply(c, i, k);
        const uint32_t nc = vrg_deferred_catchup_count(c, k), pp = ((uint32_t)k & 1u) ^ 1u;
        if (dtid < nc) { const uint32_t dw = pp ? cda1 : cda0, x = pp ? cxa1 : cxa0; if (dw != VRG_NOCHG) vrg_atomic_xor(&c.clsb[pp ^ 1u][dw], x); }
        for (uint32_t i = dtid + G; i < nc; i += G) vrg_deferred_catchup(c, i, k);
        for (uint32_t j = dtid; j < s.fr_n; j += G) vrg_deferred_free(c, s, j);
        band_deferred_done(c, k, defer_wgs);
        return;
    }
    // (many flips - a pool of hundreds of thousands of entries, thousands of pending slots: a workgroup lists its flips together, one bump of the flip
    // counter, vrg_decide_core)
    __shared__ VrgFlipSink s_sink;
    if (!pool_wg) {
        const bool sunk = s.nfx > EXACT_BLOCKS;           // (uniform; the branch of exact_wg that uses the sink)
        if (sunk) { if (tid == 0) s_sink.n = 0; __syncthreads(); }
        exact_wg(c, s, s.nfx, blockIdx.x - band_blocks, EXACT_BLOCKS, nin0, nout0, sunk ? &s_sink : nullptr);
        if (sunk) band_sink_file(c, s_sink, tid);
        if (stx && live) { vrg_drain(); VRG_STAMP_PUT(c, 3, t_entry); VRG_STAMP(c, 4); }
        return;
    }
    const bool direct = s.corr && !s.use_tab;
    VrgFlipSink* const sink = s.np > SINK_ABOVE ? &s_sink : nullptr;
    if (tid == 0) s_sink.n = 0;                           // (in place before any decision: the barriers below)
    if (!direct) {
        if (direct_hint) for (uint32_t j = tid; j < 3 * tab_n; j += TPB) s_raw[j] = c.tabC[j];      // (the hint was wrong: the memo head now)
        __syncthreads();                                  // (the memo head is in LDS)
        if (!direct_hint) {
            if (slot0 < s.np)
                vrg_item_band_fields(c, s, slot0, fl0, ip0, op0, lev0, idx0, key0, nin0, nout0, c.nz_val, c.nz_cin, c.nz_cout, c.nz_cconv, s_raw, tab_n, (double)err0, sink);
            for (uint32_t slot = slot0 + band_blocks * TPB; slot < s.np; slot += band_blocks * TPB)
                vrg_item_band(c, s, slot, c.nz_val, c.nz_cin, c.nz_cout, c.nz_cconv, s_raw, tab_n, nin0, nout0, sink);
        } else
            for (uint32_t slot = gtid; slot < s.np; slot += band_blocks * TPB)
                vrg_item_band(c, s, slot, c.nz_val, c.nz_cin, c.nz_cout, c.nz_cconv, s_raw, tab_n, nin0, nout0, sink);
        if (sink) band_sink_file(c, s_sink, tid);
        if (st0 && live) { vrg_drain(); VRG_STAMP(c, 2); }
        return;
    }
    __syncthreads();                                      // (everyone is done staging the memo head: the block changes hands)
    const double* nzv = c.nz_val; const uint32_t* nzi = c.nz_cin; const uint32_t* nzo = c.nz_cout; const uint32_t* nzc = c.nz_cconv;
    const bool nz_lds = s.nnz <= NZ_LDS;
    if (nz_lds && !was_open) {                            // (an open-ended sweep's list is in LDS already: this workgroup has just built it)
        if (direct_hint) {
#pragma unroll
            for (uint32_t k = 0; k < NZQ; k++) { const uint32_t j = tid + k * TPB; if (j < s.nnz) { if (use_ktab) s_nzl[j] = zl[k]; else s_val[j] = zv[k]; s_cin[j] = zi[k]; s_cout[j] = zo[k]; s_cconv[j] = zc[k]; } }
            for (uint32_t j = tid + NZQ * TPB; j < s.nnz; j += TPB) { if (use_ktab) s_nzl[j] = (uint32_t)c.nz_key[j]; else s_val[j] = c.nz_val[j]; s_cin[j] = c.nz_cin[j]; s_cout[j] = c.nz_cout[j]; s_cconv[j] = c.nz_cconv[j]; }
        } else
            for (uint32_t j = tid; j < s.nnz; j += TPB) { if (use_ktab) s_nzl[j] = (uint32_t)c.nz_key[j]; else s_val[j] = c.nz_val[j]; s_cin[j] = c.nz_cin[j]; s_cout[j] = c.nz_cout[j]; s_cconv[j] = c.nz_cconv[j]; }
    }
    __syncthreads();
    if (st0 && live) VRG_STAMP(c, 7);
    const uint32_t sub = tid & (LPE - 1);
    const uint32_t np_pad = (s.np + (TPB / LPE) - 1) / (TPB / LPE) * (TPB / LPE);      // whole waves stay in the loop together
    const uint32_t first = gtid / LPE;
    for (uint32_t slot = first; slot < np_pad; slot += band_blocks * TPB / LPE) {
        uint8_t fl = 0; double ip = 0, op = 0, v = 0; float err = 0; uint32_t lev = 0, idx = 0; uint64_t key = 0;
        const bool in_pool = slot < s.np, pre = direct_hint && slot == first;
        if (in_pool) {
            if (pre) { fl = fl0; ip = ip0; op = op0; err = err0; lev = lev0; idx = idx0; key = key0; }
            else { fl = c.p_flag[slot]; ip = c.p_ip[slot]; op = c.p_op[slot]; err = c.p_err[slot]; lev = c.p_lev[slot]; idx = c.p_idx[slot]; key = c.p_key[slot]; }   // one batch
        }
        const bool work = in_pool && (fl & PF_ALIVE) && !(fl & PF_PEND);
        const bool by_table = use_ktab && nz_lds;
        if (work && !by_table) v = c.lev[lev];
        double a = 0, bb = 0, d = 0;
        if (work) {
            if (by_table) {
                // kern(x_j - v) = ktab[lev][level of x_j]: eight look-ups in flight per turn - one turn for up to 128 touched
                // levels - each unconditional (index clamped: a load under a branch would wait for the ones before it)
                const double* row = c.ktab + (size_t)lev * c.L;
                const uint32_t last = s.nnz - 1u;
                constexpr int KQ = 128 / LPE < 16 ? 128 / LPE : 16;     // (one turn for up to 128 touched levels; 64 with 4 lanes per slot)
                for (uint32_t j0 = sub; j0 < s.nnz; j0 += KQ * LPE) {
                    uint32_t jj[KQ]; double kk[KQ];
#pragma unroll
                    for (int q = 0; q < KQ; q++) { const uint32_t j = j0 + q * LPE; jj[q] = j < s.nnz ? j : last; }
#pragma unroll
                    for (int q = 0; q < KQ; q++) kk[q] = row[s_nzl[jj[q]]];
#pragma unroll
                    for (int q = 0; q < KQ; q++)
                        if (j0 + q * LPE < s.nnz) { a += (double)s_cin[jj[q]] * kk[q]; bb += (double)s_cout[jj[q]] * kk[q]; d += (double)s_cconv[jj[q]] * kk[q]; }
                }
            } else if (nz_lds)
                for (uint32_t j = sub; j < s.nnz; j += LPE) {
                    const double k = vrg_kern(c, s_val[j] - v);
                    a += (double)s_cin[j] * k; bb += (double)s_cout[j] * k; d += (double)s_cconv[j] * k;
                }
            else
                for (uint32_t j = sub; j < s.nnz; j += LPE) {
                    const double k = vrg_kern(c, nzv[j] - v);
                    a += (double)nzi[j] * k; bb += (double)nzo[j] * k; d += (double)nzc[j] * k;
                }
        }
        a = group_sum<LPE>(a); bb = group_sum<LPE>(bb); d = group_sum<LPE>(d);
        if (sub == 0 && in_pool) {
            if ((fl & PF_ALIVE) && (fl & PF_PEND)) c.p_flag[slot] = (uint8_t)(fl & ~PF_PEND);   // decided by the exact half
            else if (work) {
                vrg_add_correction(a, bb, d, ip, op);
                c.p_ip[slot] = ip; c.p_op[slot] = op;
                if (s.iter < s.iterMax)              // while iterNum <= iterMax (:58)
                    vrg_decide_core(c, s, nin0, nout0, slot, fl & PF_INNER, ip, op, key, idx, lev, (double)err, sink);
            }
        }
    }
    if (sink) band_sink_file(c, s_sink, tid);
    if (st0 && live) { VRG_STAMP(c, 40); vrg_drain(); VRG_STAMP(c, 2); }
}

// ---- sorting inside one workgroup -------------------------------------------------------------------------
// ascending sort of n (key, value) pairs, keys distinct; n <= capacity of the arrays rounded up to a power of two
// (LDS arrays, or global ones for the rare long list).  All threads of the workgroup call it.
template <class K, class V>
__device__ void wg_sort_pairs(K* key, V* val, uint32_t n, bool has_val) {
    const uint32_t t = threadIdx.x, nt = blockDim.x;
    if (n <= 128u) {                                      // by counting: rank = number of smaller keys; two barriers
        K k = 0; V v = 0; uint32_t r = 0;
        if (t < n) {
            k = key[t]; if (has_val) v = val[t];
            for (uint32_t j = 0; j < n; j++) r += key[j] < k;
        }
        __syncthreads();
        if (t < n) { key[r] = k; if (has_val) val[r] = v; }
        __syncthreads();
        return;
    }
    uint32_t n2 = 1; while (n2 < n) n2 <<= 1;
    for (uint32_t i = n + t; i < n2; i += nt) key[i] = ~(K)0;
    __syncthreads();
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = t; i < n2; i += nt) {
                const uint32_t x = i ^ j;
                if (x > i) {
                    const K a = key[i], bb = key[x];
                    if ((a > bb) == ((i & k) == 0)) {
                        key[i] = bb; key[x] = a;
                        if (has_val) { const V va = val[i]; val[i] = val[x]; val[x] = va; }
                    }
                }
            }
            __syncthreads();
        }
}

// ---- update() as three launches -----------------------------------------------------------------------------
// What was measured on MI355X and shapes this (tools/latbench.hip, tools/icbench.hip, in-kernel phase stamps): a
// dependent kernel boundary costs 2.3 us; a dependent global load 0.1 us (L2) to 0.4 us (HBM); code that runs once per
// launch is fetched cold at ~30 ns per 64-B line; and ONE workgroup has four SIMDs - a whole update() inside one
// workgroup took 67 us, nearly all of it instruction issue (the label stencil of ~1600 marked voxels on 4 SIMDs).  So the
// per-voxel stencil runs on the whole chip, and only the steps that need to see every flip / every result - ordering
// the flips, closing the sweep - are single workgroups with compact code:
//   k_order        (1 workgroup)  stop tests (:91-104), flips sorted by list key in LDS = the reference's flip order
//                                 (:88), L/P bits + stamps, skip-rule prepass and fix-point
//   k_mark_relabel (chip-wide)    item = (flip, position of its 5x5x5 cube): the first marker of a voxel runs the label
//                                 stencil for it from the OLD labels and appends (voxel, new byte) to the marked list;
//                                 the stencil also files what the change means for the band pool, the class histograms
//                                 and the sweep's level deltas
//   k_close        (workgroup 0)  new label bytes in place (+ class bits, region sizes), dead slots onto the free list,
//                                 touched levels sorted, iterNum += 1, trace record;
//                  (workgroups 1..) per-level memo of the density corrections for the next k_band, one wave per level
// ---- the two edges between the streams, kept on the device -------------------------------------------------
// (a host event wait / record is a barrier packet of several microseconds in the stream; both conditions are almost
// always true already, so one thread looks at a word instead.)  Spins are bounded: a wait that does not end within
// SPIN_LIMIT raises an error instead of hanging the queue.
// band side, before the labels of sweep k are written into class copy k & 1: the recount k-2 has read that copy
__device__ __forceinline__ void wait_dense_read(const VrgCtx& c) { wait_dense_read_for(c, (int64_t)c.st->iter + 1 - 2); }
// dense side, in front of every recount: a sweep has been applied since the last recount - or the run has stopped and
// there is nothing to count.  True at once whenever the dense pass is what bounds the step (the band side runs a sweep
// ahead).  A kernel of its own (one thread), not the first thing in the recount: the recount's duration - what the HIP
// events and rocprofv3 report for the roofline - must be streaming time only, and workgroups that wait while holding
// LDS (the 16-bit variant: 64 KiB each) could leave no CU for k_close, which produces what they wait for.
__device__ __forceinline__ bool gate_dense_due(const VrgCtx& c) {
    const int64_t rseq = vrg_load_i64(&c.dctl[VD_RSEQ]);       // (only this stream changes it)
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if (vrg_load_i64(&c.gate[VG_REQ]) > rseq) return true;
        if (vrg_load_i64(&c.gate[VG_STOP])) return vrg_load_i64(&c.gate[VG_REQ]) > rseq;   // (the last applied sweep's request is older than the stop flag)
        __builtin_amdgcn_s_sleep(4);                          // (~0.1 us per look: one thread polls, two words of one cache line)
        if (wall_clock64() - t0 > SPIN_LIMIT) { c.dctl[VD_ERR] = 10; return false; }
    }
}

__global__ void k_wait_dense(VrgCtx c) { if (threadIdx.x == 0) wait_dense_read(c); }

constexpr int KO_THREADS = 256;
constexpr int TAB_BLOCKS = 256;     // k_close: memo workgroups (1024 waves, one level each at a time)
constexpr uint32_t NZ_SORT = 2048;  // touched levels one workgroup sorts in LDS

__global__ void __launch_bounds__(KO_THREADS) k_order(VrgCtx c, uint32_t small_limit) {
    VRG_CHAOS_POINT(2);
    if (threadIdx.x == 0) c.counters[48] = 0;             // (the flips k_mark_compact will leave to k_mark_relabel: none yet)

    constexpr uint32_t REC_LDS = 1024;
    __shared__ uint64_t s_key[NF_SMALL];
    __shared__ uint32_t s_slot[NF_SMALL];
    __shared__ uint32_t s_rslot[REC_LDS], s_ridx[REC_LDS], s_rlev[REC_LDS];
    __shared__ int s_go, s_changed;
    constexpr uint32_t T = KO_THREADS;
    const uint32_t t = threadIdx.x;
    // (this thread's first flip record travels with the state: k_band has appended it, whatever the state says)
    uint64_t k0 = 0; uint32_t rs0 = 0, ri0 = 0, rl0 = 0;
    const unsigned long long t_entry = t == 0 ? VRG_STAMP_NOW() : 0ull;
    if (t < c.fcap) { k0 = c.f_key[t]; rs0 = c.flist[t]; ri0 = c.fr_idx[t]; rl0 = c.fr_lev[t]; }
    // (... and so do the whole state, the region size the size stop looks at and this thread's first touched level: one
    // round trip for everything the kernel needs before it can order the flips)
    const VrgState s0 = *c.st;
    const int64_t nin0 = c.inc[VC_NIN];
    const uint64_t zk0 = t < c.zcap ? c.nz_key[t] : 0ull;
    if (s0.done || s0.bail) return;
    // (replication's per-sweep streaming: a four-launch trip's k_band may have no deferred workgroups to publish the change log's progress -
    // the sweep before this trip was closed by kernels that have ended: nothing to drain)
    if (t == 0) vrg_log_publish(c, s0.log_nsw, s0.log_pos, false);
    if (t == 0) {
        int go = 1;
        const int32_t stop = vrg_stop_test_v(s0, nin0);                  // :91-104, in the reference's order
        if (stop || s0.error) { c.stg->done = stop ? stop : -1; vrg_close_without_update(c); go = 0; }
        else {
            const int32_t bail = s0.nf > small_limit ? (int32_t)VBAIL_FLIPS : vrg_capacity_test(c, s0.nf);
            if (bail) { c.stg->bail = bail; vrg_close_without_update(c); go = 0; }
        }
        s_go = go;
    }
    __syncthreads();
    if (!s_go) return;
    if (t == 0) { VRG_STAMP_PUT(c, 8, t_entry); VRG_STAMP(c, 9); }
    const uint32_t nf = s0.nf;
    if (t < s0.nnz) { const uint32_t l = (uint32_t)zk0; c.dIn[l] = 0; c.dOut[l] = 0; c.dConv[l] = 0; c.ltouch[l] = 0; }   // level counters of the sweep before
    for (uint32_t j = t + T; j < s0.nnz; j += T) vrg_item_level_clear(c, j);
    // more flips than this workgroup orders in LDS: the chip-wide kernels behind it do the ordering (they are no-ops otherwise)
    if (nf > NF_ORDER) { if (t == 0) { vrg_open_update(c); c.stg->wide = 1; } return; }
    if (t == 0) c.stg->wide = 0;
    // the flips' records as k_band appended them; sorted by key, the payload being the record's number
    if (t < nf) { s_key[t] = k0; s_slot[t] = t; }
    for (uint32_t q = t + T; q < nf; q += T) { s_key[q] = c.f_key[q]; s_slot[q] = q; }
    // (a short list - the usual case - keeps the rest of every record in LDS too: no dependent look-up after the sort)
    const bool rec_lds = nf <= REC_LDS;
    if (rec_lds) {
        if (t < nf) { s_rslot[t] = rs0; s_ridx[t] = ri0; s_rlev[t] = rl0; }
        for (uint32_t q = t + T; q < nf; q += T) { s_rslot[q] = c.flist[q]; s_ridx[q] = c.fr_idx[q]; s_rlev[q] = c.fr_lev[q]; }
    }
    __syncthreads();
    if (t == 0) vrg_open_update(c);
    wg_sort_pairs(s_key, s_slot, nf, true);
    if (t == 0) VRG_STAMP(c, 10);
    for (uint32_t r = t; r < nf; r += T) {               // L (+P) bits, stamps, the ordered flip arrays
        const uint32_t q = s_slot[r];
        const bool inner = !(s_key[r] >> 63);
        if (rec_lds) vrg_item_list_rec(c, r, s_rslot[q], s_ridx[q], s_rlev[q], inner);
        else vrg_item_list_rec(c, r, c.flist[q], c.fr_idx[q], c.fr_lev[q], inner);
    }
    __syncthreads();
    if (t == 0) VRG_STAMP(c, 11);
    for (uint32_t r = t; r < nf; r += T) vrg_item_prepass(c, r);         // phase-A label of the flip-ins
    __syncthreads();
    if (t == 0) { vrg_drain(); VRG_STAMP(c, 12); }
    const uint32_t np_ = vrg_load_u32(&c.st->npend);
    if (np_) {                                                           // skip-rule fix-point (rare)
        for (;;) {
            __syncthreads();
            if (t == 0) s_changed = 0;
            __syncthreads();
            for (uint32_t j = t; j < np_; j += T) if (vrg_item_fix(c, j) == 2) s_changed = 1;
            __syncthreads();
            if (!s_changed) break;
        }
    }
}

// ---- k_order's work chip-wide, for sweeps of NF_SMALL .. NF_WIDE flips (VrgState::wide) --------------------------------------------
// rank of a flip = the number of smaller sort keys (the keys are distinct): every thread owns one record and looks at ALL keys, which
// pass through LDS a tile at a time (broadcast reads) - n^2 comparisons, 4 * 10^9 at 65 536 flips: ~0.1 ms on the chip; 10^8 at 10^4
// flips: a few microseconds.  No global sort, no host.  Then the flip's L (+P) bits, stamp and the ordered flip arrays (vrg_item_list_rec).
constexpr int KR_THREADS = 256;
constexpr uint32_t KR_TILE = 512;
// A workgroup = 256 records x one tile of 512 keys; the tiles of a record block run side by side on the chip and add what they find
// to the record's count (rk_part - distinct words, nothing returns).  (One workgroup per record block walking ALL tiles: 51 workgroups
// at 12 900 flips, each issuing 12 900 broadcast LDS reads per wave - 0.19 ms on a fifth of the chip.)
__global__ void __launch_bounds__(KR_THREADS) k_rank_wide(VrgCtx c) {
    __shared__ uint64_t s_k[KR_TILE];
    const VrgState& s = *c.st;
    if (s.done || s.bail || !s.wide) return;
    const uint32_t nf = min(s.nf, c.fcap), t = threadIdx.x;
    const uint32_t nib = (nf + KR_THREADS - 1) / KR_THREADS, ntl = (nf + KR_TILE - 1) / KR_TILE;
    for (uint32_t w = blockIdx.x; w < nib * ntl; w += gridDim.x) {        // (whole workgroups stay in the loop together: the tile barriers)
        const uint32_t i = (w / ntl) * KR_THREADS + t, j0 = (w % ntl) * KR_TILE;
        const uint64_t key = i < nf ? c.f_key[i] : 0ull;
        __syncthreads();
        for (uint32_t j = t; j < KR_TILE; j += KR_THREADS) s_k[j] = j0 + j < nf ? c.f_key[j0 + j] : ~0ull;
        __syncthreads();
        uint32_t r = 0;
#pragma unroll 16
        for (uint32_t j = 0; j < KR_TILE; j++) r += s_k[j] < key;         // (the padding keys are larger than every key)
        if (i < nf && r) atomicAdd(&c.rk_part[i], r);
    }
}
// ... and the flip's L (+P) bits, stamp and place in the ordered flip arrays, once every tile has reported (a kernel boundary)
__global__ void __launch_bounds__(TPB) k_list_wide(VrgCtx c) {
    const VrgState& s = *c.st;
    if (s.done || s.bail || !s.wide) return;
    ITEM_LOOP(min(s.nf, c.fcap)) {
        const uint32_t r = c.rk_part[i];
        c.rk_part[i] = 0;                                 // (all zero again for the next sweep)
        vrg_item_list_rec(c, r, c.flist[i], c.fr_idx[i], c.fr_lev[i], !(c.f_key[i] >> 63));
    }
}
// ... or, for a sweep with more flips than n^2 comparisons are worth (host-driven trips: the host knows the count and has sized a radix sort): the rank of the flip
// the sort put at place r is r
__global__ void __launch_bounds__(TPB) k_rank_scatter(VrgCtx c, const uint32_t* __restrict__ perm, uint32_t nf) {
    const VrgState& s = *c.st;
    if (s.done || s.bail || !s.wide) return;
    ITEM_LOOP(min(nf, c.fcap)) c.rk_part[perm[i]] = i;
}
__global__ void __launch_bounds__(TPB) k_prepass_wide(VrgCtx c) {               // phase-A label of the flip-ins (every L bit is in place: a kernel boundary)
    const VrgState& s = *c.st;
    if (s.done || s.bail || !s.wide) return;
    ITEM_LOOP(min(s.nf, c.fcap)) vrg_item_prepass(c, i);
}
__global__ void __launch_bounds__(1024) k_fix_wide(VrgCtx c) {                  // skip-rule fix-point (rare), one workgroup
    __shared__ int changed;
    const VrgState& s = *c.st;
    if (s.done || s.bail || !s.wide) return;
    const uint32_t np_ = s.npend;
    if (np_ == 0) return;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < np_; j += blockDim.x) if (vrg_item_fix(c, j) == 2) changed = 1;
        __syncthreads();
        if (!changed) break;
    }
}

constexpr uint32_t LEV_LDS = 2048;  // level values a workgroup of k_mark_relabel keeps in LDS
constexpr uint32_t HIST_LDS = 1024; // ... and level tables up to this size: its changes to the five per-level counters
// k_mark_relabel: ONE flip per workgroup and trip - thread p < 125 is place p of the flip's 5x5x5 cube.  Everything the
// stencils of those 125 voxels read of the LABELS lies within 4 voxels of the flip: the workgroup fetches that 9x9x9
// neighbourhood once - 81 rows of 16 bytes, one load each for 81 threads - into LDS, and the nine 3-byte rows of a voxel's
// 3x3x3 masks, its own byte and the 25 rows of an excluded voxel's 2-ring come from there.  (Round 2 / early round 3:
// every thread fetched its own 9 + 25 rows from memory, 34 scattered requests per lane through one address unit per CU -
// the "label byte + preload" phase alone took 3.1 us - and the 2-ring and the neighbour ranks were two more dependent round
// trips inside the case analysis.)  What stays per voxel - its stamp rank, slot and intensity - is requested together with
// the tile; the ranks of its listed neighbours travel together with the marking atomic.
constexpr int KM_THREADS = 128;
constexpr int KM_BLOCKS = 512;
constexpr int KM_BLOCKS_WIDE = 512;    // ... of a sweep with more flips than that (four flips at a time each: all resident, 2048 flips in flight)
typedef uint32_t km_u4 __attribute__((ext_vector_type(4)));
constexpr int KM_ROWS = 81;         // (dy, dz) in [-4, 4]^2; row bytes 0..8 = dx -4..+4 (16 bytes are fetched)
constexpr uint32_t KM_MKBUF = 512, KM_EVBUF = 256;      // marked voxels / events a workgroup keeps in LDS between two filings, per flip it handles at a time (a flip adds at most 125 of each)
struct KmEvRec { VrgEvent ev; uint32_t m, r1, rf; };    // a buffered event: its voxel, its number among the workgroup's events of its kind (new or dead) and among the pending ones
__device__ __forceinline__ uint32_t km_row(int dy, int dz) { return (uint32_t)((dz + 4) * 9 + (dy + 4)); }
// G: the flips a workgroup handles side by side, 128 threads each (1: a sweep of up to KM_BLOCKS flips, a workgroup per flip; 4: thousands
// of flips - every round trip of a flip's chain then serves four, and a workgroup files what ~25 flips add to the lists at once)
// (136 registers with the list indirection - one 512-thread workgroup per CU; it only takes what k_mark_compact leaves.  At 10^4 flips per sweep the kernel's time was its FILING - six same-address reservations per
// workgroup, queueing at the memory side - not its rounds: VrgCtx::rsv and k_mark_compact, below; profiles/NOTES_r05.md)
#if defined(VRG_STAMPS)
#define KM_STAMP_OCC __attribute__((amdgpu_waves_per_eu(4, 4)))     // (the stamps cost registers: keep the product's two 512-thread workgroups per CU, or the timeline is another kernel's)
#else
#define KM_STAMP_OCC
#endif
// A workgroup reserves its stretches of the sweep's lists: new / dead / pending events (s_n), marked voxels (s_cnt[0]), list length changes (s_d) - through VrgCtx::rsv
// (two returning 64-bit adds and two plain ones, on four cache lines) or, without it, through the state's own six words.  Bases into s_base[0..3].
__device__ __forceinline__ void km_reserve(const VrgCtx& c, uint32_t tt, const uint32_t* s_n, const uint32_t* s_cnt, const int32_t* s_d, uint32_t* s_base) {
    if (c.rsv) {
        if (tt == 0 && (s_n[0] | s_n[1])) { const unsigned long long o = atomicAdd((unsigned long long*)&c.rsv[0], (unsigned long long)s_n[0] | ((unsigned long long)s_n[1] << 32)); s_base[0] = (uint32_t)o; s_base[1] = (uint32_t)(o >> 32); }
        if (tt == 1 && (s_n[2] | s_cnt[0])) { const unsigned long long o = atomicAdd((unsigned long long*)&c.rsv[16], (unsigned long long)s_n[2] | ((unsigned long long)s_cnt[0] << 32)); s_base[2] = (uint32_t)o; s_base[3] = (uint32_t)(o >> 32); }
        if (tt == 2 && s_d[0]) atomicAdd((int*)&c.rsv[32], s_d[0]);
        if (tt == 3 && s_d[1]) atomicAdd((int*)&c.rsv[48], s_d[1]);
        return;
    }
    if (tt < 3 && s_n[tt]) s_base[tt] = vrg_atomic_add(tt == 0 ? &c.st->nalloc : tt == 1 ? &c.st->ndead : &c.st->nfresh, s_n[tt]);
    if (tt == 3 && s_cnt[0]) s_base[3] = vrg_atomic_add(&c.stg->nmk, s_cnt[0]);
    if (tt >= 4 && tt < 6 && s_d[tt - 4]) vrg_atomic_add(tt == 4 ? &c.st->d_ni : &c.st->d_no, s_d[tt - 4]);
}
// (list / list_n: null - every flip of the sweep, flip r = the r-th of the ordered list; else the flips k_mark_compact left to this kernel)
template <int G>
__global__ void __launch_bounds__(KM_THREADS * G) KM_STAMP_OCC k_mark_relabel(VrgCtx cg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_n) {
    VRG_CHAOS_POINT(3);
    // (the first flip's voxel travels with the state: k_order has written the list, whatever the state says)
    const uint32_t tt = threadIdx.x, g = tt / KM_THREADS, t = tt % KM_THREADS;     // (tt: in the workgroup; t: among the 128 threads of flip g)
    const bool st0 = blockIdx.x == 0 && tt == 0;
    const unsigned long long t_entry = tt == 0 ? VRG_STAMP_NOW() : 0ull;
    const uint32_t r_first = list ? 0xffffffffu : blockIdx.x * G + g;
    const uint32_t fidx_first = r_first < cg.fcap ? cg.f_idx[r_first] : 0u;
    const int32_t st_done = cg.st->done, st_bail = cg.st->bail;
    const uint32_t nf_all = cg.st->nf, nlist = list ? *list_n : 0u;
    asm volatile("" :: "v"(fidx_first), "v"(st_done), "v"(st_bail), "v"(nf_all), "v"(nlist));     // one wait for the five
    const uint32_t nf = list ? (nlist < nf_all ? nlist : nf_all) : nf_all;
    if (st_done || st_bail) return;
    if (list && blockIdx.x == 0 && tt == 0 && nf) atomicAdd(&cg.counters[49], nf);       // (diagnostics: flips the compact kernel left to this one, since the handle was created)
    if (st0) { VRG_STAMP_PUT(cg, 16, t_entry); VRG_STAMP(cg, 17); }
#if defined(VRG_STAMPS)
    if (list) { if (blockIdx.x * G >= nf) return; }                  // (the compact kernel's stamps stay when this launch has nothing to do)
    VRG_STAMP_WG_PUT(cg, 0, t_entry); VRG_STAMP_WG(cg, 1);
    VRG_STAMP_WG_PUT(cg, 15, ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg(63492));   // XCC_ID | HW_ID
    for (uint32_t k_ = 2; k_ < (uint32_t)VRG_DBG_PER; k_++) if (k_ != 15u) VRG_STAMP_WG_PUT(cg, k_, 0ull);
#endif
    if (blockIdx.x * G >= nf) return;                                     // (no flip for this workgroup)
    // a voxel that enters the band needs the level index of its intensity: a binary search, i.e. log2(L) DEPENDENT loads -
    // from LDS when the table fits
    extern __shared__ double s_lev[];                                     // (L doubles when the table fits LEV_LDS - the launch sizes it - else nothing)
    __shared__ uint32_t s_tile_all[G][KM_ROWS * 4];
    uint32_t* const s_tile = s_tile_all[g];
    // What the workgroup's flips add to the sweep's lists - marked voxels, new / dead / pending events, list length changes - is kept in
    // LDS and filed in one go (km_flush): ONE reservation per list and workgroup, not one per flip.  Every reservation is an atomic on
    // one of a few words of the state, and those execute one after the other in L2 (~8 ns each): at 12 900 flips per sweep and seven
    // per flip they WERE the kernel (850 us).
    __shared__ uint32_t s_n[3], s_base[4], s_cnt[2];                      // events by kind since the last flush; bases (new, dead, pending, marked); buffered marked voxels / events
    __shared__ int32_t s_d[2];                                            // list length changes since the last flush
    constexpr uint32_t MKBUF = KM_MKBUF * G, EVBUF = KM_EVBUF * G, NT = KM_THREADS * G;
    __shared__ uint32_t s_mk_idx[MKBUF];
    __shared__ uint8_t s_mk_nw[MKBUF], s_mk_old[MKBUF];
    __shared__ KmEvRec s_ev[EVBUF];
    VrgCtx c = cg;
    uint8_t* lab = c.lab[0];
    const uint32_t idx_lo = vrg_idx(c, 0, 0, 0), idx_hi = vrg_idx(c, c.nx - 1, c.ny - 1, c.nz - 1);
    const uint32_t p = t;
    const int dx = (int)(p % 5) - 2, dy = (int)((p / 5) % 5) - 2, dz = (int)(p / 25) - 2;     // vrg_mark_pos
    const int ry = (int)(t % 9) - 4, rz = (int)(t / 9) - 4;                                   // the tile row thread t < 81 fetches
    c.lev_fast = (cg.L <= LEV_LDS || cg.lev16 || cg.lev_map || cg.lidx) ? 1 : 0;   // (the direct map / the per-voxel level index: one load, requested with the rest)
    if (cg.L <= LEV_LDS) c.lev_map = nullptr;                              // (a table in LDS needs no load at all)
    if (tt < 3) s_n[tt] = 0;
    if (tt < 2) { s_d[tt] = 0; s_cnt[tt] = 0; }                           // (in place before anyone counts: the tile barrier of the first flip)
    // The class histograms (vrg_hist_change) and this sweep's innerAdded / outerAdded / addedPoints by level (vrg_note_level): counted in
    // LDS, added to the global counters when the workgroup is done - the voxels of a vessel share a handful of levels, and ten thousand
    // flips bumping those few words one after the other in L2 is what this kernel would otherwise wait for (level tables up to HIST_LDS
    // levels whose touched levels are found by scanning, lvl_scan 1; nothing in this kernel reads the counters)
    const bool lds_hist = cg.lvl_scan == 1 && cg.L <= HIST_LDS;
    uint32_t* const s_hist = reinterpret_cast<uint32_t*>(s_lev + ((cg.L <= LEV_LDS && !cg.lev16) ? cg.L : 0u));
    if (lds_hist) {
        for (uint32_t l = tt; l < 5u * cg.L; l += NT) s_hist[l] = 0;
        c.dIn = s_hist; c.dOut = s_hist + cg.L; c.dConv = s_hist + 2u * cg.L;
        c.hin = reinterpret_cast<int32_t*>(s_hist + 3u * cg.L); c.hout = reinterpret_cast<int32_t*>(s_hist + 4u * cg.L);
    }
    auto km_flush = [&]() {                                               // all threads; the buffers are complete (a barrier since the last entry)
        km_reserve(c, tt, s_n, s_cnt, s_d, s_base);
        __syncthreads();
        const uint32_t nm = s_cnt[0], ne = s_cnt[1];
        for (uint32_t i = tt; i < nm; i += NT) {
            const uint32_t q = s_base[3] + i;
            if (q < c.mcap) { c.mk_idx[q] = s_mk_idx[i]; c.mk_new[q] = s_mk_nw[i]; c.mk_old[q] = s_mk_old[i]; } else vrg_store_i32(&c.stg->error, 4);
        }
        for (uint32_t i = tt; i < ne; i += NT) {
            const KmEvRec& e = s_ev[i];
            vrg_ev_write(c, e.m, e.ev, s_base[0] + e.r1, s_base[1] + e.r1, s_base[2] + e.rf);
        }
        __syncthreads();
        if (tt < 3) s_n[tt] = 0;
        if (tt < 2) { s_d[tt] = 0; s_cnt[tt] = 0; }
        __syncthreads();
    };
    // (every thread of the workgroup makes the same number of trips: the barriers)
    for (uint32_t rb = blockIdx.x * G; rb < nf; rb += gridDim.x * G) {
        const uint32_t ri = rb + g;
        const bool have = ri < nf;                                        // (the last round of a sweep may leave some of the G places empty)
        const uint32_t r = list ? list[have ? ri : nf - 1u] : (have ? ri : nf - 1u);
        const uint32_t fidx = r == r_first ? fidx_first : c.f_idx[r];
        if (rb != blockIdx.x * G && (s_cnt[0] + 125u * G > MKBUF || s_cnt[1] + 125u * G > EVBUF)) km_flush();     // (uniform: read after the barrier that ended the round before)
        // the tile row (a row that is not wholly inside the allocation - 16 guard bytes at either end - belongs to no real
        // voxel's neighbourhood: it reads as out-of-bounds bytes)
        km_u4 row = {0x01010101u * VB_OOB, 0x01010101u * VB_OOB, 0x01010101u * VB_OOB, 0x01010101u * VB_OOB};
        if (t < KM_ROWS) {
            const int64_t a = (int64_t)fidx + ((int64_t)rz * c.PY + ry) * c.PX - 4;
            if (a >= -16 && a + 16 <= (int64_t)c.PV + 16) row = __builtin_nontemporal_load(reinterpret_cast<const km_u4*>(lab + a));
        }
        // this thread's voxel and what is kept per voxel elsewhere (a position outside the real volume is padding - never
        // relabelled - so its index is clamped to stay inside the arrays)
        const int64_t m = (int64_t)fidx + ((int64_t)dz * c.PY + dy) * c.PX + dx;
        VrgPre pre;
        pre.rank = 0; pre.vent = 0; pre.lev16 = 0; pre.val = 0.0;
        if (p < 125u && have) {
            const uint32_t ms = (uint32_t)(m < (int64_t)idx_lo ? (int64_t)idx_lo : (m > (int64_t)idx_hi ? (int64_t)idx_hi : m));
            pre.rank = (uint32_t)c.stamp[ms]; pre.vent = c.vent[ms];
            pre.lev16 = c.lev16 ? (uint32_t)c.lev16[ms] : c.lidx ? c.lidx[ms] : 0u;
            pre.val = (c.lev16 || c.lidx) ? 0.0 : vrg_voxel_value(c, ms);
        }
        if (rb == blockIdx.x * G && cg.L <= LEV_LDS && !cg.lev16) {      // (its loads queue behind those: one wait covers both)
            for (uint32_t l = tt; l < cg.L; l += NT) s_lev[l] = cg.lev[l];
            c.lev = s_lev;
        }
        if (t < KM_ROWS) { s_tile[4 * t] = row.x; s_tile[4 * t + 1] = row.y; s_tile[4 * t + 2] = row.z; s_tile[4 * t + 3] = row.w; }
        __syncthreads();
        if (st0) { vrg_drain(); VRG_STAMP(c, 18); }
#if defined(VRG_STAMPS)
        const uint32_t rnd_ = (rb - blockIdx.x * G) / (gridDim.x * G);          // (phases of rounds 0 and 2, every workgroup: words 16.. / 20..)
        const uint32_t ph_ = rnd_ == 0u ? 16u : rnd_ == 2u ? 20u : 64u;
        VRG_STAMP_WG(c, ph_);
#endif
        uint8_t mb = VB_OOB;
        if (p < 125u && have) { const uint32_t o = (uint32_t)(dx + 4); mb = (uint8_t)(s_tile[4 * km_row(dy, dz) + (o >> 2)] >> (8u * (o & 3u))); }
        const bool wanted = vrg_mark_wanted(p, mb);
        // the mark (its answer says whether this thread is the voxel's first marker) and the ranks of the listed neighbours
        // leave together; the tile work below runs while they travel
        uint32_t old = 0xffffffffu;
        const uint32_t sh = 8u * ((uint32_t)m & 3u);
        if (wanted) old = vrg_atomic_or((uint32_t*)(lab + ((uint32_t)m & ~3u)), (uint32_t)VB_M << sh);
        VrgNbr nb = {0u, 0u, 0u, 0u};
        uint32_t FO = 0, AP = 0, cand = 0, n0[VRG_RANK_BATCH], r0[VRG_RANK_BATCH];
        if (wanted) {
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const uint64_t w8 = *reinterpret_cast<const uint64_t*>(&s_tile[4 * km_row(dy + j / 3 - 1, dz + j % 3 - 1)]);
                pre.w[j] = (uint32_t)(w8 >> (8 * (dx + 3)));              // bytes x-1 .. x+2 of the row (vrg_preload)
            }
            nb = vrg_masks_of(pre.w);
            uint32_t ex, segA; vrg_nbr_sets(nb, ex, segA, FO, AP);
            cand = FO | AP;
        }
#pragma unroll
        for (int k = 0; k < VRG_RANK_BATCH; k++) { n0[k] = 32u; r0[k] = 0u; }
        if (cand) vrg_rank_batch(c, cand, (uint32_t)m, n0, r0);
        bool ring2 = false;
        if (wanted && vrg_wants_ring2(mb, nb)) {                          // an applied flip (P and not OOB) within the 2-ring? (vrg_ring2_applied)
            uint64_t any = 0;
            const uint32_t o2 = (uint32_t)(dx + 2);
#pragma unroll 5                                                       // (all 25 rows unrolled: 176 registers instead of 126 - one workgroup of four flips per CU instead of two)
            for (int j = 0; j < 25; j++) {
                const uint32_t* rw = &s_tile[4 * km_row(dy + j % 5 - 2, dz + j / 5 - 2)];
                const uint64_t lo8 = *reinterpret_cast<const uint64_t*>(rw);
                const uint64_t w8 = o2 ? (lo8 >> (8u * o2)) | ((uint64_t)rw[2] << (64u - 8u * o2)) : lo8;    // bytes x-2 .. x+2
                any |= ((w8 >> 4) & ~(w8 >> 5)) & 0x0101010101ull;
            }
            ring2 = any != 0;
        }
        const uint32_t lev_here = (wanted && c.lev_fast) ? vrg_pre_level(c, pre) : 0xffffffffu;
        const bool first = wanted && !((old >> sh) & VB_M);
        if (st0) { vrg_drain(); VRG_STAMP(c, 19); }
        // the first marker's voxel and its event take their places in the workgroup's buffers (one wave-wide count each: LDS atomics)
        if (st0) VRG_STAMP(c, 23);
#if defined(VRG_STAMPS)
        VRG_STAMP_WG(c, ph_ + 1u);                                         // (wave 0 knows who is first: the mark atomics are back)
#endif
        if (first) {
            VrgEvent ev; ev.kind = VE_NONE; ev.pend = 0;
            VrgRanks qr; vrg_ranks_none(qr);
            vrg_ranks_take(FO, n0, r0, qr);
            while (cand) { vrg_rank_batch(c, cand, (uint32_t)m, n0, r0); vrg_ranks_take(FO, n0, r0, qr); }     // (more than four listed neighbours: rare)
            const uint8_t nw = vrg_sweep_cases(c, (uint32_t)m, mb, pre, nb, qr, ring2, lev_here, ev);   // (L / P bits date from k_order: mb is current)
            const uint32_t qm = atomicAdd(&s_cnt[0], 1u);
            s_mk_idx[qm] = (uint32_t)m; s_mk_nw[qm] = nw; s_mk_old[qm] = mb;
            if (ev.kind != VE_NONE) {
                uint32_t r1 = 0, rf = 0;
                if (ev.kind == VE_NEW) r1 = atomicAdd(&s_n[0], 1u);
                if (ev.kind == VE_DIE) r1 = atomicAdd(&s_n[1], 1u);
                if (ev.kind != VE_DIE && ev.pend) rf = atomicAdd(&s_n[2], 1u);
                const int di = vrg_ev_dni(ev), dq = vrg_ev_dno(ev);
                if (di) atomicAdd(&s_d[0], di);
                if (dq) atomicAdd(&s_d[1], dq);
                KmEvRec& e = s_ev[atomicAdd(&s_cnt[1], 1u)];
                e.ev = ev; e.m = (uint32_t)m; e.r1 = r1; e.rf = rf;
            }
        }
#if defined(VRG_STAMPS)
        VRG_STAMP_WG(c, ph_ + 2u);                                         // (wave 0's own stencils are done)
#endif
        __syncthreads();                                                  // (the buffers are consistent; the tile may be overwritten)
        if (st0) { vrg_drain(); VRG_STAMP(c, 20); }
        VRG_STAMP_WG(c, min(11u, 2u + (rb - blockIdx.x * G) / (gridDim.x * G)));
    }
    VRG_STAMP_WG(c, 12);
    km_flush();
    VRG_STAMP_WG(c, 13);
    if (lds_hist)                                                         // (km_flush ends with a barrier: the counts are complete)
        for (uint32_t l = tt; l < 5u * cg.L; l += NT) {
            const uint32_t n = s_hist[l];                                 // (the histograms' changes are signed: the same bits)
            if (n) { const uint32_t k = l / cg.L, lv = l - k * cg.L; atomicAdd((k == 0 ? cg.dIn : k == 1 ? cg.dOut : k == 2 ? cg.dConv : k == 3 ? (uint32_t*)cg.hin : (uint32_t*)cg.hout) + lv, n); }
        }
    if (st0) { vrg_drain(); VRG_STAMP(c, 21); }
    if (t == 0) { vrg_drain(); VRG_STAMP_MAX(c, 22); }
#if defined(VRG_STAMPS)
    if (tt == 0) vrg_drain();
#endif
    VRG_STAMP_WG(c, 14);
}

// ---- k_mark_relabel in its COMPACT form (round 6): sweeps of thousands of flips -------------------------------------------------
// What the per-workgroup timeline of k_mark_relabel<4> showed at 12 900 flips per sweep (tools/mark_stamps.py, profiles/NOTES_r05.md): a
// round of four flips takes ~10 us whatever its loads look like, 2048 flips are in flight chip-wide, and the kernel is bound by the NUMBER of
// scattered requests a flip makes - 81 tile rows, 3 x 125 per-voxel fields, 125 marking atomics, the rank look-ups - of which most serve
// places that are not wanted at all: only the 27 voxels of the flip's 1-ring are ever relabelled unless an EXCLUDED voxel lies in its 2-ring
// (:177-179), and inside a brain mask none does.  So:
//  * ONE FLIP PER HALF-WAVE: lane l < 27 is place l of the flip's 3x3x3 box, lane l < 25 fetches row l of its 5x5 rows (all the 27 stencils
//    read: 5x5x5 voxels); a 256-thread workgroup handles eight flips side by side.  A flip with an excluded voxel anywhere in its 5x5x5 cube
//    (found in the rows by a wave ballot) is left untouched and put on the `slow` list: k_mark_relabel<4> - the general form, launched behind
//    this kernel over that list - takes it.
//  * NOTHING A FLIP NEEDS IS SHARED BETWEEN WAVES: tile and rank tile belong to the half-wave, so a round has NO workgroup barrier - a wave
//    runs through its rounds at its own pace.  The ranks of the listed flips in the cube (what the case analysis asks of a voxel's listed
//    neighbours) are fetched once per listed voxel into the rank tile, not once per (voxel, neighbour) pair.
//  * the workgroup's lists (marked voxels, events) are filed every two rounds and at the end, as k_mark_relabel<4> files them.
// Requests per flip: 25 rows + 27 x 3 fields + the cube's listed stamps + 27 marks, against 81 + 375 + 125 + the rank batches.
constexpr int KMC_GROUPS = 8;
constexpr int KMC_THREADS = 32 * KMC_GROUPS;
constexpr int KMC_ROWS = 25;                    // (dy, dz) in [-2, 2]^2; row bytes 0..15 = dx -4 .. +11 (bytes 2..6 are the cube's)
constexpr uint32_t KMC_FLUSH_ROUNDS = 2;
constexpr uint32_t KMC_BUF = KMC_FLUSH_ROUNDS * KMC_GROUPS * 27;     // marked voxels / events between two filings
constexpr int KMC_BLOCKS = 768;                 // three workgroups per CU: 6144 flips in flight
__global__ void __launch_bounds__(KMC_THREADS) k_mark_compact(VrgCtx cg, uint32_t* __restrict__ slow, uint32_t* __restrict__ slow_n) {
    VRG_CHAOS_POINT(3);
    const uint32_t tt = threadIdx.x, g = tt >> 5, l = tt & 31u;
    const uint32_t r_first = blockIdx.x * KMC_GROUPS + g;
    const uint32_t fidx_first = r_first < cg.fcap ? cg.f_idx[r_first] : 0u;
    const int32_t st_done = cg.st->done, st_bail = cg.st->bail;
    const uint32_t nf = cg.st->nf;
    asm volatile("" :: "v"(fidx_first), "v"(st_done), "v"(st_bail), "v"(nf));     // one wait for the four
    if (st_done || st_bail) return;
#if defined(VRG_STAMPS)
    VRG_STAMP_WG(cg, 0); VRG_STAMP_WG(cg, 1);
    VRG_STAMP_WG_PUT(cg, 15, ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg(63492));
    for (uint32_t k_ = 2; k_ < (uint32_t)VRG_DBG_PER; k_++) if (k_ != 15u) VRG_STAMP_WG_PUT(cg, k_, 0ull);
#endif
    if (blockIdx.x * KMC_GROUPS >= nf) return;
    extern __shared__ double s_lev[];                                     // (the level table + the per-level counts, as k_mark_relabel keeps them)
    __shared__ uint32_t s_tile_all[KMC_GROUPS][KMC_ROWS * 4];
    __shared__ uint32_t s_rank_all[KMC_GROUPS][128];                      // rank of the listed flip at place p of the 5x5x5 cube
    __shared__ uint32_t s_n[3], s_base[4], s_cnt[2];
    __shared__ int32_t s_d[2];
    __shared__ uint32_t s_mk_idx[KMC_BUF];
    __shared__ uint8_t s_mk_nw[KMC_BUF], s_mk_old[KMC_BUF];
    __shared__ KmEvRec s_ev[KMC_BUF];
    uint32_t* const s_tile = s_tile_all[g];
    uint32_t* const s_rank = s_rank_all[g];
    VrgCtx c = cg;
    uint8_t* lab = c.lab[0];
    const uint32_t idx_lo = vrg_idx(c, 0, 0, 0), idx_hi = vrg_idx(c, c.nx - 1, c.ny - 1, c.nz - 1);
    const bool ring = l < 27u;
    const int dx = (int)(l % 3u) - 1, dy = (int)((l / 3u) % 3u) - 1, dz = ring ? (int)(l / 9u) - 1 : 0;      // this lane's voxel of the 3x3x3 box
    const int ry = (int)(l % 5u) - 2, rz = l < (uint32_t)KMC_ROWS ? (int)(l / 5u) - 2 : 0;                    // the row lane l < 25 fetches
    c.lev_fast = (cg.L <= LEV_LDS || cg.lev16 || cg.lev_map || cg.lidx) ? 1 : 0;
    if (cg.L <= LEV_LDS) c.lev_map = nullptr;
    if (tt < 3) s_n[tt] = 0;
    if (tt < 2) { s_d[tt] = 0; s_cnt[tt] = 0; }
    const bool lds_hist = cg.lvl_scan == 1 && cg.L <= HIST_LDS;
    uint32_t* const s_hist = reinterpret_cast<uint32_t*>(s_lev + ((cg.L <= LEV_LDS && !cg.lev16) ? cg.L : 0u));
    if (lds_hist) {
        for (uint32_t k = tt; k < 5u * cg.L; k += KMC_THREADS) s_hist[k] = 0;
        c.dIn = s_hist; c.dOut = s_hist + cg.L; c.dConv = s_hist + 2u * cg.L;
        c.hin = reinterpret_cast<int32_t*>(s_hist + 3u * cg.L); c.hout = reinterpret_cast<int32_t*>(s_hist + 4u * cg.L);
    }
    auto km_flush = [&]() {                                               // all threads (a barrier first: every wave has finished its rounds so far)
        __syncthreads();
        km_reserve(c, tt, s_n, s_cnt, s_d, s_base);
        __syncthreads();
        const uint32_t nm = s_cnt[0], ne = s_cnt[1];
        for (uint32_t i = tt; i < nm; i += KMC_THREADS) {
            const uint32_t q = s_base[3] + i;
            if (q < c.mcap) { c.mk_idx[q] = s_mk_idx[i]; c.mk_new[q] = s_mk_nw[i]; c.mk_old[q] = s_mk_old[i]; } else vrg_store_i32(&c.stg->error, 4);
        }
        for (uint32_t i = tt; i < ne; i += KMC_THREADS) {
            const KmEvRec& e = s_ev[i];
            vrg_ev_write(c, e.m, e.ev, s_base[0] + e.r1, s_base[1] + e.r1, s_base[2] + e.rf);
        }
        __syncthreads();
        if (tt < 3) s_n[tt] = 0;
        if (tt < 2) { s_d[tt] = 0; s_cnt[tt] = 0; }
        __syncthreads();
    };
    uint32_t round = 0;
    for (uint32_t rb = blockIdx.x * KMC_GROUPS; rb < nf; rb += gridDim.x * KMC_GROUPS, round++) {      // (every wave of the workgroup makes the same number of trips)
        const uint32_t r = rb + g;
        const bool have = r < nf;
        const uint32_t fidx = r == r_first ? fidx_first : c.f_idx[have ? r : nf - 1u];
        if (round && round % KMC_FLUSH_ROUNDS == 0u) { VRG_STAMP_WG(c, 16); km_flush(); VRG_STAMP_WG(c, 17); }
        // this lane's row of the cube and, for a lane of the 3x3x3 box, what is kept per voxel elsewhere - one batch
        km_u4 row = {0x01010101u * VB_OOB, 0x01010101u * VB_OOB, 0x01010101u * VB_OOB, 0x01010101u * VB_OOB};
        if (l < (uint32_t)KMC_ROWS) {
            const int64_t a = (int64_t)fidx + ((int64_t)rz * c.PY + ry) * c.PX - 4;
            if (a >= -16 && a + 16 <= (int64_t)c.PV + 16) row = __builtin_nontemporal_load(reinterpret_cast<const km_u4*>(lab + a));
        }
        const int64_t m = (int64_t)fidx + ((int64_t)dz * c.PY + dy) * c.PX + dx;
        VrgPre pre;
        pre.rank = 0; pre.vent = 0; pre.lev16 = 0; pre.val = 0.0;
        if (ring && have) {
            const uint32_t ms = (uint32_t)(m < (int64_t)idx_lo ? (int64_t)idx_lo : (m > (int64_t)idx_hi ? (int64_t)idx_hi : m));
            pre.rank = (uint32_t)c.stamp[ms]; pre.vent = c.vent[ms];
            pre.lev16 = c.lev16 ? (uint32_t)c.lev16[ms] : c.lidx ? c.lidx[ms] : 0u;
            pre.val = (c.lev16 || c.lidx) ? 0.0 : vrg_voxel_value(c, ms);
        }
        if (round == 0u && cg.L <= LEV_LDS && !cg.lev16) {
            for (uint32_t k = tt; k < cg.L; k += KMC_THREADS) s_lev[k] = cg.lev[k];
            c.lev = s_lev;
        }
        if (l < (uint32_t)KMC_ROWS) { s_tile[4 * l] = row.x; s_tile[4 * l + 1] = row.y; s_tile[4 * l + 2] = row.z; s_tile[4 * l + 3] = row.w; }
        if (round == 0u) __syncthreads();                                 // (the level table and the zeroed counts: once)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();     // (the tile is the half-wave's own: its lanes' LDS traffic is in order)
        // the cube's 125 places, four per lane: an excluded voxel anywhere -> the flip goes to the general kernel; a listed voxel -> its rank
        bool xf = false;
        uint32_t rk[4] = {0u, 0u, 0u, 0u}, lst = 0u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t p = l + 32u * (uint32_t)q;
            if (p < 125u && have) {
                const uint32_t o = p % 5u + 2u;
                const uint8_t b = (uint8_t)(s_tile[4u * (p / 5u) + (o >> 2)] >> (8u * (o & 3u)));
                if (!(b & VB_OOB)) {
                    xf = xf || (b & VB_X);
                    if (b & VB_L) {
                        const int64_t mp = (int64_t)fidx + ((int64_t)((int)(p / 25u) - 2) * c.PY + ((int)((p / 5u) % 5u) - 2)) * c.PX + ((int)(p % 5u) - 2);
                        rk[q] = (uint32_t)c.stamp[(uint32_t)mp]; lst |= 1u << q;
                    }
                }
            }
        }
        const unsigned long long bal = __ballot(xf);
        const bool slowg = ((g & 1u) ? (uint32_t)(bal >> 32) : (uint32_t)bal) != 0u;
        if (slowg && l == 0u && have) slow[vrg_atomic_add(slow_n, 1u)] = r;                       // (rare: nothing of this flip is touched here)
        uint8_t mb = VB_OOB;
        if (ring && have) { const uint32_t o = (uint32_t)(dx + 4); mb = (uint8_t)(s_tile[4 * ((dz + 2) * 5 + (dy + 2)) + (o >> 2)] >> (8u * (o & 3u))); }
        const bool wanted = !slowg && !(mb & (VB_OOB | VB_M));             // (vrg_mark_wanted for a place of the 1-ring)
        uint32_t old = 0xffffffffu;
        const uint32_t sh = 8u * ((uint32_t)m & 3u);
        if (wanted) old = vrg_atomic_or((uint32_t*)(lab + ((uint32_t)m & ~3u)), (uint32_t)VB_M << sh);
#pragma unroll
        for (int q = 0; q < 4; q++) if ((lst >> q) & 1u) s_rank[l + 32u * (uint32_t)q] = rk[q];
        VrgNbr nb = {0u, 0u, 0u, 0u};
        uint32_t FO = 0, AP = 0, cand = 0;
        if (wanted) {
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const uint64_t w8 = *reinterpret_cast<const uint64_t*>(&s_tile[4 * ((dz + j % 3 - 1 + 2) * 5 + (dy + j / 3 - 1 + 2))]);
                pre.w[j] = (uint32_t)(w8 >> (8 * (dx + 3)));              // bytes x-1 .. x+2 of the row (vrg_preload)
            }
            nb = vrg_masks_of(pre.w);
            uint32_t ex, segA; vrg_nbr_sets(nb, ex, segA, FO, AP);
            cand = FO | AP;
        }
        const uint32_t lev_here = (wanted && c.lev_fast) ? vrg_pre_level(c, pre) : 0xffffffffu;
        const bool first = wanted && !((old >> sh) & VB_M);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();     // (the rank tile is complete)
        if (first) {
            VrgEvent ev; ev.kind = VE_NONE; ev.pend = 0;
            VrgRanks qr; vrg_ranks_none(qr);
            while (cand) {                                                // the listed neighbours' ranks, from the rank tile
                uint32_t n0[VRG_RANK_BATCH], r0[VRG_RANK_BATCH];
#pragma unroll
                for (int k = 0; k < VRG_RANK_BATCH; k++) {
                    n0[k] = cand ? vrg_ctz(cand) : 32u; if (cand) cand &= cand - 1u;
                    const uint32_t n = n0[k] < 27u ? n0[k] : 13u, j = n / 3u;           // neighbour n: ddx = n % 3 - 1, ddz = j % 3 - 1, ddy = j / 3 - 1 (vrg_noff)
                    r0[k] = s_rank[(uint32_t)((dz + (int)(j % 3u) - 1 + 2) * 25 + (dy + (int)(j / 3u) - 1 + 2) * 5 + (dx + (int)(n % 3u) - 1 + 2))];
                }
                vrg_ranks_take(FO, n0, r0, qr);
            }
            const uint8_t nw = vrg_sweep_cases(c, (uint32_t)m, mb, pre, nb, qr, false, lev_here, ev);      // (no excluded voxel in the cube: nobody asks for the 2-ring)
            const uint32_t qm = atomicAdd(&s_cnt[0], 1u);
            s_mk_idx[qm] = (uint32_t)m; s_mk_nw[qm] = nw; s_mk_old[qm] = mb;
            if (ev.kind != VE_NONE) {
                uint32_t r1 = 0, rf = 0;
                if (ev.kind == VE_NEW) r1 = atomicAdd(&s_n[0], 1u);
                if (ev.kind == VE_DIE) r1 = atomicAdd(&s_n[1], 1u);
                if (ev.kind != VE_DIE && ev.pend) rf = atomicAdd(&s_n[2], 1u);
                const int di = vrg_ev_dni(ev), dq = vrg_ev_dno(ev);
                if (di) atomicAdd(&s_d[0], di);
                if (dq) atomicAdd(&s_d[1], dq);
                KmEvRec& e = s_ev[atomicAdd(&s_cnt[1], 1u)];
                e.ev = ev; e.m = (uint32_t)m; e.r1 = r1; e.rf = rf;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();     // (before the next round overwrites the tiles)
        VRG_STAMP_WG(c, min(11u, 2u + round));
    }
    VRG_STAMP_WG(c, 12);
    km_flush();
    VRG_STAMP_WG(c, 13);
    if (lds_hist)                                                         // (km_flush ends with a barrier: the counts are complete)
        for (uint32_t k = tt; k < 5u * cg.L; k += KMC_THREADS) {
            const uint32_t n = s_hist[k];
            if (n) { const uint32_t w = k / cg.L, lv = k - w * cg.L; atomicAdd((w == 0 ? cg.dIn : w == 1 ? cg.dOut : w == 2 ? cg.dConv : w == 3 ? (uint32_t*)cg.hin : (uint32_t*)cg.hout) + lv, n); }
        }
#if defined(VRG_STAMPS)
    if (tt == 0) vrg_drain();
    VRG_STAMP_WG(c, 14);
#endif
}

// Workgroups [0, CLOSE_APPLY): the sweep's label bytes in place (+ class bits, region sizes, the class changes of the
// sweep before), dead slots onto the free list - the marked voxels spread over all their threads, one round trip
// instead of six in a single workgroup (5.6 of that workgroup's 10.7 us).  Workgroups [CLOSE_APPLY, +TAB_BLOCKS): the
// touched levels in ascending order and the per-level memo of the density corrections for the next k_band.  The
// workgroup that arrives last (a ticket; what it reads of the others' work - region sizes, error word - went through
// device-scope atomics / write-through stores) files the sizes, asks for the dense pass and closes the sweep.
constexpr int CLOSE_APPLY = 8;
// (napply: the workgroups that apply the sweep's label bytes - CLOSE_APPLY, more for a sweep with thousands of flips)
__global__ void __launch_bounds__(KC_THREADS) k_close(VrgCtx c, int dense_on, uint32_t napply) {
    VRG_CHAOS_POINT(4);
    constexpr uint32_t T = KC_THREADS;
    const uint32_t t = threadIdx.x;
    const bool st0 = blockIdx.x == 0 && t == 0, stm = blockIdx.x == napply && t == 0;
    const unsigned long long t_entry = (st0 || stm) ? VRG_STAMP_NOW() : 0ull;
    // What a thread's FIRST item of every list needs travels with the state (the lists are complete, whatever the state
    // says; any index below a list's capacity is readable): the apply workgroups' marked voxel with its current byte, the
    // class change of the sweep before, a flip's result, a dead slot - one round trip, where a loop after a loop made five.
    const uint32_t g = blockIdx.x * T + t, G = napply * T;
    uint32_t mk0 = 0, cdwA = VRG_NOCHG, cxA = 0, cdwB = VRG_NOCHG, cxB = 0, dead0 = 0, ncA = 0, ncB = 0; uint8_t mn0 = 0, old0 = 0, fres0 = FR_WRITTEN; uint64_t zk0 = 0;
    int64_t rseq0 = 0;
    if (blockIdx.x < napply) {
        if (g < c.mcap) {                              // (both parities of the change list: which one the sweep before filed follows from the state)
            mk0 = c.mk_idx[g]; mn0 = c.mk_new[g]; old0 = c.mk_old[g]; dead0 = c.dead[g];
            cdwA = c.chg_dw[0][g]; cxA = c.chg_x[0][g]; cdwB = c.chg_dw[1][g]; cxB = c.chg_x[1][g];
        }
        if (g < c.fcap) fres0 = c.f_res[g];
        ncA = c.nchg[0]; ncB = c.nchg[1];
        if (t == 0 && dense_on) rseq0 = vrg_load_i64(&c.dctl[VD_RSEQ]);
    } else if (!c.lvl_scan && t < c.zcap) zk0 = c.nz_key[t];
    // (the relabel kernels reserved their list stretches through VrgCtx::rsv, not through the state's own words: what they counted, beside the state)
    uint64_t rsvA = 0, rsvB = 0;
    if (c.rsv) { rsvA = vrg_load_u64(&c.rsv[0]); rsvB = vrg_load_u64(&c.rsv[16]); }
    const VrgState s0 = *c.st;
    if (s0.done || s0.bail) return;                    // (the same for every workgroup: the state is written by the last one to finish)
    const uint32_t k_nalloc = c.rsv ? (uint32_t)rsvA : s0.nalloc, k_ndead = c.rsv ? (uint32_t)(rsvA >> 32) : s0.ndead, k_nmk = c.rsv ? (uint32_t)(rsvB >> 32) : s0.nmk;
    const int pc = ((s0.iter + 1) & 1) ^ 1;                              // parity of the change list the sweep before filed
    const uint32_t cdw0 = pc ? cdwB : cdwA, cx0 = pc ? cxB : cxA;
    __shared__ uint64_t s_key[NZ_SORT];
    __shared__ double s_val[NZ_SORT];
    __shared__ uint32_t s_cin[NZ_SORT], s_cout[NZ_SORT], s_cconv[NZ_SORT];
    __shared__ uint32_t s_scan[T / 64];
    __shared__ int s_last;
    uint32_t nnz = c.lvl_scan ? 0u : min(s0.nnz, c.zcap);
    const bool use_tab = (c.lvl_scan || nnz <= NZ_SORT) && s0.tab_ok;   // fewer levels than entries: memoise per level
    const uint32_t nmk = min(k_nmk, c.mcap);
    if (st0) { VRG_STAMP_PUT(c, 24, t_entry); VRG_STAMP(c, 25); }
    if (stm) VRG_STAMP_PUT(c, 32, t_entry);
    if (blockIdx.x < napply) {
        const uint32_t nf = s0.nf, nd = k_ndead, nalloc = k_nalloc, nc = min(pc ? ncB : ncA, c.mcap);
        if (t == 0 && dense_on && (int64_t)s0.iter - 1 > rseq0) wait_dense_read(c);        // (the pass of two sweeps ago has read the class copy this sweep rewrites)
        __syncthreads();
        if (st0) VRG_STAMP(c, 26);
        // the sweep's label bytes (+ class bits, region sizes; the change filed at the voxel's place of the marked list)
        // (the region sizes: every thread adds up what its voxels change, the workgroup sends ONE pair of atomics - vrg_count_change_at)
        int acc[2] = {0, 0};
        if (g < nmk) vrg_apply_at(c, g, mk0, old0, mn0, s0.log_pos, acc);
        for (uint32_t i = g + G; i < nmk; i += G) vrg_apply_at(c, i, c.mk_idx[i], c.mk_old[i], c.mk_new[i], s0.log_pos, acc);
        {
            int din = acc[0], dout = acc[1];
            for (int o = 32; o > 0; o >>= 1) { din += __shfl_xor(din, o, 64); dout += __shfl_xor(dout, o, 64); }
            __shared__ int s_acc[2];
            if (t == 0) { s_acc[0] = 0; s_acc[1] = 0; }
            __syncthreads();
            if ((t & 63u) == 0u) { if (din) atomicAdd(&s_acc[0], din); if (dout) atomicAdd(&s_acc[1], dout); }
            __syncthreads();
            if (t == 0) { if (s_acc[0]) vrg_atomic_add64(&c.inc[VC_NIN], s_acc[0]); if (s_acc[1]) vrg_atomic_add64(&c.inc[VC_NOUT], s_acc[1]); }
        }
        // the class changes of the sweep before go into this sweep's copy of the class bits
        if (g < nc) vrg_catchup_entry(c, cdw0, cx0);
        for (uint32_t i = g + G; i < nc; i += G) vrg_item_catchup(c, i);
        if (g < nf && !(fres0 & FR_WRITTEN)) vrg_store_i32(&c.stg->error, 3);       // a listed flip the relabel never visited
        for (uint32_t r = g + G; r < nf; r += G) vrg_item_check_flip(c, r);
        if (g < nd) vrg_free_entry(c, g, dead0, s0.nfree, nalloc);
        for (uint32_t j = g + G; j < nd; j += G) vrg_free_entry(c, j, c.dead[j], s0.nfree, nalloc);
        if (st0) { vrg_drain(); VRG_STAMP(c, 27); }
        if (blockIdx.x == 0 && !c.lvl_scan && nnz > NZ_SORT) {           // (rare: a long level list is sorted in place in global memory)
            wg_sort_pairs(c.nz_key, (uint32_t*)nullptr, nnz, false);
            __syncthreads();
            for (uint32_t j = t; j < nnz; j += T) vrg_item_level(c, j, false);
        }
    } else if ((c.lvl_scan || nnz <= NZ_SORT) && (use_tab || blockIdx.x == napply)) {
        // this sweep's touched levels in ascending order (a fixed summation order), with their counts
        if (c.lvl_scan) {
            // small level table: every level's three counters are looked at - thread t its stretch of levels - and the
            // touched ones are listed by a block scan: in ascending order by construction, and k_mark_relabel needed no
            // list-building atomics (two dependent returning atomics per touching thread)
            const uint32_t per = (c.L + T - 1) / T, l0 = t * per, l1 = min(l0 + per, c.L);
            uint32_t ci[NZ_SORT / T], co[NZ_SORT / T], cc[NZ_SORT / T], cnt = 0;
#pragma unroll
            for (uint32_t k = 0; k < NZ_SORT / T; k++) {
                const uint32_t l = l0 + k;
                ci[k] = co[k] = cc[k] = 0;
                if (k < per && l < l1) { ci[k] = c.dIn[l]; co[k] = c.dOut[l]; cc[k] = c.dConv[l]; }
                cnt += (ci[k] | co[k] | cc[k]) ? 1u : 0u;
            }
            const uint32_t incl = wave_incl_scan(cnt);
            if ((t & 63) == 63) s_scan[t >> 6] = incl;
            __syncthreads();
            uint32_t base = 0, total = 0;
            for (uint32_t w = 0; w < T / 64; w++) { if (w < (t >> 6)) base += s_scan[w]; total += s_scan[w]; }
            uint32_t q = base + incl - cnt;
#pragma unroll
            for (uint32_t k = 0; k < NZ_SORT / T; k++)
                if (ci[k] | co[k] | cc[k]) { const uint32_t l = l0 + k; s_key[q] = l; s_val[q] = c.lev[l]; s_cin[q] = ci[k]; s_cout[q] = co[k]; s_cconv[q] = cc[k]; q++; }
            nnz = total;
            __syncthreads();
            if (stm) VRG_STAMP(c, 33);
            if (blockIdx.x == napply) {                             // the list itself: the next k_order clears these counters, an entry-by-entry k_band sums over it
                for (uint32_t j = t; j < nnz; j += T) c.nz_key[j] = s_key[j];
                if (t == 0) __hip_atomic_store(&c.stg->nnz, nnz, VRG_MO_STORE, __HIP_MEMORY_SCOPE_AGENT);   // (read by whoever closes the sweep)
            }
        } else {
            if (t < nnz) s_key[t] = zk0;
            for (uint32_t j = t + T; j < nnz; j += T) s_key[j] = c.nz_key[j];
            __syncthreads();
            wg_sort_pairs(s_key, (uint32_t*)nullptr, nnz, false);
            if (stm) VRG_STAMP(c, 33);
            for (uint32_t j = t; j < nnz; j += T) {
                const uint32_t l = (uint32_t)s_key[j];
                s_val[j] = c.lev[l]; s_cin[j] = c.dIn[l]; s_cout[j] = c.dOut[l]; s_cconv[j] = c.dConv[l];
            }
            __syncthreads();
        }
        if (blockIdx.x == napply)                                   // the ordered level list, for an entry-by-entry k_band
            // (nz_key itself stays as it is: the other workgroups may still be reading it, and only the set matters later)
            for (uint32_t j = t; j < nnz; j += T) { c.nz_val[j] = s_val[j]; c.nz_cin[j] = s_cin[j]; c.nz_cout[j] = s_cout[j]; c.nz_cconv[j] = s_cconv[j]; }
        if (use_tab) {                                                   // the memo: one wave per level
            const uint32_t lane = t & 63, wid = ((blockIdx.x - napply) * T + t) >> 6, nw = (TAB_BLOCKS * T) >> 6;
            for (uint32_t l = wid; l < c.L; l += nw) {
                const double v = c.lev[l];
                double a = 0, bb = 0, d = 0;
                for (uint32_t j = lane; j < nnz; j += 64) {
                    const double k = vrg_kern(c, s_val[j] - v);
                    a += (double)s_cin[j] * k; bb += (double)s_cout[j] * k; d += (double)s_cconv[j] * k;
                }
                a = wave_sum(a); bb = wave_sum(bb); d = wave_sum(d);
                if (lane == 0) { c.tabC[3 * (size_t)l] = a; c.tabC[3 * (size_t)l + 1] = bb; c.tabC[3 * (size_t)l + 2] = d; }
            }
        }
    }
    // everything this workgroup sent to memory has arrived before it takes its ticket
    vrg_drain();
    __syncthreads();
    if (stm) VRG_STAMP(c, 34);
    if (t == 0) {
        VRG_CHAOS_POINT(11);
        const uint32_t k = __hip_atomic_fetch_add(&c.counters[1], 1u, VRG_MO_TICKET, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (k == gridDim.x - 1);
        if (s_last) {
            VRG_STAMP(c, 28);
            c.counters[1] = 0;                                           // every workgroup has arrived: reset for the next launch
            vrg_close_sweep(c, (int64_t)nmk, use_tab);
            vrg_drain(); VRG_STAMP(c, 29);
        }
    }
}

// ---- update() of a sweep with few flips as ONE launch (vrg_items.h, "fused sweep") ------------------------------------
// One flip per workgroup of 128 threads; no seam inside the sweep - every workgroup ranks all flips and resolves the skip
// rule itself, a voxel is relabelled by the flip of smallest rank that wants it, and nothing is applied before the kernel
// ends (the next trip's k_band does that in the shadow of its decisions).  Round trips to memory per workgroup: state +
// flip records -> label tile, per-voxel fields, flip neighbourhoods -> one reservation per event list -> ticket; the
// workgroup whose ticket comes last lists the sweep's touched levels and closes it.
constexpr uint32_t FUSE_MEMO_NNZ = 1024;  // touched levels k_memo keeps in LDS; a sweep that touches more keeps no memo (corrections entry by entry)
// memo_follows: the launch behind this one is k_memo (large bands: the corrections of the sweep memoised per level)
// BIGL: a level table of more than VRG_FUSE_LEVELS values (never searched here: every voxel's level index is kept, VrgCtx::lidx):
// the touched levels are listed by their first toucher and sorted by the closing workgroup instead of found by a scan.
// open_end: the sweep stops at its commit - no ticket, no closing workgroup; the next trip's k_band derives the closed state (vrg_items.h
// "open-ended sweeps"; small level tables only)
template <bool BIGL>
__global__ void __launch_bounds__(VRG_FUSE_THREADS) k_sweep(VrgCtx cg, int memo_follows, int open_end, int zero_par) {
    VRG_CHAOS_POINT(5);
    __shared__ VrgFuseLdsT<BIGL ? 1 : VRG_FUSE_LEVELS> sh;
    __shared__ uint32_t s_keys[BIGL ? VRG_FUSE_KEYS : 1];
    __shared__ uint32_t s_scan[VRG_FUSE_THREADS / 64];
    __shared__ int s_last;
    constexpr uint32_t T = VRG_FUSE_THREADS;
    const uint32_t t = threadIdx.x, r = blockIdx.x;
    const bool st0 = r == 0 && t == 0;
    const unsigned long long t_entry = st0 ? VRG_STAMP_NOW() : 0ull;
    VrgFuseThread th;
    vrg_fuse_load1(cg, th, t);                             // (this thread's flip record travels with the state)
    const VrgState s0 = *cg.st;
    const int64_t nin0 = cg.inc[VC_NIN];
    vrg_fuse_init(sh, t);
    const int lp = (s0.iter + 1) & 1;                      // this sweep's set of per-level counters; the other set - the sweep before's - goes back to zero
    // (BEFORE anything can make this workgroup leave, and by the HOST's count of the sweeps (zero_par = the set the sweep before this trip
    // filled), not by the state this workgroup has loaded: a workgroup that starts late may find the stop or hand-back flag another one has
    // just raised - or, without a flip of its own, the state the closing workgroup has already written: sweep number advanced - and its
    // stretch of the counters still has to be zeroed, or the next sweep would add to stale counts.  Found by the interleaving campaign,
    // 2 cases in 2200.  After a stop or a hand-back the host's count runs ahead: both sets are empty then, zeroing either is harmless.)
    if constexpr (!BIGL) vrg_fuse_zero_other_levels(cg, zero_par ^ 1, r, gridDim.x, t, T);
    if (s0.done || s0.bail) return;
    if (st0) vrg_fuse_prepare_other(cg, s0);               // (the next trip's k_band counts its flips and ties into the other state buffer)
    // replication's per-sweep streaming: the change log is complete up to the sweep BEFORE this one - its records were written by the k_sweep
    // before, its header by that kernel or by the k_band in between: both have ended, nothing to drain (k_order / k_trip_open do the same for the other kinds of trip)
#if !defined(VRG_NO_PUBLISH)                                    // (A/B build of tools/ab_publish.sh: what the publishing costs a plain handle's chain)
    if (st0) vrg_log_publish(cg, s0.log_nsw, s0.log_pos, false);
#endif
    const int32_t gate = vrg_fuse_gate(cg, s0, nin0, vrg_fuse_limit(cg));      // stop tests (:91-104) / can the sweep run fused: the same answer everywhere
    if (gate) {
        if (st0) {
            if (gate > 0) cg.stg->done = gate == 1000 ? -1 : gate; else cg.stg->bail = -gate;
            vrg_close_without_update(cg);
        }
        return;
    }
    const uint32_t nf = s0.nf;
    if (r >= nf) return;                                   // (no flip for this workgroup)
    if (st0) { VRG_STAMP_PUT(cg, 8, t_entry); VRG_STAMP(cg, 9); }
    vrg_fuse_load_rows(cg, th);                            // (the label rows around this thread's record: in flight while the flips are ranked)
    VrgState sl = s0;                                      // (what the item functions read of the state: registers, not memory)
    VrgCtx c = cg;
    c.st = &sl; c.lev_fast = 1; c.lvl_scan = BIGL ? 2 : 1;
    if constexpr (!BIGL) { c.dIn = cg.dInS[lp]; c.dOut = cg.dOutS[lp]; c.dConv = cg.dConvS[lp]; }
    vrg_fuse_keys(sh, th, t, nf);
    __syncthreads();
    vrg_fuse_rank(c, sh, th, t, nf);
    __syncthreads();
    if (st0) VRG_STAMP(cg, 10);
    vrg_fuse_load2(c, sh, th, t, r, nf);
    constexpr uint32_t PER_MAX = VRG_FUSE_LEVELS / VRG_FUSE_THREADS;
    double lv[PER_MAX];                                    // the level table (a voxel that enters the band needs the level of its intensity): requested with the rest
    if constexpr (!BIGL) {
#pragma unroll
        for (uint32_t k = 0; k < PER_MAX; k++) { const uint32_t l = t + k * T; lv[k] = cg.lev[l < cg.L ? l : cg.L - 1u]; }   // (unconditional, index clamped)
    }
    vrg_fuse_listed_nbrs(sh, t, nf);                       // (LDS work while the loads travel)
    if constexpr (!BIGL) {
#pragma unroll
        for (uint32_t k = 0; k < PER_MAX; k++) { const uint32_t l = t + k * T; if (!cg.lev16 && l < cg.L) sh.lev[l] = lv[k]; }
    }
    __syncthreads();
    if (st0) VRG_STAMP(cg, 11);
    vrg_fuse_prepass(sh, th, t, nf);
    __syncthreads();
    if (st0) { vrg_drain(); VRG_STAMP(cg, 18); }
    if (sh.any_pend)                                       // skip-rule fix-point (rare)
        for (;;) {
            vrg_fuse_fix(c, sh, t, nf);
            __syncthreads();
            const uint32_t ch = sh.changed;
            __syncthreads();
            if (!ch) break;
            if (t == 0) sh.changed = 0;
            __syncthreads();
        }
    vrg_fuse_annotate(c, sh, t, r, nf);
    __syncthreads();
    if (st0) VRG_STAMP(cg, 19);
    VRG_CHAOS_POINT(14);
    vrg_fuse_stencil(c, sh, th, t, r);
    __syncthreads();
    if (st0) { vrg_drain(); VRG_STAMP(cg, 20); }
    vrg_fuse_reserve(c, sh, t);
    __syncthreads();
    vrg_fuse_commit(c, sh, th, t, r);
    // everything this workgroup sent to memory has arrived before it takes its ticket
    vrg_drain();
    __syncthreads();
    if (st0) VRG_STAMP(cg, 21);
    if (open_end) {                                        // nobody closes: the state keeps what the workgroups have added up, marked open
        if (st0) { vrg_store_i32(&cg.stg->open, 1); VRG_STAMP(cg, 28); VRG_STAMP(cg, 29); }     // (written through: the line takes the other workgroups' atomics)
        return;
    }
    if (t == 0) {
        VRG_CHAOS_POINT(12);
        const uint32_t k = __hip_atomic_fetch_add(&cg.counters[16], 1u, VRG_MO_TICKET, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (k == nf - 1u);
        if (s_last) cg.counters[16] = 0;                   // every workgroup with a flip has arrived: reset for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    if (t == 0) VRG_STAMP(cg, 28);
    int64_t fin_nin, fin_nout;
    const VrgState fin = vrg_fuse_close_load(c, fin_nin, fin_nout);    // (every thread asks - the same words, one request -: no branch around the loads, they travel with the counters)
    uint32_t total = 0;
    if constexpr (BIGL) {
        // the levels the sweep's first touchers listed (written through, read past L1), sorted in LDS, filed with their counts
        total = min(min(fin.nnz_new, cg.zcap), (uint32_t)VRG_FUSE_KEYS);
        for (uint32_t j = t; j < total; j += T) s_keys[j] = (uint32_t)vrg_load_u64(&cg.nz_key[j]);
        __syncthreads();
        wg_sort_pairs(s_keys, (uint32_t*)nullptr, total, false);
        for (uint32_t j = t; j < total; j += T) vrg_fuse_level_file_listed(c, j, s_keys[j]);
    } else {
    // the touched levels in ascending order (thread t its stretch of levels, a block scan for the places), counters zeroed
    const uint32_t per = (cg.L + T - 1u) / T, l0 = t * per;
    uint32_t ci[PER_MAX], co[PER_MAX], cc[PER_MAX], cnt = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER_MAX; k++) {
        const uint32_t l = l0 + k;
        ci[k] = co[k] = cc[k] = 0;
        if (k < per && l < cg.L) cnt += vrg_fuse_level_touched(c, l, ci[k], co[k], cc[k]) ? 1u : 0u;
    }
    const uint32_t incl = wave_incl_scan(cnt);
    if ((t & 63u) == 63u) s_scan[t >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < T / 64; w++) { if (w < (t >> 6)) base += s_scan[w]; total += s_scan[w]; }
    uint32_t q = base + incl - cnt;
#pragma unroll
    for (uint32_t k = 0; k < PER_MAX; k++)
        if (ci[k] | co[k] | cc[k]) {
            const uint32_t l = l0 + k;
            if (cg.lev16) vrg_fuse_level_file(c, q++, l, cg.lev[l], ci[k], co[k], cc[k]);     // (two calls: one pointer into LDS, one into memory - never a generic one)
            else vrg_fuse_level_file(c, q++, l, sh.lev[l], ci[k], co[k], cc[k]);
        }
    }
    if (t == 0) { vrg_fuse_close(c, fin, fin_nin, fin_nout, total, memo_follows && total <= FUSE_MEMO_NNZ); vrg_drain(); VRG_STAMP(cg, 29); }
}
// (entering fused trips after trips of another kind: the per-level counters of the last sweep are still listed, not yet zero)
__global__ void __launch_bounds__(TPB) k_levels_clear(VrgCtx c) {
    if (c.st->done || c.st->bail) return;
    for (uint32_t j = threadIdx.x, n = min(c.st->nnz, c.zcap); j < n; j += TPB) vrg_item_level_clear(c, j);
}

// Behind a fused sweep on a LARGE band (more entries than k_band decides in one round of workgroups when every entry sums its
// correction itself): the per-level memo of the corrections (:236-247) from the touched-level list the sweep's closing
// workgroup filed - one wave per level, the kernel between two levels from the table; the same terms in the same order as
// k_close's memo.  (Tried first inside k_sweep, by workgroups that wait on the device for the list: 128 polling workgroups
// slowed the dense pass beside them by 15 % and the sweep's tail by 5 us - a launch of its own costs 2.5.)
constexpr int MEMO_BLOCKS = 128;
__global__ void __launch_bounds__(TPB) k_memo(VrgCtx c) {
    VRG_CHAOS_POINT(6);
    __shared__ uint32_t s_nzl[FUSE_MEMO_NNZ], s_ci[FUSE_MEMO_NNZ], s_co[FUSE_MEMO_NNZ], s_cc[FUSE_MEMO_NNZ];
    const uint32_t t = threadIdx.x;
    // (the head of the list travels with the state)
    const uint32_t q = t < c.zcap ? t : c.zcap - 1u;
    const uint32_t l0 = (uint32_t)c.nz_key[q], i0 = c.nz_cin[q], o0 = c.nz_cout[q], c0 = c.nz_cconv[q];
    const VrgState s = *c.st;
    if (s.done || s.bail || !s.use_tab || !s.corr) return;
    const uint32_t nnz = s.nnz;
    if (t < nnz) { s_nzl[t] = l0; s_ci[t] = i0; s_co[t] = o0; s_cc[t] = c0; }
    for (uint32_t j = t + TPB; j < nnz; j += TPB) { s_nzl[j] = (uint32_t)c.nz_key[j]; s_ci[j] = c.nz_cin[j]; s_co[j] = c.nz_cout[j]; s_cc[j] = c.nz_cconv[j]; }
    __syncthreads();
    const uint32_t lane = t & 63u, wid = (blockIdx.x * TPB + t) >> 6, nw = (MEMO_BLOCKS * TPB) >> 6;
    for (uint32_t l = wid; l < c.L; l += nw) {
        double a, bb, d;
        vrg_fuse_memo_terms(c, l, lane, nnz, s_nzl, s_ci, s_co, s_cc, a, bb, d);
        a = wave_sum(a); bb = wave_sum(bb); d = wave_sum(d);
        if (lane == 0) { c.tabC[3 * (size_t)l] = a; c.tabC[3 * (size_t)l + 1] = bb; c.tabC[3 * (size_t)l + 2] = d; }
    }
}

// ---- the same update() as device-wide kernels (host-driven trips: any number of flips) ------------------------
__global__ void __launch_bounds__(TPB) k_trip_open(VrgCtx c) {   // stop tests and capacity test; opens update() (one workgroup)
    __shared__ int s_go;
    if (c.st->done || c.st->bail) return;
    if (threadIdx.x == 0) {
        vrg_log_publish(c, c.st->log_nsw, c.st->log_pos, false);      // (as k_order: the sweep before this trip was closed by kernels that have ended)
        int go = 1;
        const int32_t stop = vrg_stop_test(c);
        if (stop || c.st->error) { c.stg->done = stop ? stop : -1; vrg_close_without_update(c); go = 0; }
        else {
            const int32_t bail = vrg_capacity_test(c, c.st->nf);
            if (bail) { c.stg->bail = bail; vrg_close_without_update(c); go = 0; }
        }
        s_go = go;
    }
    __syncthreads();
    if (!s_go) return;
    for (uint32_t j = threadIdx.x, n = c.st->nnz; j < n; j += TPB) vrg_item_level_clear(c, j);   // level counters of the sweep before
    __syncthreads();
    if (threadIdx.x == 0) vrg_open_update(c);
}
__global__ void k_list(VrgCtx c, uint32_t nf) { ITEM_LOOP(nf) vrg_item_list(c, i); }
__global__ void k_marks_prepass(VrgCtx c, uint32_t nf) {
    ITEM_LOOP64((uint64_t)nf * 128u) {
        uint32_t r = (uint32_t)(i >> 7), p = (uint32_t)(i & 127u);
        if (p < 125u) vrg_item_scatter_marks(c, r, p);
        else if (p == 125u) vrg_item_prepass(c, r);
    }
}
__global__ void k_prepass(VrgCtx c, uint32_t nf) { ITEM_LOOP(nf) vrg_item_prepass(c, i); }
__global__ void __launch_bounds__(KS_THREADS) k_fix(VrgCtx c) {   // skip-rule fix-point, one workgroup
    __shared__ int changed;
    const uint32_t np_ = c.st->npend;
    if (np_ == 0) return;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < np_; j += blockDim.x)
            if (vrg_item_fix(c, j) == 2) changed = 1;
        __syncthreads();
        if (!changed) break;
    }
}
__global__ void k_relabel(VrgCtx c) { ITEM_LOOP(min(c.st->nmk, c.mcap)) vrg_item_relabel(c, i); }
__global__ void k_apply(VrgCtx c) {
    const uint32_t nm = min(c.st->nmk, c.mcap);
    ITEM_LOOP(nm + vrg_catchup_count(c)) { if (i < nm) vrg_item_apply(c, i); else vrg_item_catchup(c, i - nm); }
}
__global__ void k_close_items(VrgCtx c, uint32_t nf) {
    ITEM_LOOP(nf) vrg_item_check_flip(c, i);
    ITEM_LOOP(c.st->ndead) vrg_item_free(c, i);
}
__global__ void k_levels(VrgCtx c, uint32_t nnz) { ITEM_LOOP(nnz) vrg_item_level(c, i, false); }
// per-level memo of the three density corrections: one wave per level
__global__ void k_tab(VrgCtx c, uint32_t nnz) {
    int lane = threadIdx.x & 63;
    uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t l = wid; l < c.L; l += nw) {
        double v = c.lev[l], a = 0, bb = 0, d = 0;
        for (uint32_t i = lane; i < nnz; i += 64) {
            double k = vrg_kern(c, c.nz_val[i] - v);
            a += (double)c.nz_cin[i] * k; bb += (double)c.nz_cout[i] * k; d += (double)c.nz_cconv[i] * k;
        }
        a = wave_sum(a); bb = wave_sum(bb); d = wave_sum(d);
        if (lane == 0) { c.tabC[3 * (size_t)l] = a; c.tabC[3 * (size_t)l + 1] = bb; c.tabC[3 * (size_t)l + 2] = d; }
    }
}
__global__ void k_finalize(VrgCtx c, int use_tab) { vrg_close_sweep(c, -1, use_tab != 0); }
__global__ void k_dense_pack(VrgCtx c) { vrg_dense_pack(c); }
__global__ void k_dense_fin(VrgCtx c) { vrg_dense_fin_staged(c); }

// full-stencil check variant: every voxel runs the relabel stencil (no marks); new bytes go to lab[1]
// and are copied back, so stencil reads only ever see pre-sweep labels.
__global__ void __launch_bounds__(TPB) k_full_relabel(VrgCtx c) {
    const uint8_t* __restrict__ in = c.lab[0];
    uint8_t* __restrict__ out = c.lab[1];
    const uint32_t plane = (uint32_t)c.PY * (uint32_t)c.PX;
    const uint32_t first = 2u * plane;
    const uint32_t ndw = (uint32_t)(((uint64_t)c.nz * plane) >> 2);
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < ndw; d += gridDim.x * blockDim.x) {
        const uint32_t base = first + (d << 2);
        uint32_t v = *reinterpret_cast<const uint32_t*>(in + base);
        if ((v & 0x20202020u) != 0x20202020u)
            for (int bb = 0; bb < 4; bb++) {
                uint8_t cb = (uint8_t)(v >> (8 * bb));
                if (!(cb & VB_OOB)) {
                    VrgEvent ev;
                    uint8_t nb = vrg_sweep_core(c, in, base + bb, cb, ev);
                    vrg_commit_event(c, base + bb, ev);
                    v = (v & ~(0xffu << (8 * bb))) | ((uint32_t)nb << (8 * bb));
                }
            }
        *reinterpret_cast<uint32_t*>(out + base) = v;
    }
}
__global__ void __launch_bounds__(TPB) k_copy_back(VrgCtx c) {
    const uint32_t plane = (uint32_t)c.PY * (uint32_t)c.PX;
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(c.lab[1] + 2u * plane);
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(c.lab[0] + 2u * plane);
    const uint32_t n16 = (uint32_t)(((uint64_t)c.nz * plane) >> 4);
    ITEM_LOOP(vrg_catchup_count(c)) vrg_item_catchup(c, i);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) {
        uint4 a = src[i], bq = dst[i];
        if (a.x != bq.x || a.y != bq.y || a.z != bq.z || a.w != bq.w) {
            const uint32_t nw[4] = {a.x, a.y, a.z, a.w}, od[4] = {bq.x, bq.y, bq.z, bq.w};
            for (int k = 0; k < 16; k++) vrg_count_change(c, 2u * plane + 16u * i + (uint32_t)k, (uint8_t)(od[k >> 2] >> (8 * (k & 3))), (uint8_t)(nw[k >> 2] >> (8 * (k & 3))));
            dst[i] = a;
        }
    }
}

// ---- the dense pass ----------------------------------------------------------------------------------
// Region recount (:113-116 innerSize/outerSize, :249-250 dataArray[mask]) over every voxel, every sweep.
// Streams the interior planes of the padded volume in units of 1024 voxels (k_recount_bits below).  Sums are
// reduced lane -> wave butterfly -> LDS -> one slot per workgroup, added in fixed slot order by the last workgroup
// to finish: bit-reproducible.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));

struct SweepAcc { long long nin, nout; double sin_, sout; };

// per-workgroup slot, then the LAST workgroup to arrive adds all slots in slot order and publishes the totals.
// Hand-off without fences (cdna guide, Guideline 16 / "Valid forms", first row of the measured table): one lane stores its
// workgroup's four values write-through (sc1), drains them (vmcnt(0)), takes a ticket with an agent-scope atomic add;
// the workgroup whose add came last reads every slot with sc1 loads behind a workgroup barrier.  (An agent-scope release
// + acquire pair costs ~1.7 us each - a tenth of a slab's recount.)
__device__ __forceinline__ void st_sc1(long long* p, long long v) { __hip_atomic_store(p, v, VRG_MO_STORE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(double* p, double v) { __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), VRG_MO_STORE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long ld_sc1(const long long* p) { return __hip_atomic_load(p, VRG_MO_LOAD, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(const double* p) { return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), VRG_MO_LOAD, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void sweep_finish(const VrgCtx& c, SweepAcc a, int fin) {
    __shared__ long long sh_n[2][4];
    __shared__ double sh_s[2][4];
    __shared__ int is_last;
    a.nin = wave_sum(a.nin); a.nout = wave_sum(a.nout); a.sin_ = wave_sum(a.sin_); a.sout = wave_sum(a.sout);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh_n[0][wv] = a.nin; sh_n[1][wv] = a.nout; sh_s[0][wv] = a.sin_; sh_s[1][wv] = a.sout; }
    __syncthreads();
    if (threadIdx.x == 0) {
        st_sc1((long long*)&c.st_nin[blockIdx.x], sh_n[0][0] + sh_n[0][1] + sh_n[0][2] + sh_n[0][3]);
        st_sc1((long long*)&c.st_nout[blockIdx.x], sh_n[1][0] + sh_n[1][1] + sh_n[1][2] + sh_n[1][3]);
        st_sc1(&c.st_sin[blockIdx.x], ((sh_s[0][0] + sh_s[0][1]) + sh_s[0][2]) + sh_s[0][3]);
        st_sc1(&c.st_sout[blockIdx.x], ((sh_s[1][0] + sh_s[1][1]) + sh_s[1][2]) + sh_s[1][3]);
        vrg_drain();
        VRG_CHAOS_POINT(13);
        uint32_t t = __hip_atomic_fetch_add(&c.counters[0], 1u, VRG_MO_TICKET, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == gridDim.x - 1);
        if (is_last) c.counters[0] = 0;              // every workgroup has arrived: reset for the next launch
    }
    __syncthreads();
    if (!is_last) return;
    long long x = 0, y = 0; double sx = 0, sy = 0;
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += TPB) {
        x += ld_sc1((const long long*)&c.st_nin[i]); y += ld_sc1((const long long*)&c.st_nout[i]); sx += ld_sc1(&c.st_sin[i]); sy += ld_sc1(&c.st_sout[i]);
    }
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
        if (fin == 0) { *c.dn_part = d; if (c.world == 1) *c.dn = d; }   // init: the sizes found the incremental counts
        else if (fin == 4) vrg_follow_check(c, d, (uint32_t)c.fexp[0], c.fexp[1], c.fexp[2]);   // a follower's count of the sweep its apply step announced
        else {
            vrg_recount_done(c, d);                  // this device's slab sums of the recount
            if (fin == 2) vrg_dense_fin_one(c, d);   // nothing to sum over ranks: close the pass here
        }
    }
}

constexpr uint32_t LEV16_MAX = 16384;   // 16-bit storage: the level values sit in LDS (<= 16384 x f32)

// The recount needs two facts per voxel - inner / outer - so it streams the
// 2-bit class volume (VrgCtx::clsb, 0.25 B/voxel) instead of the label bytes: 4.25 B (fp32 storage), 2.25 B
// (16-bit storage) or 8.25 B (float64 storage) per voxel.  Units are 1024-voxel aligned in the absolute voxel
// index; lane l owns class dword l of a unit and the 4 x 4 intensities at 256*j + 4*l, i.e. one 256-B + four 1-KiB
// (or 512-B / 2-KiB) requests per wave and unit.  A slab edge that cuts a unit is handled by masking (first / last
// wave); padding planes are class 0.
// UNITS = units a wave loads per trip (bytes in flight); NT = non-temporal loads: the volume is read once per
// sweep and is far larger than the 256-MiB Infinity Cache, so nothing is worth keeping.
// MODE 0: fp32 intensities, 1: 16-bit level indices + LDS value table (floats), 2: float64 intensities, 3: as 1 with the table
// held as doubles (up to TAB64_LEVELS levels): the pass is then bound by its arithmetic, not by memory - SQ counters, 16-bit
// storage at 880x880x640: VALU busy 0.6-0.8 of all issue slots, a quarter of it the float -> double conversion of every value
// (half rate on this chip) - and the doubles come out of the table ready to be added.
// Dense pass number seq (= passes closed + 1) reads copy seq & 1 of the class bits.
template <int MODE> struct UnitVals { f4v f[4]; };
template <> struct UnitVals<1> { u2v q[4]; };           // raw level indices: the LDS look-ups wait until the sums are formed
template <> struct UnitVals<3> { u2v q[4]; };
constexpr uint32_t TAB64_LEVELS = 4096;                 // 32 KiB of LDS per workgroup
template <int MODE> __device__ __forceinline__ void lookup4(const UnitVals<MODE>& u, int j, const float* s_val, float lv[4], double dv[4]) {
    if constexpr (MODE == 1) {
        lv[0] = s_val[u.q[j].x & 0xffffu]; lv[1] = s_val[u.q[j].x >> 16];
        lv[2] = s_val[u.q[j].y & 0xffffu]; lv[3] = s_val[u.q[j].y >> 16];
    } else if constexpr (MODE == 3) {
        const double* s_dv = reinterpret_cast<const double*>(s_val);
        dv[0] = s_dv[u.q[j].x & 0xffffu]; dv[1] = s_dv[u.q[j].x >> 16];
        dv[2] = s_dv[u.q[j].y & 0xffffu]; dv[3] = s_dv[u.q[j].y >> 16];
    }
}
template <> struct UnitVals<2> { d2v f[4][2]; };
template <int MODE>
__device__ __forceinline__ void stats_group(SweepAcc& a, uint32_t wj, const UnitVals<MODE>& u, int j, const float* s_val) {
    float lv[4]; double dv[4];
    lookup4<MODE>(u, j, s_val, lv, dv);
#pragma unroll
    for (int bb = 0; bb < 4; bb++) {
        if constexpr (MODE == 2 || MODE == 3) {
            const uint32_t t = wj >> (2 * bb);
            double x;
            if constexpr (MODE == 2) x = u.f[j][bb >> 1][bb & 1]; else x = dv[bb];
            a.sin_ += (t & 1u) ? x : 0.0;
            a.sout += (t & 2u) ? x : 0.0;
        } else {
            // class bit -> all-ones / all-zeros mask over the float's bits: a masked-out voxel adds +0.0
            uint32_t xi;
            if constexpr (MODE == 1) xi = __float_as_uint(lv[bb]); else xi = __float_as_uint(u.f[j][bb]);
            const uint32_t m_in = (uint32_t)((int32_t)(wj << (31 - 2 * bb)) >> 31);
            const uint32_t m_out = (uint32_t)((int32_t)(wj << (30 - 2 * bb)) >> 31);
            a.sin_ += (double)__uint_as_float(xi & m_in);
            a.sout += (double)__uint_as_float(xi & m_out);
        }
    }
}
// the same for a group in which no lane holds an inner voxel: only the outer sum moves (same additions, same order)
template <int MODE>
__device__ __forceinline__ void stats_group_outer(SweepAcc& a, uint32_t wj, const UnitVals<MODE>& u, int j, const float* s_val) {
    float lv[4]; double dv[4];
    lookup4<MODE>(u, j, s_val, lv, dv);
#pragma unroll
    for (int bb = 0; bb < 4; bb++) {
        if constexpr (MODE == 2 || MODE == 3) {
            double x;
            if constexpr (MODE == 2) x = u.f[j][bb >> 1][bb & 1]; else x = dv[bb];
            a.sout += ((wj >> (2 * bb)) & 2u) ? x : 0.0;
        } else {
            uint32_t xi;
            if constexpr (MODE == 1) xi = __float_as_uint(lv[bb]); else xi = __float_as_uint(u.f[j][bb]);
            const uint32_t m_out = (uint32_t)((int32_t)(wj << (30 - 2 * bb)) >> 31);
            a.sout += (double)__uint_as_float(xi & m_out);
        }
    }
}
// SKIP: a group of four voxels per lane whose 256 voxels are all excluded costs the wave nothing (the branch is
// wave-uniform there); partly excluded groups run with the excluded lanes masked off.  Adding +0.0 or not adding at
// all gives the same sums (the accumulators never hold -0.0: they start at +0.0).
template <int MODE>
__device__ __forceinline__ double unit_value(const UnitVals<MODE>& u, int j, int bb, const float* lv, const double* dv) {
    if constexpr (MODE == 2) return u.f[j][bb >> 1][bb & 1];
    else if constexpr (MODE == 3) return dv[bb];
    else if constexpr (MODE == 1) return (double)lv[bb];
    else return (double)u.f[j][bb];
}
// Most groups inside the brain mask hold four outer voxels (class byte 0xAA): when every lane that takes part has
// such a group, the four values go straight into the outer sum - the same additions in the same order as the general
// path makes (which also adds +0.0 to the inner sum four times: no change).
template <int MODE, bool SKIP>
__device__ __forceinline__ void stats_bits(SweepAcc& a, uint32_t w, const UnitVals<MODE>& u, const float* s_val) {
    a.nin += __popc(w & 0x55555555u); a.nout += __popc(w & 0xAAAAAAAAu);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t wj = (w >> (8 * j)) & 0xffu;
        if (!SKIP || wj != 0u) {
            if (SKIP && __builtin_amdgcn_ballot_w64(wj != 0xAAu) == 0ull) {
                float lv[4]; double dv[4];
                lookup4<MODE>(u, j, s_val, lv, dv);
#pragma unroll
                for (int bb = 0; bb < 4; bb++) a.sout += unit_value<MODE>(u, j, bb, lv, dv);
            } else if (SKIP && __builtin_amdgcn_ballot_w64((wj & 0x55u) != 0u) == 0ull) {
                // no lane holds an inner voxel here (the rim of the brain mask: outer and excluded voxels mixed): the outer
                // sum alone, masked - the general path would add +0.0 to the inner sum sixteen times for nothing
                stats_group_outer<MODE>(a, wj, u, j, s_val);
            } else {
                stats_group<MODE>(a, wj, u, j, s_val);
            }
        }
    }
}
template <bool NT>
__device__ __forceinline__ uint32_t load_cls(const uint32_t* cls, uint32_t u, uint32_t lane) {
    const uint32_t* pc = cls + ((size_t)u << 6) + lane;
    return NT ? __builtin_nontemporal_load(pc) : *pc;
}
// the intensities of the lane's 4 x 4 voxels of unit u.  SKIP: a group of four voxels that are all excluded (class 0:
// label 4 or padding) is not fetched - the reference's dataArray[mask] gathers (:249-250) do not touch excluded voxels
// either; a 128-byte line is then not transferred when all eight lanes that share it skip it, i.e. wherever 32
// consecutive voxels are excluded (the brain mask leaves long runs).  The sums are bit-identical: a class-0 voxel
// contributes +0.0 either way.
template <int MODE, bool NT, bool SKIP>
__device__ __forceinline__ void load_vals(const VrgCtx& c, uint32_t u, uint32_t lane, uint32_t w, UnitVals<MODE>& o) {
    const uint32_t base = (u << 10) + (lane << 2);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool need = !SKIP || ((w >> (8 * j)) & 0xffu) != 0u;
        if constexpr (MODE == 1 || MODE == 3) {
            o.q[j] = u2v{0u, 0u};
            if (need) { const u2v* pq = reinterpret_cast<const u2v*>(c.lev16 + base + (j << 8)); o.q[j] = NT ? __builtin_nontemporal_load(pq) : *pq; }
        } else if constexpr (MODE == 2) {
            o.f[j][0] = d2v{0.0, 0.0}; o.f[j][1] = d2v{0.0, 0.0};
            if (need) {
                const d2v* pd = reinterpret_cast<const d2v*>(c.I64 + base + (j << 8));
                o.f[j][0] = NT ? __builtin_nontemporal_load(pd) : pd[0];
                o.f[j][1] = NT ? __builtin_nontemporal_load(pd + 1) : pd[1];
            }
        } else {
            o.f[j] = f4v{0.f, 0.f, 0.f, 0.f};
            if (need) { const f4v* pi = reinterpret_cast<const f4v*>(c.I + base + (j << 8)); o.f[j] = NT ? __builtin_nontemporal_load(pi) : *pi; }
        }
    }
}
// The pass walks the LIST of units that hold an included voxel (VrgCtx::ulist, ascending; the 28 % of the bench volume's
// units that lie wholly outside the brain mask are never visited), all waves in formation: trip t of wave w takes
// entries (t * nwaves + w) * UNITS ..., so neighbouring waves read neighbouring units at about the same time and memory
// sees one dense sweep through the volume.  (Measured at 880x880x640: each wave streaming a contiguous range of its own
// - same bytes, every trip useful - took 0.20-0.37 ms by how the ranges were cut; in formation 0.177; the all-units walk
// of round 2, which fetched the class words of the empty units too, 0.19.)
// SKIP = false (option skip_excluded = 0): the listed units are walked the same way with unpredicated loads - each lane
// makes the same additions in the same order, a class-0 voxel adding +0.0: bit-identical sums - and the units that are
// not listed are streamed afterwards for their bytes only.
template <int UNITS, bool NT, int MODE, bool SKIP>
__global__ void __launch_bounds__(TPB) k_recount_bits(VrgCtx c, int check_done) {
    VRG_CHAOS_POINT(7);
    if (check_done && check_done < 3 && !c.dctl[VD_GO]) return;     // the gate says: no sweep to count (the run has stopped) or this sweep's pass is left out
    extern __shared__ __attribute__((aligned(16))) float s_val[];   // 16-bit storage: the level values (c.L floats - MODE 3: doubles -, sized at launch)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t* __restrict__ ulist = c.ulist;
    const uint32_t n = c.uctl[UC_N];
    const uint32_t last = n ? n - 1u : 0u;
    uint32_t i = __builtin_amdgcn_readfirstlane(wave * UNITS);     // (wave-uniform: the list is read through the scalar cache)
    // (the first trip's units travel with everything else a wave reads first)
    uint32_t uu[UNITS];
#pragma unroll
    for (int q = 0; q < UNITS; q++) uu[q] = i < n ? ulist[min(i + q, last)] : 0u;     // (an empty list has no readable entry)
    if (MODE == 1) {
        for (uint32_t k = threadIdx.x; k < c.L; k += TPB) s_val[k] = (float)c.lev[k];
        __syncthreads();
    }
    if (MODE == 3) {                                 // (the stored value is the float: the same doubles as MODE 1 adds)
        double* s_dv = reinterpret_cast<double*>(s_val);
        for (uint32_t k = threadIdx.x; k < c.L; k += TPB) s_dv[k] = (double)(float)c.lev[k];
        __syncthreads();
    }
    const uint32_t* __restrict__ cls = c.clsb[(vrg_load_i64(&c.dctl[VD_RSEQ]) + (check_done >= 3 ? 0 : 1)) & 1];   // (3: the run's last sweep, counted after all)
    const uint32_t plane = (uint32_t)c.PY * (uint32_t)c.PX;
    const uint32_t lo = (2u + (uint32_t)c.z0) * plane;         // this device's Z-slab [z0, z1) as a voxel range
    const uint32_t hi = (2u + (uint32_t)c.z1) * plane;
    uint32_t f_lo = (uint32_t)(((uint64_t)lo + 1023u) >> 10), f_hi = hi >> 10;       // units wholly inside it
    if (f_hi < f_lo) f_hi = f_lo;
    SweepAcc acc = {0, 0, 0.0, 0.0};
    // the class words of a trip are fetched one trip ahead (the intensity loads depend on them), before this trip's
    // intensities: loads return in order, so they cost no wait of their own
    uint32_t w[UNITS];
#pragma unroll
    for (int q = 0; q < UNITS; q++) { uu[q] = __builtin_amdgcn_readfirstlane(uu[q]); w[q] = i < n ? load_cls<NT>(cls, uu[q], lane) : 0u; }
#pragma unroll
    for (int q = 0; q < UNITS; q++) { asm volatile("" : "+v"(w[q])); if (i + q >= n) w[q] = 0u; }   // settle the first trip's class words here: no waits in mid-loop
    while (i < n) {
        const uint32_t in = i + nwaves * UNITS;
        uint32_t un[UNITS], wn[UNITS];
#pragma unroll
        for (int q = 0; q < UNITS; q++) { un[q] = __builtin_amdgcn_readfirstlane(ulist[min(in + q, last)]); wn[q] = load_cls<NT>(cls, un[q], lane); }
#pragma unroll
        for (int q = 0; q < UNITS; q++) if (in + q >= n) wn[q] = 0u;          // (uniform: slots past the list's end hold nothing)
        UnitVals<MODE> f[UNITS];
#pragma unroll
        for (int q = 0; q < UNITS; q++) load_vals<MODE, NT, SKIP>(c, uu[q], lane, w[q], f[q]);
#pragma unroll
        for (int q = 0; q < UNITS; q++) stats_bits<MODE, SKIP>(acc, w[q], f[q], s_val);
#pragma unroll
        for (int q = 0; q < UNITS; q++) { w[q] = wn[q]; uu[q] = un[q]; }
        i = in;
    }
    if (!SKIP) {                                               // the bytes of the units that are not listed (they add nothing)
        for (uint32_t u = f_lo + wave; u < f_hi; u += nwaves) {
            if ((c.ubits[u >> 5] >> (u & 31u)) & 1u) continue;
            UnitVals<MODE> f;
            const uint32_t w1 = load_cls<NT>(cls, u, lane);
            load_vals<MODE, NT, false>(c, u, lane, w1, f);
            if constexpr (MODE == 1 || MODE == 3) { asm volatile("" :: "v"(f.q[0]), "v"(f.q[1]), "v"(f.q[2]), "v"(f.q[3]), "v"(w1)); }
            else if constexpr (MODE == 2) { asm volatile("" :: "v"(f.f[0][0]), "v"(f.f[1][1]), "v"(f.f[2][0]), "v"(f.f[3][1]), "v"(f.f[0][1]), "v"(f.f[1][0]), "v"(f.f[2][1]), "v"(f.f[3][0]), "v"(w1)); }
            else { asm volatile("" :: "v"(f.f[0]), "v"(f.f[1]), "v"(f.f[2]), "v"(f.f[3]), "v"(w1)); }
        }
    }
    // units the slab edges cut: the first and the last unit touching [lo, hi), masked to the slab
    const uint32_t e0 = lo >> 10, e1 = (hi - 1u) >> 10;
    const uint32_t edge = wave == 0 ? e0 : (wave == nwaves - 1 && e1 != e0 ? e1 : 0xffffffffu);
    if (edge != 0xffffffffu && !(edge >= f_lo && edge < f_hi)) {
        UnitVals<MODE> f;
        uint32_t w1 = load_cls<false>(cls, edge, lane);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t v = (edge << 10) + (j << 8) + (lane << 2);        // groups of 4 voxels never straddle a plane
            if (v < lo || v >= hi) w1 &= ~(0xffu << (8 * j));
        }
        load_vals<MODE, false, SKIP>(c, edge, lane, w1, f);
        stats_bits<MODE, SKIP>(acc, w1, f, s_val);
    }
    sweep_finish(c, acc, check_done == 3 ? 0 : check_done);
}
// fp32 storage + skip_excluded (the default; option dense_pipe = 0 switches it off): the same walk, software-pipelined two
// trips deep.  Every intensity load is issued unconditionally - a lane whose group is excluded reads one fixed dummy line
// (the first 16 bytes of the padded volume) instead of being predicated off - so the number of loads in flight is static,
// the compiler's wait counts are exact, and the NEXT trip's intensities are requested before this trip's sums are formed
// (with predicated loads the compiler waits for everything before it forms a sum).  Same additions in the same order:
// bit-identical to k_recount_bits.  Measured (one process, alternating): 880x880x640 0.173 vs 0.178-0.183 ms; 512x512x170
// 0.031 vs 0.036 ms; without a brain mask 0.340 vs 0.335 (nothing to hide there).
template <int UNITS, bool NT>
__device__ __forceinline__ void load_vals_uncond(const VrgCtx& c, uint32_t u, uint32_t lane, uint32_t w, UnitVals<0>& o) {
    const uint32_t base = (u << 10) + (lane << 2);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool need = ((w >> (8 * j)) & 0xffu) != 0u;
        const f4v* pi = reinterpret_cast<const f4v*>(need ? c.I + base + (j << 8) : c.I);
        o.f[j] = NT ? __builtin_nontemporal_load(pi) : *pi;
    }
}
template <int UNITS, bool NT>
__global__ void __launch_bounds__(TPB) k_recount_pipe(VrgCtx c, int check_done) {
    VRG_CHAOS_POINT(8);
    if (check_done && check_done < 3 && !c.dctl[VD_GO]) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t* __restrict__ ulist = c.ulist;
    const uint32_t n = c.uctl[UC_N];
    const uint32_t last = n ? n - 1u : 0u, stride = nwaves * UNITS;
    uint32_t i = __builtin_amdgcn_readfirstlane(wave * UNITS);     // (wave-uniform: the list is read through the scalar cache)
    const uint32_t* __restrict__ cls = c.clsb[(vrg_load_i64(&c.dctl[VD_RSEQ]) + (check_done >= 3 ? 0 : 1)) & 1];   // (3: the run's last sweep, counted after all)
    const uint32_t plane = (uint32_t)c.PY * (uint32_t)c.PX;
    const uint32_t lo = (2u + (uint32_t)c.z0) * plane, hi = (2u + (uint32_t)c.z1) * plane;
    uint32_t f_lo = (uint32_t)(((uint64_t)lo + 1023u) >> 10), f_hi = hi >> 10;
    if (f_hi < f_lo) f_hi = f_lo;
    SweepAcc acc = {0, 0, 0.0, 0.0};
    if (i < n) {
        uint32_t u0[UNITS], u1[UNITS], w0[UNITS], w1[UNITS];
        UnitVals<0> fa[UNITS];
#pragma unroll
        for (int q = 0; q < UNITS; q++) { u0[q] = __builtin_amdgcn_readfirstlane(ulist[min(i + q, last)]); w0[q] = load_cls<NT>(cls, u0[q], lane); }
#pragma unroll
        for (int q = 0; q < UNITS; q++) { u1[q] = __builtin_amdgcn_readfirstlane(ulist[min(i + stride + q, last)]); w1[q] = load_cls<NT>(cls, u1[q], lane); }
#pragma unroll
        for (int q = 0; q < UNITS; q++) { if (i + q >= n) w0[q] = 0u; if (i + stride + q >= n) w1[q] = 0u; }
#pragma unroll
        for (int q = 0; q < UNITS; q++) load_vals_uncond<UNITS, NT>(c, u0[q], lane, w0[q], fa[q]);
        while (i < n) {
            uint32_t u2[UNITS], w2[UNITS];
            UnitVals<0> fb[UNITS];
#pragma unroll
            for (int q = 0; q < UNITS; q++) { u2[q] = __builtin_amdgcn_readfirstlane(ulist[min(i + 2u * stride + q, last)]); w2[q] = load_cls<NT>(cls, u2[q], lane); }
#pragma unroll
            for (int q = 0; q < UNITS; q++) load_vals_uncond<UNITS, NT>(c, u1[q], lane, w1[q], fb[q]);      // the next trip's intensities first ...
#pragma unroll
            for (int q = 0; q < UNITS; q++) stats_bits<0, true>(acc, w0[q], fa[q], nullptr);                 // ... then this trip's sums
#pragma unroll
            for (int q = 0; q < UNITS; q++) { if (i + 2u * stride + q >= n) w2[q] = 0u; w0[q] = w1[q]; u0[q] = u1[q]; fa[q] = fb[q]; w1[q] = w2[q]; u1[q] = u2[q]; }
            i += stride;
        }
    }
    const uint32_t e0 = lo >> 10, e1 = (hi - 1u) >> 10;
    const uint32_t edge = wave == 0 ? e0 : (wave == nwaves - 1 && e1 != e0 ? e1 : 0xffffffffu);
    if (edge != 0xffffffffu && !(edge >= f_lo && edge < f_hi)) {
        UnitVals<0> f;
        uint32_t w1e = load_cls<false>(cls, edge, lane);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t v = (edge << 10) + (j << 8) + (lane << 2);
            if (v < lo || v >= hi) w1e &= ~(0xffu << (8 * j));
        }
        load_vals<0, false, true>(c, edge, lane, w1e, f);
        stats_bits<0, true>(acc, w1e, f, nullptr);
    }
    sweep_finish(c, acc, check_done == 3 ? 0 : check_done);
}
// The unit list from the bitmap, by one workgroup of 1024 threads (a few microseconds): thread t counts the set bits of
// its stretch of bitmap words, a block scan gives its place, it writes its units.  Bitmap words are read past L1 / a
// stale L2 line (sc1): band kernels of the other stream set bits with device-scope atomics.
constexpr int GATE_THREADS = 1024;
// (p = parity of the pass being prepared: the units its sweep listed for the first time are merged into the bitmap first -
// VrgCtx::unew; the sweep's labels are in place, and no other sweep of that parity can be writing)
__device__ void ulist_refresh(const VrgCtx& c, bool force, int p) {
    __shared__ uint32_t s_part[GATE_THREADS / 64];
    __shared__ uint32_t s_gen;
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) s_gen = vrg_load_u32(&c.uctl[UC_GEN + p * UC_GEN_STRIDE]);
    __syncthreads();
    if (!force && s_gen == 0u) return;                         // (uniform)
    const uint32_t plane = (uint32_t)c.PY * (uint32_t)c.PX, lo = (2u + (uint32_t)c.z0) * plane, hi = (2u + (uint32_t)c.z1) * plane;
    uint32_t f_lo = (uint32_t)(((uint64_t)lo + 1023u) >> 10), f_hi = hi >> 10;
    if (f_hi < f_lo) f_hi = f_lo;
    const uint32_t w0 = f_lo >> 5, w1 = (f_hi + 31u) >> 5, nwords = w1 - w0;
    const uint32_t per = (nwords + GATE_THREADS - 1) / GATE_THREADS;
    const uint32_t a = w0 + t * per, b = min(a + per, w1);
    for (uint32_t wi = a; wi < b; wi++) {                      // merge this sweep's new units (whole words; the slab's range is cut out below)
        const uint32_t nw = vrg_load_u32(&c.unew[p][wi]);
        if (nw) { c.ubits[wi] = c.ubits[wi] | nw; c.unew[p][wi] = 0u; }
    }
    auto word = [&](uint32_t wi) -> uint32_t {
        uint32_t bits = c.ubits[wi];
        const uint32_t u0 = wi << 5;
        if (u0 < f_lo) bits &= 0xffffffffu << (f_lo - u0);
        if (f_hi - u0 < 32u) bits &= (1u << (f_hi - u0)) - 1u;
        return bits;
    };
    uint32_t cnt = 0;
    for (uint32_t wi = a; wi < b; wi++) cnt += __popc(word(wi));
    const uint32_t incl = wave_incl_scan(cnt);
    if (lane == 63) s_part[wv] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int k = 0; k < GATE_THREADS / 64; k++) { if (k < (int)wv) base += s_part[k]; total += s_part[k]; }
    uint32_t q = base + incl - cnt;
    for (uint32_t wi = a; wi < b; wi++) {
        uint32_t bits = word(wi);
        while (bits) { c.ulist[q++] = (wi << 5) + vrg_ctz(bits); bits &= bits - 1u; }
    }
    if (t == 0) { c.uctl[UC_N] = total; c.uctl[UC_GEN + p * UC_GEN_STRIDE] = 0u; }
}
__global__ void __launch_bounds__(GATE_THREADS) k_ulist_init(VrgCtx c) { ulist_refresh(c, true, 0); }
// in front of every recount (dense stream): wait for the sweep's labels, then bring the unit list up to date if that sweep
// (or an earlier one) listed a new unit - rare: label 4 turns into 3 only next to the band
// ... or, with option verify_every, close the sweep's pass without a count (fin = 2: one GPU, the pass is closed here; 1: the
// marker travels through the staged all-reduce like a slab's sums).  VD_GO tells the recount behind the gate what to do.
__global__ void __launch_bounds__(GATE_THREADS) k_gate(VrgCtx c, int every, int fin) {
    VRG_CHAOS_POINT(9);
    __shared__ int s_due, s_par;
    if (threadIdx.x == 0) {
        const int due = gate_dense_due(c) ? 1 : 0;
        const int64_t seq = c.dctl[VD_RSEQ] + 1;               // the pass this gate stands in front of
        int go = due;
        if (due && vrg_dense_skipped(seq, every, c.ver_n, c.ver_me)) {
            vrg_recount_done(c, vrg_dense_skip_marker());
            if (fin == 2) vrg_dense_fin_one(c, vrg_dense_skip_marker());
            go = 0;
        }
        c.dctl[VD_GO] = go;
        s_due = due; s_par = (int)(seq & 1);
    }
    __syncthreads();
    if (s_due) ulist_refresh(c, false, s_par);                 // (also for a pass that is left out: its sweep's new units join the bitmap at THEIR gate)
}
__global__ void k_verify_last(VrgCtx c) { vrg_dense_verify_last(c, c.world == 1 ? *c.dn_part : *c.dn); }
__global__ void k_cls_build(VrgCtx c) {
    const uint32_t nd = (uint32_t)((((uint64_t)c.PV + 1023u) >> 10) << 6);
    // (a wave = the 64 class words of ONE 1024-voxel unit: one atomic per listed unit instead of one per non-empty word - 14 M
    // atomics on 15 K bitmap words made this 3 ms of vrg_init at 880x880x640)
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < nd; d += gridDim.x * blockDim.x) {
        const uint32_t w = vrg_cls_word_build(c, d);
        if (__ballot(w != 0u) && (threadIdx.x & 63u) == 0u) vrg_atomic_or(&c.ubits[d >> 11], 1u << ((d >> 6) & 31u));
    }
}

// Bytes one dense pass requests from memory for the class copy the last pass read: per listed unit its list entry (4 B)
// and its class words (256 B; the units the slab's faces cut: always) + every 128-byte intensity line that holds an
// included voxel (what k_recount_bits<.., SKIP> fetches).
__global__ void __launch_bounds__(TPB) k_dense_bytes(VrgCtx c, unsigned long long* out) {
    const uint32_t* __restrict__ cls = c.clsb[vrg_load_i64(&c.dctl[VD_RSEQ]) & 1];
    const uint32_t plane = (uint32_t)c.PY * (uint32_t)c.PX;
    const uint32_t lo = (2u + (uint32_t)c.z0) * plane, hi = (2u + (uint32_t)c.z1) * plane;
    const uint32_t e0 = lo >> 10, e1 = (hi - 1u) >> 10;
    uint32_t f_lo = (uint32_t)(((uint64_t)lo + 1023u) >> 10), f_hi = hi >> 10;
    if (f_hi < f_lo) f_hi = f_lo;
    const uint32_t lpl = c.lev16 ? 16u : (c.I ? 8u : 4u);          // lanes (of 4 voxels each) per 128-byte line
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    unsigned long long bytes = 0;
    for (uint32_t u = e0 + wave; u <= e1; u += nwaves) {
        const bool whole = u >= f_lo && u < f_hi;
        if (whole && !((c.ubits[u >> 5] >> (u & 31u)) & 1u)) continue;
        uint32_t w = cls[((size_t)u << 6) + lane];
        bytes += whole ? 260u : 256u;                               // class words (+ the unit's list entry)
        for (int j = 0; j < 4; j++) {
            const uint32_t v = (u << 10) + (j << 8) + (lane << 2);
            const bool need = v >= lo && v < hi && ((w >> (8 * j)) & 0xffu) != 0u;
            const uint64_t m = __ballot(need);
            for (uint32_t g = 0; g < 64u; g += lpl) bytes += ((m >> g) & ((1ull << lpl) - 1ull)) ? 128u : 0u;
        }
    }
    if (lane == 0 && bytes) atomicAdd(out, bytes);
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
// same, for level tables that fit LDS (fp32 storage): per-workgroup private histograms (the level values too),
// streamed over the padded interior 16 bytes per lane, flushed with one global atomic per non-zero bin.
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
        for (int bb = 0; bb < 4; bb++) {
            uint8_t cb = (uint8_t)(v >> (8 * bb));
            if (cb & (VB_OOB | VB_X)) continue;
            uint32_t lo = 0, hi = L - 1;
            if (c.lev16) lo = c.lev16[base + bb];
            else while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (s_lev[m] < fv[bb]) lo = m + 1; else hi = m; }
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
__global__ void k_exact_init(VrgCtx c) {               // init mode (:152-155): every band entry
    const VrgState s = *c.st;
    const uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    exact_wave(c, s, s.nfresh, wid, nw, false);
}
__global__ void k_fin_init(VrgCtx c) {
    VrgState& s = *c.st;
    s.np = s.ni + s.no; s.nfree = 0; s.nfresh = 0; s.nfx = 0; s.nf = 0; s.last_nf = 0; s.npend = 0; s.nmk = 0; s.nnz = 0;
    s.nalloc = 0; s.ndead = 0; s.d_ni = 0; s.d_no = 0; s.corr = 0; s.use_tab = 0; s.bail = 0;
    vrg_init_counts(c);
    const VrgDense& d = *c.dn;
    VrgTrace& t = c.trace[0];
    t.nflip = 0; t.nseg = (int64_t)d.n_in; t.n_in = (int64_t)d.n_in; t.n_out = (int64_t)d.n_out; t.ni = s.ni; t.no = s.no;
    t.sum_in = d.sum_in; t.sum_out = d.sum_out; t.ties = 0; t.near_ties = 0;
    s.ties = 0; s.near_ties = 0; s.ties_filed = 0; s.near_filed = 0;
}
__global__ void k_recount_hist(VrgCtx c, int32_t* rin, int32_t* rout) {
    VOXEL_LOOP(c) {
        int x, y, z; uint32_t idx = real_idx(c, t, x, y, z);
        uint8_t bb = c.lab[0][idx];
        if (bb & VB_X) continue;
        uint32_t lev = vrg_level_of(c, vrg_voxel_value(c, idx));
        atomicAdd((bb & VB_S) ? &rin[lev] : &rout[lev], 1);
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
template <class T> __global__ void k_gather_I(VrgCtx c, T* dst) {
    VOXEL_LOOP(c) { int x, y, z; uint32_t idx = real_idx(c, t, x, y, z); dst[t] = c.I ? (T)c.I[idx] : (T)c.I64[idx]; }
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
// (nz: the number of non-zero values - np.count_nonzero(dataArray) of the reference's closing message, :95 - counted on the way)
__global__ void k_pack_volume(VrgCtx c, float* dst, double* dst64, const void* src, int dtype, int64_t s0, int64_t s1, int64_t s2, int* flag, unsigned long long* nz) {
    unsigned long long mine = 0;
    VOXEL_LOOP(c) {
        int x, y, z; uint32_t idx = real_idx(c, t, x, y, z);
        double v = load_as_double(src, dtype, x * s0 + y * s1 + z * s2);
        mine += v != 0.0;
        if (dst64) { dst64[idx] = v; continue; }
        float f = (float)v;
        if ((double)f != v) *flag = 1;
        dst[idx] = f;
    }
    mine = (unsigned long long)wave_sum((long long)mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(nz, mine);
}
__global__ void k_pack_labels(VrgCtx c, uint8_t* dst, const void* src, int dtype, int64_t s0, int64_t s1, int64_t s2, int* flag) {
    VOXEL_LOOP(c) {
        int x, y, z; uint32_t idx = real_idx(c, t, x, y, z);
        double v = load_as_double(src, dtype, x * s0 + y * s1 + z * s2);
        uint8_t bb = 0;
        if (v == 0) bb = VB_S; else if (v == 3) bb = 0; else if (v == 4) bb = VB_X; else *flag = 1;
        dst[idx] = bb;
    }
}
__global__ void k_unpack_labels(VrgCtx c, const uint8_t* lab, void* dst, int dtype, int64_t s0, int64_t s1, int64_t s2, int what) {
    VOXEL_LOOP(c) {
        int x, y, z; uint32_t idx = real_idx(c, t, x, y, z);
        const int v = vrg_dec(lab[idx]);
        store_int(dst, dtype, x * s0 + y * s1 + z * s2, what ? (v <= 1 ? 1 : 0) : v);
    }
}
__global__ void k_build_lev16(VrgCtx c, uint16_t* dst) {
    VOXEL_LOOP(c) { int x, y, z; uint32_t idx = real_idx(c, t, x, y, z); dst[idx] = (uint16_t)vrg_level_of(c, vrg_voxel_value(c, idx)); }
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
// Workgroups k_band gets for the pool.  Its loops are grid-stride, so any number is correct; beside a recount only two
// of its waves fit on a SIMD, and 1024 workgroups of which 800 find nothing to do then cost it two extra rounds.  The
// pool can grow by at most a few thousand slots within a batch of trips: 1.5 x the last known size leaves room.
// (corrections entry by entry: where the engine says so - more levels than entries - and behind a fused trip that kept no memo)
bool band_direct(const VrgBackend* b) { return b->direct_hint || (b->fused_prev && !b->fused_memo); }
uint64_t band_slots(const VrgBackend* b) { return (uint64_t)b->band_hint * 9 / 8 + 2048; }      // (a batch of 64 sweeps adds a few thousand slots at most)
int band_lanes(const VrgBackend* b) {       // lanes per slot of the entry-by-entry corrections: the pool within 512 workgroups where it can be
    if (!b->band_hint) return 4;
    const uint64_t slots = band_slots(b);
    return slots * 16 <= 512u * TPB ? 16 : slots * 8 <= 512u * TPB ? 8 : 4;
}
uint32_t band_blocks(const VrgBackend* b) {
    if (!b->band_hint) return b->band_blocks_max;
    const uint64_t threads = band_slots(b) * (band_direct(b) ? band_lanes(b) : 1);
    // (a pool so large that every workgroup files its flips together - one bump of the flip counter per workgroup, SINK_ABOVE - is bound by those bumps: they
    // execute one after the other at the memory side, ~15 ns each.  512x512x170 with 1.2 M slots, 12 900 flips per sweep: 2048 / 512 / 256 / 128 workgroups ->
    // 0.253 / 0.244 / 0.226 / 0.231 ms per sweep; 880x880x640: 0.544 / 0.494 / 0.440)
    const uint64_t most = (band_slots(b) > SINK_ABOVE && std::max(b->flip_hint, b->flip_hint_min) > 2048u) ? std::min<uint64_t>(b->band_blocks_max, 256) : b->band_blocks_max;   // (few flips: nothing queues, the pool's size decides)
    return (uint32_t)std::min<uint64_t>(most, std::max<uint64_t>(32, (threads + TPB - 1) / TPB));
}

// Non-temporal loads for the dense pass?  By the bytes a pass fetches (counted when init has built the class bits): up
// to a little more than the 256-MiB Infinity Cache, ordinary loads keep most of the slab there from sweep to sweep
// (512x512x170, 102 MB: 0.0345 -> 0.0306 ms per pass, and less HBM traffic for the band kernels to queue behind; 274 MB:
// 0.0582 -> 0.0545); a larger pass would only thrash the cache (342 MB: even; 548 MB: 0.102 -> 0.112; 1.1 GB: 0.19 -> 0.204).
constexpr uint64_t NT_ABOVE_BYTES = 300ull << 20;
bool dense_nt(const VrgBackend* b, const VrgCtx&) {
    if (b->nt_loads >= 0) return b->nt_loads != 0;
    return b->pass_bytes == 0 || b->pass_bytes > NT_ABOVE_BYTES;
}

int dense_blocks(const VrgBackend* b, const VrgCtx& c) {
    if (b->sweep_blocks > 0) return b->sweep_blocks;
    uint64_t units = ((uint64_t)(c.z1 - c.z0) * c.PY * c.PX) >> 10;
    // skipping pass: few registers, many short trips - 3 waves per SIMD on a big volume, the 16-bit variant (LDS look-ups,
    // half the bytes per trip) 8; measured in DESIGN.md section 5.  A SMALL pass (512x512x170, an 80-plane slab) is not
    // what bounds the step - the band chain is, and every recount wave on a CU is a queue of loads the band kernels'
    // dependent loads wait behind: 200-280 workgroups there (512x512x170: step 0.0447 ms with 192-256, 0.0456 with 353,
    // 0.0481 with 512; 880x880x80: 0.0466 with 256, 0.0493 with 483, 0.0505 with 512).
    // Streaming pass (skip_excluded = 0): 1 resp. 2 workgroups per CU.
    if (!b->skip) return (int)std::min<uint64_t>(c.lev16 ? 2 * SWEEP_BLOCKS : SWEEP_BLOCKS, std::max<uint64_t>(64, units / 128));
    // 16-bit storage: six workgroups per CU (one session, 880x880x640 / 1024^3: 1024 -> 0.137 / -, 1280 -> 0.132 / 0.258, 1536 ->
    // 0.122 / 0.240, 1792 -> 0.122 / -, 2048 -> 0.148 / 0.281 ms)
    // (small passes, where the band chain bounds the step: 512x512x170 942 -> 0.0428 ms/step, 384-512 -> 0.0381; 80-plane slab
    // 1289 -> 0.0516, 512 -> 0.0394; 160 planes 1536 -> 0.0633, 1024 -> 0.0499)
    if (c.lev16) return (int)std::min<uint64_t>(units <= 100000 ? 2 * SWEEP_BLOCKS : units <= 200000 ? 4 * SWEEP_BLOCKS : 6 * SWEEP_BLOCKS, std::max<uint64_t>(64, units / 48));
    // fp32: whole or half multiples of the CU count only - 552 or 640 workgroups leave some CUs with a wave more than others for
    // the whole pass (880x880x160: 552 -> 0.058 ms, 384 -> 0.050; 880x880x320: 640 -> 0.103, 512 -> 0.094, 768 -> 0.091 but a
    // slower step, 0.1035 vs 0.1003, the band chain queueing behind three waves per SIMD); one session, tools/gpu_slabsweep.sh
    // (round 4, with the fused band chain beside the pass - fewer dependent round trips for the pass's loads to delay: 80-plane
    // slab 256 -> 0.0419 ms/step, 384 -> 0.0382, 512 -> 0.0392; 160 planes 384 -> 0.0577, 512 -> 0.0552, 768 -> 0.0582; 512x512x170
    // 256 -> 0.0343, 384 -> 0.0347, 512 -> 0.0509: the band kernels then wait for a place on the chip)
    // (round 6, the chain at 0.031 ms: 512x512x170 - 45 000 units - 256 -> 0.0342-0.0345 ms per step, 320-448 -> 0.0322-0.0331, three repeats each, 512 -> 0.0333, 768 -> 0.035)
    const uint64_t pick = units <= 100000 ? 384 : units <= 350000 ? 512 : 768;
    return (int)std::min<uint64_t>(pick, std::max<uint64_t>(64, units / 110));      // (512x512x170: 45 000 units -> 384, a whole multiple of half the CUs; not 282)
}

void use_device(VrgBackend* b) { HIP_CHECK(hipSetDevice(b->device)); }

void make_streams(VrgBackend* b) {
    int lo = 0, hi = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));      // hi = numerically lowest = highest priority
    if (b->sa) { HIP_CHECK(hipStreamSynchronize(b->sa)); HIP_CHECK(hipStreamDestroy(b->sa)); }
    if (b->sb) { HIP_CHECK(hipStreamSynchronize(b->sb)); HIP_CHECK(hipStreamDestroy(b->sb)); }
    // prio_mode 0: equal; 1: band stream A high; 2: dense stream B high
    // (keeping the dense pass off 1-8 CUs of every XCD with a CU-masked stream - free places for the band chain - was measured in
    // round 4: the chain beside a pass stays at 37 us, the pass gets 3-20 % slower: what the chain waits for is memory, not a place)
    HIP_CHECK(hipStreamCreateWithPriority(&b->sa, hipStreamNonBlocking, b->prio_mode == 1 ? hi : (b->prio_mode == 2 ? lo : 0)));
    HIP_CHECK(hipStreamCreateWithPriority(&b->sb, hipStreamNonBlocking, b->prio_mode == 2 ? hi : (b->prio_mode == 1 ? lo : 0)));
}

// scratch for the host-driven sorts, grown on demand
bool need_tmp(VrgBackend* b, size_t bytes) {
    if (bytes <= b->tmp_bytes) return true;
    if (b->tmp) HIP_CHECK(hipFree(b->tmp));
    b->tmp = nullptr; b->tmp_bytes = 0;
    if (hipMalloc(&b->tmp, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    b->tmp_bytes = bytes;
    return true;
}
bool need_keys2(VrgBackend* b, size_t n) {
    if (n <= b->keys2_n) return true;
    if (b->keys2) HIP_CHECK(hipFree(b->keys2));
    b->keys2 = nullptr; b->keys2_n = 0;
    if (hipMalloc(&b->keys2, n * 8) != hipSuccess) { (void)hipGetLastError(); return false; }
    b->keys2_n = n;
    return true;
}

}  // namespace

// ---- backend interface ---------------------------------------------------------------------------------
VrgBackend* be_create(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) { (void)hipGetLastError(); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    VrgBackend* b = new VrgBackend();
    b->device = device;
    make_streams(b);
    if (b->err[0]) { be_destroy(b); return nullptr; }
    return b;
}
void be_destroy(VrgBackend* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->rsv) (void)hipFree(b->rsv);
    if (b->sa) (void)hipStreamSynchronize(b->sa);
    if (b->sb) (void)hipStreamSynchronize(b->sb);
    if (b->comm) { ncclCommDestroy(b->comm); b->comm = nullptr; }
    for (auto& p : b->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (int j = 0; j < 4; j++) if (b->mark[j]) (void)hipEventDestroy(b->mark[j]);
    if (b->tmp) (void)hipFree(b->tmp);
    for (int j = 0; j < 2; j++) if (b->xfer[j]) (void)hipHostFree(b->xfer[j]);
    if (b->keys2) (void)hipFree(b->keys2);
    if (b->sa) (void)hipStreamDestroy(b->sa);
    if (b->sb) (void)hipStreamDestroy(b->sb);
    if (b->sc) { (void)hipStreamSynchronize(b->sc); (void)hipStreamDestroy(b->sc); }
    if (b->sd) { (void)hipStreamSynchronize(b->sd); (void)hipStreamDestroy(b->sd); }
    delete b;
}
void be_set_tuning(VrgBackend* b, const char* name, long long v) {
    use_device(b);
    if (std::strcmp(name, "sweep_blocks") == 0 && v >= 0 && v <= 4096) b->sweep_blocks = (int)v;
    if (std::strcmp(name, "serial_streams") == 0) b->serial = v != 0;
    if (std::strcmp(name, "repl") == 0) b->repl = v != 0;
    if (std::strcmp(name, "skip_excluded") == 0) b->skip = v != 0;
    if (std::strcmp(name, "nt_loads") == 0) b->nt_loads = v < 0 ? -1 : (v != 0);
    if (std::strcmp(name, "iter_hint") == 0) b->iter_hint = (int)v;
    if (std::strcmp(name, "open_sweeps") == 0) b->open_sweeps = v != 0;
    if (std::strcmp(name, "mark_compact") == 0) b->mark_compact = v != 0;
    if (std::strcmp(name, "band_blocks_max") == 0 && v >= 32 && v <= BAND_BLOCKS) b->band_blocks_max = (uint32_t)v;
    if (std::strcmp(name, "band_hint") == 0) b->band_hint = (uint32_t)std::min<long long>(std::max<long long>(v, 0), 0x7fffffff);
    if (std::strcmp(name, "direct_hint") == 0) b->direct_hint = v != 0;
    if (std::strcmp(name, "dense_pipe") == 0) b->dense_pipe = (int)v;
    if (std::strcmp(name, "memo_above") == 0 && v >= 0) b->memo_above = (uint32_t)std::min<long long>(v, 0x7fffffff);
    if (std::strcmp(name, "verify_every") == 0 && v >= 0) b->verify_every = (int)std::min<long long>(v, 1 << 20);
    if (std::strcmp(name, "small_flips") == 0 && v >= 0) b->small_flips = (uint32_t)std::min<long long>(v, NF_WIDE);
    if (std::strcmp(name, "flip_hint") == 0) {           // (a sweep as large as the one that came back has been applied: the floor has done its job)
        const uint32_t f = (uint32_t)std::min<long long>(std::max<long long>(v, 0), 0x7fffffff);
        if (f >= b->flip_hint_min) b->flip_hint_min = 0;
        b->flip_hint = std::max(f, b->flip_hint_min);
    }
    if (std::strcmp(name, "flip_hint_min") == 0) { b->flip_hint_min = (uint32_t)std::min<long long>(std::max<long long>(v, 0), 0x7fffffff); b->flip_hint = std::max(b->flip_hint, b->flip_hint_min); }
    if (std::strcmp(name, "prio_mode") == 0 && v >= 0 && v <= 2 && v != b->prio_mode) { b->prio_mode = (int)v; make_streams(b); }
}
uint32_t be_small_flip_limit(VrgBackend* b) { return b->small_flips; }
uint32_t be_fuse_limit(VrgBackend*, const VrgCtx& c) { return vrg_fuse_limit(c); }
// fused trips need the level table in the workgroup's LDS (or 16-bit level indices)
bool be_fuse_ok(VrgBackend*, const VrgCtx& c) { return c.L <= (uint32_t)VRG_FUSE_LEVELS || c.lidx != nullptr; }     // (a large level table: with every voxel's level index at hand)
void be_fuse_enter(VrgBackend* b, const VrgCtx& c) { use_device(b); k_levels_clear<<<1, TPB, 0, b->sa>>>(c); }
bool be_wants_sync(VrgBackend*, const VrgCtx&) { return false; }     // (every level-table size runs batched trips: large tables evaluate their exact densities through the bins)

void* be_alloc(VrgBackend* b, size_t bytes) { use_device(b); void* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; } return p; }
void be_free(VrgBackend* b, void* p) { use_device(b); HIP_CHECK(hipFree(p)); }
void be_fill(VrgBackend* b, void* p, int byte, size_t bytes) { use_device(b); HIP_CHECK(hipMemsetAsync(p, byte, bytes, b->sa)); }
void be_upload(VrgBackend* b, void* dst, const void* src, size_t bytes) { use_device(b); HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, b->sa)); HIP_CHECK(hipStreamSynchronize(b->sa)); }
void be_download(VrgBackend* b, void* dst, const void* src, size_t bytes) { use_device(b); HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, b->sa)); HIP_CHECK(hipStreamSynchronize(b->sa)); }
void be_copy(VrgBackend* b, void* dst, const void* src, size_t bytes) { use_device(b); HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, b->sa)); }
const char* be_last_error(VrgBackend* b) {
    if (!b->err[0]) { (void)hipSetDevice(b->device); hipError_t e = hipGetLastError(); if (e != hipSuccess) std::snprintf(b->err, sizeof(b->err), "HIP error '%s' (asynchronous)", hipGetErrorString(e)); }
    return b->err[0] ? b->err : nullptr;
}
void be_clear_error(VrgBackend* b) { b->err[0] = 0; }
// (the engine synchronises when a run ends or a trip was handed back: no fused sweep is waiting for its dense pass then)
bool be_band_busy(VrgBackend* b) { use_device(b); const hipError_t e = hipStreamQuery(b->sa); if (e == hipErrorNotReady) { (void)hipGetLastError(); return true; } return false; }
void be_sync(VrgBackend* b) { use_device(b); HIP_CHECK(hipStreamSynchronize(b->sa)); HIP_CHECK(hipStreamSynchronize(b->sb)); if (b->sd) HIP_CHECK(hipStreamSynchronize(b->sd)); b->fused_prev = false; b->prev_open = false; }

// A device-resident input is read on the library's own stream: the caller's producer must have finished (vrg.h).
// ---- host arrays in and out --------------------------------------------------------------------------------------------------------
// The reference's own calling convention is int64 valueMap and int / float64 dataArray (variationalRegionGrowing.py:44-46, :288): 8 bytes per
// voxel each way over PCIe from pageable memory.  A HOST array wider than what the device keeps is therefore narrowed on the host first -
// labels to one byte, intensities to fp32 when every value survives that (else the raw array travels: the volume is kept as float64) - by a
// few threads, a chunk at a time through two page-locked buffers, so that the narrowing of one chunk overlaps the copy of the chunk before;
// results go the other way: one byte per voxel comes back and is widened into the caller's array on the host.
constexpr size_t XFER_CHUNK = 32u << 20;              // elements per chunk
static int host_threads() { const unsigned n = std::thread::hardware_concurrency(); return (int)std::min<unsigned>(16u, std::max<unsigned>(1u, n)); }
template <class F> static void parallel_chunks(size_t n, F f) {                  // f(begin, end) on a few threads
    const int nt = n < (1u << 20) ? 1 : host_threads();
    if (nt == 1) { f((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; t++) { const size_t a = std::min(n, t * per), e = std::min(n, a + per); if (a < e) th.emplace_back([=] { f(a, e); }); }
    for (auto& x : th) x.join();
}
static bool xfer_buffers(VrgBackend* b, size_t bytes) {
    if (b->xfer_bytes >= bytes) return true;
    for (int j = 0; j < 2; j++) { if (b->xfer[j]) (void)hipHostFree(b->xfer[j]); b->xfer[j] = nullptr; }
    b->xfer_bytes = 0;
    for (int j = 0; j < 2; j++) if (hipHostMalloc(&b->xfer[j], bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
    b->xfer_bytes = bytes;
    return true;
}
template <class T> static double host_load(const void* p, size_t i) { return (double)((const T*)p)[i]; }
static double host_load_as_double(const void* p, int dtype, size_t i) {
    switch (dtype) { case 0: return host_load<uint8_t>(p, i); case 1: return host_load<int16_t>(p, i); case 2: return host_load<uint16_t>(p, i); case 3: return host_load<int32_t>(p, i);
                     case 4: return host_load<int64_t>(p, i); case 5: return host_load<float>(p, i); default: return host_load<double>(p, i); }
}
// a host array of V elements narrowed to `out_elem`-byte elements (1: label bytes, 255 for a value that is no label; 4: fp32) and copied to
// device memory `dev`, chunk by chunk; *flag: a value did not survive (labels: not 0 / 3 / 4; intensities: not exact in fp32 - the copy stops)
static bool narrow_to_device(VrgBackend* b, void* dev, const void* src, int dtype, size_t V, int out_elem, int* flag) {
    if (!xfer_buffers(b, XFER_CHUNK * 4)) return false;
    *flag = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int j = 0; j < 2; j++) HIP_CHECK(hipEventCreateWithFlags(&ev[j], hipEventDisableTiming));
    int k = 0;
    for (size_t i0 = 0; i0 < V; i0 += XFER_CHUNK, k ^= 1) {
        const size_t n = std::min(XFER_CHUNK, V - i0);
        HIP_CHECK(hipEventSynchronize(ev[k]));           // (the copy that last used this buffer is done)
        std::atomic<int> bad{0};
        void* buf = b->xfer[k];
        parallel_chunks(n, [&](size_t a, size_t e) {
            int mine = 0;
            if (out_elem == 1) { uint8_t* o = (uint8_t*)buf; for (size_t i = a; i < e; i++) { const double v = host_load_as_double(src, dtype, i0 + i); const bool ok = v == 0 || v == 3 || v == 4; o[i] = ok ? (uint8_t)v : 255; mine |= !ok; } }
            else { float* o = (float*)buf; for (size_t i = a; i < e; i++) { const double v = host_load_as_double(src, dtype, i0 + i); const float f = (float)v; o[i] = f; mine |= ((double)f != v); } }
            if (mine) bad.store(1);
        });
        if (bad.load()) { *flag = 1; if (out_elem == 4) break; }
        HIP_CHECK(hipMemcpyAsync((uint8_t*)dev + i0 * out_elem, buf, n * out_elem, hipMemcpyHostToDevice, b->sa));
        HIP_CHECK(hipEventRecord(ev[k], b->sa));
    }
    HIP_CHECK(hipStreamSynchronize(b->sa));
    for (int j = 0; j < 2; j++) (void)hipEventDestroy(ev[j]);
    return true;
}
// A device-resident input is read on the library's own stream: the caller's producer must have finished (vrg.h).
// *dtype_dev: the element type of what is on the device (a narrowed host array: VRG_U8 / VRG_F32); *early: the narrowing already
// answered the question the kernel would have answered (an intensity that fp32 cannot hold: nothing was copied)
static const void* stage_in(VrgBackend* b, const VrgCtx& c, const void* src, int dtype, void** tmp, int* dtype_dev, int narrow_to, int* early) {
    *tmp = nullptr; *dtype_dev = dtype; *early = 0;
    if (is_device_ptr(src)) return src;
    const size_t V = (size_t)c.nx * c.ny * c.nz;
    if (narrow_to && (int)kElem[dtype] > narrow_to) {
        if (hipMalloc(tmp, V * narrow_to) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        int flag = 0;
        if (!narrow_to_device(b, *tmp, src, dtype, V, narrow_to, &flag)) { HIP_CHECK(hipFree(*tmp)); *tmp = nullptr; return nullptr; }
        if (flag && narrow_to == 4) { *early = 1; return *tmp; }           // (the volume is kept as float64: the caller comes again for the raw array)
        *dtype_dev = narrow_to == 1 ? VRG_U8 : VRG_F32;
        return *tmp;
    }
    size_t bytes = V * kElem[dtype];
    if (hipMalloc(tmp, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    HIP_CHECK(hipMemcpyAsync(*tmp, src, bytes, hipMemcpyHostToDevice, b->sa));
    return *tmp;
}

int be_pack_volume(VrgBackend* b, const VrgCtx& c, float* dst, double* dst64, const void* src, int dtype, const int64_t st[3], int* inexact, long long* nonzero) {
    use_device(b);
    if (!dense_strides(c, st)) return -1;
    void* tmp; int dt = dtype, early = 0;
    const void* d = stage_in(b, c, src, dtype, &tmp, &dt, dst64 ? 0 : 4, &early);     // (the fp32 attempt narrows a wide host array; the float64 pass takes it raw)
    if (!d) return -1;
    if (early) { *inexact = 1; HIP_CHECK(hipFree(tmp)); return 0; }
    struct { int flag; int pad; unsigned long long nz; } host = {0, 0, 0}, *dev = nullptr;
    HIP_CHECK(hipMalloc(&dev, sizeof(host))); HIP_CHECK(hipMemsetAsync(dev, 0, sizeof(host), b->sa));
    k_pack_volume<<<voxel_blocks(c), TPB, 0, b->sa>>>(c, dst, dst64, d, dt, st[0], st[1], st[2], &dev->flag, &dev->nz);
    HIP_CHECK(hipMemcpyAsync(&host, dev, sizeof(host), hipMemcpyDeviceToHost, b->sa));
    HIP_CHECK(hipStreamSynchronize(b->sa));
    *inexact = host.flag; if (nonzero) *nonzero = (long long)host.nz;
    HIP_CHECK(hipFree(dev)); if (tmp) HIP_CHECK(hipFree(tmp));
    return 0;
}
int be_pack_labels(VrgBackend* b, const VrgCtx& c, uint8_t* dst, const void* src, int dtype, const int64_t st[3], int* bad) {
    use_device(b);
    if (!dense_strides(c, st)) return -1;
    void* tmp; int dt = dtype, early = 0;
    const void* d = stage_in(b, c, src, dtype, &tmp, &dt, 1, &early);
    if (!d) return -1;
    int* flag; HIP_CHECK(hipMalloc(&flag, sizeof(int))); HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), b->sa));
    k_pack_labels<<<voxel_blocks(c), TPB, 0, b->sa>>>(c, dst, d, dt, st[0], st[1], st[2], flag);
    HIP_CHECK(hipMemcpyAsync(bad, flag, sizeof(int), hipMemcpyDeviceToHost, b->sa));
    HIP_CHECK(hipStreamSynchronize(b->sa));
    HIP_CHECK(hipFree(flag)); if (tmp) HIP_CHECK(hipFree(tmp));
    return 0;
}
template <class T> static void host_widen(void* dst, const uint8_t* src, size_t a, size_t e) { T* o = (T*)dst; for (size_t i = a; i < e; i++) o[i] = (T)src[i]; }
// what: 0 = the labels 0..4 (valueMap on return, :33-36), 1 = segmentedMap (labels <= 1 -> 1, else 0: :31-32)
int be_unpack_labels(VrgBackend* b, const VrgCtx& c, const uint8_t* lab, void* dst, int dtype, const int64_t st[3], int what) {
    use_device(b);
    if (!dense_strides(c, st)) return -1;
    bool dev = is_device_ptr(dst);
    const size_t V = (size_t)c.nx * c.ny * c.nz;
    if (dev) {
        k_unpack_labels<<<voxel_blocks(c), TPB, 0, b->sa>>>(c, lab, dst, dtype, st[0], st[1], st[2], what);
        HIP_CHECK(hipStreamSynchronize(b->sa));
        return 0;
    }
    // a host array: one byte per voxel in the caller's layout comes back, widened on the host chunk by chunk
    void* d = nullptr;
    if (hipMalloc(&d, V) != hipSuccess) { (void)hipGetLastError(); return -1; }
    k_unpack_labels<<<voxel_blocks(c), TPB, 0, b->sa>>>(c, lab, d, VRG_U8, st[0], st[1], st[2], what);
    if (kElem[dtype] == 1) { HIP_CHECK(hipMemcpyAsync(dst, d, V, hipMemcpyDeviceToHost, b->sa)); HIP_CHECK(hipStreamSynchronize(b->sa)); HIP_CHECK(hipFree(d)); return 0; }
    if (!xfer_buffers(b, XFER_CHUNK * 4)) { HIP_CHECK(hipFree(d)); return -1; }
    int k = 0;
    size_t prev0 = 0, prevn = 0; int prevk = -1;
    auto widen = [&](size_t i0, size_t n, int kk) {
        const uint8_t* srcb = (const uint8_t*)b->xfer[kk];
        uint8_t* base = (uint8_t*)dst + i0 * kElem[dtype];
        parallel_chunks(n, [&](size_t a, size_t e) {
            switch (dtype) { case 1: host_widen<int16_t>(base, srcb, a, e); break; case 2: host_widen<uint16_t>(base, srcb, a, e); break; case 3: host_widen<int32_t>(base, srcb, a, e); break;
                             case 4: host_widen<int64_t>(base, srcb, a, e); break; case 5: host_widen<float>(base, srcb, a, e); break; default: host_widen<double>(base, srcb, a, e); break; }
        });
    };
    for (size_t i0 = 0; i0 < V; i0 += XFER_CHUNK, k ^= 1) {
        const size_t n = std::min(XFER_CHUNK, V - i0);
        HIP_CHECK(hipMemcpyAsync(b->xfer[k], (const uint8_t*)d + i0, n, hipMemcpyDeviceToHost, b->sa));
        if (prevk >= 0) widen(prev0, prevn, prevk);        // (the chunk before, while this one travels)
        HIP_CHECK(hipStreamSynchronize(b->sa));
        prev0 = i0; prevn = n; prevk = k;
    }
    if (prevk >= 0) widen(prev0, prevn, prevk);
    HIP_CHECK(hipFree(d));
    return 0;
}

// sorted distinct intensity values (rocPRIM radix sort + unique), as float64
template <class T> static int build_levels_t(VrgBackend* b, const VrgCtx& c, double** lev, uint32_t* L) {
    size_t V = (size_t)c.nx * c.ny * c.nz;
    T *a = nullptr, *bb = nullptr; uint32_t* cnt = nullptr; void* tmp = nullptr; size_t tb = 0, tb2 = 0;
    int rc = -1;
    double* out = nullptr;
    if (hipMalloc(&a, V * sizeof(T)) == hipSuccess && hipMalloc(&bb, V * sizeof(T)) == hipSuccess && hipMalloc(&cnt, 4) == hipSuccess) {
        k_gather_I<T><<<voxel_blocks(c), TPB, 0, b->sa>>>(c, a);
        HIP_CHECK(rocprim::radix_sort_keys(nullptr, tb, a, bb, V, 0, 8 * sizeof(T), b->sa));
        HIP_CHECK(rocprim::unique(nullptr, tb2, bb, a, cnt, V, rocprim::equal_to<T>(), b->sa));
        tb = std::max(tb, tb2);
        if (hipMalloc(&tmp, tb) == hipSuccess) {
            HIP_CHECK(rocprim::radix_sort_keys(tmp, tb, a, bb, V, 0, 8 * sizeof(T), b->sa));
            HIP_CHECK(rocprim::unique(tmp, tb, bb, a, cnt, V, rocprim::equal_to<T>(), b->sa));
            uint32_t n = 0;
            HIP_CHECK(hipMemcpyAsync(&n, cnt, 4, hipMemcpyDeviceToHost, b->sa));
            HIP_CHECK(hipStreamSynchronize(b->sa));
            if (n && hipMalloc(&out, (size_t)n * 8) == hipSuccess) {
                if (sizeof(T) == 4) k_f2d<<<256, TPB, 0, b->sa>>>((const float*)a, out, n);
                else HIP_CHECK(hipMemcpyAsync(out, a, (size_t)n * 8, hipMemcpyDeviceToDevice, b->sa));
                HIP_CHECK(hipStreamSynchronize(b->sa));
                *lev = out; *L = n; rc = 0;
            }
        }
    }
    (void)hipGetLastError();
    if (a) HIP_CHECK(hipFree(a)); if (bb) HIP_CHECK(hipFree(bb)); if (cnt) HIP_CHECK(hipFree(cnt)); if (tmp) HIP_CHECK(hipFree(tmp));
    return rc;
}
int be_build_levels(VrgBackend* b, const VrgCtx& c, double** lev, uint32_t* L) {
    use_device(b);
    return c.I ? build_levels_t<float>(b, c, lev, L) : build_levels_t<double>(b, c, lev, L);
}

__global__ void k_lev_map(VrgCtx c, uint16_t* map, int* bad) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < c.L; k += gridDim.x * blockDim.x) {
        const double v = c.lev[k];
        if (v != floor(v)) *bad = 1; else map[(uint32_t)(v - c.lev[0])] = (uint16_t)k;
    }
}
bool be_build_lev_map(VrgBackend* b, const VrgCtx& c, uint16_t* map, uint32_t span) {
    use_device(b);
    int* bad = nullptr; int hbad = 1;
    if (hipMalloc(&bad, sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return false; }
    HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), b->sa));
    HIP_CHECK(hipMemsetAsync(map, 0xff, (size_t)span * 2, b->sa));
    k_lev_map<<<(c.L + TPB - 1) / TPB, TPB, 0, b->sa>>>(c, map, bad);
    HIP_CHECK(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, b->sa));
    HIP_CHECK(hipStreamSynchronize(b->sa));
    HIP_CHECK(hipFree(bad));
    return hbad == 0;
}

__global__ void k_ktab(VrgCtx c, double* ktab) {
    const uint64_t n = (uint64_t)c.L * c.L;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = (uint32_t)(i / c.L), b = (uint32_t)(i - (uint64_t)a * c.L);
        ktab[i] = vrg_kern(c, c.lev[b] - c.lev[a]);
    }
}
void be_build_ktab(VrgBackend* b, const VrgCtx& c, double* ktab) { use_device(b); k_ktab<<<1024, TPB, 0, b->sa>>>(c, ktab); }

// the bins' moments from the per-level class histograms (init; fixed-point integer adds: any order gives the same bits)
__global__ void k_bins_build(VrgCtx c) {
    for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < c.L; l += gridDim.x * blockDim.x) {
        const int32_t a = c.hin[l], b = c.hout[l];
        if (a | b) vrg_bin_add(c, c.lev[l], a, b);
    }
}
// ... and how many of them differ from `ref_in` / `ref_out` built the same way from other histograms (verification aid)
__global__ void k_bins_diff(VrgCtx c, const int64_t* ref_in, const int64_t* ref_out, unsigned long long* out) {
    const uint64_t n = (uint64_t)c.nb * (VRG_BIN_K + 1);
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) bad += (c.bm_in[i] != ref_in[i]) + (c.bm_out[i] != ref_out[i]);
    if (bad) atomicAdd(out, bad);
}
void be_build_bins(VrgBackend* b, const VrgCtx& c) {
    use_device(b);
    HIP_CHECK(hipMemsetAsync(c.bm_in, 0, (size_t)c.nb * (VRG_BIN_K + 1) * 8, b->sa)); HIP_CHECK(hipMemsetAsync(c.bm_out, 0, (size_t)c.nb * (VRG_BIN_K + 1) * 8, b->sa));
    k_bins_build<<<2048, TPB, 0, b->sa>>>(c);
}
long long be_check_bins(VrgBackend* b, const VrgCtx& c, const int32_t* rin, const int32_t* rout) {
    use_device(b);
    if (!c.nb) return 0;
    const size_t bytes = (size_t)c.nb * (VRG_BIN_K + 1) * 8;
    int64_t *ri = nullptr, *ro = nullptr; unsigned long long* d = nullptr; unsigned long long bad = ~0ull;
    if (hipMalloc(&ri, bytes) == hipSuccess && hipMalloc(&ro, bytes) == hipSuccess && hipMalloc(&d, 8) == hipSuccess) {
        VrgCtx r = c;
        r.bm_in = ri; r.bm_out = ro; r.hin = const_cast<int32_t*>(rin); r.hout = const_cast<int32_t*>(rout);
        HIP_CHECK(hipMemsetAsync(ri, 0, bytes, b->sa)); HIP_CHECK(hipMemsetAsync(ro, 0, bytes, b->sa)); HIP_CHECK(hipMemsetAsync(d, 0, 8, b->sa));
        k_bins_build<<<2048, TPB, 0, b->sa>>>(r);
        k_bins_diff<<<256, TPB, 0, b->sa>>>(c, ri, ro, d);
        HIP_CHECK(hipMemcpyAsync(&bad, d, 8, hipMemcpyDeviceToHost, b->sa));
        HIP_CHECK(hipStreamSynchronize(b->sa));
    }
    (void)hipGetLastError();
    if (ri) HIP_CHECK(hipFree(ri)); if (ro) HIP_CHECK(hipFree(ro)); if (d) HIP_CHECK(hipFree(d));
    return (long long)bad;
}

__global__ void k_build_lidx(VrgCtx c, uint32_t* dst) {
    VOXEL_LOOP(c) { int x, y, z; uint32_t idx = real_idx(c, t, x, y, z); dst[idx] = vrg_level_of(c, vrg_voxel_value(c, idx)); }
}
void be_build_lidx(VrgBackend* b, const VrgCtx& c, uint32_t* dst) {
    use_device(b);
    HIP_CHECK(hipMemsetAsync(dst, 0, ((size_t)c.PV + 1023) / 1024 * 1024 * 4, b->sa));
    k_build_lidx<<<voxel_blocks(c), TPB, 0, b->sa>>>(c, dst);
}
void be_build_lev16(VrgBackend* b, const VrgCtx& c, uint16_t* dst) {
    use_device(b);
    HIP_CHECK(hipMemsetAsync(dst, 0, ((size_t)c.PV + 1023) / 1024 * 1024 * 2, b->sa));
    k_build_lev16<<<voxel_blocks(c), TPB, 0, b->sa>>>(c, dst);
}

void be_init_band(VrgBackend* b, const VrgCtx& c) {
    use_device(b);
    k_init_voxel<<<voxel_blocks(c), TPB, 0, b->sa>>>(c);
}

void be_init_sort(VrgBackend* b, const VrgCtx& c, uint32_t n_in, uint32_t n_out) {
    use_device(b);
    uint32_t nmax = std::max(n_in, n_out);
    if (nmax == 0) return;
    uint64_t* kout = nullptr; void* tmp = nullptr; size_t tb = 0;
    HIP_CHECK(hipMalloc(&kout, (size_t)nmax * 8));
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tb, c.init_key, kout, c.init_idx, c.p_idx, nmax, 0, 64, b->sa));
    HIP_CHECK(hipMalloc(&tmp, tb));
    if (n_in) HIP_CHECK(rocprim::radix_sort_pairs(tmp, tb, c.init_key, kout, c.init_idx, c.p_idx, n_in, 0, 64, b->sa));
    if (n_out) HIP_CHECK(rocprim::radix_sort_pairs(tmp, tb, c.init_key + (c.bcap - n_out), kout, c.init_idx + (c.bcap - n_out),
                                                   c.p_idx + n_in, n_out, 0, 64, b->sa));
    HIP_CHECK(hipStreamSynchronize(b->sa));
    HIP_CHECK(hipFree(kout)); HIP_CHECK(hipFree(tmp));
}

// sum the slab statistics over the ranks: RCCL on the stream, or the host callback (synchronises)
static void reduce_dense(VrgBackend* b, const VrgCtx& c, be_reduce_fn cb, void* user, hipStream_t st) {
    if (b->repl) return;                             // (every rank of a leader / follower group counts whole volumes: nothing to sum)
    if (b->comm) {
        ncclResult_t r = ncclAllReduce(c.dn_part, c.dn, 4, ncclDouble, ncclSum, b->comm, st);
        if (r != ncclSuccess && !b->err[0]) {        // sticky: the engine turns it into VRG_E_INTERNAL at its next synchronisation point
            std::snprintf(b->err, sizeof(b->err), "RCCL all-reduce of the slab statistics failed: %s", ncclGetErrorString(r));
            std::fprintf(stderr, "%s\n", b->err);
        }
    } else if (cb) {
        double v[4];
        HIP_CHECK(hipMemcpyAsync(v, c.dn_part, sizeof(v), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        cb(v, user);
        HIP_CHECK(hipMemcpyAsync(c.dn, v, sizeof(v), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
}

// Z-slabs: pack the slab sums of the recounts not yet closed, sum them over the ranks (RCCL on the dense stream, or the
// host callback once per entry), close those passes
constexpr int DENSE_GROUP = 8;
static_assert(DENSE_GROUP <= VRG_STAGE, "staging buffer");
static void reduce_staged(VrgBackend* b, const VrgCtx& c, be_reduce_fn cb, void* user) {
    b->dense_pending = 0;
    k_dense_pack<<<1, 1, 0, b->sb>>>(c);
    if (b->comm) {
        ncclResult_t r = ncclAllReduce(c.stage_in, c.stage_out, 4 * VRG_STAGE, ncclDouble, ncclSum, b->comm, b->sb);
        if (r != ncclSuccess && !b->err[0]) {        // sticky: the engine turns it into VRG_E_INTERNAL at its next synchronisation point
            std::snprintf(b->err, sizeof(b->err), "RCCL all-reduce of the slab statistics failed: %s", ncclGetErrorString(r));
            std::fprintf(stderr, "%s\n", b->err);
        }
    } else if (cb) {
        double v[4 * VRG_STAGE]; int64_t n = 0;
        HIP_CHECK(hipMemcpyAsync(v, c.stage_in, sizeof(v), hipMemcpyDeviceToHost, b->sb));
        HIP_CHECK(hipMemcpyAsync(&n, c.dctl + VD_NST, sizeof(n), hipMemcpyDeviceToHost, b->sb));
        HIP_CHECK(hipStreamSynchronize(b->sb));
        for (int64_t j = 0; j < n && j < VRG_STAGE; j++) cb(v + 4 * j, user);
        HIP_CHECK(hipMemcpyAsync(c.stage_out, v, sizeof(v), hipMemcpyHostToDevice, b->sb));
        HIP_CHECK(hipStreamSynchronize(b->sb));
    } else {
        HIP_CHECK(hipMemcpyAsync(c.stage_out, c.stage_in, VRG_STAGE * sizeof(VrgDense), hipMemcpyDeviceToDevice, b->sb));
    }
    k_dense_fin<<<1, 1, 0, b->sb>>>(c);
}

int be_comm_unique_id(void* id128) {
    static_assert(sizeof(ncclUniqueId) == 128, "id size");
    return ncclGetUniqueId((ncclUniqueId*)id128) == ncclSuccess ? 0 : -1;
}
int be_comm_init(VrgBackend* b, int nranks, int rank, const void* id128) {
    use_device(b);
    if (b->comm) { ncclCommDestroy(b->comm); b->comm = nullptr; }
    ncclUniqueId id; std::memcpy(&id, id128, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&b->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        b->comm = nullptr;
        if (!b->err[0]) std::snprintf(b->err, sizeof(b->err), "ncclCommInitRank(%d ranks, rank %d) failed: %s", nranks, rank, ncclGetErrorString(r));
        return -1;
    }
    return 0;
}

// The start / stop events ride on the dispatch itself (hipExtLaunchKernel): no separate event packets in the stream,
// which cost ~4 us each between two back-to-back recounts.
template <bool NT, bool SKIP>
static void launch_recount_as(const VrgCtx& c, int blocks, int check, hipStream_t st, hipEvent_t e_start, hipEvent_t e_stop) {
    if (c.lev16 && c.L <= TAB64_LEVELS) hipExtLaunchKernelGGL((k_recount_bits<3, NT, 3, SKIP>), dim3(blocks), dim3(TPB), c.L * sizeof(double), st, e_start, e_stop, 0, c, check);
    else if (c.lev16) hipExtLaunchKernelGGL((k_recount_bits<3, NT, 1, SKIP>), dim3(blocks), dim3(TPB), c.L * sizeof(float), st, e_start, e_stop, 0, c, check);
    else if (c.I) hipExtLaunchKernelGGL((k_recount_bits<3, NT, 0, SKIP>), dim3(blocks), dim3(TPB), 0, st, e_start, e_stop, 0, c, check);
    else hipExtLaunchKernelGGL((k_recount_bits<2, NT, 2, SKIP>), dim3(blocks), dim3(TPB), 0, st, e_start, e_stop, 0, c, check);
}
// nt: non-temporal loads - for a pass that is larger than the 256-MiB Infinity Cache, where nothing is worth keeping;
// a smaller slab is read with ordinary loads and then comes out of that cache sweep after sweep.
static void launch_recount(const VrgCtx& c, int blocks, int check, hipStream_t st, bool skip, bool nt, hipEvent_t e_start = nullptr, hipEvent_t e_stop = nullptr, int every = 1) {
    if (check == 1 || check == 2) k_gate<<<1, GATE_THREADS, 0, st>>>(c, every, check);        // waits (on the device) until the sweep's labels are in place; keeps the unit list current
    if (skip) { if (nt) launch_recount_as<true, true>(c, blocks, check, st, e_start, e_stop); else launch_recount_as<false, true>(c, blocks, check, st, e_start, e_stop); }
    else { if (nt) launch_recount_as<true, false>(c, blocks, check, st, e_start, e_stop); else launch_recount_as<false, false>(c, blocks, check, st, e_start, e_stop); }
}

void be_init_finish(VrgBackend* b, const VrgCtx& c, be_reduce_fn cb, void* user) {
    use_device(b);
    b->dense_pending = 0;
    HIP_CHECK(hipStreamSynchronize(b->sb));     // both class copies are rebuilt: no dense pass may be in flight
    k_init_entry<<<ITEM_BLOCKS, TPB, 0, b->sa>>>(c);
    if (c.I && c.L <= HIST_LDS_LEVELS) k_hist_lds<<<1024, TPB, 0, b->sa>>>(c);
    else k_hist_voxel<<<voxel_blocks(c), TPB, 0, b->sa>>>(c);
    if (c.nb) be_build_bins(b, c);              // (large level table: the histograms also as bin moments, before the first exact densities)
    k_exact_init<<<1024, TPB, 0, b->sa>>>(c);
    k_cls_build<<<2048, TPB, 0, b->sa>>>(c);
    k_ulist_init<<<1, GATE_THREADS, 0, b->sa>>>(c);
    launch_recount(c, dense_blocks(b, c), 0, b->sa, b->skip != 0, dense_nt(b, c));
    reduce_dense(b, c, cb, user, b->sa);
    k_fin_init<<<1, 1, 0, b->sa>>>(c);
    b->pass_bytes = be_dense_bytes(b, c);       // (decides between ordinary and non-temporal loads for the sweeps' passes)
}

// ---- one trip ---------------------------------------------------------------------------------------------
// Stream A ("band") carries k_band and the update() kernels of every trip in program order; stream B ("dense") carries the recounts
// (+ the slab all-reduce and k_dense_fin on several GPUs).  The only edges between them:
//   recount(k) waits for k_close(k)        (labels of sweep k in class copy k & 1, expected sizes filed)
//   k_close(k) waits for recount(k-2)      (it rewrites class copy k & 1, which pass k-2 was reading)
// Neither wait blocks in steady state: on one big volume stream A is a sweep ahead and stream B runs its recounts
// back to back; on small slabs stream B is idle most of the time and stream A never finds pass k-2 unfinished.
// The band kernels read and write the label BYTES only; the dense pass reads the class bits only.

// update() driven from the host: any number of flips.  Returns after the band side of the trip is enqueued.
static void host_driven_update(VrgBackend* b, const VrgCtx& c, int flags) {
    k_trip_open<<<1, TPB, 0, b->sa>>>(c);
    VrgState s;
    HIP_CHECK(hipMemcpyAsync(&s, c.st, sizeof(s), hipMemcpyDeviceToHost, b->sa));
    HIP_CHECK(hipStreamSynchronize(b->sa));
    if (s.done || s.bail || b->err[0]) return;
    const uint32_t nf = s.nf;
    // the flip list in the reference's order: device-wide sort by (list, key)
    size_t tb = 0;
    if (!need_keys2(b, nf)) { std::snprintf(b->err, sizeof(b->err), "out of device memory (flip sort)"); return; }
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tb, c.f_key, b->keys2, c.flist, c.f_slot, nf, 0, 64, b->sa));
    if (!need_tmp(b, tb)) { std::snprintf(b->err, sizeof(b->err), "out of device memory (flip sort)"); return; }
    HIP_CHECK(rocprim::radix_sort_pairs(b->tmp, tb, c.f_key, b->keys2, c.flist, c.f_slot, nf, 0, 64, b->sa));
    k_list<<<ITEM_BLOCKS, TPB, 0, b->sa>>>(c, nf);
    if (flags & VRG_SWEEP_FULL) k_prepass<<<ITEM_BLOCKS, TPB, 0, b->sa>>>(c, nf);
    else k_marks_prepass<<<4 * ITEM_BLOCKS, TPB, 0, b->sa>>>(c, nf);
    k_fix<<<1, KS_THREADS, 0, b->sa>>>(c);
    if (flags & VRG_SWEEP_FULL) k_full_relabel<<<2048, TPB, 0, b->sa>>>(c);
    else k_relabel<<<ITEM_BLOCKS, TPB, 0, b->sa>>>(c);
    // the labels change now, in the class copy the dense pass of two sweeps ago was reading
    if (!(flags & VRG_SWEEP_NODENSE)) k_wait_dense<<<1, 64, 0, b->sa>>>(c);
    if (flags & VRG_SWEEP_FULL) k_copy_back<<<2048, TPB, 0, b->sa>>>(c);
    else k_apply<<<ITEM_BLOCKS, TPB, 0, b->sa>>>(c);
    k_close_items<<<ITEM_BLOCKS, TPB, 0, b->sa>>>(c, nf);
    HIP_CHECK(hipMemcpyAsync(&s, c.st, sizeof(s), hipMemcpyDeviceToHost, b->sa));
    HIP_CHECK(hipStreamSynchronize(b->sa));
    const uint32_t nnz = std::min(s.nnz, c.zcap);
    if (nnz > 1) {                                   // touched levels in ascending order
        if (!need_keys2(b, nnz)) { std::snprintf(b->err, sizeof(b->err), "out of device memory (level sort)"); return; }
        HIP_CHECK(rocprim::radix_sort_keys(nullptr, tb, c.nz_key, b->keys2, nnz, 0, 32, b->sa));
        if (!need_tmp(b, tb)) { std::snprintf(b->err, sizeof(b->err), "out of device memory (level sort)"); return; }
        HIP_CHECK(rocprim::radix_sort_keys(b->tmp, tb, c.nz_key, b->keys2, nnz, 0, 32, b->sa));
        HIP_CHECK(hipMemcpyAsync(c.nz_key, b->keys2, (size_t)nnz * 8, hipMemcpyDeviceToDevice, b->sa));
    }
    k_levels<<<ITEM_BLOCKS, TPB, 0, b->sa>>>(c, nnz);
    // memoise the corrections per level when there are far fewer levels than band entries
    const uint64_t band = (uint64_t)((int64_t)s.ni + s.d_ni) + (uint64_t)((int64_t)s.no + s.d_no);
    const int use_tab = (uint64_t)c.L * 8u <= band;
    if (use_tab) k_tab<<<ITEM_BLOCKS, TPB, 0, b->sa>>>(c, nnz);
    k_finalize<<<1, 1, 0, b->sa>>>(c, use_tab);
}

// update() for a sweep with few flips: three launches, nothing from the host in between
// (host_nf > 0: a host-driven trip - the host has read the flip count: the launches are sized for it, and the flips are ranked by a radix sort instead of k_rank_wide's
// n^2 comparisons, which at 10^5 flips would take milliseconds)
static void small_update(VrgBackend* b, const VrgCtx& c0, bool dense, hipEvent_t e_chain_stop = nullptr, uint32_t host_nf = 0) {
    VrgCtx c = c0;
    if (!b->rsv) { HIP_CHECK(hipMalloc((void**)&b->rsv, 64 * sizeof(uint64_t))); if (b->rsv) HIP_CHECK(hipMemsetAsync(b->rsv, 0, 64 * sizeof(uint64_t), b->sa)); }
    c.rsv = b->rsv;                                  // (the relabel kernels' list reservations: lines of their own)
    c.lvl_scan = c.L <= NZ_SORT ? 1 : 0;             // small level table: the touched levels are found by scanning the counters (k_close)
    // (sized by the flips of the last sweep the engine saw: a sweep of thousands of flips gets a workgroup per flip, not a queue of them;
    // a sweep with more flips than its launches can order is handed back - VBAIL_FLIPS - and enqueued again with launches that can)
    const uint32_t fh = host_nf ? host_nf : std::max<uint32_t>(b->flip_hint, 1u);
    const bool sorted = host_nf > NF_WIDE;             // (more flips than the device-resident chain takes on its own)
    const bool wide = sorted || 2 * (uint64_t)fh > NF_ORDER;
    k_order<<<1, KO_THREADS, 0, b->sa>>>(c, sorted ? host_nf : wide ? b->small_flips : std::min<uint32_t>(b->small_flips, NF_ORDER));
    if (sorted) {
        size_t tb = 0;
        if (!need_keys2(b, host_nf)) { std::snprintf(b->err, sizeof(b->err), "out of device memory (flip sort)"); return; }
        HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tb, c.f_key, b->keys2, rocprim::counting_iterator<uint32_t>(0u), c.slow, host_nf, 0, 64, b->sa));
        if (!need_tmp(b, tb)) { std::snprintf(b->err, sizeof(b->err), "out of device memory (flip sort)"); return; }
        HIP_CHECK(rocprim::radix_sort_pairs(b->tmp, tb, c.f_key, b->keys2, rocprim::counting_iterator<uint32_t>(0u), c.slow, host_nf, 0, 64, b->sa));
        k_rank_scatter<<<std::min<uint32_t>(1024u, (host_nf + TPB - 1) / TPB), TPB, 0, b->sa>>>(c, c.slow, host_nf);      // (c.slow: free until k_mark_compact fills it)
    }
    if (wide) {                  // its ordering step chip-wide (no-ops when k_order did the ordering itself)
        const uint64_t nrec = (2 * (uint64_t)fh + KR_THREADS - 1) / KR_THREADS, ntile = (2 * (uint64_t)fh + KR_TILE - 1) / KR_TILE;
        if (!sorted) k_rank_wide<<<(uint32_t)std::min<uint64_t>(16384u, nrec * ntile), KR_THREADS, 0, b->sa>>>(c);
        k_list_wide<<<std::min<uint32_t>(1024u, (2 * fh + TPB - 1) / TPB), TPB, 0, b->sa>>>(c);
        k_prepass_wide<<<std::min<uint32_t>(1024u, (2 * fh + TPB - 1) / TPB), TPB, 0, b->sa>>>(c);
        k_fix_wide<<<1, 1024, 0, b->sa>>>(c);
    }
    // (a workgroup per flip up to KM_BLOCKS flips; beyond, every workgroup takes several and files what they add to the lists together)
    const size_t lev_lds = ((c.L <= LEV_LDS && !c.lev16) ? (size_t)c.L * sizeof(double) : 0) + ((c.lvl_scan == 1 && c.L <= HIST_LDS) ? 5 * (size_t)c.L * sizeof(uint32_t) : 0);   // level table + per-level counts
    // (measured at 12 900 flips, ms per sweep: 4 flips at a time x 256 / 512 / 1024 workgroups 0.340 / 0.348 / 0.380; 2 x 1024 / 2048: 0.40 / 0.50; 1 x 2048 / 4096:
    // 0.52 / 0.70 - every workgroup more is six more reservations on the same few words)
    // (thousands of flips: the compact form first - a flip per half-wave, everything but flips with an excluded voxel in their cube - then the general
    // form over the flips it left; option "mark_compact" = 0: the general form alone, as up to round 5)
    if (2 * (uint64_t)fh > KM_BLOCKS) {
        if (b->mark_compact) {
            k_mark_compact<<<KMC_BLOCKS, KMC_THREADS, lev_lds, b->sa>>>(c, c.slow, c.counters + 48);
            k_mark_relabel<4><<<KM_BLOCKS_WIDE, 4 * KM_THREADS, lev_lds, b->sa>>>(c, c.slow, c.counters + 48);
        } else k_mark_relabel<4><<<KM_BLOCKS_WIDE, 4 * KM_THREADS, lev_lds, b->sa>>>(c, nullptr, nullptr);
    }
    else k_mark_relabel<1><<<KM_BLOCKS, KM_THREADS, lev_lds, b->sa>>>(c, nullptr, nullptr);
    // (waits on the device for the dense pass of two sweeps ago)
    const uint32_t napply = std::max<uint32_t>(CLOSE_APPLY, std::min<uint32_t>(1024u, fh / 8u));
    hipExtLaunchKernelGGL(k_close, dim3(napply + TAB_BLOCKS), dim3(KC_THREADS), 0, b->sa, nullptr, e_chain_stop, 0, c, dense ? 1 : 0, napply);
}

static void enqueue_dense(VrgBackend* b, const VrgCtx& c, hipEvent_t e_start, hipEvent_t e_stop, be_reduce_fn cb, void* user);

// (before the first fused trip of a batch: the live counters of the buffer its k_band decides into - inside a run of fused trips every
// k_sweep sets them up for the trip after it, vrg_fuse_prepare_other; the host may have rewritten the state in between)
__global__ void k_state_prep(const VrgState* in, VrgState* out) { out->nf = 0; out->ties = in->ties; out->near_ties = in->near_ties; out->error = in->error; }
constexpr uint32_t OPEN_LEVELS = 1024;               // level tables up to this size run open-ended sweeps (k_band lists the touched levels from 4 counters per thread)

void be_sweep_once(VrgBackend* b, VrgCtx& c, int flags, VrgEvents* ev, be_reduce_fn cb, void* user, bool first, bool last) {
    use_device(b);
    hipEvent_t e_start = nullptr, e_stop = nullptr;
    const bool dense = !(flags & VRG_SWEEP_NODENSE);
    const long long trip = b->ev_trip++;
    auto take_pair = [&](int kind) -> EvPair& {
        // (32 pairs at a time: an event costs ~10-20 us to create, and a short run - the driver's 20 steps - should not pay
        // for its events inside its timed sweeps; the first sweep of a handle that times anything creates the lot)
        if (b->ev_used == b->ev_pool.size())
            for (int k = 0; k < 32; k++) { EvPair n; HIP_CHECK(hipEventCreate(&n.a)); HIP_CHECK(hipEventCreate(&n.b)); n.trip = 0; n.kind = 0; n.ntrips = 1; b->ev_pool.push_back(n); }
        EvPair& p = b->ev_pool[b->ev_used++];
        p.trip = trip; p.kind = kind; p.ntrips = 1;
        return p;
    };
    size_t dense_pair = (size_t)-1;
    if (dense && ev && ev->enabled > 0 && trip % ev->enabled == 0) {     // (every enabled-th trip: an event pair costs the dense stream a few us)
        EvPair& p = take_pair(0);
        e_start = p.a; e_stop = p.b; dense_pair = b->ev_used - 1;
    }
    hipEvent_t e_c0 = nullptr, e_c1 = nullptr;                           // the band chain of this trip: k_band's start to k_close's end
    if (ev && ev->chain_enabled > 0 && !(flags & VRG_SWEEP_SYNC) && trip % ev->chain_enabled == 0) {
        EvPair& p = take_pair(1);
        e_c0 = p.a; e_c1 = p.b;
    }
    const uint32_t nbb = band_blocks(b);
    const bool fused_trip = (flags & VRG_SWEEP_FUSED) && !(flags & (VRG_SWEEP_SYNC | VRG_SWEEP_FULL));
    // a fused trip reads the state in one buffer and files it into the other (vrg_items.h "open-ended sweeps"); every other kind works in place
    VrgState* const st_in = c.st;
    VrgState* const st_out = fused_trip ? (c.st == c.stb[0] ? c.stb[1] : c.stb[0]) : c.st;
    if (fused_trip && (first || !b->fused_prev)) k_state_prep<<<1, 1, 0, b->sa>>>(st_in, st_out);
    // (grid: the pool's workgroups, the exact-density ones, and - in and behind a fused trip - the ones that file the state and carry out what the sweep before deferred)
    {
        VrgCtx cb_ = c;
        cb_.st = st_in; cb_.stg = st_out; cb_.lvl_par = b->prev_open ? b->open_par : -1;
        cb_.inc_in = c.inc; cb_.inc = fused_trip ? (c.inc == c.incb[0] ? c.incb[1] : c.incb[0]) : c.inc;
        const dim3 grid(nbb + EXACT_BLOCKS + ((b->fused_prev || fused_trip) ? DEFER_WGS : 0));
        const int lanes = band_lanes(b), dh = band_direct(b) ? 1 : 0, don = dense ? 1 : 0;
        if (lanes == 16) hipExtLaunchKernelGGL(k_band<16>, grid, dim3(TPB), 0, b->sa, e_c0, nullptr, 0, cb_, nbb, don, dh);
        else if (lanes == 8) hipExtLaunchKernelGGL(k_band<8>, grid, dim3(TPB), 0, b->sa, e_c0, nullptr, 0, cb_, nbb, don, dh);
        else hipExtLaunchKernelGGL(k_band<4>, grid, dim3(TPB), 0, b->sa, e_c0, nullptr, 0, cb_, nbb, don, dh);
    }
    c.st = c.stg = st_out; c.st_other = st_in;             // (where the state is from here on; k_sweep sets up the buffer just read for the next trip's decisions)
    if (fused_trip) c.inc = c.inc == c.incb[0] ? c.incb[1] : c.incb[0];
    c.inc_in = c.inc;
    const int sweep_par = (b->iter_hint + 1) & 1;          // the sweep this trip applies, if it applies one
    b->iter_hint++;
    b->prev_open = false;
    // A fused trip leaves the labels of the sweep it applies to the NEXT trip's k_band, which also asks for that sweep's dense
    // pass: the pass is therefore enqueued here, right behind the k_band that raises its request - never earlier: a gate that
    // waits for a request nobody has enqueued yet would block every host synchronisation of the dense stream.
    if (b->fused_prev && dense) {
        if (dense_pair != (size_t)-1) b->ev_pool[dense_pair].trip = trip - 1;     // (the pass of the sweep BEFORE this trip: it counts if that sweep was applied)
        enqueue_dense(b, c, e_start, e_stop, cb, user);
        e_start = e_stop = nullptr;
    }
    b->fused_prev = false;
    if (fused_trip) {
        // update() as ONE launch; on a large band a second one memoises the sweep's corrections per level
        const bool memo = !b->direct_hint && b->band_hint > b->memo_above && c.ktab;
        b->memo_trips += memo;
        // open-ended: no closing workgroup - the next trip's k_band derives the closed state.  Not the last trip of a batch (the host reads
        // closed states only), not in front of the memo kernel, small level tables only.
        const bool open = b->open_sweeps && !last && !memo && c.L <= OPEN_LEVELS && c.ktab;
        if (c.L > (uint32_t)VRG_FUSE_LEVELS) hipExtLaunchKernelGGL(k_sweep<true>, dim3(VRG_FUSE_MAX_BIG), dim3(VRG_FUSE_THREADS), 0, b->sa, nullptr, memo ? nullptr : e_c1, 0, c, 0, 0, sweep_par ^ 1);
        else hipExtLaunchKernelGGL(k_sweep<false>, dim3(VRG_FUSE_MAX), dim3(VRG_FUSE_THREADS), 0, b->sa, nullptr, memo ? nullptr : e_c1, 0, c, memo ? 1 : 0, open ? 1 : 0, sweep_par ^ 1);
        if (memo) hipExtLaunchKernelGGL(k_memo, dim3(MEMO_BLOCKS), dim3(TPB), 0, b->sa, nullptr, e_c1, 0, c);
        b->fused_prev = true; b->fused_memo = memo; b->prev_open = open; b->open_par = sweep_par;
        return;
    }
    if (flags & VRG_SWEEP_SYNC) {
        VrgState s;
        HIP_CHECK(hipMemcpyAsync(&s, c.st, sizeof(s), hipMemcpyDeviceToHost, b->sa));
        HIP_CHECK(hipStreamSynchronize(b->sa));
        if (s.done || s.bail) return;
        // (more flips than the device-resident chain takes - 65 536 - : the same chip-wide kernels with a host-sized radix sort for the ranking; the item kernels
        // of host_driven_update remain for the full-stencil check variant and for a handle whose "small_flips" was lowered - the tests do that to run them)
        if (flags & VRG_SWEEP_FULL) host_driven_update(b, c, flags);
        else if (s.nf > b->small_flips) { if (b->small_flips >= NF_WIDE) small_update(b, c, dense, nullptr, s.nf); else host_driven_update(b, c, flags); }
        else small_update(b, c, dense);
    } else {
        small_update(b, c, dense, e_c1);
    }
    if (!dense) return;
    enqueue_dense(b, c, e_start, e_stop, cb, user);
}

// dense stream: every voxel once, read-only; k_gate in front of the recount waits until the sweep's labels are in place.
// (Option "serial_streams", for tools that run one kernel at a time - rocprofv3 --pmc does: a kernel that waits on the
// device for another one could then wait for ever, so the host orders the two streams instead.)
static void enqueue_dense(VrgBackend* b, const VrgCtx& c, hipEvent_t e_start, hipEvent_t e_stop, be_reduce_fn cb, void* user) {
    if (b->serial) HIP_CHECK(hipStreamSynchronize(b->sa));
    const bool ranks = !b->repl && (c.world > 1 || b->comm || cb);
    if (b->dense_pipe && c.I && !c.lev16 && b->skip) {
        const int check = ranks ? 1 : 2;
        k_gate<<<1, GATE_THREADS, 0, b->sb>>>(c, b->verify_every, check);        // (also when every pass is counted: with several verifiers this handle counts its share)
        if (dense_nt(b, c)) hipExtLaunchKernelGGL((k_recount_pipe<3, true>), dim3(dense_blocks(b, c)), dim3(TPB), 0, b->sb, e_start, e_stop, 0, c, check);
        else hipExtLaunchKernelGGL((k_recount_pipe<3, false>), dim3(dense_blocks(b, c)), dim3(TPB), 0, b->sb, e_start, e_stop, 0, c, check);
    } else
    launch_recount(c, dense_blocks(b, c), ranks ? 1 : 2, b->sb, b->skip != 0, dense_nt(b, c), e_start, e_stop, b->verify_every);
    if (b->serial) HIP_CHECK(hipStreamSynchronize(b->sb));
    // one GPU: the last workgroup of the recount closes the pass itself.  Z-slabs: the slab sums of DENSE_GROUP recounts
    // are summed over the ranks by ONE all-reduce (nothing on the band side waits for it: the decisions use the
    // incremental sizes; the totals are only cross-checked against them and filed in the trace)
    if (ranks && ++b->dense_pending >= DENSE_GROUP) reduce_staged(b, c, cb, user);
}

// n trips in a row: what the engine enqueues between two looks at the state
void be_sweep_batch(VrgBackend* b, VrgCtx& c, int flags, int n, VrgEvents* ev, be_reduce_fn cb, void* user) {
    for (int i = 0; i < n; i++) be_sweep_once(b, c, flags, ev, cb, user, i == 0, i == n - 1);
}

// option verify_every != 1, at the end of a run (both streams idle, every pass closed): the labels of the last sweep counted
// after all and compared with the sizes kept by increments (collective on several ranks)
void be_verify_last(VrgBackend* b, const VrgCtx& c, be_reduce_fn cb, void* user) {
    use_device(b);
    launch_recount(c, dense_blocks(b, c), 3, b->sa, b->skip != 0, dense_nt(b, c));      // (check 3: no gate - launch_recount puts one in front of checks 1 and 2 only)
    reduce_dense(b, c, cb, user, b->sa);
    k_verify_last<<<1, 1, 0, b->sa>>>(c);
    HIP_CHECK(hipStreamSynchronize(b->sa));
}

void be_dense_flush(VrgBackend* b, const VrgCtx& c, be_reduce_fn cb, void* user) {
    use_device(b);
    const bool ranks = !b->repl && (c.world > 1 || b->comm || cb);
    if (ranks && b->dense_pending) reduce_staged(b, c, cb, user);
}

// ---- leader / follower replication: the follower's side, and the transports -----------------------------------------------------
struct FollowGroup { VrgLogSweep h[8]; int n; int count_last; };
// label bytes and stamps: ONE workgroup, the sweeps in order (a voxel may change in consecutive sweeps) with a barrier between them.
// Runs beside a dense pass, where every dependent load takes 2-3 us: a thread's records of a sweep are fetched together, then the label
// bytes they name, then the stores go out - two round trips per sweep whatever its length (up to FQ x 1024 records; more: another turn).
constexpr int FQ = 8;
typedef uint32_t fu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ VrgLogRec follow_rec(const VrgLogRec* p) {
    const fu4 v = *reinterpret_cast<const fu4*>(p);
    VrgLogRec r; r.idx = v.x; r.rank = v.y; r.old = (uint8_t)v.z; r.nw = (uint8_t)(v.z >> 8); r.pad = 0; r.pad2 = 0;
    return r;
}
__global__ void __launch_bounds__(GATE_THREADS) k_follow_labels(VrgCtx c, const VrgLogRec* __restrict__ recs, FollowGroup g) {
    const uint32_t t = threadIdx.x;
    const uint32_t safe = vrg_idx(c, 0, 0, 0);
    for (int s = 0; s < g.n; s++) {
        const VrgLogRec* r = recs + g.h[s].rec0;
        const uint32_t n = g.h[s].nrec, k = g.h[s].sweep;
        for (uint32_t i0 = 0; i0 < n; i0 += FQ * GATE_THREADS) {
            VrgLogRec q[FQ]; uint8_t have[FQ];
#pragma unroll
            for (int j = 0; j < FQ; j++) { const uint32_t i = i0 + j * GATE_THREADS + t; q[j] = follow_rec(r + (i < n ? i : n - 1u)); if (i >= n) q[j].idx = VRG_NONE; }
#pragma unroll
            for (int j = 0; j < FQ; j++) have[j] = vrg_load_coherent(c.lab[0] + (q[j].idx != VRG_NONE ? q[j].idx : safe));      // (unconditional: a load under a branch would wait for the ones before it; past L1: another wave of this workgroup may have written the byte a sweep ago)
#pragma unroll
            for (int j = 0; j < FQ; j++) {
                if (q[j].idx == VRG_NONE) continue;
                if ((uint8_t)(have[j] & (VB_LABEL | VB_OOB)) != q[j].old) {      // this rank's labels have drifted from the leader's
                    if (c.dctl[VD_ERR] == 0) { c.dctl[VD_ERR] = 12; c.fexp[3] = (int64_t)k; c.fexp[4] = (int64_t)q[j].idx; c.fexp[5] = (int64_t)have[j]; c.fexp[6] = (int64_t)q[j].old; c.fexp[7] = (int64_t)q[j].nw; }
                    continue;
                }
                c.lab[0][q[j].idx] = q[j].nw;
                if ((q[j].nw & VB_S) && !(q[j].old & VB_S)) c.stamp[q[j].idx] = ((uint64_t)k << 32) | q[j].rank;
            }
        }
        __syncthreads();
    }
}
// class bits (their changes commute: no order between the sweeps - the group's records are one stretch of the batch), trace records;
// then - when the last sweep of the group is counted next - the unit list and what the count has to reproduce
__global__ void __launch_bounds__(GATE_THREADS) k_follow_classes(VrgCtx c, const VrgLogRec* __restrict__ recs, FollowGroup g) {
    const uint32_t t = threadIdx.x;
    if (t < (uint32_t)g.n) vrg_follow_trace(c, g.h[t]);
    const uint32_t first = g.h[0].rec0, n = g.h[g.n - 1].rec0 + g.h[g.n - 1].nrec - first;
    const VrgLogRec* r = recs + first;
    for (uint32_t i0 = 0; i0 < n; i0 += FQ * GATE_THREADS) {
        VrgLogRec q[FQ];
#pragma unroll
        for (int j = 0; j < FQ; j++) { const uint32_t i = i0 + j * GATE_THREADS + t; q[j] = follow_rec(r + (i < n ? i : n - 1u)); if (i >= n) q[j].idx = VRG_NONE; }
#pragma unroll
        for (int j = 0; j < FQ; j++) vrg_follow_class_rec(c, q[j]);
    }
    if (!g.count_last) return;
    if (t == 0) vrg_follow_expect(c, g.h[g.n - 1]);
    vrg_drain();
    __syncthreads();
    ulist_refresh(c, false, 0);
}

static hipStream_t label_stream(VrgBackend* b) {
    if (!b->sd) HIP_CHECK(hipStreamCreateWithFlags(&b->sd, hipStreamNonBlocking));
    return b->sd;
}
void be_follow_apply(VrgBackend* b, const VrgCtx& c, const VrgLogRec* recs, const VrgLogSweep* hdr, int n, int count_last) {
    use_device(b);
    for (int i0 = 0; i0 < n; i0 += 8) {
        FollowGroup g; g.n = std::min(8, n - i0); g.count_last = (count_last && i0 + g.n == n) ? 1 : 0;
        for (int i = 0; i < g.n; i++) g.h[i] = hdr[i0 + i];
        k_follow_labels<<<1, GATE_THREADS, 0, label_stream(b)>>>(c, recs, g);
        k_follow_classes<<<1, GATE_THREADS, 0, b->sa>>>(c, recs, g);
    }
}
void be_follow_count(VrgBackend* b, const VrgCtx& c, VrgEvents* ev) {
    use_device(b);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ev && ev->enabled > 0 && (b->follow_counts++ % ev->enabled) == 0) {     // (every enabled-th count, as on one GPU: an event pair costs the stream a few us)
        if (b->ev_used == b->ev_pool.size())
            for (int k = 0; k < 32; k++) { EvPair n; HIP_CHECK(hipEventCreate(&n.a)); HIP_CHECK(hipEventCreate(&n.b)); n.trip = 0; n.kind = 0; n.ntrips = 1; b->ev_pool.push_back(n); }
        EvPair& p = b->ev_pool[b->ev_used++];
        p.trip = 0; p.kind = 0; p.ntrips = 1; e0 = p.a; e1 = p.b;
    }
    // the very pass a single GPU runs for this sweep (same kernel, same workgroups, same unit list: the same sums bit for bit); its closing
    // workgroup compares the totals with what the leader filed (check 4)
    if (b->dense_pipe && c.I && !c.lev16 && b->skip) {
        if (dense_nt(b, c)) hipExtLaunchKernelGGL((k_recount_pipe<3, true>), dim3(dense_blocks(b, c)), dim3(TPB), 0, b->sa, e0, e1, 0, c, 4);
        else hipExtLaunchKernelGGL((k_recount_pipe<3, false>), dim3(dense_blocks(b, c)), dim3(TPB), 0, b->sa, e0, e1, 0, c, 4);
    } else launch_recount(c, dense_blocks(b, c), 4, b->sa, b->skip != 0, dense_nt(b, c), e0, e1);
}
void be_follow_mark(VrgBackend* b, int slot) {
    use_device(b);
    for (int q = 0; q < 2; q++) {                      // (both streams read the staging buffer: the class bits' and the label bytes')
        hipEvent_t& e = b->mark[2 * slot + q];
        if (q == 1 && !b->sd) continue;
        if (!e) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(e, q ? b->sd : b->sa));
    }
}
void be_follow_wait(VrgBackend* b, int slot) { use_device(b); for (int q = 0; q < 2; q++) if (b->mark[2 * slot + q]) HIP_CHECK(hipEventSynchronize(b->mark[2 * slot + q])); }
static hipStream_t repl_stream(VrgBackend* b) {
    if (!b->sc) HIP_CHECK(hipStreamCreateWithFlags(&b->sc, hipStreamNonBlocking));
    return b->sc;
}
int be_repl_bcast(VrgBackend* b, void* dev_buf, size_t bytes, int root) {
    use_device(b);
    if (!b->comm) return -1;
    const ncclResult_t r = ncclBroadcast(dev_buf, dev_buf, bytes, ncclChar, root, b->comm, repl_stream(b));
    if (r != ncclSuccess) { if (!b->err[0]) std::snprintf(b->err, sizeof(b->err), "RCCL broadcast of the change log failed: %s", ncclGetErrorString(r)); return -1; }
    return 0;
}
int be_repl_allsum(VrgBackend* b, double* dev_buf, size_t n) {
    use_device(b);
    if (!b->comm) return -1;
    const ncclResult_t r = ncclAllReduce(dev_buf, dev_buf, n, ncclDouble, ncclSum, b->comm, repl_stream(b));
    if (r != ncclSuccess) { if (!b->err[0]) std::snprintf(b->err, sizeof(b->err), "RCCL all-reduce of the trace sums failed: %s", ncclGetErrorString(r)); return -1; }
    return 0;
}
void be_repl_wait(VrgBackend* b) { use_device(b); if (b->sc) HIP_CHECK(hipStreamSynchronize(b->sc)); }
void be_repl_copy(VrgBackend* b, void* dst, const void* src, size_t bytes) {
    use_device(b);
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, repl_stream(b)));
    HIP_CHECK(hipStreamSynchronize(b->sc));
}
void* be_host_alloc(VrgBackend* b, size_t bytes) { use_device(b); void* p = nullptr; if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; } return p; }
void be_host_free(VrgBackend* b, void* p) { use_device(b); if (p) (void)hipHostFree(p); }
int be_ipc_export(VrgBackend* b, void* dev_ptr, void* handle64) {
    use_device(b);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "ipc handle size");
    if (hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, dev_ptr) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return 0;
}
void* be_ipc_open(VrgBackend* b, const void* handle64) {
    use_device(b);
    hipIpcMemHandle_t h; std::memcpy(&h, handle64, sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void be_ipc_close(VrgBackend* b, void* mapped) { use_device(b); if (mapped && hipIpcCloseMemHandle(mapped) != hipSuccess) (void)hipGetLastError(); }

void be_events_collect(VrgBackend* b, VrgEvents* ev, long long n_valid) {
    if (!ev) return;
    use_device(b);
    if (b->ev_used) { HIP_CHECK(hipStreamSynchronize(b->sb)); HIP_CHECK(hipStreamSynchronize(b->sa)); }   // the dense stream may trail the band stream by one pass
    for (size_t i = 0; i < b->ev_used; i++) {
        if (b->ev_pool[i].trip < n_valid) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, b->ev_pool[i].a, b->ev_pool[i].b) == hipSuccess) {
                if (b->ev_pool[i].kind == 0) { ev->ms_total += ms; ev->launches++; } else { ev->chain_ms_total += ms; ev->chain_launches += b->ev_pool[i].ntrips; }
            }
            else (void)hipGetLastError();
        }
    }
    b->ev_used = 0; b->ev_trip = 0;
}

void be_recount_hist(VrgBackend* b, const VrgCtx& c, int32_t* rin, int32_t* rout) {
    use_device(b);
    k_recount_hist<<<voxel_blocks(c), TPB, 0, b->sa>>>(c, rin, rout);
    HIP_CHECK(hipStreamSynchronize(b->sa));
}

// what the dense pass of this handle is launched as: {non-temporal loads, storage mode (0 fp32, 1 u16 level index, 2 f64),
// workgroups, skip_excluded, k_recount_pipe instead of k_recount_bits}
static bool dense_is_pipe(VrgBackend* b, const VrgCtx& c) { return b->dense_pipe && c.I && !c.lev16 && b->skip; }
long long be_slow_flips(VrgBackend* b, const VrgCtx& c) { use_device(b); uint32_t v = 0; HIP_CHECK(hipMemcpyAsync(&v, c.counters + 49, 4, hipMemcpyDeviceToHost, b->sa)); HIP_CHECK(hipStreamSynchronize(b->sa)); return (long long)v; }
long long be_memo_trips(VrgBackend* b) { return b->memo_trips; }
void be_dense_info(VrgBackend* b, const VrgCtx& c, int64_t out[5]) {
    out[0] = dense_nt(b, c) ? 1 : 0; out[1] = c.lev16 ? (c.L <= TAB64_LEVELS ? 3 : 1) : (c.I ? 0 : 2); out[2] = dense_blocks(b, c); out[3] = b->skip ? 1 : 0;
    out[4] = dense_is_pipe(b, c) ? 1 : 0;
}
uint64_t be_dense_bytes(VrgBackend* b, const VrgCtx& c) {
    use_device(b);
    if (!b->skip) {            // every voxel of the slab's units is streamed
        const uint64_t plane = (uint64_t)c.PY * c.PX, lo = (2u + (uint64_t)c.z0) * plane, hi = (2u + (uint64_t)c.z1) * plane;
        const uint64_t bpv4 = c.lev16 ? 9 : (c.I ? 17 : 33);      // 4 x (intensity bytes + 0.25)
        return (hi - lo) * bpv4 / 4;
    }
    HIP_CHECK(hipStreamSynchronize(b->sb));
    unsigned long long* d = nullptr; unsigned long long v = 0;
    HIP_CHECK(hipMalloc(&d, 8));
    if (!d) return 0;
    HIP_CHECK(hipMemsetAsync(d, 0, 8, b->sa));
    k_dense_bytes<<<ITEM_BLOCKS, TPB, 0, b->sa>>>(c, d);
    HIP_CHECK(hipMemcpyAsync(&v, d, 8, hipMemcpyDeviceToHost, b->sa));
    HIP_CHECK(hipStreamSynchronize(b->sa));
    HIP_CHECK(hipFree(d));
    return v;
}

uint32_t be_collect_segmented(VrgBackend* b, const VrgCtx& c, uint64_t* stamps, uint32_t* idxs, uint32_t cap) {
    use_device(b);
    uint64_t* ds = nullptr; uint32_t* di = nullptr; uint32_t* dc = nullptr;
    HIP_CHECK(hipMalloc(&ds, (size_t)(cap + 1) * 8)); HIP_CHECK(hipMalloc(&di, (size_t)(cap + 1) * 4)); HIP_CHECK(hipMalloc(&dc, 4));
    if (!ds || !di || !dc) { if (ds) (void)hipFree(ds); if (di) (void)hipFree(di); if (dc) (void)hipFree(dc); return 0xffffffffu; }
    HIP_CHECK(hipMemsetAsync(dc, 0, 4, b->sa));
    k_collect_seg<<<voxel_blocks(c), TPB, 0, b->sa>>>(c, ds, di, cap, dc);
    uint32_t n = 0;
    HIP_CHECK(hipMemcpyAsync(&n, dc, 4, hipMemcpyDeviceToHost, b->sa));
    HIP_CHECK(hipStreamSynchronize(b->sa));
    uint32_t m = std::min(n, cap);
    HIP_CHECK(hipMemcpy(stamps, ds, (size_t)m * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(idxs, di, (size_t)m * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipFree(ds)); HIP_CHECK(hipFree(di)); HIP_CHECK(hipFree(dc));
    return n;
}
