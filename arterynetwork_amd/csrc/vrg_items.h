// vrg_items.h - per-item device functions of the VRG sweep (one band slot, one listed flip, one voxel).
//
// These restate variationalRegionGrowing.py's SEQUENTIAL update() (:124-261) as order-independent
// local rules so that every item can run in parallel.  Derivation (validated against the oracle and
// the reference goldens in tests/):
//
//  * flip list = concat(innerBnd, outerBnd)[mask] (:48,:88,:111): all flip-outs (label 1) precede all
//    flip-ins (label 2).  Only the ORDER of that list matters, and only between 26-neighbours; the "rank" of a
//    flip is its position in the list.
//  * phase A (flip-outs, :170-196), always applied:  P: 1->2.  A label-0 neighbour becomes 1 (:194).
//    A label-2 voxel next to a flip-out becomes 3 iff no segmented neighbour is left after all
//    flip-outs (:186-190) - for a flipped-out voxel itself only if a flip-out neighbour of larger rank
//    re-examines it; otherwise it stays a "ghost" 2.
//  * phase B (flip-ins, :198-230): a flip-in whose label is still 2 after phase A is applied; one that
//    dropped to 3 is skipped (both branches :170/:198 fail) unless an applied flip-in neighbour of
//    smaller rank re-promoted it first (:210-213) - a monotone fix-point along rank order (P bit).
//    Applied P: 2->1; its label-3 (or freshly included label-4) neighbours become 2; a label-1
//    neighbour becomes 0 iff it has no non-segmented neighbour left (:223-228) - the flipped-in voxel
//    itself only if an applied flip-in neighbour of larger rank re-examines it ("ghost" 1 otherwise).
//  * 4->3 inclusion: 1-ring of every listed flip (:166-168) and 2-ring of every applied flip
//    (:177-179, :206-208).
//  * LIST ORDER (:257-258).  After a sweep: survivors keep their order; appended in order:
//      inner: phase-A promotions keyed (rank of first flip-out neighbour, k), then applied flip-ins by rank;
//      outer: flip-outs still labelled 2 by rank, then phase-B promotions keyed (rank of first applied
//      flip-in neighbour, k); k = position of the promoted voxel in get_neighbours(promoter) (:263-282).
//    So a 64-bit key per band entry,
//          key = sweep that appended it << 40 | phase << 39 | rank of the appending flip << 5 | k
//    (phase 0 = the first group of each list above, 1 = the second; init: sweep 0, rest = init position), orders
//    each list exactly as the reference does, and an entry keeps its key for as long as it stays in its list.
//    Nothing is rebuilt per sweep: the flips (tens) are sorted by (list, key) to get their ranks, everything
//    else keeps its pool slot.
//  * densities (:232-255): incremental correction for entries that stayed in the band for the whole
//    sweep (added by the NEXT trip's k_band on its way through the pool), exact recomputation for entries that
//    (re-)entered it (newInnerBndList/newOuterBndList; slot flag PF_PEND).
#pragma once
#include <math.h>
#include <stddef.h>
#include "vrg_types.h"

// ------------------------------------------------------------------ backend shims
// The hand-offs between workgroups, kernels and streams are built from relaxed agent-scope atomics, write-through stores and an explicit drain
// (s_waitcnt vmcnt(0)) - no acquire / release fence, each of which costs ~1.7 us on this part.  The FENCED TWIN of the same
// sources (-DVRG_FENCES, tools/build_fenced.sh) gives every one of them its textbook ordering instead - polls and tickets acquire, announcing stores
// and tickets release, a release fence where the product only drains: slow, and by the memory model right.  tools/gpu.sh <tag> fenced runs
// the fuzz campaign on both builds: any difference in a result would be a hand-off the product gets wrong.
#if defined(VRG_FENCES)
#define VRG_MO_LOAD __ATOMIC_ACQUIRE
#define VRG_MO_STORE __ATOMIC_RELEASE
#define VRG_MO_TICKET __ATOMIC_ACQ_REL
#else
#define VRG_MO_LOAD __ATOMIC_RELAXED
#define VRG_MO_STORE __ATOMIC_RELAXED
#define VRG_MO_TICKET __ATOMIC_RELAXED
#endif
#if defined(__HIP_DEVICE_COMPILE__)
VRG_HD uint32_t vrg_atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
VRG_HD int32_t vrg_atomic_add(int32_t* p, int32_t v) { return atomicAdd(p, v); }
VRG_HD uint32_t vrg_atomic_or(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
VRG_HD void vrg_atomic_add64(int64_t* p, int64_t v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }
VRG_HD void vrg_atomic_xor(uint32_t* p, uint32_t v) { atomicXor(p, v); }
// loads that are served by L2: for bytes / counters that atomics of the SAME kernel may have changed (atomics execute
// in L2; a plain load could be answered from a line this CU cached before)
VRG_HD uint8_t vrg_load_coherent(const uint8_t* p) {
    return __hip_atomic_load(p, VRG_MO_LOAD, __HIP_MEMORY_SCOPE_AGENT);
}
VRG_HD uint32_t vrg_load_u32(const uint32_t* p) { return __hip_atomic_load(p, VRG_MO_LOAD, __HIP_MEMORY_SCOPE_AGENT); }
VRG_HD int32_t vrg_load_i32(const int32_t* p) { return __hip_atomic_load(p, VRG_MO_LOAD, __HIP_MEMORY_SCOPE_AGENT); }
VRG_HD int64_t vrg_load_i64(const int64_t* p) { return __hip_atomic_load(p, VRG_MO_LOAD, __HIP_MEMORY_SCOPE_AGENT); }
// the error word is written through: another workgroup of the SAME kernel may be the one that reads it (k_close)
VRG_HD void vrg_store_i32(int32_t* p, int32_t v) { __hip_atomic_store(p, v, VRG_MO_STORE, __HIP_MEMORY_SCOPE_AGENT); }
// words another STREAM's kernel polls or reads while this kernel is still running (the dense side's gate and expected
// sizes): written through to memory, and drained before the word that announces them
VRG_HD void vrg_store_i64(int64_t* p, int64_t v) { __hip_atomic_store(p, v, VRG_MO_STORE, __HIP_MEMORY_SCOPE_AGENT); }
#if defined(VRG_FENCES)
VRG_HD void vrg_drain() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
VRG_HD void vrg_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
VRG_HD void vrg_store_u32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, VRG_MO_STORE, __HIP_MEMORY_SCOPE_AGENT); }
VRG_HD void vrg_store_u64(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, VRG_MO_STORE, __HIP_MEMORY_SCOPE_AGENT); }
// ... and a word the HOST (page-locked memory) or another device polls: system scope
VRG_HD void vrg_store_u64_sys(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, VRG_MO_STORE, __HIP_MEMORY_SCOPE_SYSTEM); }
VRG_HD uint64_t vrg_load_u64(const uint64_t* p) { return __hip_atomic_load(p, VRG_MO_LOAD, __HIP_MEMORY_SCOPE_AGENT); }
// atomics on a workgroup's LDS arrays (fused sweep)
VRG_HD uint32_t vrg_lds_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
VRG_HD int32_t vrg_lds_add(int32_t* p, int32_t v) { return atomicAdd(p, v); }
VRG_HD uint32_t vrg_lds_or(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
VRG_HD uint32_t vrg_lds_cas(uint32_t* p, uint32_t expect, uint32_t want) { return atomicCAS(p, expect, want); }
#elif defined(VRG_HOSTMODEL)
// tests/hostmodel only (sequential test model of the kernels; never part of the product library)
VRG_HD uint32_t vrg_atomic_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
VRG_HD int32_t vrg_atomic_add(int32_t* p, int32_t v) { int32_t o = *p; *p = o + v; return o; }
VRG_HD uint32_t vrg_atomic_or(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
VRG_HD void vrg_atomic_add64(int64_t* p, int64_t v) { *p += v; }
VRG_HD void vrg_atomic_xor(uint32_t* p, uint32_t v) { *p ^= v; }
VRG_HD uint8_t vrg_load_coherent(const uint8_t* p) { return *(const volatile uint8_t*)p; }
VRG_HD uint32_t vrg_load_u32(const uint32_t* p) { return *p; }
VRG_HD int32_t vrg_load_i32(const int32_t* p) { return *p; }
VRG_HD int64_t vrg_load_i64(const int64_t* p) { return *p; }
VRG_HD void vrg_store_i32(int32_t* p, int32_t v) { *p = v; }
VRG_HD void vrg_store_i64(int64_t* p, int64_t v) { *p = v; }
VRG_HD void vrg_drain() {}
VRG_HD void vrg_store_u32(uint32_t* p, uint32_t v) { *p = v; }
VRG_HD void vrg_store_u64(uint64_t* p, uint64_t v) { *p = v; }
VRG_HD void vrg_store_u64_sys(uint64_t* p, uint64_t v) { *p = v; }
VRG_HD uint64_t vrg_load_u64(const uint64_t* p) { return *p; }
VRG_HD uint32_t vrg_lds_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
VRG_HD int32_t vrg_lds_add(int32_t* p, int32_t v) { int32_t o = *p; *p = o + v; return o; }
VRG_HD uint32_t vrg_lds_or(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
VRG_HD uint32_t vrg_lds_cas(uint32_t* p, uint32_t expect, uint32_t want) { uint32_t o = *p; if (o == expect) *p = want; return o; }
#else
// host pass of the product build (hipcc compiles __host__ __device__ functions for both sides): the product has no CPU
// path - the item functions are never called on the host there, and if one ever were it stops right here
VRG_HD uint32_t vrg_atomic_add(uint32_t*, uint32_t) { __builtin_trap(); }
VRG_HD int32_t vrg_atomic_add(int32_t*, int32_t) { __builtin_trap(); }
VRG_HD uint32_t vrg_atomic_or(uint32_t*, uint32_t) { __builtin_trap(); }
VRG_HD void vrg_atomic_add64(int64_t*, int64_t) { __builtin_trap(); }
VRG_HD void vrg_atomic_xor(uint32_t*, uint32_t) { __builtin_trap(); }
VRG_HD uint8_t vrg_load_coherent(const uint8_t*) { __builtin_trap(); }
VRG_HD uint32_t vrg_load_u32(const uint32_t*) { __builtin_trap(); }
VRG_HD int32_t vrg_load_i32(const int32_t*) { __builtin_trap(); }
VRG_HD int64_t vrg_load_i64(const int64_t*) { __builtin_trap(); }
VRG_HD void vrg_store_i32(int32_t*, int32_t) { __builtin_trap(); }
VRG_HD void vrg_store_i64(int64_t*, int64_t) { __builtin_trap(); }
VRG_HD void vrg_drain() { __builtin_trap(); }
VRG_HD void vrg_store_u32(uint32_t*, uint32_t) { __builtin_trap(); }
VRG_HD void vrg_store_u64(uint64_t*, uint64_t) { __builtin_trap(); }
VRG_HD void vrg_store_u64_sys(uint64_t*, uint64_t) { __builtin_trap(); }
VRG_HD uint64_t vrg_load_u64(const uint64_t*) { __builtin_trap(); }
VRG_HD uint32_t vrg_lds_add(uint32_t*, uint32_t) { __builtin_trap(); }
VRG_HD int32_t vrg_lds_add(int32_t*, int32_t) { __builtin_trap(); }
VRG_HD uint32_t vrg_lds_or(uint32_t*, uint32_t) { __builtin_trap(); }
VRG_HD uint32_t vrg_lds_cas(uint32_t*, uint32_t, uint32_t) { __builtin_trap(); }
#endif

// OR bits into one label byte without disturbing concurrent ORs into its neighbours
VRG_HD void vrg_or_byte(uint8_t* lab, uint32_t idx, uint8_t bits) {
    uint32_t* w = (uint32_t*)(lab + (idx & ~3u));
    vrg_atomic_or(w, (uint32_t)bits << (8 * (idx & 3u)));
}

// the whole state in one batch of loads
VRG_HD VrgState vrg_load_state(const VrgState* g) {
    VrgState s;
    uint32_t w[sizeof(VrgState) / 4];
    static_assert(sizeof(VrgState) % 4 == 0, "state is a whole number of words");
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (unsigned i = 0; i < sizeof(VrgState) / 4; i++) w[i] = reinterpret_cast<const uint32_t*>(g)[i];
    __builtin_memcpy(&s, w, sizeof(s));
    return s;
}

// ------------------------------------------------------------------ geometry
VRG_HD uint32_t vrg_idx(const VrgCtx& c, int x, int y, int z) {
    return ((uint32_t)(z + 2) * (uint32_t)c.PY + (uint32_t)(y + 2)) * (uint32_t)c.PX + (uint32_t)x;
}
VRG_HD void vrg_coords(const VrgCtx& c, uint32_t idx, int& x, int& y, int& z) {
    x = (int)(idx % (uint32_t)c.PX);
    uint32_t r = idx / (uint32_t)c.PX;
    y = (int)(r % (uint32_t)c.PY) - 2;
    z = (int)(r / (uint32_t)c.PY) - 2;
}
VRG_HD uint64_t vrg_lex(const VrgCtx& c, uint32_t idx) {   // np.where order (:44): x slowest, z fastest
    int x, y, z; vrg_coords(c, idx, x, y, z);
    return ((uint64_t)x * (uint64_t)c.ny + (uint64_t)y) * (uint64_t)c.nz + (uint64_t)z;
}
// k-th offset of get_neighbours (:266-269): lexicographic in (dx,dy,dz), k = 13 is the centre
VRG_HD int32_t vrg_off(const VrgCtx& c, int k) {
    int dx = k / 9 - 1, dy = (k / 3) % 3 - 1, dz = k % 3 - 1;
    return (dz * c.PY + dy) * c.PX + dx;
}
VRG_HD double vrg_kern(const VrgCtx& c, double d) { return c.A * exp(-0.5 * c.H * (d * d)); }   // :154

VRG_HD double vrg_voxel_value(const VrgCtx& c, uint32_t idx) { return c.I ? (double)c.I[idx] : c.I64[idx]; }
VRG_HD uint32_t vrg_level_of(const VrgCtx& c, double v) {   // index of v in the sorted distinct values
    if (c.lev_map) return (uint32_t)c.lev_map[(uint32_t)(v - c.lev_min)];       // (v is a voxel's value: an integer, a level)
    uint32_t lo = 0, hi = c.L - 1;
    while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (c.lev[m] < v) lo = m + 1; else hi = m; }
    return lo;
}
// level index of a voxel's intensity: stored (16-bit mode) or looked up in the sorted level table
VRG_HD uint32_t vrg_voxel_level(const VrgCtx& c, uint32_t idx) {
    return c.lev16 ? (uint32_t)c.lev16[idx] : c.lidx ? c.lidx[idx] : vrg_level_of(c, vrg_voxel_value(c, idx));
}

// The 3x3x3 neighbourhood of a voxel as four 27-bit masks (S, L, P, OOB bit of every neighbour).  The labels are
// fetched as nine 4-byte rows (x-1 .. x+2 of the nine (dy,dz) lines; x is the fastest axis, and the padding makes every
// row readable); neighbour n = 3*j + (dx+1) with j = 3*(dy+1) + (dz+1) sits at bit n, the centre at bit 13.  All the
// stencil logic below is bit arithmetic on these masks, and the few neighbours it has to visit (promoters, later
// flips) are found by iterating set bits - compact code: one workgroup's instruction stream is fetched cold on every
// launch, and the band kernels are bound by that and by dependent loads, not by arithmetic.
struct VrgNbr { uint32_t S, L, P, O; };
VRG_HD uint32_t vrg_load_row(const uint8_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    // served by L2 (not by a line this CU cached before atomics of the same kernel changed it); rows are not aligned
    return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p));
#else
    uint32_t w; __builtin_memcpy(&w, p, 4); return w;
#endif
}
VRG_HD uint64_t vrg_load_row8(const uint8_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(p));
#else
    uint64_t w; __builtin_memcpy(&w, p, 8); return w;
#endif
}
// bits 0, 8, 16 of t (one label bit of the three bytes of a row) gathered into bits 0..2
VRG_HD uint32_t vrg_gather3(uint32_t t) { return (((t & 0x010101u) * 0x00010204u) >> 16) & 7u; }
// everything the stencil of one voxel reads at addresses that follow from the voxel alone - fetched in one batch, so
// that it travels together (the band kernels are bound by DEPENDENT loads): the nine label rows, the voxel's stamp
// rank, its band slot, its intensity
struct VrgPre { uint32_t w[9]; uint32_t rank, vent; double val; uint32_t lev16; };
VRG_HD void vrg_preload(const VrgCtx& c, const uint8_t* lab, uint32_t idx, VrgPre& p) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 9; j++) p.w[j] = vrg_load_row(lab + ((int64_t)idx + ((j % 3 - 1) * c.PY + (j / 3 - 1)) * c.PX - 1));
    p.rank = (uint32_t)c.stamp[idx]; p.vent = c.vent[idx];
    p.lev16 = c.lev16 ? (uint32_t)c.lev16[idx] : c.lidx ? c.lidx[idx] : 0u;
    p.val = (c.lev16 || c.lidx) ? 0.0 : vrg_voxel_value(c, idx);
}
VRG_HD uint32_t vrg_pre_level(const VrgCtx& c, const VrgPre& p) { return (c.lev16 || c.lidx) ? p.lev16 : vrg_level_of(c, p.val); }
VRG_HD VrgNbr vrg_masks_of(const uint32_t* w) {
    VrgNbr m = {0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 9; j++) {
        m.S |= vrg_gather3(w[j]) << (3 * j);      m.L |= vrg_gather3(w[j] >> 3) << (3 * j);
        m.P |= vrg_gather3(w[j] >> 4) << (3 * j); m.O |= vrg_gather3(w[j] >> 5) << (3 * j);
    }
    return m;
}
VRG_HD VrgNbr vrg_load_masks(const VrgCtx& c, const uint8_t* lab, uint32_t idx) {
    uint32_t w[9];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 9; j++) w[j] = vrg_load_row(lab + ((int64_t)idx + ((j % 3 - 1) * c.PY + (j / 3 - 1)) * c.PX - 1));
    return vrg_masks_of(w);
}
// neighbour n of the masks: its voxel offset, and its position k in get_neighbours' order (:266-269: dx slowest)
VRG_HD int32_t vrg_noff(const VrgCtx& c, uint32_t n) {
    const int j = (int)(n / 3u);
    return ((j % 3 - 1) * c.PY + (j / 3 - 1)) * c.PX + ((int)(n % 3u) - 1);
}
VRG_HD uint32_t vrg_nk(uint32_t n) { return (n % 3u) * 9u + n / 3u; }
VRG_HD uint32_t vrg_ctz(uint32_t v) { return (uint32_t)__builtin_ctz(v); }

// the 27 label bytes around idx one by one (skip-rule fix-point only)
VRG_HD void vrg_load_nbrs(const VrgCtx& c, const uint8_t* lab, uint32_t idx, uint8_t nb[27]) {
    for (int k = 0; k < 27; k++) nb[k] = vrg_load_coherent(lab + ((int64_t)idx + vrg_off(c, k)));
}

VRG_HD uint8_t vrg_enc(uint8_t ext) {     // reference label -> byte
    return ext == 0 ? VB_S : ext == 1 ? (VB_S | VB_B) : ext == 2 ? VB_B : ext == 4 ? VB_X : 0;
}
VRG_HD uint8_t vrg_dec(uint8_t b) {       // byte -> reference label
    if (b & VB_S) return (b & VB_B) ? 1 : 0;
    if (b & VB_B) return 2;
    return (b & VB_X) ? 4 : 3;
}

// The state out, field by field: everything in front of the padding and the four live words - never the 196 bytes of padding (a whole-struct
// assignment copies them too: 96 words held across a kernel, which the compiler then keeps in scratch - k_band, 2 us per launch).
VRG_HD void vrg_state_store(VrgState* dst, const VrgState& w) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&w); uint32_t* d = reinterpret_cast<uint32_t*>(dst);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (unsigned i = 0; i < offsetof(VrgState, pad_live) / 4; i++) d[i] = src[i];
    dst->nf = w.nf; dst->ties = w.ties; dst->near_ties = w.near_ties; dst->error = w.error;
}
// ------------------------------------------------------------------ list order
VRG_HD uint64_t vrg_key(const VrgState& s, uint32_t phase, uint32_t rank, uint32_t k) {
    return ((uint64_t)(uint32_t)(s.iter + 1) << 40) | ((uint64_t)phase << 39) | ((uint64_t)rank << 5) | (uint64_t)k;
}
// sort key of a listed flip: inner list before outer list (:48), each in list order
VRG_HD uint64_t vrg_flip_key(bool inner, uint64_t key) { return (inner ? 0ull : (1ull << 63)) | key; }

// ------------------------------------------------------------------ k_band: correction of the sweep before + decide (:79-88)
// density correction of one intensity value (:236-247)
VRG_HD void vrg_corrections(const VrgCtx& c, uint32_t nnz, const double* nz_val, const uint32_t* nz_cin, const uint32_t* nz_cout,
                            const uint32_t* nz_cconv, double v, double& ic, double& oc, double& ac) {
    double a = 0, b = 0, d = 0;
    for (uint32_t i = 0; i < nnz; i++) {
        double k = vrg_kern(c, nz_val[i] - v);
        a += (double)nz_cin[i] * k; b += (double)nz_cout[i] * k; d += (double)nz_cconv[i] * k;
    }
    ic = a; oc = b; ac = d;
}
VRG_HD void vrg_add_correction(double ic, double oc, double ac, double& ip, double& op) {
    ip += ic; ip -= oc;             // :243-244
    op -= ic; op += oc; op += ac;   // :245-247
}
// A flip is listed by an unordered append of its record (slot, sort key, voxel, level); k_order orders the list.
// n_in / n_out: the region sizes (VrgCtx::inc), read once per thread by the caller
// err: bound on the absolute error of ip and of op (binned exact densities, VrgCtx::p_err; 0 without bins)
// sink: a workgroup's own list of flip records (LDS), filed with ONE bump of the flip counter when the workgroup is done (k_band, pools of
// hundreds of thousands of entries: every flip bumping the same word - 12 900 of them, one after the other in L2 - was most of the kernel)
constexpr uint32_t VRG_SINK_CAP = 512;
struct VrgFlipSink { uint64_t key[VRG_SINK_CAP]; uint32_t slot[VRG_SINK_CAP], idx[VRG_SINK_CAP], lev[VRG_SINK_CAP]; uint32_t n, base; };
VRG_HD void vrg_decide_core(const VrgCtx& c, const VrgState& s, int64_t n_in, int64_t n_out, uint32_t slot, bool inner, double ip, double op,
                            uint64_t key, uint32_t idx, uint32_t lev, double err = 0.0, VrgFlipSink* sink = nullptr) {
    double inN = ip / (double)n_in;                       // :81
    double outN = op / (double)n_out;                     // :82
    bool ge = inN >= outN;
    {   // a decision at rounding level is the reference's summation order's to make, not ours: count it (VRG_TIE_REL) - and so is
        // one the error of a binned evaluation could turn
        const double d = fabs(inN - outN), m = fmax(fabs(inN), fabs(outN));
        if (n_in == 0 || n_out == 0 || !(d > VRG_TIE_REL * m + (err / (double)n_in + err / (double)n_out))) vrg_atomic_add(&c.stg->ties, 1u);
        else if (!(d > VRG_TIE_NEAR_REL * m)) vrg_atomic_add(&c.stg->near_ties, 1u);
    }
    if (inner == ge) return;                              // :87 xor(segmentedMap, inner >= outer)
    const bool count_only = s.time_up || n_in >= s.maxSegmentSize;      // :97 / :101 fire before update()
    if (sink && !count_only) {
        const uint32_t l = vrg_lds_add(&sink->n, 1u);
        if (l < VRG_SINK_CAP) { sink->slot[l] = slot; sink->key[l] = vrg_flip_key(inner, key); sink->idx[l] = idx; sink->lev[l] = lev; return; }
    }                                                     // (a full sink: the record goes the direct way)
    uint32_t q = vrg_atomic_add(&c.stg->nf, 1u);
    if (count_only) return;
    if (q >= c.fcap) { vrg_store_i32(&c.stg->error, 2); return; }
    c.flist[q] = slot; c.f_key[q] = vrg_flip_key(inner, key); c.fr_idx[q] = idx; c.fr_lev[q] = lev;
}
// one pool slot whose fields the caller has fetched; nz_* = this trip's view of the touched-level list (LDS copy on the
// device), tab = the per-level memo (the device may pass an LDS copy of its first tab_n levels; the rest comes from c.tabC)
VRG_HD void vrg_item_band_fields(const VrgCtx& c, const VrgState& s, uint32_t slot, uint8_t fl, double ip, double op, uint32_t lev, uint32_t idx,
                                 uint64_t key, int64_t n_in, int64_t n_out, const double* nz_val, const uint32_t* nz_cin,
                                 const uint32_t* nz_cout, const uint32_t* nz_cconv, const double* tab, uint32_t tab_n, double err = 0.0, VrgFlipSink* sink = nullptr) {
    if (!(fl & PF_ALIVE)) return;
    // an entry that (re-)entered the band in the sweep before takes no correction; it is decided by whoever computes
    // its exact densities (the other half of this launch, which reads only the list bit of the flag)
    if (fl & PF_PEND) { c.p_flag[slot] = (uint8_t)(fl & ~PF_PEND); return; }
    if (s.corr) {
        double ic, oc, ac;
        if (s.use_tab) {
            if (lev < tab_n) { const double* t3 = tab + 3 * (size_t)lev; ic = t3[0]; oc = t3[1]; ac = t3[2]; }
            else { ic = c.tabC[3 * (size_t)lev]; oc = c.tabC[3 * (size_t)lev + 1]; ac = c.tabC[3 * (size_t)lev + 2]; }
        }
        else vrg_corrections(c, s.nnz, nz_val, nz_cin, nz_cout, nz_cconv, c.lev[lev], ic, oc, ac);
        vrg_add_correction(ic, oc, ac, ip, op);
        c.p_ip[slot] = ip; c.p_op[slot] = op;
    }
    if (s.iter < s.iterMax) vrg_decide_core(c, s, n_in, n_out, slot, fl & PF_INNER, ip, op, key, idx, lev, err, sink);   // while iterNum <= iterMax (:58)
}
VRG_HD void vrg_item_band(const VrgCtx& c, const VrgState& s, uint32_t slot, const double* nz_val, const uint32_t* nz_cin,
                          const uint32_t* nz_cout, const uint32_t* nz_cconv, const double* tab = nullptr, uint32_t tab_n = 0, int64_t sizes_in = -1, int64_t sizes_out = -1,
                          VrgFlipSink* sink = nullptr) {
    // the slot's fields in one batch (a dead slot's are read for nothing): the kernel is bound by dependent loads
    const uint8_t fl = c.p_flag[slot];
    const double ip = c.p_ip[slot], op = c.p_op[slot];
    const uint32_t lev = c.p_lev[slot], idx = c.p_idx[slot];
    const uint64_t key = c.p_key[slot];
    // (the region sizes: the caller's - k_band has them in registers, possibly derived from an open-ended sweep - or the ones that go with the
    // state the kernel READS; never c.inc there: k_band files into the other buffer while it decides)
    const int64_t n_in = sizes_in >= 0 ? sizes_in : c.inc_in[VC_NIN], n_out = sizes_in >= 0 ? sizes_out : c.inc_in[VC_NOUT];
    const double err = (double)c.p_err[slot];
    vrg_item_band_fields(c, s, slot, fl, ip, op, lev, idx, key, n_in, n_out, nz_val, nz_cin, nz_cout, nz_cconv, tab, tab_n, err, sink);
}
// ------------------------------------------------------------------ binned exact densities (large level tables)
// The exact densities of an entry that (re-)enters the band (:252-255) are sums over the whole inner / outer regions:
//     S(v) = sum over voxels q of the class   A * exp(-0.5 * H * (x_q - v)^2).
// The reference evaluates them voxel by voxel; regrouped by distinct value that is (new entries x L) kernel evaluations - fine for
// quantised data, 4 ms per sweep for a continuous-valued 512x512x170 volume (L = 3e7).  Here the intensity axis is cut into uniform
// bins of half width h; for a voxel with x = c + delta in the bin centred at c, and d = c - v,
//     exp(-0.5*H*(d + delta)^2) = exp(-0.5*H*d^2) * exp(-0.5*H*delta^2) * exp(-H*d*delta),
// exactly.  The first factor is per bin; the second (w) goes into the bin's moments; only the third is expanded:
//     exp(t) = sum_{k <= K} t^k / k! + R,   t = -H*d*delta,   |R| <= |t|^(K+1) / (K+1)! * max(1, e^t).
// With |t| <= theta the truncation error relative to the TRUE term is <= theta^(K+1)/(K+1)! * e^(2*theta) (the true factor is
// >= e^-theta, the remainder <= theta^(K+1)/(K+1)! * e^theta).  Every term of S is positive, so that is also the bound on
// the relative error of the whole sum.  K = 8, theta = 0.5:  0.5^9 / 9! * e = 1.46e-8 - two orders inside 1e-6, three inside north_star's
// 1e-5.  |t| <= theta has to hold for every bin that contributes at all: a bin with 0.5*H*d^2 > T = 745.2 contributes exactly 0.0 in
// double arithmetic (exp underflows, in the reference too), so d ranges up to D = sqrt(2T/H) and h = theta / (H * D) = theta /
// sqrt(2*T*H) (H = 2.25: h = 0.00863).  An evaluation visits at most 2*D / (2*h) + 2 = 2T/theta + 2 = 2983 bins whatever H and the data's scale are.
// The moments sum_q w_q * (delta_q / h)^k are 64-bit fixed-point integers (2^-30 per unit; |w * (delta/h)^k| <= 1): adding and
// removing voxels commutes exactly - no drift, bit-reproducible - at a rounding of 2^-31 per voxel and moment, i.e. <= 5e-10 of the
// class's count in the bin, times (theta^k / k!) in the sum: below the truncation bound.  Total: <= 2e-8 relative (VRG_TIE_NEAR_REL is 2e-5).
VRG_HD uint32_t vrg_bin_of(const VrgCtx& c, double v) {
    const double q = (v - c.bin_lo) / (2.0 * c.bin_h);
    const uint32_t b = q <= 0.0 ? 0u : (uint32_t)q;
    return b < c.nb ? b : c.nb - 1u;
}
VRG_HD double vrg_bin_centre(const VrgCtx& c, uint32_t b) { return c.bin_lo + (2.0 * (double)b + 1.0) * c.bin_h; }
// a voxel of value v joins (n > 0) or leaves (n < 0) the inner / outer class n_in / n_out times: its terms into the bin's moments
VRG_HD void vrg_bin_add(const VrgCtx& c, double v, int64_t n_in, int64_t n_out) {
    const uint32_t b = vrg_bin_of(c, v);
    const double delta = v - vrg_bin_centre(c, b), u = delta / c.bin_h;
    double term = exp(-0.5 * c.H * (delta * delta));
    for (int k = 0; k <= VRG_BIN_K; k++) {
        const int64_t q = (int64_t)llrint(term * VRG_BIN_SCALE);
        if (n_in) vrg_atomic_add64(&c.bm_in[(size_t)b * (VRG_BIN_K + 1) + k], q * n_in);
        if (n_out) vrg_atomic_add64(&c.bm_out[(size_t)b * (VRG_BIN_K + 1) + k], q * n_out);
        term *= u;
    }
}
// first / last bin that can contribute to an entry of value v (those with 0.5*H*d^2 <= T, a bin of margin either side)
VRG_HD void vrg_bin_range(const VrgCtx& c, double v, uint32_t& b0, uint32_t& b1) {
    const double D = sqrt(2.0 * VRG_BIN_T / c.H) + 2.0 * c.bin_h;
    b0 = vrg_bin_of(c, v - D); b1 = vrg_bin_of(c, v + D);
}
// contribution of bin b to the two exact densities of an entry of value v (0 for an empty or too distant bin)
VRG_HD void vrg_bin_terms(const VrgCtx& c, double v, uint32_t b, double& ti, double& to) {
    ti = 0; to = 0;
    const int64_t* mi = c.bm_in + (size_t)b * (VRG_BIN_K + 1);
    const int64_t* mo = c.bm_out + (size_t)b * (VRG_BIN_K + 1);
    if (!(mi[0] | mo[0])) return;                          // (moment 0 is a sum of positive terms: zero means nobody is there)
    const double d = vrg_bin_centre(c, b) - v, g = vrg_kern(c, d);
    if (g == 0.0) return;
    const double t = -c.H * d * c.bin_h;                   // exp(-H*d*delta) = sum_k (t * delta/h)^k / k!
    double pi = 0, po = 0, f = 1.0;                        // f = t^k / k!
    for (int k = 0; k <= VRG_BIN_K; k++) {
        pi += f * (double)mi[k]; po += f * (double)mo[k];
        f *= t / (double)(k + 1);
    }
    ti = g * (pi / VRG_BIN_SCALE); to = g * (po / VRG_BIN_SCALE);
}
// a voxel's class histogram entry changes: the per-level histograms and - with bins - the moments
VRG_HD void vrg_hist_change(const VrgCtx& c, uint32_t lev, int din, int dout) {
    if (din) vrg_atomic_add(&c.hin[lev], din);
    if (dout) vrg_atomic_add(&c.hout[lev], dout);
    if (c.nb) vrg_bin_add(c, c.lev[lev], din, dout);
}
// what an exact evaluation leaves in VrgCtx::p_err: nothing without bins (sums over the levels: rounding only), the proved bound with them
VRG_HD float vrg_exact_err(const VrgCtx& c, double si, double so) {
    if (!c.nb) return 0.0f;
    const float e = (float)(VRG_BIN_REL_ERR * fmax(si, so));
    return e * 1.000001f + 1e-37f;                          // (the float rounding goes up)
}
// exact densities over the whole inner / outer regions (:152-155, :252-255), regrouped by level - or, with bins, by bin
VRG_HD void vrg_exact_serial(const VrgCtx& c, const VrgState& s, uint32_t slot, bool then_decide) {
    double v = c.lev[c.p_lev[slot]];
    double si = 0, so = 0;
    if (c.nb) {
        uint32_t b0, b1; vrg_bin_range(c, v, b0, b1);
        for (uint32_t b = b0; b <= b1; b++) { double ti, to; vrg_bin_terms(c, v, b, ti, to); si += ti; so += to; }
    } else
    for (uint32_t l = 0; l < c.L; l++) {
        int32_t a = c.hin[l], b = c.hout[l];
        if (!(a | b)) continue;
        double k = vrg_kern(c, c.lev[l] - v);
        si += (double)a * k; so += (double)b * k;
    }
    const float err = vrg_exact_err(c, si, so);
    c.p_ip[slot] = si; c.p_op[slot] = so; c.p_err[slot] = err;
    if (then_decide && s.iter < s.iterMax)
        vrg_decide_core(c, s, c.inc[VC_NIN], c.inc[VC_NOUT], slot, c.p_flag[slot] & PF_INNER, si, so, c.p_key[slot], c.p_idx[slot], c.p_lev[slot], (double)err);
}

// ------------------------------------------------------------------ the sweep (update(), :156-259)
// the stop tests in the reference's order, once every entry has decided (the host raises time_up)
VRG_HD int32_t vrg_stop_test_v(const VrgState& s, int64_t n_in) {
    if (s.iter >= s.iterMax) return VRG_STOP_ITERMAX;                    // :58
    if (s.nf == 0) return VRG_STOP_CONVERGED;                            // :91
    if (s.time_up) return VRG_STOP_TIME;                                 // :97
    if (n_in >= s.maxSegmentSize) return VRG_STOP_SIZE;                  // :101
    return 0;
}
VRG_HD int32_t vrg_stop_test(const VrgCtx& c) { return vrg_stop_test_v(*c.st, c.inc[VC_NIN]); }
// can this trip's update() run with the arrays as they are?  (checked before anything is modified)
VRG_HD int32_t vrg_capacity_test(const VrgCtx& c, uint64_t nf) {
    const VrgState& s = *c.st;
    if (nf * 125u > (uint64_t)c.mcap) return VBAIL_MARKS;                // 5x5x5 marks per flip, as many class changes at most
    if ((uint64_t)s.np + nf * 27u > (uint64_t)c.bcap) return VBAIL_POOL; // a flip promotes at most its 26 neighbours
    if (c.log_rec && (uint64_t)(s.log_pos - c.log_pos0) + nf * 125u > (uint64_t)c.log_cap) return VBAIL_LOG;   // the sweep's records (one per place of its marked list)
    return 0;
}
// the trip stops (or is handed back): nothing pending for the next k_band
VRG_HD void vrg_close_without_update(const VrgCtx& c) { c.stg->corr = 0; c.stg->nfx = 0; vrg_store_i64(&c.gate[VG_STOP], 1); }
// update() begins: k_band has consumed the touched-level list of the sweep before
// (memoise the corrections per level when there are at least VRG_TAB_RATIO band entries per level: the memo is built by 256
// workgroups on the sweep's critical path, the entries sum their own corrections chip-wide in the shadow of the decisions -
// 512x512x170, 6111 levels, 20 000 entries: 0.0464 ms/step with the memo, measured against the entry-by-entry rule)
#define VRG_TAB_RATIO 4u
VRG_HD bool vrg_tab_pays(uint32_t L, uint32_t band) { return (uint64_t)L * VRG_TAB_RATIO <= (uint64_t)band; }
VRG_HD void vrg_open_update(const VrgCtx& c) { c.stg->nnz = 0; c.stg->tab_ok = vrg_tab_pays(c.L, c.st->ni + c.st->no); }

// flip r of the ordered list: L bit (+P for flip-outs, which are always applied), stamp = (sweep, rank)
VRG_HD void vrg_item_list_rec(const VrgCtx& c, uint32_t r, uint32_t slot, uint32_t idx, uint32_t lev, bool inner) {
    c.f_slot[r] = slot; c.f_idx[r] = idx; c.f_lev[r] = lev; c.f_res[r] = 0;
    vrg_or_byte(c.lab[0], idx, (uint8_t)(VB_L | (inner ? VB_P : 0)));
    c.stamp[idx] = ((uint64_t)(uint32_t)(c.st->iter + 1) << 32) | r;
}
// ... when only the slot order is known (f_slot[r]; host-driven sort)
VRG_HD void vrg_item_list(const VrgCtx& c, uint32_t r) {
    const uint32_t slot = c.f_slot[r];
    vrg_item_list_rec(c, r, slot, c.p_idx[slot], c.p_lev[slot], (c.p_flag[slot] & PF_INNER) != 0);
}

// flip-ins: label after phase A (:183-190) decides whether the flip is applied at once
VRG_HD void vrg_item_prepass(const VrgCtx& c, uint32_t r) {
    const uint32_t idx = c.f_idx[r];
    const VrgNbr m = vrg_load_masks(c, c.lab[0], idx);
    if (m.S & (1u << 13)) return;                          // a flip-out
    const uint32_t ex = ~m.O & 0x7ffdfffu;                 // existing neighbours (centre excluded)
    const bool nFO = (m.S & m.L & ex) != 0, nSegA = (m.S & ~m.L & ex) != 0;
    if (nFO && !nSegA) c.pend[vrg_atomic_add(&c.stg->npend, 1u)] = r;   // dropped to 3: skipped unless re-promoted
    else vrg_or_byte(c.lab[0], idx, VB_P);
}

// one relaxation of the skip rule: applied if an applied flip-in neighbour of smaller rank exists
// returns 0: still skipped, 1: found applied (by whoever), 2: applied by this call
VRG_HD int vrg_item_fix(const VrgCtx& c, uint32_t j) {
    const uint32_t r = c.pend[j], idx = c.f_idx[r];
    uint8_t* lab = c.lab[0];
    if (vrg_load_coherent(lab + idx) & VB_P) return 1;
    for (int k = 0; k < 27; k++) {
        if (k == 13) continue;
        uint32_t m = (uint32_t)((int64_t)idx + vrg_off(c, k));
        uint8_t mb = vrg_load_coherent(lab + m);
        if (!(mb & VB_S) && (mb & VB_L) && (mb & VB_P) && (uint32_t)c.stamp[m] < r) {
            vrg_or_byte(lab, idx, VB_P);
            return 2;
        }
    }
    return 0;
}

// scatter the "needs the stencil" mark: 1-ring of every listed flip (incl. itself) and the excluded
// voxels of its 2-ring (needed for applied flips; harmless for a skipped flip-in, whose 2-ring voxels
// then simply keep their label); the first marker of a voxel appends it to the marked list.
// Everything else keeps its label this sweep.  Item = (listed flip r, position p of the 5x5x5 cube).
#define VRG_NONE 0xffffffffu
// position p of the 5x5x5 cube around flip voxel idx: does that voxel need the stencil? (its byte mb is passed in)
VRG_HD bool vrg_mark_wanted(uint32_t p, uint8_t mb) {
    int dx = (int)(p % 5) - 2, dy = (int)((p / 5) % 5) - 2, dz = (int)(p / 25) - 2;
    bool ring1 = dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
    if (mb & (VB_OOB | VB_M)) return false;
    return ring1 || (mb & VB_X);
}
VRG_HD int64_t vrg_mark_pos(const VrgCtx& c, uint32_t idx, uint32_t p) {
    int dx = (int)(p % 5) - 2, dy = (int)((p / 5) % 5) - 2, dz = (int)(p / 25) - 2;
    return (int64_t)idx + (dz * c.PY + dy) * c.PX + dx;    // may be -1,-2 (guard bytes) at voxel (0,0,0)
}
// set the mark; true: this caller is the first marker (it appends the voxel to the marked list)
VRG_HD bool vrg_mark_set(const VrgCtx& c, int64_t m) {
    uint32_t sh = 8 * ((uint32_t)m & 3u);
    uint32_t old = vrg_atomic_or((uint32_t*)(c.lab[0] + ((uint32_t)m & ~3u)), (uint32_t)VB_M << sh);
    return !((old >> sh) & VB_M);
}
VRG_HD void vrg_item_scatter_marks(const VrgCtx& c, uint32_t r, uint32_t p) {
    if (p >= 125) return;
    const int64_t m = vrg_mark_pos(c, c.f_idx[r], p);
    if (!vrg_mark_wanted(p, vrg_load_coherent(c.lab[0] + m))) return;
    if (vrg_mark_set(c, m)) {
        uint32_t q = vrg_atomic_add(&c.stg->nmk, 1u);
        if (q < c.mcap) c.mk_idx[q] = (uint32_t)m; else vrg_store_i32(&c.stg->error, 4);
    }
}

// ---- what a relabelled voxel does to the band pool and the density bookkeeping
// this sweep's innerAdded / outerAdded / addedPoints (:232-235), by level; the first toucher lists the level
VRG_HD void vrg_note_level(const VrgCtx& c, uint32_t* cnt, uint32_t lev) {
    vrg_atomic_add(&cnt[lev], 1u);
    if (c.lvl_scan == 1) return;                      // (whoever closes the sweep scans the counters: no returning atomics here)
    if (vrg_atomic_or(&c.ltouch[lev], 1u) == 0u) {
        // (the entry is written THROUGH: in a fused sweep the workgroup that closes the sweep - another CU, another XCD - reads it)
        uint32_t q = vrg_atomic_add(c.lvl_scan == 2 ? &c.stg->nnz_new : &c.stg->nnz, 1u);
        if (q < c.zcap) vrg_store_u64(&c.nz_key[q], (uint64_t)lev); else vrg_store_i32(&c.stg->error, 8);
    }
}
// What the relabel of one voxel means for the band pool: at most one event per voxel.  The stencil only DESCRIBES it;
// committing it - slot allocation, lists of dead / pending slots, list lengths - is separate, so that a workgroup can
// commit all its events with one reservation per list (every event bumping the same few words would serialise in L2).
enum : uint8_t { VE_NONE = 0, VE_NEW = 1, VE_DIE = 2, VE_MOVE = 3 };
struct VrgEvent {
    uint64_t key;                // list-order key of a new / re-appended entry
    uint32_t slot, lev;          // slot concerned (DIE, MOVE); level of a new entry
    uint8_t kind, from_inner, to_inner, pend;   // pend: (re-)entered the band - exact densities due (:252-255)
};
VRG_HD void vrg_ev_new(VrgEvent& e, uint32_t lev, bool inner, uint64_t key) { e.kind = VE_NEW; e.lev = lev; e.to_inner = inner; e.key = key; e.pend = 1; }
VRG_HD void vrg_ev_die(VrgEvent& e, uint32_t slot, bool inner) { e.kind = VE_DIE; e.slot = slot; e.from_inner = inner; }
VRG_HD void vrg_ev_move(VrgEvent& e, uint32_t slot, bool from_inner, bool to_inner, uint64_t key, bool pend) {
    e.kind = VE_MOVE; e.slot = slot; e.from_inner = from_inner; e.to_inner = to_inner; e.key = key; e.pend = pend;
}
// list length changes of an event
VRG_HD int vrg_ev_dni(const VrgEvent& e) { return e.kind == VE_NEW ? (e.to_inner ? 1 : 0) : e.kind == VE_DIE ? (e.from_inner ? -1 : 0) : e.kind == VE_MOVE ? (int)(e.to_inner != 0) - (int)(e.from_inner != 0) : 0; }
VRG_HD int vrg_ev_dno(const VrgEvent& e) { return e.kind == VE_NEW ? (e.to_inner ? 0 : 1) : e.kind == VE_DIE ? (e.from_inner ? 0 : -1) : e.kind == VE_MOVE ? (int)(e.from_inner != 0) - (int)(e.to_inner != 0) : 0; }
// the writes of a commit, once the positions are known: q = this event's number among the sweep's allocations,
// qd / qf = its place in the dead / pending lists
VRG_HD void vrg_ev_write(const VrgCtx& c, uint32_t idx, const VrgEvent& e, uint32_t q, uint32_t qd, uint32_t qf) {
    const VrgState& s = *c.st;
    uint32_t slot = e.slot;
    if (e.kind == VE_NEW) {                               // a voxel enters the band (newInnerBndList / newOuterBndList, :196, :213)
        slot = q < s.nfree ? c.freel[s.nfree - 1u - q] : s.np + (q - s.nfree);
        if (slot >= c.bcap) { vrg_store_i32(&c.stg->error, 1); return; }
        c.p_idx[slot] = idx; c.p_lev[slot] = e.lev; c.p_ip[slot] = 0; c.p_op[slot] = 0; c.p_err[slot] = 0; c.p_key[slot] = e.key;
        c.p_flag[slot] = (uint8_t)(PF_ALIVE | PF_PEND | (e.to_inner ? PF_INNER : 0));
        c.vent[idx] = slot;
    } else if (e.kind == VE_DIE) {                        // ... leaves it (list.remove, :171-172, :188, :199, :226)
        c.p_flag[slot] = 0;
        c.dead[qd] = slot;
    } else if (e.kind == VE_MOVE) {                       // ... is re-appended to a list (a carried flip; or left and re-entered: pend)
        c.p_key[slot] = e.key;
        c.p_flag[slot] = (uint8_t)(PF_ALIVE | (e.to_inner ? PF_INNER : 0) | (e.pend ? PF_PEND : 0));
    }
    if (e.kind != VE_DIE && e.pend) c.fresh[qf] = slot;
}
// one event committed on its own (host-driven kernels, test model)
VRG_HD void vrg_commit_event(const VrgCtx& c, uint32_t idx, const VrgEvent& e) {
    if (e.kind == VE_NONE) return;
    uint32_t q = 0, qd = 0, qf = 0;
    if (e.kind == VE_NEW) q = vrg_atomic_add(&c.stg->nalloc, 1u);
    if (e.kind == VE_DIE) qd = vrg_atomic_add(&c.stg->ndead, 1u);
    if (e.kind != VE_DIE && e.pend) qf = vrg_atomic_add(&c.stg->nfresh, 1u);
    const int di = vrg_ev_dni(e), dq = vrg_ev_dno(e);
    if (di) vrg_atomic_add(&c.stg->d_ni, di);
    if (dq) vrg_atomic_add(&c.stg->d_no, dq);
    vrg_ev_write(c, idx, e, q, qd, qf);
}

// ------------------------------------------------------------------ the relabel stencil for one voxel
// who promotes this voxel - `cand` = its applied flip-in neighbours (phase B, 3 -> 2, :210-213) or its flip-out
// neighbours (phase A, 0 -> 1, :194-196): the one of smallest rank; (rank, k) is the list key of the promoted voxel,
// k = its position in get_neighbours(promoter) = 26 - the promoter's position seen from here
// (The candidates' ranks are fetched four at a time: a loop that loads one stamp per turn - or leaves at the first hit - is a
// chain of dependent round trips, one per flip neighbour; the usual one to three neighbours now cost one.)
constexpr int VRG_RANK_BATCH = 4;
VRG_HD void vrg_rank_batch(const VrgCtx& c, uint32_t& cand, uint32_t idx, uint32_t n[VRG_RANK_BATCH], uint32_t r[VRG_RANK_BATCH]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < VRG_RANK_BATCH; k++) { n[k] = cand ? vrg_ctz(cand) : 32u; if (cand) cand &= cand - 1u; }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < VRG_RANK_BATCH; k++) r[k] = n[k] < 27u ? (uint32_t)c.stamp[(uint32_t)((int64_t)idx + vrg_noff(c, n[k]))] : 0u;
}
// an excluded voxel: is an applied flip within its 2-ring (:177-179, :206-208)?  25 rows of 5 bytes
VRG_HD bool vrg_ring2_applied(const VrgCtx& c, const uint8_t* lab, uint32_t idx) {
    uint64_t any = 0;                                      // (no early exit: the 25 rows are then requested together, not one after the other)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 25; j++) {
        const uint64_t w = vrg_load_row8(lab + ((int64_t)idx + ((j / 5 - 2) * c.PY + (j % 5 - 2)) * c.PX - 2));
        any |= ((w >> 4) & ~(w >> 5)) & 0x0101010101ull;                   // P and not OOB, bytes 0..4
    }
    return any != 0;
}

// The ranks of a voxel's flip-out neighbours (FO) and applied flip-in neighbours (AP), all fetched in one go BEFORE the case
// analysis below: smallest rank of each set with the k of its owner (the promoter of a new band voxel and its list key),
// largest rank (is the voxel re-examined by a later flip?).  The cases of the stencil diverge inside a wave; loads inside
// them run one case after the other - a chain of round trips per wave - while these travel together for all lanes.
struct VrgRanks { uint32_t minFO, kFO, maxFO, minAP, kAP, maxAP; };
VRG_HD void vrg_ranks_none(VrgRanks& q) { q.minFO = q.minAP = 0xffffffffu; q.kFO = q.kAP = 0; q.maxFO = q.maxAP = 0; }
VRG_HD void vrg_ranks_take(uint32_t FO, const uint32_t n[VRG_RANK_BATCH], const uint32_t r[VRG_RANK_BATCH], VrgRanks& q) {
    for (int k = 0; k < VRG_RANK_BATCH; k++) {
        if (n[k] >= 27u) continue;
        if ((FO >> n[k]) & 1u) { if (r[k] < q.minFO) { q.minFO = r[k]; q.kFO = 26u - vrg_nk(n[k]); } if (r[k] > q.maxFO) q.maxFO = r[k]; }
        else { if (r[k] < q.minAP) { q.minAP = r[k]; q.kAP = 26u - vrg_nk(n[k]); } if (r[k] > q.maxAP) q.maxAP = r[k]; }
    }
}
VRG_HD void vrg_nbr_ranks(const VrgCtx& c, uint32_t FO, uint32_t AP, uint32_t idx, VrgRanks& q) {
    vrg_ranks_none(q);
    uint32_t cand = FO | AP;
    while (cand) {
        uint32_t n[VRG_RANK_BATCH], r[VRG_RANK_BATCH];
        vrg_rank_batch(c, cand, idx, n, r);
        vrg_ranks_take(FO, n, r, q);
    }
}

// Returns the voxel's byte after the sweep and files what the change means for the band pool (slot born / dead /
// re-appended), the class histograms and the sweep's level deltas.  `lab` = this sweep's input labels (L/P bits
// set); nothing is written to the label volume here, so every stencil read sees the pre-sweep state.
// (the case analysis proper: everything it reads from memory is handed in - the neighbourhood masks, the neighbour ranks,
// whether an applied flip lies within the 2-ring (consulted for an excluded voxel without a listed neighbour only), the
// voxel's level (lev_fast) - so that a kernel can fetch all of it together, or from a tile it keeps in LDS)
VRG_HD void vrg_nbr_sets(const VrgNbr& m, uint32_t& ex, uint32_t& segA, uint32_t& FO, uint32_t& AP) {
    ex = ~m.O & 0x7ffdfffu;                               // neighbours that exist (:278-280), centre excluded
    segA = m.S & ~m.L & ex; FO = m.S & m.L & ex; AP = ~m.S & m.P & ex;
}
VRG_HD bool vrg_wants_ring2(uint8_t cb, const VrgNbr& m) {
    return !(cb & (VB_S | VB_B)) && (cb & VB_X) && (m.L & ~m.O & 0x7ffdfffu) == 0u;
}
VRG_HD uint8_t vrg_sweep_cases(const VrgCtx& c, uint32_t idx, uint8_t cb, const VrgPre& pre, const VrgNbr& m, const VrgRanks& q, bool ring2,
                               uint32_t lev_here, VrgEvent& ev) {
    ev.kind = VE_NONE; ev.pend = 0;
    uint32_t ex, segA, FO, AP; vrg_nbr_sets(m, ex, segA, FO, AP);
    const bool nSegA = segA != 0, nFO = FO != 0, nAP = AP != 0, nNonSegB = (ex & ~(segA | AP)) != 0, nListed = (m.L & ex) != 0;
    const VrgState& s = *c.st;
    if (cb & VB_S) {
        if (cb & VB_L) {                              // flip-out (:170-175), always applied
            const uint32_t r = pre.rank, slot = pre.vent, lev = c.lev_fast ? lev_here : c.f_lev[r];   // (a flip is a band entry: its slot is the voxel's)
            vrg_hist_change(c, lev, -1, 1);
            const bool to3 = !nSegA && q.maxFO > r;   // re-examined by a later flip-out neighbour? (:183-190)
            if (!to3) {                               // stays 2, carried to the outer list (by rank)
                c.f_res[r] = FR_WRITTEN | 2;
                vrg_note_level(c, c.dOut, lev);
                vrg_ev_move(ev, slot, true, false, vrg_key(s, 0, r, 0), false);
                return VB_B;
            }
            if (nAP) {                                // 3 -> 2 again (:210-213): a new outer entry
                c.f_res[r] = FR_WRITTEN | 2 | FR_FRESH;
                vrg_note_level(c, c.dOut, lev);
                vrg_ev_move(ev, slot, true, false, vrg_key(s, 1, q.minAP, q.kAP), true);
                return VB_B;
            }
            c.f_res[r] = FR_WRITTEN | 3;
            vrg_ev_die(ev, slot, true);
            return 0;
        }
        bool is1 = (cb & VB_B) || nFO;                // label after phase A (:194)
        if (!is1) return VB_S;
        if (nAP && !nNonSegB) {                       // 1 -> 0 (:223-228)
            if (cb & VB_B) vrg_ev_die(ev, pre.vent, true);
            return VB_S;
        }
        if (!(cb & VB_B)) {                           // newly on the inner boundary
            vrg_ev_new(ev, c.lev_fast ? lev_here : vrg_pre_level(c, pre), true, vrg_key(s, 0, q.minFO, q.kFO));
        }
        return VB_S | VB_B;
    }
    if (cb & VB_B) {
        if ((cb & VB_L) && (cb & VB_P)) {             // applied flip-in (:198-204)
            const uint32_t r = pre.rank, slot = pre.vent, lev = c.lev_fast ? lev_here : c.f_lev[r];   // (a flip is a band entry: its slot is the voxel's)
            vrg_hist_change(c, lev, 1, -1);
            const bool to0 = !nNonSegB && q.maxAP > r;   // re-examined by a later applied flip-in nbr? (:219-228)
            bool fresh = nFO && !nSegA;               // had dropped to 3 in phase A: exact density (:212,:251)
            c.f_res[r] = (uint8_t)(FR_WRITTEN | (to0 ? 0 : 1) | (fresh ? FR_FRESH : 0));
            if (to0) { vrg_ev_die(ev, slot, false); return VB_S; }
            vrg_note_level(c, c.dIn, lev);
            vrg_ev_move(ev, slot, false, true, vrg_key(s, 1, r, 0), fresh);   // appended to the inner list by rank
            return VB_S | VB_B;
        }
        bool to3 = nFO && !nSegA;                     // :183-190
        uint8_t out, res;
        if (!to3) { out = VB_B; res = 2; }
        else if (nAP) {                               // left the band and re-entered: a new outer entry
            out = VB_B; res = 2 | FR_FRESH;
            vrg_ev_move(ev, pre.vent, false, false, vrg_key(s, 1, q.minAP, q.kAP), true);
        } else { out = 0; res = 3; vrg_ev_die(ev, pre.vent, false); }
        if (cb & VB_L) {                              // skipped flip-in
            const uint32_t r = pre.rank;
            c.f_res[r] = (uint8_t)(FR_WRITTEN | res);
            if ((res & FR_FINAL) == 2) vrg_note_level(c, c.dOut, c.lev_fast ? lev_here : c.f_lev[r]);
        }
        return out;
    }
    // labels 3 and 4
    bool conv = false;
    uint32_t lev = 0xffffffffu;
    if (cb & VB_X) {
                conv = nListed || ring2;                      // 1-ring of any listed flip (:166-168), 2-ring of any applied flip
        if (conv) {                                   // addedPoints (:235); the voxel joins the outer region
            lev = c.lev_fast ? lev_here : vrg_pre_level(c, pre);
            vrg_note_level(c, c.dConv, lev);
            vrg_hist_change(c, lev, 0, 1);
        }
    }
    if (nAP) {                                        // 3 -> 2 (:210-213)
        if (lev == 0xffffffffu) lev = c.lev_fast ? lev_here : vrg_pre_level(c, pre);
        vrg_ev_new(ev, lev, false, vrg_key(s, 1, q.minAP, q.kAP));
        return VB_B;
    }
    return (uint8_t)(((cb & VB_X) && !conv) ? VB_X : 0);
}

VRG_HD uint8_t vrg_sweep_core_pre(const VrgCtx& c, const uint8_t* lab, uint32_t idx, uint8_t cb, const VrgPre& pre, VrgEvent& ev) {
    const VrgNbr m = vrg_masks_of(pre.w);
    uint32_t ex, segA, FO, AP; vrg_nbr_sets(m, ex, segA, FO, AP);
    VrgRanks q; vrg_nbr_ranks(c, FO, AP, idx, q);         // (before the cases: see VrgRanks)
    const uint32_t lev_here = c.lev_fast ? vrg_pre_level(c, pre) : 0xffffffffu;   // the voxel's level where that costs no global load
    const bool ring2 = vrg_wants_ring2(cb, m) && vrg_ring2_applied(c, lab, idx);
    return vrg_sweep_cases(c, idx, cb, pre, m, q, ring2, lev_here, ev);
}

VRG_HD uint8_t vrg_sweep_core(const VrgCtx& c, const uint8_t* lab, uint32_t idx, uint8_t cb, VrgEvent& ev) {
    VrgPre pre; vrg_preload(c, lab, idx, pre);
    return vrg_sweep_core_pre(c, lab, idx, cb, pre, ev);
}

// ---- the change log (leader / follower replication, vrg_types.h VrgLogRec): written where a label byte is written
// place i of the sweep being applied (its records start at log position `base` = VrgState::log_pos when the apply began)
// (rank: the voxel's rank in the sweep's flip list where the caller has it at hand - VRG_NONE: look it up in the voxel's stamp)
VRG_HD void vrg_log_record(const VrgCtx& c, uint32_t base, uint32_t i, uint32_t idx, uint8_t old, uint8_t nw, uint32_t rank = VRG_NONE) {
    if (!c.log_rec) return;
    const uint32_t q = base - c.log_pos0 + i;
    if (q >= c.log_cap) { vrg_store_i32(&c.stg->error, 11); return; }
    VrgLogRec r; r.idx = idx; r.rank = 0; r.old = (uint8_t)(old & (VB_LABEL | VB_OOB)); r.nw = (uint8_t)(nw & (VB_LABEL | VB_OOB)); r.pad = 0; r.pad2 = 0;
    if (idx != VRG_NONE && (nw & VB_S) && !(old & VB_S)) r.rank = rank != VRG_NONE ? rank : (uint32_t)c.stamp[idx];      // an applied flip-in: its rank (stamped when the flips were ordered)
    c.log_rec[q] = r;
}
// sparse relabel, phase 1: new byte of every marked voxel from the OLD labels
VRG_HD void vrg_item_relabel(const VrgCtx& c, uint32_t i) {
    const uint32_t idx = c.mk_idx[i];
    VrgEvent ev;
    c.mk_new[i] = vrg_sweep_core(c, c.lab[0], idx, vrg_load_coherent(c.lab[0] + idx), ev);
    vrg_commit_event(c, idx, ev);
}
// phase 2: write the new bytes (this also clears the L / P / mark bits)
// class of a label for the region statistics (:113-116): 1 inner (S), 2 outer (neither S nor excluded), 0 neither
VRG_HD uint32_t vrg_cls_of(uint8_t b) { return (b & VB_S) ? 1u : ((b & (VB_X | VB_OOB)) ? 0u : 2u); }
// where voxel idx keeps its two class bits (layout: VrgCtx::clsb)
VRG_HD void vrg_cls_pos(uint32_t idx, uint32_t& dw, uint32_t& sh) {
    uint32_t o = idx & 1023u;
    dw = ((idx >> 10) << 6) | ((o & 255u) >> 2);
    sh = 2u * (((o >> 8) << 2) | (o & 3u));
}
// a voxel of a unit that held no included voxel so far becomes an outer one (sweep parity p): the unit goes onto the list of
// units the dense pass walks - through the bitmap of THIS sweep's new units (VrgCtx::unew), which the gate of the sweep's own
// pass merges; the first lister says that the list is out of date
VRG_HD void vrg_list_unit(const VrgCtx& c, uint32_t idx, int p, uint32_t bit) {
    if (c.ubits[idx >> 15] & bit) return;                  // (listed by an earlier sweep)
    if (!(vrg_atomic_or(&c.unew[p][idx >> 15], bit) & bit)) vrg_atomic_add(&c.uctl[UC_GEN + p * UC_GEN_STRIDE], 1u);
}
// a label byte changes during sweep iter+1: keep the region sizes and that sweep's copy of the class bits in step,
// and note the change for the other copy
VRG_HD void vrg_count_change(const VrgCtx& c, uint32_t idx, uint8_t old, uint8_t nw) {
    uint32_t a = vrg_cls_of(old), b = vrg_cls_of(nw);
    if (a == b) return;
    const int p = (c.st->iter + 1) & 1;
    uint32_t dw, sh; vrg_cls_pos(idx, dw, sh);
    const uint32_t x = (a ^ b) << sh;
    if (a == 0u) {             // (before the class bits: a listed unit may read as empty, never the reverse)
        const uint32_t bit = 1u << ((idx >> 10) & 31u);
        vrg_list_unit(c, idx, p, bit);
    }
    vrg_atomic_xor(&c.clsb[p][dw], x);
    uint32_t q = vrg_atomic_add(&c.nchg[p], 1u);
    if (q < c.mcap) { c.chg_dw[p][q] = dw; c.chg_x[p][q] = x; } else vrg_store_i32(&c.stg->error, 7);
    int din = (int)(b == 1u) - (int)(a == 1u), dout = (int)(b == 2u) - (int)(a == 2u);
    if (din) vrg_atomic_add64(&c.inc[VC_NIN], din);
    if (dout) vrg_atomic_add64(&c.inc[VC_NOUT], dout);
}
// the same with the change filed at a given place of the sweep's change list (the voxel's place in the marked list) instead
// of appended through a counter - no returning atomic; a voxel whose class did not change files VRG_NOCHG.  Whoever closes
// the sweep sets the list's length (vrg_post_apply).
#define VRG_NOCHG 0xffffffffu
// (acc: null - the region sizes move at once, two atomics on the same cache line per change; else the caller adds acc[0] / acc[1] up over its items
// and sends ONE pair per workgroup: same-address atomics execute one after the other at the memory side, and a sweep of 12 900 flips made
// 8 000 of them in k_close - its whole 35 us)
VRG_HD void vrg_count_change_at(const VrgCtx& c, uint32_t idx, uint8_t old, uint8_t nw, uint32_t pos, int* acc = nullptr) {
    uint32_t a = vrg_cls_of(old), b = vrg_cls_of(nw);
    const int p = (c.st->iter + 1) & 1;
    if (a == b) { c.chg_dw[p][pos] = VRG_NOCHG; return; }
    uint32_t dw, sh; vrg_cls_pos(idx, dw, sh);
    const uint32_t x = (a ^ b) << sh;
    if (a == 0u) {
        const uint32_t bit = 1u << ((idx >> 10) & 31u);
        vrg_list_unit(c, idx, p, bit);
    }
    vrg_atomic_xor(&c.clsb[p][dw], x);
    c.chg_dw[p][pos] = dw; c.chg_x[p][pos] = x;
    int din = (int)(b == 1u) - (int)(a == 1u), dout = (int)(b == 2u) - (int)(a == 2u);
    if (acc) { acc[0] += din; acc[1] += dout; return; }
    if (din) vrg_atomic_add64(&c.inc[VC_NIN], din);
    if (dout) vrg_atomic_add64(&c.inc[VC_NOUT], dout);
}
// change i of the sweep before: this sweep's copy of the class bits sat that sweep out
VRG_HD void vrg_catchup_entry(const VrgCtx& c, uint32_t dw, uint32_t x) {
    if (dw != VRG_NOCHG) vrg_atomic_xor(&c.clsb[(c.st->iter + 1) & 1][dw], x);
}
VRG_HD void vrg_item_catchup(const VrgCtx& c, uint32_t i) {
    const int p = (c.st->iter + 1) & 1;
    vrg_catchup_entry(c, c.chg_dw[p ^ 1][i], c.chg_x[p ^ 1][i]);
}
VRG_HD uint32_t vrg_catchup_count(const VrgCtx& c) { uint32_t n = vrg_load_u32(&c.nchg[((c.st->iter + 1) & 1) ^ 1]); return n < c.mcap ? n : c.mcap; }
VRG_HD void vrg_apply_voxel(const VrgCtx& c, uint32_t idx, uint8_t old, uint8_t nw) {
    c.lab[0][idx] = nw;
    vrg_count_change(c, idx, old, nw);
}
VRG_HD void vrg_item_apply(const VrgCtx& c, uint32_t i) {
    const uint32_t idx = c.mk_idx[i];
    const uint8_t old = vrg_load_coherent(c.lab[0] + idx), nw = c.mk_new[i];
    vrg_log_record(c, c.st->log_pos, i, idx, old, nw);
    vrg_apply_voxel(c, idx, old, nw);
}
// ... with the class change filed at place i of the change list (see vrg_count_change_at)
VRG_HD void vrg_apply_at(const VrgCtx& c, uint32_t i, uint32_t idx, uint8_t old, uint8_t nw, uint32_t log_base, int* acc = nullptr) {
    vrg_log_record(c, log_base, i, idx, old, nw);
    c.lab[0][idx] = nw;
    vrg_count_change_at(c, idx, old, nw, i, acc);
}
// one caller, after every label of sweep iter+1 is written and before anything of the next sweep: file the sizes
// that sweep's dense pass has to reproduce; the change list just consumed becomes the next sweep's
// nchg_at >= 0: the sweep filed its changes by place (vrg_count_change_at): that many places
VRG_HD void vrg_post_apply(const VrgCtx& c, int64_t nchg_at = -1) {
    const int64_t k = (int64_t)c.st->iter + 1;
    vrg_store_i64(&c.exp_ring[2 * (k % VRG_RING)], vrg_load_i64(&c.inc[VC_NIN])); vrg_store_i64(&c.exp_ring[2 * (k % VRG_RING) + 1], vrg_load_i64(&c.inc[VC_NOUT]));
    c.nchg[(k & 1) ^ 1] = 0;
    if (nchg_at >= 0) c.nchg[k & 1] = (uint32_t)nchg_at;
}
// init: class dword d from the labels (16 voxels), both copies
VRG_HD uint32_t vrg_cls_word_build(const VrgCtx& c, uint32_t d) {
    uint32_t base = ((d >> 6) << 10) | ((d & 63u) << 2), w = 0;
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t i4 = base + 256u * j;              // four voxels = one aligned word of labels (PV is a multiple of 16)
        if (i4 >= c.PV) continue;
        uint32_t l4; __builtin_memcpy(&l4, c.lab[0] + i4, 4);
        for (uint32_t b = 0; b < 4; b++) w |= vrg_cls_of((uint8_t)(l4 >> (8u * b))) << (2u * (4u * j + b));
    }
    c.clsb[0][d] = w; c.clsb[1][d] = w;
    return w;
}
VRG_HD void vrg_item_cls_build(const VrgCtx& c, uint32_t d) {
    if (vrg_cls_word_build(c, d)) vrg_atomic_or(&c.ubits[d >> 11], 1u << ((d >> 6) & 31u));     // (the bitmap was zeroed before)
}
// the unit list from the bitmap, sequentially (init of the test model; the device builds it in parallel: ulist_refresh)
VRG_HD void vrg_ulist_rebuild_serial(const VrgCtx& c) {
    const uint32_t plane = (uint32_t)c.PY * (uint32_t)c.PX, lo = (2u + (uint32_t)c.z0) * plane, hi = (2u + (uint32_t)c.z1) * plane;
    uint32_t f_lo = (uint32_t)(((uint64_t)lo + 1023u) >> 10), f_hi = hi >> 10, n = 0;
    for (uint32_t u = f_lo; u < f_hi; u++) if ((c.ubits[u >> 5] >> (u & 31u)) & 1u) c.ulist[n++] = u;
    c.uctl[UC_N] = n;
}
// one caller per applied sweep: the labels of sweep iter+1 are in place, a dense pass over them is due
// (the expected sizes and every class bit of the sweep have reached memory before the request does)
VRG_HD void vrg_request_dense(const VrgCtx& c) { vrg_drain(); vrg_store_i64(&c.gate[VG_REQ], (int64_t)c.st->iter + 1); }
VRG_HD bool vrg_dense_due(const VrgCtx& c) { return vrg_load_i64(&c.gate[VG_REQ]) > vrg_load_i64(&c.dctl[VD_RSEQ]); }
// Option verify_every = n: the dense pass (:113-116 - here a CHECK of the sizes kept by increments, and the source of the trace's
// intensity sums) runs on every n-th sweep only (0: never; the run's last sweep is then checked when it ends).  Is sweep number k's
// pass left out?
// With several verifiers (leader / follower replication) the passes that are due go round robin: pass number q = k / every - 1 is counted by
// verifier q % ver_n; ver_me < 0: this handle counts none.
VRG_HD bool vrg_dense_skipped(int64_t k, int64_t every, int32_t ver_n = 1, int32_t ver_me = 0) {
    if (every == 0 || ver_me < 0 || (every > 1 && k % every != 0)) return true;
    if (ver_n <= 1) return false;
    const int64_t q = (every > 1 ? k / every : k) - 1;
    return (int32_t)(q % ver_n) != ver_me;
}
// ... then the pass is closed without a count: a marker (negative sizes) takes the place of the slab sums, so that the sequence
// of passes - which class copy the next one reads, what k_close / k_band wait for - stays what it is
VRG_HD VrgDense vrg_dense_skip_marker() { VrgDense d; d.n_in = -1.0; d.n_out = -1.0; d.sum_in = 0.0; d.sum_out = 0.0; return d; }
// recount number rseq = recounts done + 1 (it read class copy rseq & 1) has this device's slab sums: keep them for the pass
VRG_HD void vrg_recount_done(const VrgCtx& c, const VrgDense& part) {
    const int64_t rseq = c.dctl[VD_RSEQ] + 1;
    c.dn_ring[rseq % VRG_RING] = part;
    c.dctl[VD_RSEQ] = rseq;
}
// close pass seq = passes closed + 1 with its totals over all slabs: cross-check the sizes it had to reproduce, file the sums
VRG_HD void vrg_dense_fin_one(const VrgCtx& c, const VrgDense& d) {
    const int64_t seq = c.dctl[VD_SEQ] + 1;
    if (d.n_in < 0.0) {                                   // a pass that was left out (verify_every): no sums for this sweep's trace record
        if ((uint64_t)seq < c.trace_cap) { c.trace[seq].sum_in = __builtin_nan(""); c.trace[seq].sum_out = __builtin_nan(""); }
        c.dctl[VD_SEQ] = seq;
        return;
    }
    if ((int64_t)d.n_in != c.exp_ring[2 * (seq % VRG_RING)] || (int64_t)d.n_out != c.exp_ring[2 * (seq % VRG_RING) + 1]) c.dctl[VD_ERR] = 5;
    if ((uint64_t)seq < c.trace_cap) { c.trace[seq].sum_in = d.sum_in; c.trace[seq].sum_out = d.sum_out; }
    *c.dn = d;
    c.dctl[VD_SEQ] = seq;
}
// the run has ended with passes left out: the labels of the last sweep counted after all (its class copy: the one pass number
// RSEQ read or would have read) - `d` = the totals over all slabs - against the sizes kept by increments; the sums go to the trace
VRG_HD void vrg_dense_verify_last(const VrgCtx& c, const VrgDense& d) {
    const int64_t seq = c.dctl[VD_RSEQ];
    if ((int64_t)d.n_in != vrg_load_i64(&c.inc[VC_NIN]) || (int64_t)d.n_out != vrg_load_i64(&c.inc[VC_NOUT])) c.dctl[VD_ERR] = 5;
    if (seq > 0 && (uint64_t)seq < c.trace_cap) { c.trace[seq].sum_in = d.sum_in; c.trace[seq].sum_out = d.sum_out; }
    *c.dn = d;
}
// Z-slabs: the partial sums of the recounts not yet closed, packed (zero-padded to a fixed length) for one all-reduce
VRG_HD void vrg_dense_pack(const VrgCtx& c) {
    int64_t n = c.dctl[VD_RSEQ] - c.dctl[VD_SEQ];
    if (n > VRG_STAGE) n = VRG_STAGE;
    for (int64_t j = 0; j < VRG_STAGE; j++) {
        VrgDense z = {0.0, 0.0, 0.0, 0.0};
        c.stage_in[j] = j < n ? c.dn_ring[(c.dctl[VD_SEQ] + 1 + j) % VRG_RING] : z;
    }
    c.dctl[VD_NST] = n;
}
VRG_HD void vrg_dense_fin_staged(const VrgCtx& c) {
    for (int64_t j = 0, n = c.dctl[VD_NST]; j < n; j++) vrg_dense_fin_one(c, c.stage_out[j]);
    c.dctl[VD_NST] = 0;
}
// init: the dense pass founds the incremental sizes
VRG_HD void vrg_init_counts(const VrgCtx& c) {
    c.inc[VC_NIN] = (int64_t)c.dn->n_in; c.inc[VC_NOUT] = (int64_t)c.dn->n_out; c.gate[VG_REQ] = 0; c.gate[VG_STOP] = 0;
    c.dctl[VD_SEQ] = 0; c.dctl[VD_ERR] = 0; c.dctl[VD_RSEQ] = 0; c.dctl[VD_NST] = 0; c.dctl[VD_GO] = 0; c.nchg[0] = 0; c.nchg[1] = 0;
}

// ------------------------------------------------------------------ the change log (leader / follower replication)
// the header of sweep k (its labels are in place, its nrec records written), by ONE caller; the state's log counters move on
// (t: the sweep's trace record as the caller has just filed it)
VRG_HD void vrg_log_sweep(const VrgCtx& c, int64_t k, uint32_t base, uint32_t nsw, uint32_t nrec, int64_t n_in, int64_t n_out, const VrgTrace& t) {
    if (!c.log_rec) return;
    const uint32_t q = nsw - c.log_nsw0;
    if (q >= c.log_swcap) { vrg_store_i32(&c.stg->error, 11); return; }
    VrgLogSweep w;
    w.nflip = t.nflip; w.nseg = n_in; w.n_in = n_in; w.n_out = n_out; w.ni = t.ni; w.no = t.no; w.ties = t.ties; w.near_ties = t.near_ties;
    w.sweep = (uint32_t)k; w.nrec = nrec; w.rec0 = base - c.log_pos0; w.pad = 0;
    c.log_sw[q] = w;
    c.stg->log_pos = base + nrec; c.stg->log_nsw = nsw + 1u;
}
// The batch's log is complete up to `nsw` sweep headers / `pos` records (VrgState::log_nsw / log_pos of a CLOSED state as a kernel finds it
// at its entry): by ONE thread of the FIRST kernel of a trip's update() - k_sweep, k_order, k_trip_open - for the sweep before its trip: every
// record and header up to there was written by kernels that have ENDED (a kernel's plain stores reach memory when it ends), so nothing has
// to be drained or written through but the word itself.  (Publishing from k_band's filing thread - one kernel earlier - cost k_band 984 bytes
// of scratch per thread: it runs at the limit of its scalar registers.)
VRG_HD void vrg_log_publish(const VrgCtx& c, uint32_t nsw, uint32_t pos, bool drain = true) {
    if (!c.log_ready || !c.log_rec) return;
    if (drain) vrg_drain();                            // (false: the caller has stored nothing the word announces - everything was written by kernels that have ended)
    vrg_store_u64_sys(c.log_ready, vrg_log_progress(c.log_seq, nsw - c.log_nsw0, pos - c.log_pos0));
}
// (a leader that enqueues no dense pass at all: the pass counters follow the sweeps, so that the handle stays consistent - which class copy
// is current, what a later pass would wait for - and the sweep's trace record says that nobody here summed its intensities)
VRG_HD void vrg_dense_none_step(const VrgCtx& c, int64_t k) {
    if (!c.dense_none) return;
    vrg_store_i64(&c.dctl[VD_RSEQ], k); vrg_store_i64(&c.dctl[VD_SEQ], k);
    if ((uint64_t)k < c.trace_cap) { c.trace[k].sum_in = __builtin_nan(""); c.trace[k].sum_out = __builtin_nan(""); }
}

// ---- a follower applies record r of sweep k: label byte, stamp of a voxel that became segmented, class bits (BOTH copies: a follower's
// passes and applies are in stream order, so the copies never differ), the unit bitmap through unew[0] (merged by the next refresh)
// (two halves: the label byte and stamp - sweeps in order, a voxel may change in consecutive sweeps - and the class bits, whose
// changes commute; a follower's dense pass reads the class bits only, so the first half may run beside a pass)
VRG_HD void vrg_follow_label_rec(const VrgCtx& c, const VrgLogRec& r, uint32_t k) {
    if (r.idx == VRG_NONE) return;
    const uint8_t have = (uint8_t)(c.lab[0][r.idx] & (VB_LABEL | VB_OOB));
    if (have != r.old) { c.dctl[VD_ERR] = 12; return; }       // this rank's labels have drifted from the leader's: nothing it counts can be trusted
    c.lab[0][r.idx] = r.nw;
    if ((r.nw & VB_S) && !(r.old & VB_S)) c.stamp[r.idx] = ((uint64_t)k << 32) | r.rank;
}
VRG_HD void vrg_follow_class_rec(const VrgCtx& c, const VrgLogRec& r) {
    if (r.idx == VRG_NONE) return;
    const uint32_t a = vrg_cls_of(r.old), b = vrg_cls_of(r.nw);
    if (a == b) return;
    uint32_t dw, sh; vrg_cls_pos(r.idx, dw, sh);
    const uint32_t x = (a ^ b) << sh;
    if (a == 0u) vrg_list_unit(c, r.idx, 0, 1u << ((r.idx >> 10) & 31u));
    vrg_atomic_xor(&c.clsb[0][dw], x); vrg_atomic_xor(&c.clsb[1][dw], x);
}
// the sweep whose count is next: what it has to reproduce (read by the pass's closing workgroup)
VRG_HD void vrg_follow_expect(const VrgCtx& c, const VrgLogSweep& w) { c.fexp[0] = (int64_t)w.sweep; c.fexp[1] = w.n_in; c.fexp[2] = w.n_out; }
// ... and files the sweep's trace record from its header (one caller); the sums come from whoever counts the sweep
VRG_HD void vrg_follow_trace(const VrgCtx& c, const VrgLogSweep& w) {
    if ((uint64_t)w.sweep >= c.trace_cap) return;
    VrgTrace& t = c.trace[w.sweep];
    t.nflip = w.nflip; t.nseg = w.nseg; t.n_in = w.n_in; t.n_out = w.n_out; t.ni = w.ni; t.no = w.no; t.ties = w.ties; t.near_ties = w.near_ties;
    t.sum_in = __builtin_nan(""); t.sum_out = __builtin_nan("");
}
// the pass over sweep k's labels is done (totals d): against the sizes the leader filed; the sums into the trace
VRG_HD void vrg_follow_check(const VrgCtx& c, const VrgDense& d, uint32_t k, int64_t n_in, int64_t n_out) {
    if (((int64_t)d.n_in != n_in || (int64_t)d.n_out != n_out) && c.dctl[VD_ERR] == 0) {      // (the first mismatch is kept for the error message)
        c.dctl[VD_ERR] = 5; c.fexp[3] = (int64_t)k; c.fexp[4] = (int64_t)d.n_in; c.fexp[5] = (int64_t)d.n_out; c.fexp[6] = n_in; c.fexp[7] = n_out;
    }
    if ((uint64_t)k < c.trace_cap) { c.trace[k].sum_in = d.sum_in; c.trace[k].sum_out = d.sum_out; }
    *c.dn = d;
}

// ------------------------------------------------------------------ closing the sweep
// the slots that died go onto the free list, below the entries this sweep's allocations consumed from its top
VRG_HD uint32_t vrg_free_used(uint32_t nalloc, uint32_t nfree) { return nalloc < nfree ? nalloc : nfree; }
VRG_HD void vrg_item_free(const VrgCtx& c, uint32_t j) {
    const uint32_t nfree = c.st->nfree;                   // (not changed during the sweep)
    c.freel[nfree - vrg_free_used(vrg_load_u32(&c.stg->nalloc), nfree) + j] = c.dead[j];
}
VRG_HD void vrg_free_entry(const VrgCtx& c, uint32_t j, uint32_t dead_slot, uint32_t nfree, uint32_t nalloc) {
    c.freel[nfree - vrg_free_used(nalloc, nfree) + j] = dead_slot;
}
// touched level j of the level-sorted list: value and counts out of the per-level counters.  clear: the counters go
// back to zero at once; otherwise whoever opens the next update() clears them (vrg_item_level_clear) - the two-launch
// closing kernel lets other workgroups read the counters while this runs.
VRG_HD void vrg_item_level_clear(const VrgCtx& c, uint32_t j) {
    const uint32_t l = (uint32_t)c.nz_key[j];
    c.dIn[l] = 0; c.dOut[l] = 0; c.dConv[l] = 0; c.ltouch[l] = 0;
}
VRG_HD void vrg_item_level(const VrgCtx& c, uint32_t j, bool clear) {
    const uint32_t l = (uint32_t)c.nz_key[j];
    c.nz_val[j] = c.lev[l];
    c.nz_cin[j] = vrg_load_u32(&c.dIn[l]); c.nz_cout[j] = vrg_load_u32(&c.dOut[l]); c.nz_cconv[j] = vrg_load_u32(&c.dConv[l]);
    if (clear) vrg_item_level_clear(c, j);
}
// a listed flip the relabel never visited would be an internal error
VRG_HD void vrg_item_check_flip(const VrgCtx& c, uint32_t r) { if (!(c.f_res[r] & FR_WRITTEN)) vrg_store_i32(&c.stg->error, 3); }
// iterNum += 1 (:117), list lengths, free list, trace record; what the next k_band finds pending
// (in three parts, so that whoever closes a sweep on the device can request everything it reads in ONE batch and send
// everything it writes together: vrg_close_sweep)
VRG_HD VrgState vrg_finalize_load(const VrgCtx& c, int64_t& n_in, int64_t& n_out) {
    VrgState s = vrg_load_state(c.stg);               // one round trip for the whole state (past L1), one to write it back
    // ... the counters this kernel's atomics moved come from L2
    s.nalloc = vrg_load_u32(&c.stg->nalloc); s.ndead = vrg_load_u32(&c.stg->ndead); s.nfresh = vrg_load_u32(&c.stg->nfresh);
    s.nnz = vrg_load_u32(&c.stg->nnz); s.nmk = vrg_load_u32(&c.stg->nmk); s.npend = vrg_load_u32(&c.stg->npend);
    s.d_ni = vrg_load_i32(&c.stg->d_ni); s.d_no = vrg_load_i32(&c.stg->d_no); s.error = vrg_load_i32(&c.stg->error);
    s.ties = vrg_load_u32(&c.stg->ties); s.near_ties = vrg_load_u32(&c.stg->near_ties);
    n_in = vrg_load_i64(&c.inc[VC_NIN]); n_out = vrg_load_i64(&c.inc[VC_NOUT]);
    if (c.rsv) {                                      // (four-launch trips: the relabel kernels reserved through VrgCtx::rsv; whoever closes the sweep zeroes it)
        const uint64_t a = vrg_load_u64(&c.rsv[0]), b = vrg_load_u64(&c.rsv[16]);
        s.nalloc = (uint32_t)a; s.ndead = (uint32_t)(a >> 32); s.nfresh = (uint32_t)b; s.nmk = (uint32_t)(b >> 32);
        s.d_ni = (int32_t)(uint32_t)vrg_load_u64(&c.rsv[32]); s.d_no = (int32_t)(uint32_t)vrg_load_u64(&c.rsv[48]);
        c.rsv[0] = 0; c.rsv[16] = 0; c.rsv[32] = 0; c.rsv[48] = 0;
    }
    return s;
}
// (tr: the integer fields of the sweep's trace record; no memory is touched here: vrg_finalize_store files what has to be filed)
VRG_HD void vrg_finalize_core(VrgState& s, int64_t n_in, int64_t n_out, bool use_tab, VrgTrace& tr) {
    const uint32_t used = vrg_free_used(s.nalloc, s.nfree);
    s.np += s.nalloc - used; s.nfree = s.nfree - used + s.ndead;
    s.ni = (uint32_t)((int32_t)s.ni + s.d_ni); s.no = (uint32_t)((int32_t)s.no + s.d_no);
    s.iter++;
    tr.nflip = s.nf; tr.nseg = n_in; tr.n_in = n_in; tr.n_out = n_out; tr.ni = s.ni; tr.no = s.no;
    tr.ties = s.ties - s.ties_filed; tr.near_ties = s.near_ties - s.near_filed; tr.sum_in = 0; tr.sum_out = 0;
    s.ties_filed = s.ties; s.near_filed = s.near_ties;
    s.last_nf = s.nf;
    s.nf = 0; s.npend = 0; s.nmk = 0; s.nalloc = 0; s.ndead = 0; s.d_ni = 0; s.d_no = 0;
    s.nfx = s.nfresh; s.nfresh = 0;                   // exact densities of the new entries: first thing next trip
    s.corr = 1; s.use_tab = use_tab ? 1 : 0;          // nnz stays: the next k_band reads the touched-level list
    s.apply_pending = 0; s.ap_n = 0; s.fr_n = 0; s.d_nin = 0; s.d_nout = 0; s.nvisit = 0; s.nnz_new = 0;   // (the fused sweep's closing thread sets its own afterwards)
    s.open = 0; s.log_n = 0;
    if (s.error) s.done = -1;
}
// ... the sweep's trace record (the intensity sums are filed by the dense pass, vrg_dense_fin), the dense side's stop word
VRG_HD void vrg_finalize_store(const VrgCtx& c, const VrgState& s, const VrgTrace& tr) {
    if ((uint32_t)s.iter < c.trace_cap) {
        VrgTrace& t = c.trace[s.iter];
        t.nflip = tr.nflip; t.nseg = tr.nseg; t.n_in = tr.n_in; t.n_out = tr.n_out; t.ni = tr.ni; t.no = tr.no;
        t.ties = tr.ties; t.near_ties = tr.near_ties;
    }
    if (s.error) vrg_store_i64(&c.gate[VG_STOP], 1);
}
VRG_HD void vrg_finalize_update(const VrgCtx& c, VrgState& s, int64_t n_in, int64_t n_out, bool use_tab, VrgTrace& tr) {
    vrg_finalize_core(s, n_in, n_out, use_tab, tr);
    vrg_finalize_store(c, s, tr);
}
VRG_HD void vrg_finalize(const VrgCtx& c, bool use_tab) {
    int64_t n_in, n_out;
    VrgState s = vrg_finalize_load(c, n_in, n_out);
    const uint32_t nrec = s.nmk < c.mcap ? s.nmk : c.mcap;
    VrgTrace tr;
    vrg_finalize_update(c, s, n_in, n_out, use_tab, tr);
    vrg_state_store(c.stg, s);
    vrg_log_sweep(c, (int64_t)s.iter, s.log_pos, s.log_nsw, nrec, n_in, n_out, tr);
    vrg_dense_none_step(c, (int64_t)s.iter);
}
// vrg_post_apply + vrg_request_dense + vrg_finalize by the ONE thread that closes a sweep in a kernel, as two round trips
// instead of three: what they read in one batch; the expected sizes, the change lists' lengths and the new state out
// together; the request word last, once all of that (and every class bit of the sweep) has reached memory
VRG_HD void vrg_close_sweep(const VrgCtx& c, int64_t nchg_at, bool use_tab) {
    int64_t n_in, n_out;
    VrgState s = vrg_finalize_load(c, n_in, n_out);
    const int64_t k = (int64_t)s.iter + 1;
    const uint32_t nrec = s.nmk < c.mcap ? s.nmk : c.mcap;   // (the sweep's marked list = its change log records)
    vrg_store_i64(&c.exp_ring[2 * (k % VRG_RING)], n_in); vrg_store_i64(&c.exp_ring[2 * (k % VRG_RING) + 1], n_out);
    c.nchg[(k & 1) ^ 1] = 0;
    if (nchg_at >= 0) c.nchg[k & 1] = (uint32_t)nchg_at;
    VrgTrace tr;
    vrg_finalize_update(c, s, n_in, n_out, use_tab, tr);
    vrg_state_store(c.stg, s);
    vrg_log_sweep(c, k, s.log_pos, s.log_nsw, nrec, n_in, n_out, tr);
    vrg_dense_none_step(c, k);
    vrg_drain(); vrg_store_i64(&c.gate[VG_REQ], k);
}

// ------------------------------------------------------------------ fused sweep (k_sweep): update() of a sweep with few flips as ONE launch
// The four-launch chain (k_band, k_order, k_mark_relabel, k_close) is bound by its three grid-wide seams and by dependent
// round trips on shared words.  For a sweep with at most VRG_FUSE_MAX flips - the regime of every
// bench volume and slab - update() needs no seam at all:
//  * ORDER WITHOUT A SEAM.  Every workgroup (one per flip) fetches ALL flip records k_band appended, ranks them itself
//    (counting sort in LDS = the reference's flip order, :88) and resolves the skip rule (:198, the P bit) for every flip-in
//    from the nine label rows around it and a hash set of the flip voxels: the same answer in every workgroup, so nobody has
//    to publish L / P bits or ranks through memory - they are written into the workgroup's own 9x9x9 label tile and a rank
//    tile in LDS, where the stencils of its 125 voxels read them.
//  * ONE OWNER PER VOXEL WITHOUT AN ATOMIC.  A voxel is relabelled by the flip of SMALLEST RANK that wants it (1-ring; 2-ring
//    for an excluded voxel) - every workgroup sees all flips within reach of its cube, so ownership needs no election.
//  * NOTHING IS APPLIED INSIDE THE SWEEP.  The owner files (voxel, old byte, new byte) at a fixed place (rank * 125 + cube
//    position) of the marked list; the label bytes, the class bits of the dense pass and the free list are brought up to date
//    by the NEXT trip's k_band, which reads no labels (vrg_deferred_*): the stencils of this sweep all read the pre-sweep labels
//    because nobody writes labels while they run.  What the next decisions DO need - region sizes, class histograms, band pool,
//    the sweep's level deltas - is complete when the kernel ends: the workgroup that finishes last (ticket) lists the touched
//    levels for the next k_band's corrections and closes the sweep (vrg_fuse_close).
// Same data structures as the four-launch chain, so a trip can go either way (k_sweep hands a sweep with more flips back:
// VBAIL_FUSE) and both are checked against the oracle by the same tests.
template <int NLEV> struct VrgFuseLdsT {
    uint64_t key[VRG_FUSE_MAX];                                        // sort keys as appended (place = append order)
    uint32_t f_idx[VRG_FUSE_MAX], f_slot[VRG_FUSE_MAX], f_lev[VRG_FUSE_MAX];   // by rank: voxel, slot, level
    uint32_t f_L[VRG_FUSE_MAX], f_FI[VRG_FUSE_MAX];                     // ... which of its 26 neighbours are listed flips / listed flip-ins (27-bit masks)
    uint8_t f_inner[VRG_FUSE_MAX], f_P[VRG_FUSE_MAX], f_pend[VRG_FUSE_MAX];
    uint16_t f_x[VRG_FUSE_MAX], f_y[VRG_FUSE_MAX], f_z[VRG_FUSE_MAX];  // ... its coordinates
    uint32_t tile[81 * 4];                                             // 9x9x9 label bytes around the workgroup's flip (rows of 16 bytes)
    uint16_t rank[729];                                                // ... rank of the flip sitting there (0xffff: none)
    uint32_t any_pend, changed;
    uint32_t n[4], base[4], nvis; int32_t d[4];                        // this workgroup's new / dead / pending events / change-log records, visited flips, list and size changes
    double lev[NLEV];                                                  // the level table (small level tables; a large one is never searched: VrgCtx::lidx)
};
typedef VrgFuseLdsT<VRG_FUSE_LEVELS> VrgFuseLds;
struct VrgFuseThread {                                                 // what a thread keeps in registers between the phases
    uint64_t key; uint32_t slot, idx, lev;                             // the flip record k_band appended at place t
    uint32_t row[4];                                                   // tile row t
    uint32_t frow[9]; uint32_t r;                                      // the nine label rows around the flip of RECORD t (requested as soon as the record is there), its rank
    VrgPre pre;                                                        // per-voxel fields of cube place t
    float valf; double val64; uint16_t l16; uint32_t l32;              // ... its intensity as loaded (whichever storage the volume has: converted when used), its level index
    VrgEvent ev; uint32_t rn, rd, rf;
    uint32_t lg_idx, lg_rank, lg_at; uint8_t lg_old, lg_new, lg_on;    // this place's change-log record (a label byte that changes), its number inside the workgroup
};
template <class LDS>
VRG_HD uint32_t vrg_fuse_level_of(const LDS& sh, uint32_t L, double v) {   // vrg_level_of on the LDS copy of the table
    uint32_t lo = 0, hi = L - 1u;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (sh.lev[m] < v) lo = m + 1u; else hi = m; }
    return lo;
}
VRG_HD uint32_t vrg_fuse_tile_row(int dy, int dz) { return (uint32_t)((dz + 4) * 9 + (dy + 4)); }
VRG_HD uint32_t vrg_fuse_tile_pos(int dx, int dy, int dz) { return (uint32_t)(((dz + 4) * 9 + (dy + 4)) * 9 + (dx + 4)); }
template <class LDS>
VRG_HD uint8_t vrg_fuse_tile_byte(const LDS& sh, int dx, int dy, int dz) {
    const uint32_t o = (uint32_t)(dx + 4);
    return (uint8_t)(sh.tile[4 * vrg_fuse_tile_row(dy, dz) + (o >> 2)] >> (8u * (o & 3u)));
}
VRG_HD void vrg_load_row16(const uint8_t* p, uint32_t out[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p));
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
#else
    __builtin_memcpy(out, p, 16);
#endif
}
// can this trip run fused?  0: yes; > 0: stop reason; < 0: -(bail reason).  The same answer in every workgroup (same inputs).
VRG_HD uint32_t vrg_fuse_limit(const VrgCtx& c) { return c.L > (uint32_t)VRG_FUSE_LEVELS ? (uint32_t)VRG_FUSE_MAX_BIG : (uint32_t)VRG_FUSE_MAX; }
VRG_HD int32_t vrg_fuse_gate(const VrgCtx& c, const VrgState& s0, int64_t n_in, uint32_t fuse_max) {
    const int32_t stop = vrg_stop_test_v(s0, n_in);
    if (stop) return stop;
    if (s0.error) return 1000;
    if (s0.nf > fuse_max) return -(int32_t)VBAIL_FUSE;
    const int32_t bail = vrg_capacity_test(c, s0.nf);
    return bail ? -bail : 0;
}
// phase 0: LDS arrays that must start empty (runs while the first loads travel)
template <class LDS>
VRG_HD void vrg_fuse_init(LDS& sh, uint32_t t) {
    for (uint32_t i = t; i < 729u; i += VRG_FUSE_THREADS) sh.rank[i] = 0xffffu;
    if (t < (uint32_t)VRG_FUSE_MAX) { sh.f_L[t] = 0; sh.f_FI[t] = 0; }
    if (t < 4) sh.n[t] = 0;
    if (t < 4) sh.d[t] = 0;
    if (t == 0) { sh.any_pend = 0; sh.changed = 0; sh.nvis = 0; }
}
// first batch of loads: the flip record at place t (any place below the capacity is readable, whatever the state says)
// (every load of the two batches is UNCONDITIONAL, its index clamped into the array instead: a load under a branch makes the
// compiler wait for everything in flight before the next one - a batch would become a chain of round trips)
VRG_HD void vrg_fuse_load1(const VrgCtx& c, VrgFuseThread& th, uint32_t t) {
    const uint32_t q = t < c.fcap ? t : c.fcap - 1u;
    th.key = c.f_key[q]; th.slot = c.flist[q]; th.idx = c.fr_idx[q]; th.lev = c.fr_lev[q];
}
// ... and, as soon as the record is there - before the flips are ranked: nothing here depends on the order -, the nine label rows
// around the record's voxel (only a flip-in's are looked at - flip-outs are always applied; unconditional: see above)
VRG_HD void vrg_fuse_load_rows(const VrgCtx& c, VrgFuseThread& th) {
    const uint8_t* lab = c.lab[0];
    // (a record beyond the sweep's flips is stale: any index - kept inside the padded volume, its rows are never looked at)
    const uint32_t lo = (uint32_t)((c.PY + 1) * c.PX + 1), hi = c.PV - lo - 4u;
    const uint32_t idx = th.idx < lo ? lo : th.idx > hi ? hi : th.idx;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 9; j++) th.frow[j] = vrg_load_row(lab + ((int64_t)idx + ((j % 3 - 1) * c.PY + (j / 3 - 1)) * c.PX - 1));
}
template <class LDS>
VRG_HD void vrg_fuse_keys(LDS& sh, const VrgFuseThread& th, uint32_t t, uint32_t nf) { if (t < nf) sh.key[t] = th.key; }
// rank of record t = number of smaller keys (keys are distinct): the reference's flip order (:48, :88); the records by rank,
// the voxel -> rank hash set
template <class LDS>
VRG_HD void vrg_fuse_rank(const VrgCtx& c, LDS& sh, VrgFuseThread& th, uint32_t t, uint32_t nf) {
    if (t >= nf) return;
    uint32_t r = 0;
    for (uint32_t j = 0; j < nf; j++) r += sh.key[j] < th.key;
    th.r = r;
    int x, y, z; vrg_coords(c, th.idx, x, y, z);
    sh.f_idx[r] = th.idx; sh.f_slot[r] = th.slot; sh.f_lev[r] = th.lev; sh.f_inner[r] = (uint8_t)!(th.key >> 63);
    sh.f_x[r] = (uint16_t)x; sh.f_y[r] = (uint16_t)(y + 2); sh.f_z[r] = (uint16_t)(z + 2);    // (+2: never negative, differences unchanged)
}
// second batch of loads (addresses follow from the ranked records): tile row t of the workgroup's flip, the per-voxel
// fields of cube place t, the nine label rows around flip t
template <class LDS>
VRG_HD void vrg_fuse_load2(const VrgCtx& c, const LDS& sh, VrgFuseThread& th, uint32_t t, uint32_t r, uint32_t nf) {
    const uint8_t* lab = c.lab[0];
    const uint32_t fidx = sh.f_idx[r];
    {   // tile row (t < 81; the other threads re-read row 80).  A row that is not wholly inside the allocation - 16 guard bytes at
        // either end - belongs to no real voxel's neighbourhood: it reads as out-of-bounds bytes (the load goes to the array's start)
        const uint32_t tr = t < 81u ? t : 80u;
        const int ry = (int)(tr % 9u) - 4, rz = (int)(tr / 9u) - 4;
        const int64_t a = (int64_t)fidx + ((int64_t)rz * c.PY + ry) * c.PX - 4;
        const bool ok = a >= -16 && a + 16 <= (int64_t)c.PV + 16;
        vrg_load_row16(lab + (ok ? a : 0), th.row);
        if (!ok) th.row[0] = th.row[1] = th.row[2] = th.row[3] = 0x01010101u * VB_OOB;
    }
    {   // per-voxel fields of cube place t (t < 125; a place outside the real volume is padding - never relabelled: its index is
        // clamped to stay inside the arrays)
        const uint32_t tp = t < 125u ? t : 124u;
        const int dx = (int)(tp % 5u) - 2, dy = (int)((tp / 5u) % 5u) - 2, dz = (int)(tp / 25u) - 2;
        const int64_t m = (int64_t)fidx + ((int64_t)dz * c.PY + dy) * c.PX + dx;
        const int64_t lo = (int64_t)vrg_idx(c, 0, 0, 0), hi = (int64_t)vrg_idx(c, c.nx - 1, c.ny - 1, c.nz - 1);
        const uint32_t ms = (uint32_t)(m < lo ? lo : (m > hi ? hi : m));
        th.pre.rank = 0; th.pre.lev16 = 0; th.pre.val = 0.0;
        th.pre.vent = c.vent[ms];
        // the intensity in all three storage forms, each through a pointer that is valid whatever the volume's storage is (a form
        // the volume does not have reads bytes of another array, never used): no branch, no conversion here - either would
        // make the compiler wait for every load in flight
        const float* pf = c.I ? c.I : reinterpret_cast<const float*>(c.I64);
        const double* pd = c.I64 ? c.I64 : reinterpret_cast<const double*>(c.I);
        // (a form that is not there reads the cache line the intensity itself comes from: no extra line leaves memory)
        const uint16_t* p16 = c.lev16 ? c.lev16 : reinterpret_cast<const uint16_t*>(c.I ? (const void*)c.I : (const void*)c.I64);
        const uint32_t* p32 = c.lidx ? c.lidx : reinterpret_cast<const uint32_t*>(c.I ? (const void*)c.I : (const void*)c.I64);
        th.valf = pf[ms]; th.val64 = pd[c.I64 ? ms : (ms >> 1)]; th.l16 = p16[c.lev16 ? ms : 2u * ms]; th.l32 = p32[ms];
    }
}
// which neighbours of a flip-in are listed flips (f_L), and which of those flip-ins (f_FI): every pair of flips is looked at,
// `parts` threads sharing a flip-in's partners.  (Pairs, not a voxel -> flip hash set probed 26 times per flip: a probe is a
// chain of dependent LDS reads of ~100 cycles each, while the partners' coordinates are read one after the other, independent
// of each other.)  Neighbour n of the masks = 3 * j + (dx + 1) with j = 3 * (dy + 1) + (dz + 1).
template <class LDS>
VRG_HD void vrg_fuse_listed_nbrs(LDS& sh, uint32_t t, uint32_t nf) {
    const uint32_t parts = nf >= (uint32_t)VRG_FUSE_THREADS ? 1u : (uint32_t)VRG_FUSE_THREADS / nf;
    const uint32_t f = t / parts, part = t - f * parts;
    if (f >= nf || sh.f_inner[f]) return;                  // (flip-outs are always applied: only a flip-in's neighbourhood is looked at)
    const int x = sh.f_x[f], y = sh.f_y[f], z = sh.f_z[f];
    uint32_t L = 0, FI = 0;
    for (uint32_t g = part; g < nf; g += parts) {
        const int dx = (int)sh.f_x[g] - x + 1, dy = (int)sh.f_y[g] - y + 1, dz = (int)sh.f_z[g] - z + 1;
        if ((uint32_t)dx > 2u || (uint32_t)dy > 2u || (uint32_t)dz > 2u || g == f) continue;
        const uint32_t bit = 1u << (3u * (3u * (uint32_t)dy + (uint32_t)dz) + (uint32_t)dx);
        L |= bit;
        if (!sh.f_inner[g]) FI |= bit;
    }
    if (L) vrg_lds_or(&sh.f_L[f], L);
    if (FI) vrg_lds_or(&sh.f_FI[f], FI);
}
// the tile into LDS; the skip rule for flip t (:183-190, :198): a flip-in that dropped to 3 in phase A (a flip-out neighbour and
// no segmented neighbour left) is pending, every other flip is applied
template <class LDS>
VRG_HD void vrg_fuse_prepass(LDS& sh, const VrgFuseThread& th, uint32_t t, uint32_t nf) {
    if (t < 81u) { sh.tile[4 * t] = th.row[0]; sh.tile[4 * t + 1] = th.row[1]; sh.tile[4 * t + 2] = th.row[2]; sh.tile[4 * t + 3] = th.row[3]; }
    if (t >= nf) return;
    const uint32_t r = th.r;                               // (thread t holds the rows of RECORD t: its flip has rank r)
    if (sh.f_inner[r]) { sh.f_P[r] = 1; sh.f_pend[r] = 0; return; }
    uint32_t S = 0, O = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 9; j++) { S |= vrg_gather3(th.frow[j]) << (3 * j); O |= vrg_gather3(th.frow[j] >> 5) << (3 * j); }
    const uint32_t ex = ~O & 0x7ffdfffu, L = sh.f_L[r];
    const bool nFO = (S & L & ex) != 0, nSegA = (S & ~L & ex) != 0;     // (a listed segmented neighbour is a flip-out)
    const bool pend = nFO && !nSegA;
    sh.f_P[r] = (uint8_t)!pend; sh.f_pend[r] = (uint8_t)pend;
    if (pend) vrg_lds_or(&sh.any_pend, 1u);
}
// one relaxation of the skip rule's fix-point (vrg_item_fix): applied if an applied flip-in neighbour of smaller rank exists
template <class LDS>
VRG_HD void vrg_fuse_fix(const VrgCtx& c, LDS& sh, uint32_t t, uint32_t nf) {
    if (t >= nf || !sh.f_pend[t] || sh.f_P[t]) return;
    const int x = sh.f_x[t], y = sh.f_y[t], z = sh.f_z[t];
    for (uint32_t g = 0; g < t; g++) {
        const int dx = (int)sh.f_x[g] - x + 1, dy = (int)sh.f_y[g] - y + 1, dz = (int)sh.f_z[g] - z + 1;
        if ((uint32_t)dx > 2u || (uint32_t)dy > 2u || (uint32_t)dz > 2u) continue;
        if (!sh.f_inner[g] && sh.f_P[g]) { sh.f_P[t] = 1; vrg_lds_or(&sh.changed, 1u); return; }
    }
}
// flip t, if it lies inside the workgroup's tile: its L (+ P) bits into the tile, its rank into the rank tile
template <class LDS>
VRG_HD void vrg_fuse_annotate(const VrgCtx& c, LDS& sh, uint32_t t, uint32_t r, uint32_t nf) {
    if (t >= nf) return;
    const int dx = (int)sh.f_x[t] - (int)sh.f_x[r], dy = (int)sh.f_y[t] - (int)sh.f_y[r], dz = (int)sh.f_z[t] - (int)sh.f_z[r];
    if (dx < -4 || dx > 4 || dy < -4 || dy > 4 || dz < -4 || dz > 4) return;
    const uint32_t o = (uint32_t)(dx + 4), bits = (uint32_t)(VB_L | (sh.f_P[t] ? VB_P : 0));
    vrg_lds_or(&sh.tile[4 * vrg_fuse_tile_row(dy, dz) + (o >> 2)], bits << (8u * (o & 3u)));
    sh.rank[vrg_fuse_tile_pos(dx, dy, dz)] = (uint16_t)t;
}
// cube place t: is this workgroup's flip (rank r) the owner of the voxel - the flip of smallest rank that wants it
// (vrg_mark_wanted: 1-ring of a listed flip; 2-ring too for an excluded voxel)?  Then the relabel stencil from the tile.
template <class LDS>
VRG_HD void vrg_fuse_stencil(const VrgCtx& c, LDS& sh, VrgFuseThread& th, uint32_t t, uint32_t r) {
    th.ev.kind = VE_NONE; th.ev.pend = 0; th.rn = th.rd = th.rf = 0; th.lg_on = 0;
    if (t >= 125u) return;
    const uint32_t place = r * (uint32_t)VRG_FUSE_PLACES + t;
    const int dx = (int)(t % 5u) - 2, dy = (int)((t / 5u) % 5u) - 2, dz = (int)(t / 25u) - 2;
    const uint8_t mb = vrg_fuse_tile_byte(sh, dx, dy, dz);
    // (smallest rank in the voxel's 3x3x3 - 5x5x5 for an excluded voxel - box of the rank tile; 0xffff = no flip there)
    uint32_t owner = 0xffffu;
    uint32_t rk[27];                                       // the ranks of the flips in the voxel's 3x3x3 box, in the masks' neighbour order n = 9 * (dy+1) + 3 * (dz+1) + (dx+1)
    if (!(mb & VB_OOB)) {
        const uint16_t* r0 = &sh.rank[vrg_fuse_tile_pos(dx - 1, dy - 1, dz - 1)];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int n = 0; n < 27; n++) rk[n] = r0[((n / 3) % 3) * 81 + (n / 9) * 9 + (n % 3)];      // 27 independent reads
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int n = 0; n < 27; n++) owner = rk[n] < owner ? rk[n] : owner;
        if (mb & VB_X)                                     // an excluded voxel is also wanted by the flips of its 2-ring
            for (int ez = -2; ez <= 2; ez++)
                for (int ey = -2; ey <= 2; ey++) {
                    const uint16_t* row = &sh.rank[vrg_fuse_tile_pos(dx - 2, dy + ey, dz + ez)];
                    for (int ex = 0; ex < 5; ex++) owner = row[ex] < owner ? row[ex] : owner;
                }
    }
    if (owner != r) { c.mk_idx[place] = VRG_NONE; return; }
    const uint32_t m = (uint32_t)((int64_t)sh.f_idx[r] + ((int64_t)dz * c.PY + dy) * c.PX + dx);
    // the voxel's 3x3x3 masks from the tile rows (bytes x-1 .. x+2 of the nine rows: vrg_preload's layout)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 9; j++) {
        const uint32_t* rw = &sh.tile[4 * vrg_fuse_tile_row(dy + j / 3 - 1, dz + j % 3 - 1)];
        const uint64_t w8 = (uint64_t)rw[0] | ((uint64_t)rw[1] << 32);
        th.pre.w[j] = (uint32_t)(w8 >> (8 * (dx + 3)));
    }
    const VrgNbr nb = vrg_masks_of(th.pre.w);
    uint32_t ex, segA, FO, AP; vrg_nbr_sets(nb, ex, segA, FO, AP);
    VrgRanks q; vrg_ranks_none(q);                                     // the ranks of the listed neighbours: already in registers (no loop over set bits: no dependent reads)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int n = 0; n < 27; n++) {
        const uint32_t kk = 26u - vrg_nk((uint32_t)n);
        if ((FO >> n) & 1u) { if (rk[n] < q.minFO) { q.minFO = rk[n]; q.kFO = kk; } if (rk[n] > q.maxFO) q.maxFO = rk[n]; }
        if ((AP >> n) & 1u) { if (rk[n] < q.minAP) { q.minAP = rk[n]; q.kAP = kk; } if (rk[n] > q.maxAP) q.maxAP = rk[n]; }
    }
    bool ring2 = false;
    if (vrg_wants_ring2(mb, nb)) {                                     // an applied flip (P and not OOB) within the 2-ring? (vrg_ring2_applied)
        uint64_t any = 0;
        const uint32_t o2 = (uint32_t)(dx + 2);
        for (int j = 0; j < 25; j++) {
            const uint32_t* rw = &sh.tile[4 * vrg_fuse_tile_row(dy + j % 5 - 2, dz + j / 5 - 2)];
            const uint64_t lo8 = (uint64_t)rw[0] | ((uint64_t)rw[1] << 32);
            const uint64_t w8 = o2 ? (lo8 >> (8u * o2)) | ((uint64_t)rw[2] << (64u - 8u * o2)) : lo8;    // bytes x-2 .. x+2
            any |= ((w8 >> 4) & ~(w8 >> 5)) & 0x0101010101ull;
        }
        ring2 = any != 0;
    }
    th.pre.rank = sh.rank[vrg_fuse_tile_pos(dx, dy, dz)];              // (its own rank, if the voxel is a listed flip)
    th.pre.lev16 = c.lev16 ? (uint32_t)th.l16 : th.l32; th.pre.val = c.I ? (double)th.valf : th.val64;
    // the voxel's level, where the cases will ask for it (c.lev_fast = 1): a flip's from its record; a voxel that may enter the
    // band or the outer region searches the level table in LDS (through sh, not through the context's generic pointer: a
    // flat load waits for every atomic the wave has in flight); nobody else pays for a search
    uint32_t lev_here = 0xffffffffu;
    if (mb & VB_L) lev_here = sh.f_lev[th.pre.rank];
    else if (!(mb & VB_B) && ((mb & VB_S) ? FO != 0u : (AP != 0u || (mb & VB_X)))) lev_here = (c.lev16 || c.lidx) ? th.pre.lev16 : vrg_fuse_level_of(sh, c.L, th.pre.val);
    const uint8_t nw = vrg_sweep_cases(c, m, mb, th.pre, nb, q, ring2, lev_here, th.ev);
    c.mk_idx[place] = m; c.mk_new[place] = nw; c.mk_old[place] = mb;
    // the change log of a fused sweep: only label bytes that change, compacted - the record takes a number inside the workgroup here, the
    // workgroup's stretch of the log is reserved with its event lists (vrg_fuse_reserve), the record written with them (vrg_fuse_commit)
    if (c.log_rec && ((mb ^ nw) & (VB_LABEL | VB_OOB))) {
        th.lg_on = 1; th.lg_idx = m; th.lg_rank = th.pre.rank; th.lg_old = mb; th.lg_new = nw;
        th.lg_at = vrg_lds_add(&sh.n[3], 1u);
    }
    if (mb & VB_L) vrg_lds_add(&sh.nvis, 1u);
    const uint32_t a = vrg_cls_of(mb), b = vrg_cls_of(nw);             // region sizes (:113-116): kept by increments
    const int din = (int)(b == 1u) - (int)(a == 1u), dout = (int)(b == 2u) - (int)(a == 2u);
    if (din) vrg_lds_add(&sh.d[2], din);
    if (dout) vrg_lds_add(&sh.d[3], dout);
    // the event takes a number inside the workgroup (vrg_ev_write's q, qd, qf)
    if (th.ev.kind == VE_NEW) th.rn = vrg_lds_add(&sh.n[0], 1u);
    if (th.ev.kind == VE_DIE) th.rd = vrg_lds_add(&sh.n[1], 1u);
    if (th.ev.kind != VE_NONE && th.ev.kind != VE_DIE && th.ev.pend) th.rf = vrg_lds_add(&sh.n[2], 1u);
    const int di = vrg_ev_dni(th.ev), dq = vrg_ev_dno(th.ev);
    if (di) vrg_lds_add(&sh.d[0], di);
    if (dq) vrg_lds_add(&sh.d[1], dq);
}
// the workgroup reserves its stretch of every list with ONE atomic each (threads 0..8)
template <class LDS>
VRG_HD void vrg_fuse_reserve(const VrgCtx& c, LDS& sh, uint32_t t) {
    if (t < 3u) { if (sh.n[t]) sh.base[t] = vrg_atomic_add(t == 0 ? &c.stg->nalloc : t == 1 ? &c.stg->ndead : &c.stg->nfresh, sh.n[t]); }
    else if (t < 7u) { if (sh.d[t - 3u]) vrg_atomic_add(t == 3 ? &c.stg->d_ni : t == 4 ? &c.stg->d_no : t == 5 ? &c.stg->d_nin : &c.stg->d_nout, sh.d[t - 3u]); }
    else if (t == 7u) { if (sh.nvis) vrg_atomic_add(&c.stg->nvisit, sh.nvis); }
    else if (t == 8u) { if (sh.n[3]) sh.base[3] = c.st->log_pos + vrg_atomic_add(&c.stg->log_n, sh.n[3]); }     // (log_pos: the same in every workgroup's snapshot - only the closing thread moves it)
}
// ... and every event is written at its place; the workgroup's flip gets its stamp = (sweep, rank) (:200: segmented's list order)
template <class LDS>
VRG_HD void vrg_fuse_commit(const VrgCtx& c, const LDS& sh, const VrgFuseThread& th, uint32_t t, uint32_t r) {
    if (th.ev.kind != VE_NONE) {
        const int dx = (int)(t % 5u) - 2, dy = (int)((t / 5u) % 5u) - 2, dz = (int)(t / 25u) - 2;
        const uint32_t m = (uint32_t)((int64_t)sh.f_idx[r] + ((int64_t)dz * c.PY + dy) * c.PX + dx);
        vrg_ev_write(c, m, th.ev, sh.base[0] + th.rn, sh.base[1] + th.rd, sh.base[2] + th.rf);
    }
    if (th.lg_on) vrg_log_record(c, sh.base[3], th.lg_at, th.lg_idx, th.lg_old, th.lg_new, th.lg_rank);
    if (t == 0) c.stamp[sh.f_idx[r]] = ((uint64_t)(uint32_t)(c.st->iter + 1) << 32) | r;
}
// the touched levels of the sweep in ascending order with their counts (for the next k_band's corrections, :236-247), the
// per-level counters back to zero: level l by whoever closes the sweep (counters read past L1: other workgroups' atomics)
VRG_HD bool vrg_fuse_level_touched(const VrgCtx& c, uint32_t l, uint32_t& ci, uint32_t& co, uint32_t& cc) {
    ci = vrg_load_u32(&c.dIn[l]); co = vrg_load_u32(&c.dOut[l]); cc = vrg_load_u32(&c.dConv[l]);
    return (ci | co | cc) != 0u;
}
VRG_HD void vrg_fuse_level_file(const VrgCtx& c, uint32_t q, uint32_t l, double v, uint32_t ci, uint32_t co, uint32_t cc) {
    c.nz_key[q] = l; c.nz_val[q] = v; c.nz_cin[q] = ci; c.nz_cout[q] = co; c.nz_cconv[q] = cc;
    c.dIn[l] = 0; c.dOut[l] = 0; c.dConv[l] = 0;
}
// (large level table: entry q of the SORTED list of the levels the sweep's first touchers listed)
VRG_HD void vrg_fuse_level_file_listed(const VrgCtx& c, uint32_t q, uint32_t l) {
    uint32_t ci, co, cc; (void)vrg_fuse_level_touched(c, l, ci, co, cc);
    vrg_fuse_level_file(c, q, l, c.lev[l], ci, co, cc);
    c.ltouch[l] = 0;
}
// the per-level memo of the three corrections (:236-247) for level l from the touched-level list (in LDS: levels' indices and
// counts), lanes striding over the list, the kernel between two levels from the table - the same terms in the same order as
// the four-launch chain's closing kernel adds them: the memo is bit-identical whichever kind of trip wrote it
VRG_HD void vrg_fuse_memo_terms(const VrgCtx& c, uint32_t l, uint32_t lane, uint32_t nnz, const uint32_t* nzl, const uint32_t* cin, const uint32_t* cout,
                                const uint32_t* cconv, double& a, double& b, double& d) {
    const double* row = c.ktab + (size_t)l * c.L;
    a = 0; b = 0; d = 0;
    for (uint32_t j = lane; j < nnz; j += 64u) {
        const double k = row[nzl[j]];
        a += (double)cin[j] * k; b += (double)cout[j] * k; d += (double)cconv[j] * k;
    }
}
// iterNum += 1 (:117) by the ONE thread that closes a fused sweep: region sizes from the increments the stencils added up,
// the sizes the sweep's dense pass has to reproduce, trace record, and what the next trip's k_band finds to do
// (in two parts, so that the closing workgroup can request what the first part reads together with the level counters)
VRG_HD VrgState vrg_fuse_close_load(const VrgCtx& c, int64_t& n_in, int64_t& n_out) {
    VrgState s = vrg_finalize_load(c, n_in, n_out);
    s.d_nin = vrg_load_i32(&c.stg->d_nin); s.d_nout = vrg_load_i32(&c.stg->d_nout); s.nvisit = vrg_load_u32(&c.stg->nvisit);
    s.nnz_new = vrg_load_u32(&c.stg->nnz_new); s.log_n = vrg_load_u32(&c.stg->log_n);
    return s;
}
// (the sweep's change-log records, compacted by the workgroups' reservations: s.log_n of them from s.log_pos on)
// In two parts.  vrg_fuse_close_core: the state after the sweep from the state it ran on + what its workgroups added up - no memory
// is touched, so EVERY workgroup of the next trip's k_band can do this for itself when the sweep was open-ended.  vrg_fuse_close_store:
// what has to reach memory, by one thread - the sweep's own closing thread, or one thread of that k_band.
struct VrgFuseClosed { VrgTrace tr; int64_t n_in, n_out; uint32_t places, log_open, log_n; };
VRG_HD void vrg_fuse_close_core(const VrgCtx& c, VrgState& s, int64_t n_in, int64_t n_out, uint32_t nnz, bool use_tab, VrgFuseClosed& f) {
    n_in += s.d_nin; n_out += s.d_nout;
    if (s.nvisit != s.nf && !s.error) s.error = 3;        // a listed flip the stencils never visited
    f.n_in = n_in; f.n_out = n_out;
    f.places = s.nf * (uint32_t)VRG_FUSE_PLACES;
    const uint32_t used = vrg_free_used(s.nalloc, s.nfree), fr_base = s.nfree - used, fr_n = s.ndead;
    s.nnz = nnz;
    f.log_open = s.log_pos; f.log_n = s.log_n;
    vrg_finalize_core(s, n_in, n_out, use_tab, f.tr);
    s.apply_pending = 1; s.ap_n = f.places; s.fr_base = fr_base; s.fr_n = fr_n;
    if (c.log_rec) { s.log_pos += f.log_n; s.log_nsw += 1u; }     // (the state's words; vrg_log_sweep files the header)
}
// (s: the closed state, its `iter` the sweep's number; the state itself is stored by the caller - whole, or all but its live counters)
VRG_HD void vrg_fuse_close_store(const VrgCtx& c, const VrgState& s, const VrgFuseClosed& f) {
    const int64_t k = (int64_t)s.iter;
    vrg_store_i64(&c.exp_ring[2 * (k % VRG_RING)], f.n_in); vrg_store_i64(&c.exp_ring[2 * (k % VRG_RING) + 1], f.n_out);
    c.inc[VC_NIN] = f.n_in; c.inc[VC_NOUT] = f.n_out;
    c.nchg[k & 1] = f.places;                             // (filed by the deferred apply: a change sits at its voxel's place)
    vrg_finalize_store(c, s, f.tr);
    vrg_log_sweep(c, k, f.log_open, s.log_nsw - (c.log_rec ? 1u : 0u), f.log_n, f.n_in, f.n_out, f.tr);
}
VRG_HD void vrg_fuse_close(const VrgCtx& c, VrgState s, int64_t n_in, int64_t n_out, uint32_t nnz, bool use_tab) {
    VrgFuseClosed f;
    vrg_fuse_close_core(c, s, n_in, n_out, nnz, use_tab, f);
    vrg_state_store(c.stg, s);
    vrg_fuse_close_store(c, s, f);
}

// ---- open-ended sweeps ------------------------------------------------------------------------------------------------------------
// The closing workgroup of k_sweep is a serial tail of the band chain: a ticket, a round trip for what the others added up, the state
// out, the kernel's end - 4.5 us of a 29-us trip (in-kernel stamps).  An OPEN-ENDED sweep stops at its commit:
// nobody takes a ticket, nobody closes.  The state then holds what the sweep ran on plus the sums its workgroups added (VrgState::open
// = 1), its per-level counters stay where they are, and the NEXT trip's k_band - which loads the state anyway - derives the closed state
// in every workgroup (vrg_fuse_close_core: arithmetic on registers), its pool workgroups list the touched levels from the counters into
// LDS themselves, and ONE thread files the closed state, the expected sizes, the trace record and the change log's header
// (vrg_fuse_persist).  Two rules keep that free of races without a fence:
//  * a state is never filed into the buffer that workgroups of the same kernel still read: fused trips swap two state buffers
//    (VrgCtx::stb) - k_band reads one and files into the other, where its decisions already count flips and ties (those four live words
//    are not filed: k_sweep sets them up for the next trip, vrg_fuse_prepare_other);
//  * the per-level counters come in two sets by sweep parity: the set an open-ended sweep filled is read by the next k_band and zeroed
//    by the k_sweep after it, which fills the other one (vrg_fuse_zero_other_levels).
// The host only ever meets closed states: the last trip of every batch closes its sweep in the old way (and so does a trip that is
// followed by the memo kernel, or one on a level table too large for a workgroup's LDS list).
VRG_HD void vrg_state_store_but_live(VrgState* dst, const VrgState& w) {       // all words but the live line (nf, ties, near_ties, error: vrg_types.h)
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&w); uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    // (the words in front of the padding only, the loop unrolled: a loop the compiler keeps indexes the caller's by-value state - which then lives in scratch)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (unsigned i = 0; i < offsetof(VrgState, pad_live) / 4; i++) d[i] = src[i];
    if (w.error) vrg_store_i32(&dst->error, w.error);
}
// k_band's one filing thread.  s: the state this trip works on (closed: as found, or derived from an open-ended sweep: f, was_open).
// The label bytes of that sweep are applied by this very kernel, so the filed state says so.
// (n_in / n_out: the region sizes that go with s - as read beside the state, or derived)
VRG_HD void vrg_fuse_persist(const VrgCtx& c, const VrgState& s, const VrgFuseClosed& f, bool was_open, int64_t n_in, int64_t n_out) {
    if (c.st == c.stg && !was_open) return;               // (in place, nothing derived: the state is where it belongs)
    if (c.inc != c.inc_in) { c.inc[VC_NIN] = n_in; c.inc[VC_NOUT] = n_out; }     // (the sizes swap buffers with the state)
    VrgState w = s;
    w.apply_pending = 0; w.fr_n = 0;
    if (c.st == c.stg) { w.nf = 0; vrg_state_store(c.stg, w); }         // (in place - the sequential test model: the sweep's flips are consumed, the decisions count from zero)
    else vrg_state_store_but_live(c.stg, w);
    if (was_open) vrg_fuse_close_store(c, s, f);
}
// k_sweep, before anything else: the NEXT trip's k_band will count its flips and ties into the other buffer
VRG_HD void vrg_fuse_prepare_other(const VrgCtx& c, const VrgState& s0) {
    if (!c.st_other || c.st_other == c.stg) return;
    c.st_other->nf = 0; c.st_other->ties = s0.ties; c.st_other->near_ties = s0.near_ties; c.st_other->error = s0.error;
}
// ... and the per-level counters of the sweep before (parity p ^ 1) go back to zero: workgroup wg of nwg its stretch of the levels
VRG_HD void vrg_fuse_zero_other_levels(const VrgCtx& c, int p, uint32_t wg, uint32_t nwg, uint32_t t, uint32_t nt) {
    const uint32_t per = (c.L + nwg - 1u) / nwg, l0 = wg * per, l1 = l0 + per < c.L ? l0 + per : c.L;
    for (uint32_t l = l0 + t; l < l1; l += nt) { c.dInS[p ^ 1][l] = 0; c.dOutS[p ^ 1][l] = 0; c.dConvS[p ^ 1][l] = 0; }
}

// ---- what the fused sweep left for the next trip's k_band (sweep k = the state's iter: already counted)
// (place i of the marked list, its three fields fetched by the caller - several places in one batch)
VRG_HD void vrg_deferred_apply_vals(const VrgCtx& c, uint32_t i, int k, uint32_t idx, uint8_t old, uint8_t nw) {
    const int p = k & 1;
    if (idx == VRG_NONE) { c.chg_dw[p][i] = VRG_NOCHG; return; }
    c.lab[0][idx] = nw;                                    // (clean: no L / P / mark bits - the fused sweep never wrote any)
    const uint32_t a = vrg_cls_of(old), b = vrg_cls_of(nw);
    if (a == b) { c.chg_dw[p][i] = VRG_NOCHG; return; }
    uint32_t dw, sh; vrg_cls_pos(idx, dw, sh);
    const uint32_t x = (a ^ b) << sh;
    if (a == 0u) {
        const uint32_t bit = 1u << ((idx >> 10) & 31u);
        vrg_list_unit(c, idx, p, bit);
    }
    vrg_atomic_xor(&c.clsb[p][dw], x);
    c.chg_dw[p][i] = dw; c.chg_x[p][i] = x;
}
VRG_HD void vrg_deferred_apply(const VrgCtx& c, uint32_t i, int k) { vrg_deferred_apply_vals(c, i, k, c.mk_idx[i], c.mk_old[i], c.mk_new[i]); }
VRG_HD void vrg_deferred_catchup(const VrgCtx& c, uint32_t i, int k) {   // change i of sweep k-1: class copy k & 1 sat that sweep out
    const int p = k & 1;
    const uint32_t dw = c.chg_dw[p ^ 1][i], x = c.chg_x[p ^ 1][i];
    if (dw != VRG_NOCHG) vrg_atomic_xor(&c.clsb[p][dw], x);
}
VRG_HD uint32_t vrg_deferred_catchup_count(const VrgCtx& c, int k) { const uint32_t n = c.nchg[(k & 1) ^ 1]; return n < c.mcap ? n : c.mcap; }
VRG_HD void vrg_deferred_free(const VrgCtx& c, const VrgState& s, uint32_t j) { c.freel[s.fr_base + j] = c.dead[j]; }
// one caller, once all of the above has reached memory: the list of sweep k-1 is consumed, the labels of sweep k are in place -
// its dense pass is due
VRG_HD void vrg_deferred_done(const VrgCtx& c, int k) {
    c.nchg[(k & 1) ^ 1] = 0;
    if (c.st == c.stg) { c.stg->apply_pending = 0; c.stg->fr_n = 0; }     // (fused trips file the state into the other buffer: vrg_fuse_persist says so there)
    vrg_dense_none_step(c, k);
    vrg_drain(); vrg_store_i64(&c.gate[VG_REQ], (int64_t)k);
}

// ------------------------------------------------------------------ init mode (:129-155)
// labels are pure morphology (verified against the reference): seed with a non-seed neighbour -> 1,
// non-seed next to a seed -> 2 (4 -> 3 -> 2 included), list orders: inner = seeds in np.where order,
// outer = first-seen order keyed (lex rank of the first seed that touches it, k).
VRG_HD void vrg_item_init_voxel(const VrgCtx& c, uint32_t idx) {
    uint8_t* lab = c.lab[0];
    uint8_t cb = lab[idx];
    if (cb & VB_OOB) return;
    VrgState& s = *c.st;
    if (cb & VB_S) {
        uint64_t lex = vrg_lex(c, idx);
        c.stamp[idx] = lex;
        vrg_atomic_add(&s.nseed, 1u);
        bool bnd = false;
        for (int k = 0; k < 27; k++) {
            if (k == 13) continue;
            uint8_t m = lab[(int64_t)idx + vrg_off(c, k)];
            if (!(m & (VB_S | VB_OOB))) bnd = true;
        }
        if (bnd) {
            uint32_t p = vrg_atomic_add(&s.ninit_in, 1u);     // inner entries staged from the bottom
            if (p < c.bcap) { c.init_key[p] = lex; c.init_idx[p] = idx; } else s.error = 1;
            vrg_or_byte(lab, idx, VB_B);
        }
        return;
    }
    // (nearly every voxel has no seed among its 26 neighbours: nine 4-byte rows say so - 26 byte loads per voxel made this
    // kernel 14.5 ms of a 40-ms vrg_init at 880x880x640.  S bits do not change during init.)
    uint32_t any = 0;
    for (int j = 0; j < 9; j++) {
        uint32_t w; __builtin_memcpy(&w, lab + ((int64_t)idx + ((j % 3 - 1) * c.PY + (j / 3 - 1)) * c.PX - 1), 4);
        any |= w;
    }
    if (!(any & (0x010101u * VB_S))) return;              // bytes x-1 .. x+1 of the nine (dy, dz) rows
    uint64_t best = ~0ull; int bk = 0;
    for (int k = 0; k < 27; k++) {
        if (k == 13) continue;
        uint32_t m = (uint32_t)((int64_t)idx + vrg_off(c, k));
        if (lab[m] & VB_S) {
            uint64_t lx = vrg_lex(c, m);
            if (lx < best) { best = lx; bk = 26 - k; }
        }
    }
    if (best == ~0ull) return;
    // outer entries are staged from the top of the init arrays downwards
    uint32_t p = vrg_atomic_add(&s.ninit_out, 1u);
    if (p < c.bcap) { c.init_key[c.bcap - 1 - p] = best * 27u + (uint64_t)bk; c.init_idx[c.bcap - 1 - p] = idx; }
    else s.error = 1;
    // B on, X off; neighbours only ever read the S bit, so a plain byte store is safe
    lab[idx] = VB_B;
}

// slot e of the initial pool = position e of concat(innerBnd, outerBnd) (p_idx holds the sorted voxels): key = position
VRG_HD void vrg_item_init_entry(const VrgCtx& c, uint32_t e) {
    uint32_t idx = c.p_idx[e];
    c.p_lev[e] = vrg_voxel_level(c, idx);
    c.p_key[e] = e;
    c.p_flag[e] = (uint8_t)(PF_ALIVE | (e < c.st->ni ? PF_INNER : 0));     // (init computes its densities at once: not pending)
    c.p_ip[e] = 0; c.p_op[e] = 0; c.p_err[e] = 0;
    c.fresh[e] = e;
    c.vent[idx] = e;
}

VRG_HD void vrg_item_hist_voxel(const VrgCtx& c, uint32_t idx) {
    uint8_t b = c.lab[0][idx];
    if (b & (VB_OOB | VB_X)) return;
    uint32_t lev = vrg_voxel_level(c, idx);
    if (b & VB_S) vrg_atomic_add(&c.hin[lev], 1); else vrg_atomic_add(&c.hout[lev], 1);
}
