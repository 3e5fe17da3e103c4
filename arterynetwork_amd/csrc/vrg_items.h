// vrg_items.h - per-item device functions of the VRG sweep (one band entry, one listed flip, one voxel).
//
// These restate variationalRegionGrowing.py's SEQUENTIAL update() (:124-261) as order-independent
// local rules so that every item can run in parallel.  Derivation (validated against the oracle and
// the reference goldens in tests/):
//
//  * flip list = concat(innerBnd, outerBnd)[mask] (:48,:88,:111): all flip-outs (label 1) precede all
//    flip-ins (label 2).  Only the ORDER of that list matters, and only between 26-neighbours, so the
//    position e of a voxel in concat(innerBnd, outerBnd) (its band entry index) serves as its "rank".
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
//  * list order after the sweep (:257-258): survivors keep their order; appended in order:
//    inner: phase-A promotions keyed (rank of first flip-out neighbour, k) then applied flip-ins by rank;
//    outer: flip-outs still labelled 2 by rank, then phase-B promotions keyed (rank of first applied
//    flip-in neighbour, k); k = position of the promoted voxel in get_neighbours(promoter) (:263-282).
//  * densities (:232-255): incremental correction for entries that stayed in the band for the whole
//    sweep, exact recomputation for entries that (re-)entered it (newInnerBndList/newOuterBndList).
#pragma once
#include <math.h>
#include "vrg_types.h"

// ------------------------------------------------------------------ backend shims
#if defined(__HIP_DEVICE_COMPILE__)
VRG_HD uint32_t vrg_atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
VRG_HD int32_t vrg_atomic_add(int32_t* p, int32_t v) { return atomicAdd(p, v); }
VRG_HD uint32_t vrg_atomic_or(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
VRG_HD void vrg_atomic_add64(int64_t* p, int64_t v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }
VRG_HD void vrg_atomic_xor(uint32_t* p, uint32_t v) { atomicXor(p, v); }
VRG_HD uint8_t vrg_load_coherent(const uint8_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
VRG_HD uint32_t vrg_atomic_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
VRG_HD int32_t vrg_atomic_add(int32_t* p, int32_t v) { int32_t o = *p; *p = o + v; return o; }
VRG_HD uint32_t vrg_atomic_or(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
VRG_HD void vrg_atomic_add64(int64_t* p, int64_t v) { *p += v; }
VRG_HD void vrg_atomic_xor(uint32_t* p, uint32_t v) { *p ^= v; }
VRG_HD uint8_t vrg_load_coherent(const uint8_t* p) { return *(const volatile uint8_t*)p; }
#endif

// OR bits into one label byte without disturbing concurrent ORs into its neighbours
VRG_HD void vrg_or_byte(uint8_t* lab, uint32_t idx, uint8_t bits) {
    uint32_t* w = (uint32_t*)(lab + (idx & ~3u));
    vrg_atomic_or(w, (uint32_t)bits << (8 * (idx & 3u)));
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
// the 27 label bytes around idx in one go: straight-line loads, so that the device issues them back to back and
// waits once (a loop with tests between the loads waits for every byte: 26 dependent L2 round trips)
VRG_HD void vrg_load_nbrs(const VrgCtx& c, const uint8_t* lab, uint32_t idx, uint8_t nb[27]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 27; k++) nb[k] = lab[(int64_t)idx + vrg_off(c, k)];
}

VRG_HD uint8_t vrg_enc(uint8_t ext) {     // reference label -> byte
    return ext == 0 ? VB_S : ext == 1 ? (VB_S | VB_B) : ext == 2 ? VB_B : ext == 4 ? VB_X : 0;
}
VRG_HD uint8_t vrg_dec(uint8_t b) {       // byte -> reference label
    if (b & VB_S) return (b & VB_B) ? 1 : 0;
    if (b & VB_B) return 2;
    return (b & VB_X) ? 4 : 3;
}

VRG_HD double vrg_kern(const VrgCtx& c, double d) { return c.A * exp(-0.5 * c.H * (d * d)); }   // :154

VRG_HD uint32_t vrg_level_of(const VrgCtx& c, double v) {   // index of v in the sorted distinct values
    uint32_t lo = 0, hi = c.L - 1;
    while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (c.lev[m] < v) lo = m + 1; else hi = m; }
    return lo;
}

// the per-level delta counters (dIn, dOut, dConv) exist twice: sweep iter+1 counts into the copy of parity iter & 1,
// so that the kernel that consumes them can leave clearing to the sweep after (see k_levels_tab_scan)
VRG_HD uint32_t vrg_delta_off(const VrgCtx& c) { return (c.st->iter & 1) ? c.L : 0u; }
// level index of a voxel's intensity: stored (16-bit mode) or looked up in the sorted level table
VRG_HD uint32_t vrg_voxel_level(const VrgCtx& c, uint32_t idx) {
    return c.lev16 ? (uint32_t)c.lev16[idx] : vrg_level_of(c, (double)c.I[idx]);
}

// ------------------------------------------------------------------ decide (:79-88) + listing
// One item per band entry.  A flip is listed at once: L bit (+P for flip-outs, which are always
// applied), stamp = (sweep, entry index) and an unordered append to the flip list.
VRG_HD void vrg_decide_core(const VrgCtx& c, const VrgState& s, uint32_t e, double ip, double op) {
    int cur = s.iter & 1;
    const uint32_t idx = c.b_idx[cur][e];
    const int64_t n_in = c.inc[VC_NIN], n_out = c.inc[VC_NOUT];
    double inN = ip / (double)n_in;                       // :81
    double outN = op / (double)n_out;                     // :82
    bool ge = inN >= outN;
    bool inner = e < s.ni;
    bool flip = inner != ge;                              // :87 xor(segmentedMap, inner >= outer)
    c.e_flag[e] = flip ? 1 : 0; c.e_res[e] = 0; c.e_mask[e] = 0; c.e_new[e] = 0;
    if (!flip) return;
    uint32_t q = vrg_atomic_add(&c.st->nf, 1u);
    if (s.time_up || n_in >= s.maxSegmentSize) return;    // :97 / :101 fire before update(): count only
    if (q >= c.fcap) { c.st->error = 2; return; }
    c.flist[q] = e; c.fidx[q] = idx;
    vrg_or_byte(c.lab[0], idx, (uint8_t)(VB_L | (inner ? VB_P : 0)));
    c.stamp[idx] = ((uint64_t)(uint32_t)(s.iter + 1) << 32) | e;
}
// skip_pending (device): an entry whose densities are still to be computed exactly is decided by the wave that
// computes them (exact half of k_decide_exact), not here
VRG_HD void vrg_item_decide(const VrgCtx& c, uint32_t e, bool skip_pending) {
    const VrgState s = *c.st;                             // a copy (nf is only ever bumped atomically)
    if (s.iter >= s.iterMax) return;                      // while iterNum <= iterMax (:58)
    const double ip = c.b_ip[s.iter & 1][e], op = c.b_op[s.iter & 1][e];
    if (skip_pending && c.b_pend[s.iter & 1][e]) return;
    vrg_decide_core(c, s, e, ip, op);
}

// the stop tests in the reference's order, once every entry has decided (the host raises time_up)
VRG_HD int32_t vrg_stop_test(const VrgCtx& c) {
    const VrgState& s = *c.st;
    if (s.iter >= s.iterMax) return VRG_STOP_ITERMAX;                    // :58
    if (s.nf == 0) return VRG_STOP_CONVERGED;                            // :91
    if (s.time_up) return VRG_STOP_TIME;                                 // :97
    if (c.inc[VC_NIN] >= s.maxSegmentSize) return VRG_STOP_SIZE;           // :101
    return 0;
}

// flip-ins: label after phase A (:183-190) decides whether the flip is applied at once
VRG_HD void vrg_item_prepass(const VrgCtx& c, uint32_t r) {
    VrgState& s = *c.st;
    const uint32_t e = c.flist[r], idx = c.fidx[r];
    if (e < s.ni) return;
    const uint8_t* lab = c.lab[0];
    bool nFO = false, nSegA = false;
    uint8_t nb[27]; vrg_load_nbrs(c, lab, idx, nb);
    for (int k = 0; k < 27; k++) {
        if (k == 13) continue;
        uint8_t m = nb[k];
        if (m & VB_S) { if (m & VB_L) nFO = true; else nSegA = true; }
    }
    if (nFO && !nSegA) c.pend[vrg_atomic_add(&s.npend, 1u)] = e;   // dropped to 3: skipped unless re-promoted
    else vrg_or_byte(c.lab[0], idx, VB_P);
}

// one relaxation of the skip rule: applied if an applied flip-in neighbour of smaller rank exists
// returns 0: still skipped, 1: found applied (by whoever), 2: applied by this call
VRG_HD int vrg_item_fix(const VrgCtx& c, uint32_t j) {
    uint32_t e = c.pend[j], idx = c.b_idx[c.st->iter & 1][e];
    uint8_t* lab = c.lab[0];
    if (vrg_load_coherent(lab + idx) & VB_P) return 1;
    for (int k = 0; k < 27; k++) {
        if (k == 13) continue;
        uint32_t m = (uint32_t)((int64_t)idx + vrg_off(c, k));
        uint8_t mb = vrg_load_coherent(lab + m);
        if (!(mb & VB_S) && (mb & VB_L) && (mb & VB_P) && (uint32_t)c.stamp[m] < e) {
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
#define VB_M 128
VRG_HD void vrg_item_scatter_marks(const VrgCtx& c, uint32_t r, uint32_t p) {
    if (p >= 125) return;
    uint8_t* lab = c.lab[0];
    uint32_t idx = c.fidx[r];
    int dx = (int)(p % 5) - 2, dy = (int)((p / 5) % 5) - 2, dz = (int)(p / 25) - 2;
    bool ring1 = dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
    int64_t m = (int64_t)idx + (dz * c.PY + dy) * c.PX + dx;   // may be -1,-2 (guard bytes) at voxel (0,0,0)
    uint8_t mb = lab[m];
    if (mb & (VB_OOB | VB_M)) return;
    if (!ring1 && !(mb & VB_X)) return;
    uint32_t sh = 8 * ((uint32_t)m & 3u);
    uint32_t old = vrg_atomic_or((uint32_t*)(lab + ((uint32_t)m & ~3u)), (uint32_t)VB_M << sh);
    if (!((old >> sh) & VB_M)) {
        uint32_t q = vrg_atomic_add(&c.st->nmk, 1u);
        if (q < c.mcap) c.mk_idx[q] = (uint32_t)m; else c.st->error = 4;
    }
}
// sparse relabel, phase 1: new byte of every marked voxel from the OLD labels (nothing is written to
// the label volume yet, so all stencil reads see the pre-sweep state)
VRG_HD uint8_t vrg_sweep_core(const VrgCtx& c, const uint8_t* lab, uint32_t idx, uint8_t cb);
#define VE_VALID 0x80          // e_new: the relabel visited the entry's voxel (new bytes never carry bit 7 = VB_M)
VRG_HD void vrg_item_relabel(const VrgCtx& c, uint32_t i) {
    uint32_t idx = c.mk_idx[i];
    const uint32_t e = c.vent[idx];                        // meaningful only for a band voxel (B bit)
    const uint8_t cb = c.lab[0][idx];
    const uint8_t nw = vrg_sweep_core(c, c.lab[0], idx, cb);
    c.mk_new[i] = nw;
    if (cb & VB_B) c.e_new[e] = (uint8_t)(nw | VE_VALID);  // the entry's survivor test need not wait for k_apply
}
// phase 2: write the new bytes (this also clears the L / P / mark bits)
// class of a label for the region statistics (:113-116): 1 inner (S), 2 outer (neither S nor excluded), 0 neither
VRG_HD uint32_t vrg_cls_of(uint8_t b) { return (b & VB_S) ? 1u : ((b & (VB_X | VB_OOB)) ? 0u : 2u); }
// where voxel idx keeps its two class bits (layout: VrgCtx::cls)
VRG_HD void vrg_cls_pos(uint32_t idx, uint32_t& dw, uint32_t& sh) {
    uint32_t o = idx & 1023u;
    dw = ((idx >> 10) << 6) | ((o & 255u) >> 2);
    sh = 2u * (((o >> 8) << 2) | (o & 3u));
}
// a label byte changes during sweep iter+1: keep the region sizes and that sweep's copy of the class bits in step,
// and note the change for the other copy
VRG_HD void vrg_count_change(const VrgCtx& c, uint32_t idx, uint8_t old, uint8_t nw) {
    uint32_t a = vrg_cls_of(old), b = vrg_cls_of(nw);
    if (a == b) return;
    const int p = (c.st->iter + 1) & 1;
    uint32_t dw, sh; vrg_cls_pos(idx, dw, sh);
    const uint32_t x = (a ^ b) << sh;
    vrg_atomic_xor(&c.clsb[p][dw], x);
    uint32_t q = vrg_atomic_add(&c.nchg[p], 1u);
    if (q < c.ccap) { c.chg_dw[p][q] = dw; c.chg_x[p][q] = x; } else c.st->error = 7;
    int din = (int)(b == 1u) - (int)(a == 1u), dout = (int)(b == 2u) - (int)(a == 2u);
    if (din) vrg_atomic_add64(&c.inc[VC_NIN], din);
    if (dout) vrg_atomic_add64(&c.inc[VC_NOUT], dout);
}
// change i of the sweep before: this sweep's copy of the class bits sat that sweep out
VRG_HD void vrg_item_catchup(const VrgCtx& c, uint32_t i) {
    const int p = (c.st->iter + 1) & 1;
    vrg_atomic_xor(&c.clsb[p][c.chg_dw[p ^ 1][i]], c.chg_x[p ^ 1][i]);
}
VRG_HD uint32_t vrg_catchup_count(const VrgCtx& c) { uint32_t n = c.nchg[((c.st->iter + 1) & 1) ^ 1]; return n < c.ccap ? n : c.ccap; }
VRG_HD void vrg_item_apply(const VrgCtx& c, uint32_t i) {
    uint32_t idx = c.mk_idx[i];
    uint8_t old = c.lab[0][idx], nw = c.mk_new[i];
    c.lab[0][idx] = (uint8_t)(nw & ~VB_F);                 // F is a note to the entry's survivor test (e_new), not a label bit
    vrg_count_change(c, idx, old, nw);
}
// one caller, after every label of sweep iter+1 is written and before anything of the next sweep: file the sizes
// that sweep's dense pass has to reproduce; the change list just consumed becomes the next sweep's
VRG_HD void vrg_post_apply(const VrgCtx& c) {
    const int64_t k = (int64_t)c.st->iter + 1;
    c.inc[VC_EXP + 2 * (k & 3)] = c.inc[VC_NIN]; c.inc[VC_EXP + 2 * (k & 3) + 1] = c.inc[VC_NOUT];
    c.nchg[(k & 1) ^ 1] = 0;
}
// init: class dword d from the labels (16 voxels), both copies
VRG_HD void vrg_item_cls_build(const VrgCtx& c, uint32_t d) {
    uint32_t base = ((d >> 6) << 10) | ((d & 63u) << 2), w = 0;
    for (uint32_t j = 0; j < 4; j++)
        for (uint32_t b = 0; b < 4; b++) {
            uint32_t idx = base + 256u * j + b;
            if (idx < c.PV) w |= vrg_cls_of(c.lab[0][idx]) << (2u * (4u * j + b));
        }
    c.clsb[0][d] = w; c.clsb[1][d] = w;
}
// one caller per applied sweep: the labels of sweep iter+1 are in place, a dense pass over them is due
VRG_HD void vrg_request_dense(const VrgCtx& c) { c.inc[VC_REQ] = (int64_t)c.st->iter + 1; }
VRG_HD bool vrg_dense_due(const VrgCtx& c) { return c.inc[VC_REQ] > c.dctl[VD_SEQ]; }
// the dense pass (number seq = passes closed + 1, reading clsb[seq & 1]) has the totals in c.dn: cross-check the sizes
// it had to reproduce, file the sums, close the pass
VRG_HD void vrg_dense_fin(const VrgCtx& c) {
    if (!vrg_dense_due(c)) return;
    int64_t seq = c.dctl[VD_SEQ] + 1;
    const VrgDense& d = *c.dn;
    if ((int64_t)d.n_in != c.inc[VC_EXP + 2 * (seq & 3)] || (int64_t)d.n_out != c.inc[VC_EXP + 2 * (seq & 3) + 1]) c.dctl[VD_ERR] = 5;
    if ((uint64_t)seq < c.trace_cap) { c.trace[seq].sum_in = d.sum_in; c.trace[seq].sum_out = d.sum_out; }
    c.dctl[VD_SEQ] = seq;
}
// init: the dense pass founds the incremental sizes
VRG_HD void vrg_init_counts(const VrgCtx& c) {
    c.inc[VC_NIN] = (int64_t)c.dn->n_in; c.inc[VC_NOUT] = (int64_t)c.dn->n_out; c.inc[VC_REQ] = 0;
    c.dctl[VD_SEQ] = 0; c.dctl[VD_ERR] = 0; c.nchg[0] = 0; c.nchg[1] = 0;
}

// ------------------------------------------------------------------ the relabel stencil for one voxel
// phase-B promotion (3 -> 2, :210-213): list key (first applied flip-in neighbour, k)
VRG_HD void vrg_promote_b(const VrgCtx& c, const uint8_t* nb, uint32_t idx) {
    uint32_t best = 0xffffffffu; int bk = 0;
    for (int k = 0; k < 27; k++) {
        if (k == 13) continue;
        uint32_t m = (uint32_t)((int64_t)idx + vrg_off(c, k));
        uint8_t mb = nb[k];
        if (!(mb & VB_S) && (mb & VB_P)) {
            uint32_t r = (uint32_t)c.stamp[m];
            if (r < best) { best = r; bk = 26 - k; }
        }
    }
    vrg_atomic_or(&c.e_mask[best], 1u << bk);
}
// phase-A promotion (0 -> 1, :194-196): list key (first flip-out neighbour, k)
VRG_HD void vrg_promote_a(const VrgCtx& c, const uint8_t* nb, uint32_t idx) {
    uint32_t best = 0xffffffffu; int bk = 0;
    for (int k = 0; k < 27; k++) {
        if (k == 13) continue;
        uint32_t m = (uint32_t)((int64_t)idx + vrg_off(c, k));
        uint8_t mb = nb[k];
        if ((mb & VB_S) && (mb & VB_L)) {
            uint32_t r = (uint32_t)c.stamp[m];
            if (r < best) { best = r; bk = 26 - k; }
        }
    }
    vrg_atomic_or(&c.e_mask[best], 1u << bk);
}

// Returns the voxel's byte after the sweep.  Side effects for the rare cases: e_res (listed flips),
// e_mask (promotions), dConv (4->3 inclusions).  Ranks (low stamp word) are band entry indices.  `lab` = this sweep's input labels (L/P bits set).
VRG_HD uint8_t vrg_sweep_core(const VrgCtx& c, const uint8_t* lab, uint32_t idx, uint8_t cb) {
    bool nSegA = false, nFO = false, nAP = false, nNonSegB = false, nListed = false;
    uint8_t nb[27]; vrg_load_nbrs(c, lab, idx, nb);
    for (int k = 0; k < 27; k++) {
        if (k == 13) continue;
        uint8_t m = nb[k];
        if (m & VB_OOB) continue;                     // neighbour does not exist (:278-280)
        bool mS = m & VB_S, mL = m & VB_L, mP = m & VB_P;
        bool segA = mS && !mL, ap = !mS && mP;
        nSegA |= segA; nFO |= (mS && mL); nAP |= ap; nNonSegB |= !(segA || ap); nListed |= mL;
    }
    if (cb & VB_S) {
        if (cb & VB_L) {                              // flip-out (:170-175), always applied
            uint32_t r = (uint32_t)c.stamp[idx];
            bool to3 = false;
            if (!nSegA)                               // re-examined by a later flip-out neighbour? (:183-190)
                for (int k = 0; k < 27 && !to3; k++) {
                    if (k == 13) continue;
                    uint32_t m = (uint32_t)((int64_t)idx + vrg_off(c, k));
                    uint8_t mb = nb[k];
                    if ((mb & VB_S) && (mb & VB_L) && (uint32_t)c.stamp[m] > r) to3 = true;
                }
            if (!to3) { c.e_res[r] = FR_WRITTEN | 2; return VB_B; }          // stays 2, carried to the outer list
            if (nAP) {                                                      // 3 -> 2 again (:210-213): fresh
                c.e_res[r] = FR_WRITTEN | 2 | FR_FRESH;
                vrg_promote_b(c, nb, idx);
                return VB_B;
            }
            c.e_res[r] = FR_WRITTEN | 3;
            return 0;
        }
        bool is1 = (cb & VB_B) || nFO;                // label after phase A (:194)
        if (!is1) return VB_S;
        if (nAP && !nNonSegB) return VB_S;            // 1 -> 0 (:223-228)
        if (!(cb & VB_B)) vrg_promote_a(c, nb, idx); // newly on the inner boundary
        return VB_S | VB_B;
    }
    if (cb & VB_B) {
        if ((cb & VB_L) && (cb & VB_P)) {             // applied flip-in (:198-204)
            uint32_t r = (uint32_t)c.stamp[idx];
            bool to0 = false;
            if (!nNonSegB)                            // re-examined by a later applied flip-in nbr? (:219-228)
                for (int k = 0; k < 27 && !to0; k++) {
                    if (k == 13) continue;
                    uint32_t m = (uint32_t)((int64_t)idx + vrg_off(c, k));
                    uint8_t mb = nb[k];
                    if (!(mb & VB_S) && (mb & VB_P) && (uint32_t)c.stamp[m] > r) to0 = true;
                }
            bool fresh = nFO && !nSegA;               // had dropped to 3 in phase A: exact density (:212,:251)
            c.e_res[r] = (uint8_t)(FR_WRITTEN | (to0 ? 0 : 1) | (fresh ? FR_FRESH : 0));
            return to0 ? (uint8_t)VB_S : (uint8_t)(VB_S | VB_B);
        }
        bool to3 = nFO && !nSegA;                     // :183-190
        uint8_t out, res;
        if (!to3) { out = VB_B; res = 2; }
        else if (nAP) { out = VB_B | VB_F; res = 2 | FR_FRESH; vrg_promote_b(c, nb, idx); }
        else { out = 0; res = 3; }
        if (cb & VB_L) c.e_res[(uint32_t)c.stamp[idx]] = (uint8_t)(FR_WRITTEN | res);   // skipped flip-in
        return out;
    }
    // labels 3 and 4
    bool conv = false;
    if (cb & VB_X) {
        conv = nListed;                               // 1-ring of any listed flip (:166-168)
        if (!conv)                                    // 2-ring of any applied flip (:177-179,:206-208)
            for (int dz = -2; dz <= 2 && !conv; dz++)
                for (int dy = -2; dy <= 2 && !conv; dy++)
                    for (int dx = -2; dx <= 2; dx++) {
                        uint8_t mb = lab[(int64_t)idx + (dz * c.PY + dy) * c.PX + dx];
                        if ((mb & VB_P) && !(mb & VB_OOB)) { conv = true; break; }
                    }
        if (conv) vrg_atomic_add(&c.dConv[vrg_delta_off(c) + vrg_voxel_level(c, idx)], 1u);   // addedPoints (:235)
    }
    if (nAP) { vrg_promote_b(c, nb, idx); return VB_B; }   // 3 -> 2 (:210-213)
    return (uint8_t)(((cb & VB_X) && !conv) ? VB_X : 0);
}

// ------------------------------------------------------------------ after the relabel
// Rebuild count array (length 3n, exclusive-scanned into new list positions), n = ni + no, j = e - ni:
//   A0[e]  inner survivors          A1[e]  voxels promoted to 1 by flip-out e     A2[j]  flip-ins now labelled 1
//   B0[j]  outer survivors          B1[e]  flip-outs still labelled 2 (carried)   B2[j]  voxels promoted to 2 by flip-in j
// laid out [A0|A1|A2|B0|B1|B2]: exactly the append order of innerBndList / outerBndList (:257-258).
VRG_HD uint32_t vrg_slot_A0(const VrgState&, uint32_t e) { return e; }
VRG_HD uint32_t vrg_slot_A1(const VrgState& s, uint32_t e) { return s.ni + e; }
VRG_HD uint32_t vrg_slot_A2(const VrgState& s, uint32_t j) { return 2 * s.ni + j; }
VRG_HD uint32_t vrg_slot_B0(const VrgState& s, uint32_t j) { return 2 * s.ni + s.no + j; }
VRG_HD uint32_t vrg_slot_B1(const VrgState& s, uint32_t e) { return 2 * s.ni + 2 * s.no + e; }
VRG_HD uint32_t vrg_slot_B2(const VrgState& s, uint32_t j) { return 3 * s.ni + 2 * s.no + j; }

// per old band entry: survivor test, and for listed flips the density bookkeeping sets (:232-233) and
// the class histograms
VRG_HD void vrg_item_entry_post(const VrgCtx& c, uint32_t e) {
    const VrgState s = *c.st;                                 // a copy: no reloads after the stores below
    int cur = s.iter & 1;
    const uint8_t* lab = c.lab[0];
    // independent loads first (one round trip), then the one that depends on idx
    uint32_t idx = c.b_idx[cur][e], lev = c.b_lev[cur][e], mask = c.e_mask[e];
    uint8_t res = c.e_res[e], en = c.e_new[e];
    bool inner = e < s.ni, flag = c.e_flag[e] != 0;
    // the voxel's byte after the sweep: from the relabel if it visited the voxel, else the label it kept (k_apply,
    // which may run beside this kernel, only rewrites visited voxels)
    uint8_t nb = (en & VE_VALID) ? (uint8_t)(en & ~VE_VALID) : lab[idx];
    uint8_t fin = res & FR_FINAL;
    bool fresh = res & FR_FRESH;
    if (!flag) mask = 0u;
    if (flag) {
        if (!(res & FR_WRITTEN)) c.st->error = 3;             // a listed flip the relabel never visited
        if (fin == 1) vrg_atomic_add(&c.dIn[vrg_delta_off(c) + lev], 1u);        // innerAdded: listed flips labelled 1 at the end
        else if (fin == 2) vrg_atomic_add(&c.dOut[vrg_delta_off(c) + lev], 1u);  // outerAdded: ... labelled 2
        if (inner) { vrg_atomic_add(&c.hin[lev], -1); vrg_atomic_add(&c.hout[lev], 1); }             // flip-out
        else if (fin <= 1) { vrg_atomic_add(&c.hin[lev], 1); vrg_atomic_add(&c.hout[lev], -1); }     // applied flip-in
    }
    bool surv;
    if (inner) {
        surv = !flag && (nb & VB_LABEL) == (VB_S | VB_B);
        c.scan[vrg_slot_A0(s, e)] = surv ? 1u : 0u;
        c.scan[vrg_slot_A1(s, e)] = (uint32_t)__builtin_popcount(mask);
        c.scan[vrg_slot_B1(s, e)] = (flag && fin == 2 && !fresh) ? 1u : 0u;
    } else {
        uint32_t j = e - s.ni;
        surv = !flag && (nb & VB_LABEL) == VB_B && !(nb & VB_F);
        c.scan[vrg_slot_B0(s, j)] = surv ? 1u : 0u;
        c.scan[vrg_slot_A2(s, j)] = (flag && fin == 1) ? 1u : 0u;
        c.scan[vrg_slot_B2(s, j)] = (uint32_t)__builtin_popcount(mask);
    }
    c.e_surv[e] = surv;
}

// density correction of one intensity value (:236-247)
VRG_HD void vrg_corrections(const VrgCtx& c, double v, double& ic, double& oc, double& ac) {
    const VrgState& s = *c.st;
    double a = 0, b = 0, d = 0;
    for (uint32_t i = 0; i < s.nnz; i++) {
        double k = vrg_kern(c, c.nz_val[i] - v);
        a += (double)c.nz_cin[i] * k; b += (double)c.nz_cout[i] * k; d += (double)c.nz_cconv[i] * k;
    }
    ic = a; oc = b; ac = d;
}
VRG_HD void vrg_apply_correction(const VrgCtx& c, uint32_t lev, double& ip, double& op) {
    double ic, oc, ac;
    if (c.st->use_tab) { ic = c.tabC[3 * (size_t)lev]; oc = c.tabC[3 * (size_t)lev + 1]; ac = c.tabC[3 * (size_t)lev + 2]; }
    else vrg_corrections(c, c.lev[lev], ic, oc, ac);
    ip += ic; ip -= oc;             // :243-244
    op -= ic; op += oc; op += ac;   // :245-247
}

VRG_HD void vrg_new_fresh(const VrgCtx& c, int nx, uint32_t pos, uint32_t idx, uint32_t lev) {
    if (pos >= c.bcap) { c.st->error = 1; return; }
    c.b_idx[nx][pos] = idx; c.b_lev[nx][pos] = lev; c.b_ip[nx][pos] = 0; c.b_op[nx][pos] = 0; c.b_pend[nx][pos] = 1;
    c.vent[idx] = pos;
    c.fresh[vrg_atomic_add(&c.st->nfresh, 1u)] = pos;
}

// per old band entry: survivors and carried flips copy themselves to their new position with the
// incremental correction; a flip that left the band during the sweep re-enters as a fresh entry
VRG_HD void vrg_item_scatter_entry(const VrgCtx& c, uint32_t e) {
    const VrgState s = *c.st;                                 // a copy: no reloads after the stores below
    int cur = s.iter & 1, nx = cur ^ 1;
    bool inner = e < s.ni;
    uint32_t j = e - s.ni;
    // independent loads first (one round trip); the survivor's slot is known without looking anything up
    const uint8_t surv = c.e_surv[e], flag = c.e_flag[e], res = c.e_res[e];
    const uint32_t lev = c.b_lev[cur][e], idx = c.b_idx[cur][e];
    double ip = c.b_ip[cur][e], op = c.b_op[cur][e];
    uint32_t pos = c.scan[inner ? vrg_slot_A0(s, e) : vrg_slot_B0(s, j)];
    bool fresh = false;
    if (!surv) {
        if (!flag) return;
        uint8_t fin = res & FR_FINAL;
        fresh = res & FR_FRESH;
        if (inner) { if (!(fin == 2 && !fresh)) return; pos = c.scan[vrg_slot_B1(s, e)]; }
        else { if (fin != 1) return; pos = c.scan[vrg_slot_A2(s, j)]; }
    }
    if (fresh) { vrg_new_fresh(c, nx, pos, idx, lev); return; }
    if (pos >= c.bcap) { c.st->error = 1; return; }
    vrg_apply_correction(c, lev, ip, op);
    c.b_idx[nx][pos] = idx; c.b_lev[nx][pos] = lev; c.b_ip[nx][pos] = ip; c.b_op[nx][pos] = op; c.b_pend[nx][pos] = 0;
    c.vent[idx] = pos;
}

// the voxels a listed flip promoted (item k = neighbour k of the flip, :263-282 order): fresh entries
VRG_HD void vrg_item_scatter_promo(const VrgCtx& c, uint32_t r, uint32_t k) {
    if (k >= 27) return;
    const VrgState& s = *c.st;
    int cur = s.iter & 1, nx = cur ^ 1;
    uint32_t e = c.flist[r], fi = c.fidx[r];
    uint32_t mask = c.e_mask[e];
    if (!(mask & (1u << k))) return;
    uint32_t slot = e < s.ni ? vrg_slot_A1(s, e) : vrg_slot_B2(s, e - s.ni);
    uint32_t pos = c.scan[slot] + (uint32_t)__builtin_popcount(mask & ((1u << k) - 1u));
    uint32_t m = (uint32_t)((int64_t)fi + vrg_off(c, (int)k));
    vrg_new_fresh(c, nx, pos, m, vrg_voxel_level(c, m));
}

// exact densities over the whole inner / outer regions (:152-155, :252-255), regrouped by level
VRG_HD void vrg_exact_serial(const VrgCtx& c, int par, uint32_t pos) {
    double v = c.lev[c.b_lev[par][pos]];
    double si = 0, so = 0;
    for (uint32_t l = 0; l < c.L; l++) {
        int32_t a = c.hin[l], b = c.hout[l];
        if (!(a | b)) continue;
        double k = vrg_kern(c, c.lev[l] - v);
        si += (double)a * k; so += (double)b * k;
    }
    c.b_ip[par][pos] = si; c.b_op[par][pos] = so;
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

VRG_HD void vrg_item_init_entry(const VrgCtx& c, uint32_t e) {
    uint32_t idx = c.b_idx[0][e];
    c.b_lev[0][e] = vrg_voxel_level(c, idx);
    c.fresh[e] = e;
    c.vent[idx] = e;
}

VRG_HD void vrg_item_hist_voxel(const VrgCtx& c, uint32_t idx) {
    uint8_t b = c.lab[0][idx];
    if (b & (VB_OOB | VB_X)) return;
    uint32_t lev = vrg_voxel_level(c, idx);
    if (b & VB_S) vrg_atomic_add(&c.hin[lev], 1); else vrg_atomic_add(&c.hout[lev], 1);
}
