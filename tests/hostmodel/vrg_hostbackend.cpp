// vrg_hostbackend.cpp - TEST INFRASTRUCTURE: a sequential implementation of vrg_backend.h.
//
// It runs the very same item functions as the HIP kernels (arterynetwork_amd/csrc/vrg_items.h), one
// item at a time, so the parallel restatement of the reference's sequential update() - local label
// rules, list-order keys, band pool, density bookkeeping - can be validated against the oracle on a
// machine without a GPU.  Built only by tests/hostmodel/Makefile into libvrg_hostmodel.so with the
// vrgm_ symbol prefix; the product package never loads it (the product has no CPU path).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../arterynetwork_amd/csrc/vrg_backend.h"
#include "../../arterynetwork_amd/csrc/vrg_items.h"

// low limits, so that the hand-back protocols (VBAIL_FUSE: fused -> four-launch trips, VBAIL_FLIPS: -> host-driven) are exercised all the time
struct VrgBackend { uint32_t small_flips = 8; uint32_t fuse_max = 5; int verify_every = 1; int open_sweeps = 1; };

VrgBackend* be_create(int) { return new VrgBackend(); }
void be_destroy(VrgBackend* b) { delete b; }
void be_set_tuning(VrgBackend* b, const char* name, long long v) {
    if (std::strcmp(name, "small_flips") == 0 && v >= 0) b->small_flips = (uint32_t)v;
    if (std::strcmp(name, "fuse_max") == 0 && v >= 0 && v <= VRG_FUSE_MAX) b->fuse_max = (uint32_t)v;
    if (std::strcmp(name, "verify_every") == 0 && v >= 0) b->verify_every = (int)v;
    if (std::strcmp(name, "open_sweeps") == 0) b->open_sweeps = v != 0;
}
void* be_alloc(VrgBackend*, size_t bytes) { return std::aligned_alloc(256, (bytes + 255) / 256 * 256); }     // (VrgState: a cache line of its own for the live words)
void be_free(VrgBackend*, void* p) { std::free(p); }
void be_fill(VrgBackend*, void* p, int byte, size_t bytes) { std::memset(p, byte, bytes); }
void be_upload(VrgBackend*, void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); }
void be_download(VrgBackend*, void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); }
void be_copy(VrgBackend*, void* dst, const void* src, size_t bytes) { std::memmove(dst, src, bytes); }
void be_sync(VrgBackend*) {}
bool be_band_busy(VrgBackend*) { return false; }
const char* be_last_error(VrgBackend*) { return nullptr; }
void be_clear_error(VrgBackend*) {}
uint32_t be_small_flip_limit(VrgBackend* b) { return b->small_flips; }
uint32_t be_fuse_limit(VrgBackend* b, const VrgCtx& c) { return std::min(b->fuse_max, vrg_fuse_limit(c)); }
bool be_fuse_ok(VrgBackend*, const VrgCtx& c) { return c.L <= (uint32_t)VRG_FUSE_LEVELS || c.lidx != nullptr; }
void be_fuse_enter(VrgBackend*, const VrgCtx& c) { if (!c.st->done && !c.st->bail) for (uint32_t j = 0; j < c.st->nnz && j < c.zcap; j++) vrg_item_level_clear(c, j); }
bool be_wants_sync(VrgBackend*, const VrgCtx&) { return false; }

namespace {
double load_as_double(const void* p, int dtype, int64_t i) {
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
void store_int(void* p, int dtype, int64_t i, int v) {
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
template <class F> void for_real_voxels(const VrgCtx& c, F f) {
    for (int z = 0; z < c.nz; z++)
        for (int y = 0; y < c.ny; y++)
            for (int x = 0; x < c.nx; x++) f(vrg_idx(c, x, y, z), x, y, z);
}
}  // namespace

int be_pack_volume(VrgBackend*, const VrgCtx& c, float* dst, double* dst64, const void* src, int dtype, const int64_t st[3], int* inexact, long long* nonzero) {
    *inexact = 0;
    long long nz = 0;
    for_real_voxels(c, [&](uint32_t idx, int x, int y, int z) {
        double v = load_as_double(src, dtype, x * st[0] + y * st[1] + z * st[2]);
        nz += v != 0.0;
        if (dst64) { dst64[idx] = v; return; }
        float f = (float)v;
        if ((double)f != v) *inexact = 1;
        dst[idx] = f;
    });
    if (nonzero) *nonzero = nz;
    return 0;
}
int be_pack_labels(VrgBackend*, const VrgCtx& c, uint8_t* dst, const void* src, int dtype, const int64_t st[3], int* bad) {
    *bad = 0;
    for_real_voxels(c, [&](uint32_t idx, int x, int y, int z) {
        double v = load_as_double(src, dtype, x * st[0] + y * st[1] + z * st[2]);
        if (v == 0) dst[idx] = VB_S; else if (v == 3) dst[idx] = 0; else if (v == 4) dst[idx] = VB_X;
        else { dst[idx] = 0; *bad = 1; }
    });
    return 0;
}
int be_unpack_labels(VrgBackend*, const VrgCtx& c, const uint8_t* lab, void* dst, int dtype, const int64_t st[3], int what) {
    for_real_voxels(c, [&](uint32_t idx, int x, int y, int z) {
        const int v = vrg_dec(lab[idx]);
        store_int(dst, dtype, x * st[0] + y * st[1] + z * st[2], what ? (v <= 1 ? 1 : 0) : v);
    });
    return 0;
}

int be_build_levels(VrgBackend*, const VrgCtx& c, double** lev, uint32_t* L) {
    std::vector<double> v;
    v.reserve((size_t)c.nx * c.ny * c.nz);
    for_real_voxels(c, [&](uint32_t idx, int, int, int) { v.push_back(vrg_voxel_value(c, idx)); });
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    *L = (uint32_t)v.size();
    *lev = (double*)std::malloc(sizeof(double) * v.size());
    if (!*lev) return -1;
    std::memcpy(*lev, v.data(), sizeof(double) * v.size());
    return 0;
}

bool be_build_lev_map(VrgBackend*, const VrgCtx& c, uint16_t* map, uint32_t span) {
    std::memset(map, 0xff, (size_t)span * 2);
    for (uint32_t k = 0; k < c.L; k++) {
        if (c.lev[k] != std::floor(c.lev[k])) return false;
        map[(uint32_t)(c.lev[k] - c.lev[0])] = (uint16_t)k;
    }
    return true;
}

void be_build_ktab(VrgBackend*, const VrgCtx& c, double* ktab) {
    for (uint32_t a = 0; a < c.L; a++) for (uint32_t b = 0; b < c.L; b++) ktab[(size_t)a * c.L + b] = vrg_kern(c, c.lev[b] - c.lev[a]);
}

void be_build_bins(VrgBackend*, const VrgCtx& c) {
    std::memset(c.bm_in, 0, (size_t)c.nb * (VRG_BIN_K + 1) * 8); std::memset(c.bm_out, 0, (size_t)c.nb * (VRG_BIN_K + 1) * 8);
    for (uint32_t l = 0; l < c.L; l++) if (c.hin[l] | c.hout[l]) vrg_bin_add(c, c.lev[l], c.hin[l], c.hout[l]);
}
long long be_check_bins(VrgBackend* b, const VrgCtx& c, const int32_t* rin, const int32_t* rout) {
    if (!c.nb) return 0;
    const size_t n = (size_t)c.nb * (VRG_BIN_K + 1);
    std::vector<int64_t> ri(n, 0), ro(n, 0);
    VrgCtx r = c;
    r.bm_in = ri.data(); r.bm_out = ro.data(); r.hin = const_cast<int32_t*>(rin); r.hout = const_cast<int32_t*>(rout);
    for (uint32_t l = 0; l < c.L; l++) if (r.hin[l] | r.hout[l]) vrg_bin_add(r, c.lev[l], r.hin[l], r.hout[l]);
    long long bad = 0;
    for (size_t i = 0; i < n; i++) bad += (c.bm_in[i] != ri[i]) + (c.bm_out[i] != ro[i]);
    return bad;
}

void be_build_lidx(VrgBackend*, const VrgCtx& c, uint32_t* dst) {
    std::memset(dst, 0, (size_t)c.PV * 4);
    for_real_voxels(c, [&](uint32_t idx, int, int, int) { dst[idx] = vrg_level_of(c, vrg_voxel_value(c, idx)); });
}
void be_build_lev16(VrgBackend*, const VrgCtx& c, uint16_t* dst) {
    std::memset(dst, 0, (size_t)c.PV * 2);
    for_real_voxels(c, [&](uint32_t idx, int, int, int) { dst[idx] = (uint16_t)vrg_level_of(c, vrg_voxel_value(c, idx)); });
}

void be_init_band(VrgBackend*, const VrgCtx& c) {
    for_real_voxels(c, [&](uint32_t idx, int, int, int) { vrg_item_init_voxel(c, idx); });
}

void be_init_sort(VrgBackend*, const VrgCtx& c, uint32_t n_in, uint32_t n_out) {
    std::vector<std::pair<uint64_t, uint32_t>> a(n_in), b(n_out);
    for (uint32_t i = 0; i < n_in; i++) a[i] = {c.init_key[i], c.init_idx[i]};
    for (uint32_t i = 0; i < n_out; i++) b[i] = {c.init_key[c.bcap - 1 - i], c.init_idx[c.bcap - 1 - i]};
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    for (uint32_t i = 0; i < n_in; i++) c.p_idx[i] = a[i].second;
    for (uint32_t i = 0; i < n_out; i++) c.p_idx[n_in + i] = b[i].second;
}

// the dense recount over this handle's Z-slab, then the sum over the slabs (callback) if there are several
static void dense_stats(const VrgCtx& c, const uint8_t* lab, be_reduce_fn cb, void* user, int last = 0) {
    int64_t a = 0, b = 0; double sa = 0, sb = 0;
    const uint32_t* cls = c.clsb[(c.dctl[VD_RSEQ] + (last ? 0 : 1)) & 1];
    {   // (the device: k_gate, before every recount) the units this pass's sweep listed for the first time join the bitmap, then the list
        const int p = (int)((c.dctl[VD_RSEQ] + (last ? 0 : 1)) & 1);
        if (c.uctl[UC_GEN + p * UC_GEN_STRIDE]) {
            for (size_t w = 0, nw = (((size_t)c.PV + 1023) >> 10) / 32 + 1; w < nw; w++) { c.ubits[w] |= c.unew[p][w]; c.unew[p][w] = 0; }
            c.uctl[UC_GEN + p * UC_GEN_STRIDE] = 0;
            vrg_ulist_rebuild_serial(c);
        }
    }
    std::vector<uint8_t> in_list((((size_t)c.PV + 1023) >> 10) + 1, 0);
    for (uint32_t i = 0; i < c.uctl[UC_N]; i++) { if (i && c.ulist[i] <= c.ulist[i - 1]) c.st->error = 6; in_list[c.ulist[i]] = 1; }
    const uint32_t plane_ = (uint32_t)c.PY * (uint32_t)c.PX, lo_ = (2u + (uint32_t)c.z0) * plane_, hi_ = (2u + (uint32_t)c.z1) * plane_;
    const uint32_t f_lo_ = (uint32_t)(((uint64_t)lo_ + 1023u) >> 10), f_hi_ = hi_ >> 10;
    for_real_voxels(c, [&](uint32_t idx, int, int, int z) {
        if (z < c.z0 || z >= c.z1) return;
        uint32_t dw, sh; vrg_cls_pos(idx, dw, sh);
        uint32_t k = (cls[dw] >> sh) & 3u;                   // the dense pass reads its copy of the class bits ...
        if (k != vrg_cls_of(lab[idx])) c.st->error = 6;      // ... which every label write must have kept in step
        if (k != 0u && (idx >> 10) >= f_lo_ && (idx >> 10) < f_hi_ && !in_list[idx >> 10]) c.st->error = 6;   // ... and walks the unit list only
        double v = c.lev16 ? (c.I ? (double)(float)c.lev[c.lev16[idx]] : c.lev[c.lev16[idx]]) : vrg_voxel_value(c, idx);
        if (k == 1u) { a++; sa += v; }
        else if (k == 2u) { b++; sb += v; }
    });
    VrgDense& p = *c.dn_part;
    p.n_in = (double)a; p.n_out = (double)b; p.sum_in = sa; p.sum_out = sb;
    *c.dn = p;
    if (cb) cb(&c.dn->n_in, user);
}

void be_init_finish(VrgBackend*, const VrgCtx& c, be_reduce_fn cb, void* user) {
    VrgState& s = *c.st;
    uint32_t n = s.ni + s.no;
    for (uint32_t e = 0; e < n; e++) vrg_item_init_entry(c, e);
    for_real_voxels(c, [&](uint32_t idx, int, int, int) { vrg_item_hist_voxel(c, idx); });
    if (c.nb) be_build_bins(nullptr, c);
    for (uint32_t i = 0; i < n; i++) vrg_exact_serial(c, s, c.fresh[i], false);
    for (uint32_t d = 0; d < (((c.PV + 1023u) >> 10) << 6); d++) vrg_item_cls_build(c, d);
    vrg_ulist_rebuild_serial(c);
    dense_stats(c, c.lab[0], cb, user);
    s.np = n; s.nfree = 0; s.nfresh = 0; s.nfx = 0; s.nf = 0; s.npend = 0; s.nmk = 0; s.nnz = 0; s.nalloc = 0; s.ndead = 0;
    s.d_ni = 0; s.d_no = 0; s.corr = 0; s.use_tab = 0; s.bail = 0;
    vrg_init_counts(c);
    const VrgDense& d = *c.dn;
    VrgTrace& t = c.trace[0];
    t.nflip = 0; t.nseg = (int64_t)d.n_in; t.n_in = (int64_t)d.n_in; t.n_out = (int64_t)d.n_out; t.ni = s.ni; t.no = s.no;
    t.sum_in = d.sum_in; t.sum_out = d.sum_out; t.ties = 0; t.near_ties = 0;
    s.ties = 0; s.near_ties = 0; s.ties_filed = 0; s.near_filed = 0;
}

int be_comm_unique_id(void*) { return -1; }
int be_comm_init(VrgBackend*, int, int, const void*) { return -1; }

// the dense pass of the sweep just applied - or, with option verify_every, its marker (k_gate)
static void dense_pass(VrgBackend* b, const VrgCtx& c, be_reduce_fn cb, void* user) {
    if (vrg_dense_skipped(c.dctl[VD_RSEQ] + 1, b->verify_every, c.ver_n, c.ver_me)) {
        vrg_recount_done(c, vrg_dense_skip_marker());
        vrg_dense_fin_one(c, vrg_dense_skip_marker());
        return;
    }
    dense_stats(c, c.lab[0], cb, user);
    vrg_recount_done(c, *c.dn_part);
    VrgDense tot = *c.dn;
    vrg_dense_fin_one(c, tot);
}
void be_verify_last(VrgBackend* b, const VrgCtx& c, be_reduce_fn cb, void* user) {
    dense_stats(c, c.lab[0], cb, user, 1);
    vrg_dense_verify_last(c, *c.dn);
}

// k_sweep (the fused sweep), workgroup by workgroup and - inside a workgroup - phase by phase over its 128 threads: the very
// phase functions the kernel runs between its barriers
// open_end: the sweep stops at its commit (vrg_items.h "open-ended sweeps"); the next trip's k_band part derives the closed state
static void fused_update(VrgBackend* b, const VrgCtx& c00, bool open_end) {
    VrgState& g = *c00.st;
    const bool bigl0 = c00.L > (uint32_t)VRG_FUSE_LEVELS;
    VrgCtx c0 = c00;
    const int lp = (g.iter + 1) & 1;                   // this sweep's set of per-level counters; the other one goes back to zero (every workgroup its stretch)
    if (!bigl0) {
        for (uint32_t wg = 0; wg < (uint32_t)VRG_FUSE_MAX; wg++) vrg_fuse_zero_other_levels(c00, lp, wg, VRG_FUSE_MAX, 0, 1);
        c0.dIn = c00.dInS[lp]; c0.dOut = c00.dOutS[lp]; c0.dConv = c00.dConvS[lp];
    }
    vrg_fuse_prepare_other(c00, g);                    // (a no-op here: this model keeps the state in place)
    const VrgState snap = g;                           // the state as it was when the sweep opened (the kernel's LDS copy)
    const bool bigl = c0.L > (uint32_t)VRG_FUSE_LEVELS;
    const int32_t gate = vrg_fuse_gate(c0, snap, c0.inc[VC_NIN], std::min(b->fuse_max, vrg_fuse_limit(c0)));
    if (gate) {
        if (gate > 0) g.done = gate == 1000 ? -1 : gate; else g.bail = -gate;
        vrg_close_without_update(c0);
        return;
    }
    const uint32_t nf = snap.nf, T = VRG_FUSE_THREADS;
    VrgState st = snap;
    std::vector<VrgFuseThread> th(T);
    VrgFuseLds* shp = new VrgFuseLds();
    VrgFuseLds& sh = *shp;
    for (uint32_t r = 0; r < nf; r++) {
        VrgCtx c = c0;
        c.st = &st; c.lev_fast = 1; c.lvl_scan = bigl ? 2 : 1;
        for (uint32_t t = 0; t < T; t++) { vrg_fuse_load1(c, th[t], t); vrg_fuse_load_rows(c, th[t]); }
        for (uint32_t t = 0; t < T; t++) vrg_fuse_init(sh, t);
        for (uint32_t t = 0; t < T; t++) vrg_fuse_keys(sh, th[t], t, nf);
        for (uint32_t t = 0; t < T; t++) vrg_fuse_rank(c, sh, th[t], t, nf);
        for (uint32_t t = 0; t < T; t++) vrg_fuse_load2(c, sh, th[t], t, r, nf);
        if (!c.lev16 && !bigl) for (uint32_t l = 0; l < c.L; l++) sh.lev[l] = c0.lev[l];
        for (uint32_t t = 0; t < T; t++) vrg_fuse_listed_nbrs(sh, t, nf);
        for (uint32_t t = 0; t < T; t++) vrg_fuse_prepass(sh, th[t], t, nf);
        if (sh.any_pend)
            for (;;) {
                for (uint32_t t = 0; t < T; t++) vrg_fuse_fix(c, sh, t, nf);
                if (!sh.changed) break;
                sh.changed = 0;
            }
        for (uint32_t t = 0; t < T; t++) vrg_fuse_annotate(c, sh, t, r, nf);
        for (uint32_t t = 0; t < T; t++) vrg_fuse_stencil(c, sh, th[t], t, r);
        for (uint32_t t = 0; t < T; t++) vrg_fuse_reserve(c, sh, t);
        for (uint32_t t = 0; t < T; t++) vrg_fuse_commit(c, sh, th[t], t, r);
    }
    delete shp;
    if (open_end) { g.open = 1; return; }
    // the workgroup that finishes last: touched levels in ascending order, counters zeroed, the sweep closed
    uint32_t q = 0;
    if (bigl) {                                        // (large level table: the first touchers' list, sorted)
        q = std::min(std::min(g.nnz_new, c0.zcap), (uint32_t)VRG_FUSE_KEYS);
        std::vector<uint32_t> keys(q);
        for (uint32_t j = 0; j < q; j++) keys[j] = (uint32_t)c0.nz_key[j];
        std::sort(keys.begin(), keys.end());
        for (uint32_t j = 0; j < q; j++) vrg_fuse_level_file_listed(c0, j, keys[j]);
    } else
    for (uint32_t l = 0; l < c0.L; l++) { uint32_t ci, co, cc; if (vrg_fuse_level_touched(c0, l, ci, co, cc)) vrg_fuse_level_file(c0, q++, l, c0.lev[l], ci, co, cc); }
    int64_t n_in, n_out;
    const VrgState fin = vrg_fuse_close_load(c0, n_in, n_out);
    const bool use_tab = vrg_tab_pays(c0.L, snap.ni + snap.no) && c0.ktab != nullptr;
    if (use_tab) {                                     // (the memo workgroups: one wave per level, lanes striding over the list, a fixed butterfly)
        std::vector<uint32_t> nzl(q);
        for (uint32_t j = 0; j < q; j++) nzl[j] = (uint32_t)c0.nz_key[j];
        for (uint32_t l = 0; l < c0.L; l++) {
            double a[64], b2[64], d[64];
            for (uint32_t lane = 0; lane < 64; lane++) vrg_fuse_memo_terms(c0, l, lane, q, nzl.data(), c0.nz_cin, c0.nz_cout, c0.nz_cconv, a[lane], b2[lane], d[lane]);
            for (int o = 32; o > 0; o >>= 1) for (int i = 0; i < 64; i++) if (!(i & o)) { a[i] += a[i ^ o]; b2[i] += b2[i ^ o]; d[i] += d[i ^ o]; a[i ^ o] = a[i]; b2[i ^ o] = b2[i]; d[i ^ o] = d[i]; }
            c0.tabC[3 * (size_t)l] = a[0]; c0.tabC[3 * (size_t)l + 1] = b2[0]; c0.tabC[3 * (size_t)l + 2] = d[0];
        }
    }
    vrg_fuse_close(c0, fin, n_in, n_out, q, use_tab);
}


void be_sweep_once(VrgBackend* b, VrgCtx& c0, int flags, VrgEvents*, be_reduce_fn cb, void* user, bool first, bool last) {
    VrgState& s = *c0.st;
    if (s.done || s.bail) return;
    VrgState derived = s;
    if (s.open) {                                      // (k_band after an open-ended sweep: the touched levels listed from the counters - the device: in every pool workgroup's
        const int p = (s.iter + 1) & 1;                //  LDS -, the closed state derived, filed by one thread)
        uint32_t q = 0;
        for (uint32_t l = 0; l < c0.L; l++) {
            const uint32_t ci = c0.dInS[p][l], co = c0.dOutS[p][l], cc = c0.dConvS[p][l];
            if (ci | co | cc) { c0.nz_key[q] = l; c0.nz_val[q] = c0.lev[l]; c0.nz_cin[q] = ci; c0.nz_cout[q] = co; c0.nz_cconv[q] = cc; q++; }
        }
        VrgFuseClosed f;
        vrg_fuse_close_core(c0, derived, c0.inc[VC_NIN], c0.inc[VC_NOUT], q, false, f);
        vrg_fuse_persist(c0, derived, f, true, f.n_in, f.n_out);
    }
    if (derived.apply_pending) {                       // (k_band: what the fused sweep before this trip left to do)
        const int k = derived.iter;
        const VrgState snap0 = derived;
        for (uint32_t i = 0; i < snap0.ap_n; i++) vrg_deferred_apply(c0, i, k);
        for (uint32_t i = 0, nc = vrg_deferred_catchup_count(c0, k); i < nc; i++) vrg_deferred_catchup(c0, i, k);
        for (uint32_t j = 0; j < snap0.fr_n; j++) vrg_deferred_free(c0, snap0, j);
        vrg_deferred_done(c0, k);
        if (!(flags & VRG_SWEEP_NODENSE)) dense_pass(b, c0, cb, user);   // its dense pass (the device: gate + recount on the other stream, asked for by vrg_deferred_done)
    }
    vrg_log_publish(c0, derived.log_nsw, derived.log_pos, false);      // (the first kernel of the trip's update(): the change log is complete up to the sweep before this trip)
    // the per-launch modes of the batched kernels, alternated so that both forms of every item function run here: the
    // touched levels listed by atomics / found by scanning the counters; a flip's level fetched through its rank / looked up
    VrgCtx c = c0;
    c.lvl_scan = (s.iter & 1) && c.L <= 2048;
    c.lev_fast = (s.iter >> 1) & 1;
    // ---- k_band: corrections of the sweep before, decisions, exact densities of the entries that sweep added
    {
        const VrgState snap = s;
        for (uint32_t slot = 0; slot < snap.np; slot++) vrg_item_band(c, snap, slot, c.nz_val, c.nz_cin, c.nz_cout, c.nz_cconv);
        for (uint32_t i = 0; i < snap.nfx; i++) vrg_exact_serial(c, snap, c.fresh[i], true);
    }
    if ((flags & VRG_SWEEP_FUSED) && !(flags & (VRG_SWEEP_SYNC | VRG_SWEEP_FULL))) {
        // (as the device: not the last trip of a batch, small level tables with the kernel table only)
        fused_update(b, c0, b->open_sweeps && !last && c0.L <= 1024 && c0.ktab != nullptr);
        return;
    }
    // ---- update() as the four-launch chain / host-driven
    if (int32_t stop = vrg_stop_test(c)) { s.done = stop; vrg_close_without_update(c); return; }
    if (s.error) { s.done = -1; return; }
    const uint32_t nf = s.nf;
    int32_t bail = (!(flags & VRG_SWEEP_SYNC) && nf > b->small_flips) ? (int32_t)VBAIL_FLIPS : vrg_capacity_test(c, nf);
    if (bail) { s.bail = bail; vrg_close_without_update(c); return; }
    for (uint32_t j = 0; j < s.nnz; j++) vrg_item_level_clear(c, j);   // (a no-op here: this model clears when it files the levels)
    vrg_open_update(c);
    {   // the flip list in the reference's order (:88)
        std::vector<std::pair<uint64_t, uint32_t>> v(nf);
        for (uint32_t q = 0; q < nf; q++) v[q] = {c.f_key[q], q};
        std::sort(v.begin(), v.end());
        for (uint32_t r = 0; r < nf; r++) {             // alternately from the flip's record / through its slot
            const uint32_t q = v[r].second;
            if (r & 1) vrg_item_list_rec(c, r, c.flist[q], c.fr_idx[q], c.fr_lev[q], !(c.f_key[q] >> 63));
            else { c.f_slot[r] = c.flist[q]; vrg_item_list(c, r); }
        }
    }
    for (uint32_t r = 0; r < nf; r++) vrg_item_prepass(c, r);
    for (bool changed = true; changed;) {
        changed = false;
        for (uint32_t j = 0; j < s.npend; j++) changed |= vrg_item_fix(c, j) == 2;
    }
    uint8_t* lab = c.lab[0];
    if (!(flags & VRG_SWEEP_FULL)) {
        // marks -> sparse two-phase relabel, in place
        for (uint32_t r = 0; r < nf; r++) for (uint32_t p = 0; p < 125; p++) vrg_item_scatter_marks(c, r, p);
        if (s.nmk > c.mcap) { s.error = 4; s.done = -1; return; }
        for (uint32_t i = 0; i < s.nmk; i++) vrg_item_relabel(c, i);
        // (as k_close does: the class change filed at the voxel's place of the marked list, no counter)
        for (uint32_t i = 0; i < s.nmk; i++) { const uint32_t idx = c.mk_idx[i]; vrg_apply_at(c, i, idx, lab[idx], c.mk_new[i], s.log_pos); }
        for (uint32_t i = 0, nc = vrg_catchup_count(c); i < nc; i++) vrg_item_catchup(c, i);
    } else {
        // full-stencil check variant: every voxel, through the scratch volume
        for_real_voxels(c, [&](uint32_t idx, int, int, int) { VrgEvent ev; c.lab[1][idx] = vrg_sweep_core(c, lab, idx, lab[idx], ev); vrg_commit_event(c, idx, ev); });
        for_real_voxels(c, [&](uint32_t idx, int, int, int) { vrg_count_change(c, idx, lab[idx], c.lab[1][idx]); lab[idx] = c.lab[1][idx]; });
        for (uint32_t i = 0, nc = vrg_catchup_count(c); i < nc; i++) vrg_item_catchup(c, i);
    }
    vrg_request_dense(c);
    vrg_post_apply(c, (flags & VRG_SWEEP_FULL) ? -1 : (int64_t)s.nmk);
    if (!(flags & VRG_SWEEP_NODENSE)) dense_pass(b, c, cb, user);   // the dense recount (:113-116): cross-checks the incremental sizes and files the sums
    // closing: flips all visited, dead slots onto the free list, this sweep's level deltas in level order
    for (uint32_t r = 0; r < nf; r++) vrg_item_check_flip(c, r);
    for (uint32_t j = 0; j < s.ndead; j++) vrg_item_free(c, j);
    if (c.lvl_scan) {                          // (k_close, small level table: the touched levels by a scan of the counters)
        s.nnz = 0;
        for (uint32_t l = 0; l < c.L; l++) if (c.dIn[l] | c.dOut[l] | c.dConv[l]) c.nz_key[s.nnz++] = l;
    }
    std::sort(c.nz_key, c.nz_key + s.nnz);
    for (uint32_t j = 0; j < s.nnz; j++) vrg_item_level(c, j, (s.iter & 1) != 0);     // alternate: cleared at once / by the next update()
    const bool use_tab = s.tab_ok != 0;          // either way the same sums; the model alternates with the band size
    if (use_tab)
        for (uint32_t l = 0; l < c.L; l++)
            vrg_corrections(c, s.nnz, c.nz_val, c.nz_cin, c.nz_cout, c.nz_cconv, c.lev[l], c.tabC[3 * (size_t)l], c.tabC[3 * (size_t)l + 1], c.tabC[3 * (size_t)l + 2]);
    vrg_finalize(c, use_tab);
}

// ---- leader / follower replication: a follower's apply and verify steps, sequentially (the transports are the engine's callbacks) ----
void be_follow_apply(VrgBackend*, const VrgCtx& c, const VrgLogRec* recs, const VrgLogSweep* hdr, int n, int count_last) {
    for (int s = 0; s < n; s++) {
        vrg_follow_trace(c, hdr[s]);
        for (uint32_t i = 0; i < hdr[s].nrec; i++) { vrg_follow_label_rec(c, recs[hdr[s].rec0 + i], hdr[s].sweep); vrg_follow_class_rec(c, recs[hdr[s].rec0 + i]); }
    }
    if (!count_last) return;
    vrg_follow_expect(c, hdr[n - 1]);
    for (int p = 0; p < 2; p++)                          // (k_follow_classes) the units the applied sweeps listed join the bitmap, then the list
        if (c.uctl[UC_GEN + p * UC_GEN_STRIDE]) {
            for (size_t w = 0, nw = (((size_t)c.PV + 1023) >> 10) / 32 + 1; w < nw; w++) { c.ubits[w] |= c.unew[p][w]; c.unew[p][w] = 0; }
            c.uctl[UC_GEN + p * UC_GEN_STRIDE] = 0;
            vrg_ulist_rebuild_serial(c);
        }
}
void be_follow_count(VrgBackend*, const VrgCtx& c, VrgEvents*) {
    dense_stats(c, c.lab[0], nullptr, nullptr, 1);
    vrg_follow_check(c, *c.dn_part, (uint32_t)c.fexp[0], c.fexp[1], c.fexp[2]);
}
void be_follow_mark(VrgBackend*, int) {}
void be_follow_wait(VrgBackend*, int) {}
int be_repl_bcast(VrgBackend*, void*, size_t, int) { return -1; }
int be_repl_allsum(VrgBackend*, double*, size_t) { return -1; }
void be_repl_wait(VrgBackend*) {}
void be_repl_copy(VrgBackend*, void* dst, const void* src, size_t bytes) { std::memmove(dst, src, bytes); }
void* be_host_alloc(VrgBackend*, size_t bytes) { return std::malloc(bytes); }
void be_host_free(VrgBackend*, void* p) { std::free(p); }
int be_ipc_export(VrgBackend*, void*, void*) { return -1; }
void* be_ipc_open(VrgBackend*, const void*) { return nullptr; }
void be_ipc_close(VrgBackend*, void*) {}

void be_events_collect(VrgBackend*, VrgEvents*, long long) {}
void be_sweep_batch(VrgBackend* b, VrgCtx& c, int flags, int n, VrgEvents* ev, be_reduce_fn cb, void* user) { for (int i = 0; i < n; i++) be_sweep_once(b, c, flags, ev, cb, user, i == 0, i == n - 1); }
void be_dense_info(VrgBackend*, const VrgCtx& c, int64_t out[5]) { out[0] = 0; out[1] = c.lev16 ? 1 : (c.I ? 0 : 2); out[2] = 1; out[3] = 1; out[4] = 0; }
void be_dense_flush(VrgBackend*, const VrgCtx&, be_reduce_fn, void*) {}
long long be_memo_trips(VrgBackend*) { return 0; }
long long be_slow_flips(VrgBackend*, const VrgCtx&) { return 0; }

void be_recount_hist(VrgBackend*, const VrgCtx& c, int32_t* rin, int32_t* rout) {
    for_real_voxels(c, [&](uint32_t idx, int, int, int) {
        uint8_t b = c.lab[0][idx];
        if (b & VB_X) return;
        uint32_t lev = vrg_level_of(c, vrg_voxel_value(c, idx));
        if (b & VB_S) rin[lev]++; else rout[lev]++;
    });
}

// same definition as the device kernel (k_dense_bytes), as a plain loop over the class words
uint64_t be_dense_bytes(VrgBackend*, const VrgCtx& c) {
    const uint32_t* cls = c.clsb[c.dctl[VD_RSEQ] & 1];
    const uint32_t plane = (uint32_t)c.PY * (uint32_t)c.PX;
    const uint32_t lo = (2u + (uint32_t)c.z0) * plane, hi = (2u + (uint32_t)c.z1) * plane;
    const uint32_t lpl = c.lev16 ? 16u : (c.I ? 8u : 4u);
    uint32_t f_lo = (uint32_t)(((uint64_t)lo + 1023u) >> 10), f_hi = hi >> 10;
    if (f_hi < f_lo) f_hi = f_lo;
    uint64_t bytes = 0;
    for (uint32_t u = lo >> 10; u <= (hi - 1u) >> 10; u++) {
        const bool whole = u >= f_lo && u < f_hi;
        if (whole && !((c.ubits[u >> 5] >> (u & 31u)) & 1u)) continue;
        bytes += whole ? 260u : 256u;                     // class words (+ the unit's list entry)
        for (int j = 0; j < 4; j++)
            for (uint32_t g = 0; g < 64u; g += lpl) {
                bool any = false;
                for (uint32_t lane = g; lane < g + lpl; lane++) {
                    const uint32_t v = (u << 10) + (j << 8) + (lane << 2);
                    if (v >= lo && v < hi && ((cls[((size_t)u << 6) + lane] >> (8 * j)) & 0xffu)) any = true;
                }
                if (any) bytes += 128u;
            }
    }
    return bytes;
}

uint32_t be_collect_segmented(VrgBackend*, const VrgCtx& c, uint64_t* stamps, uint32_t* idxs, uint32_t cap) {
    uint32_t n = 0;
    for_real_voxels(c, [&](uint32_t idx, int, int, int) {
        if (c.lab[0][idx] & VB_S) { if (n < cap) { stamps[n] = c.stamp[idx]; idxs[n] = idx; } n++; }
    });
    return n;
}
