// vrg_hostbackend.cpp - TEST INFRASTRUCTURE: a sequential implementation of vrg_backend.h.
//
// It runs the very same item functions as the HIP kernels (arterynetwork_amd/csrc/vrg_items.h), one
// item at a time, so the parallel restatement of the reference's sequential update() - local label
// rules, rank keys, list rebuild, density bookkeeping - can be validated against the oracle on a
// machine without a GPU.  Built only by tests/hostmodel/Makefile into libvrg_hostmodel.so with the
// vrgm_ symbol prefix; the product package never loads it (the product has no CPU path).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../arterynetwork_amd/csrc/vrg_backend.h"
#include "../../arterynetwork_amd/csrc/vrg_items.h"

int be_set_device(int) { return 0; }
void be_set_tuning(const char*, long long) {}
void* be_alloc(size_t bytes) { return std::malloc(bytes); }
void be_free(void* p) { std::free(p); }
void be_fill(void* p, int byte, size_t bytes) { std::memset(p, byte, bytes); }
void be_upload(void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); }
void be_download(void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); }
void be_sync() {}
const char* be_last_error() { return nullptr; }

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
uint32_t exclusive_scan(uint32_t* a, uint32_t n) {
    uint32_t run = 0;
    for (uint32_t i = 0; i < n; i++) { uint32_t v = a[i]; a[i] = run; run += v; }
    return run;
}
}  // namespace

int be_pack_volume(const VrgCtx& c, float* dst, const void* src, int dtype, const int64_t st[3], int* inexact) {
    *inexact = 0;
    for_real_voxels(c, [&](uint32_t idx, int x, int y, int z) {
        double v = load_as_double(src, dtype, x * st[0] + y * st[1] + z * st[2]);
        float f = (float)v;
        if ((double)f != v) *inexact = 1;
        dst[idx] = f;
    });
    return 0;
}
int be_pack_labels(const VrgCtx& c, uint8_t* dst, const void* src, int dtype, const int64_t st[3], int* bad) {
    *bad = 0;
    for_real_voxels(c, [&](uint32_t idx, int x, int y, int z) {
        double v = load_as_double(src, dtype, x * st[0] + y * st[1] + z * st[2]);
        if (v == 0) dst[idx] = VB_S; else if (v == 3) dst[idx] = 0; else if (v == 4) dst[idx] = VB_X;
        else { dst[idx] = 0; *bad = 1; }
    });
    return 0;
}
int be_unpack_labels(const VrgCtx& c, const uint8_t* lab, void* dst, int dtype, const int64_t st[3]) {
    for_real_voxels(c, [&](uint32_t idx, int x, int y, int z) {
        store_int(dst, dtype, x * st[0] + y * st[1] + z * st[2], vrg_dec(lab[idx]));
    });
    return 0;
}

int be_build_levels(const VrgCtx& c, double** lev, uint32_t* L) {
    std::vector<double> v;
    v.reserve((size_t)c.nx * c.ny * c.nz);
    for_real_voxels(c, [&](uint32_t idx, int, int, int) { v.push_back((double)c.I[idx]); });
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    *L = (uint32_t)v.size();
    *lev = (double*)std::malloc(sizeof(double) * v.size());
    if (!*lev) return -1;
    std::memcpy(*lev, v.data(), sizeof(double) * v.size());
    return 0;
}

void be_build_lev16(const VrgCtx& c, uint16_t* dst) {
    std::memset(dst, 0, (size_t)c.PV * 2);
    for_real_voxels(c, [&](uint32_t idx, int, int, int) { dst[idx] = (uint16_t)vrg_level_of(c, (double)c.I[idx]); });
}

void be_init_band(const VrgCtx& c) {
    for_real_voxels(c, [&](uint32_t idx, int, int, int) { vrg_item_init_voxel(c, idx); });
}

void be_init_sort(const VrgCtx& c, uint32_t n_in, uint32_t n_out) {
    std::vector<std::pair<uint64_t, uint32_t>> a(n_in), b(n_out);
    for (uint32_t i = 0; i < n_in; i++) a[i] = {c.init_key[i], c.init_idx[i]};
    for (uint32_t i = 0; i < n_out; i++) b[i] = {c.init_key[c.bcap - 1 - i], c.init_idx[c.bcap - 1 - i]};
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    for (uint32_t i = 0; i < n_in; i++) c.b_idx[0][i] = a[i].second;
    for (uint32_t i = 0; i < n_out; i++) c.b_idx[0][n_in + i] = b[i].second;
}

// the dense recount over this handle's Z-slab, then the sum over the slabs (callback) if there are several
static void dense_stats(const VrgCtx& c, const uint8_t* lab, be_reduce_fn cb, void* user) {
    int64_t a = 0, b = 0; double sa = 0, sb = 0;
    const uint32_t* cls = c.clsb[(c.dctl[VD_SEQ] + 1) & 1];
    for_real_voxels(c, [&](uint32_t idx, int, int, int z) {
        if (z < c.z0 || z >= c.z1) return;
        uint32_t dw, sh; vrg_cls_pos(idx, dw, sh);
        uint32_t k = (cls[dw] >> sh) & 3u;                   // the dense pass reads its copy of the class bits ...
        if (k != vrg_cls_of(lab[idx])) c.st->error = 6;      // ... which every label write must have kept in step
        double v = c.lev16 ? (double)(float)c.lev[c.lev16[idx]] : (double)c.I[idx];
        if (k == 1u) { a++; sa += v; }
        else if (k == 2u) { b++; sb += v; }
    });
    VrgDense& p = *c.dn_part;
    p.n_in = (double)a; p.n_out = (double)b; p.sum_in = sa; p.sum_out = sb;
    *c.dn = p;
    if (cb) cb(&c.dn->n_in, user);
}

void be_init_finish(const VrgCtx& c, be_reduce_fn cb, void* user) {
    VrgState& s = *c.st;
    uint32_t n = s.ni + s.no;
    for (uint32_t e = 0; e < n; e++) vrg_item_init_entry(c, e);
    for_real_voxels(c, [&](uint32_t idx, int, int, int) { vrg_item_hist_voxel(c, idx); });
    for (uint32_t i = 0; i < n; i++) vrg_exact_serial(c, 0, c.fresh[i]);
    for (uint32_t d = 0; d < (((c.PV + 1023u) >> 10) << 6); d++) vrg_item_cls_build(c, d);
    dense_stats(c, c.lab[0], cb, user);
    s.nfresh = 0;
    vrg_init_counts(c);
    const VrgDense& d = *c.dn;
    VrgTrace& t = c.trace[0];
    t.nflip = 0; t.nseg = (int64_t)d.n_in; t.n_in = (int64_t)d.n_in; t.n_out = (int64_t)d.n_out; t.ni = s.ni; t.no = s.no;
    t.sum_in = d.sum_in; t.sum_out = d.sum_out;
}

int be_comm_unique_id(void*) { return -1; }
int be_comm_init(int, int, const void*) { return -1; }

void be_sweep_once(const VrgCtx& c, int variant, VrgEvents*, be_reduce_fn cb, void* user) {
    VrgState& s = *c.st;
    if (s.done) return;
    int nxt = (s.iter & 1) ^ 1;
    uint32_t n = s.ni + s.no;
    for (uint32_t e = 0; e < n; e++) vrg_item_decide(c, e, false);
    if (int32_t stop = vrg_stop_test(c)) { s.done = stop; return; }
    if (s.error) { s.done = -1; return; }
    for (uint32_t r = 0; r < s.nf; r++) vrg_item_prepass(c, r);
    for (bool changed = true; changed;) {
        changed = false;
        for (uint32_t j = 0; j < s.npend; j++) changed |= vrg_item_fix(c, j) == 2;
    }
    const bool full = variant & 1;
    uint8_t* lab = c.lab[0];
    if (!full) {
        // marks -> sparse two-phase relabel, in place
        for (uint32_t r = 0; r < s.nf; r++) for (uint32_t p = 0; p < 125; p++) vrg_item_scatter_marks(c, r, p);
        if (s.nmk > c.mcap) { s.error = 4; s.done = -1; return; }
        for (uint32_t i = 0; i < s.nmk; i++) vrg_item_relabel(c, i);
        for (uint32_t i = 0; i < s.nmk; i++) vrg_item_apply(c, i);
        for (uint32_t i = 0, nc = vrg_catchup_count(c); i < nc; i++) vrg_item_catchup(c, i);
    } else {
        // full-stencil check variant: every voxel, through the scratch volume
        for_real_voxels(c, [&](uint32_t idx, int, int, int) {
            uint8_t nb = vrg_sweep_core(c, lab, idx, lab[idx]);
            if (lab[idx] & VB_B) c.e_new[c.vent[idx]] = (uint8_t)(nb | VE_VALID);
            c.lab[1][idx] = nb;
        });
        for_real_voxels(c, [&](uint32_t idx, int, int, int) { vrg_count_change(c, idx, lab[idx], c.lab[1][idx]); lab[idx] = (uint8_t)(c.lab[1][idx] & ~VB_F); });
        for (uint32_t i = 0, nc = vrg_catchup_count(c); i < nc; i++) vrg_item_catchup(c, i);
    }
    vrg_request_dense(c);
    vrg_post_apply(c);
    if (!(variant & 4)) {
        dense_stats(c, lab, cb, user);      // the dense recount (:113-116) ...
        vrg_dense_fin(c);                   // ... cross-checks the incremental sizes and files the sums
    }
    // band bookkeeping
    for (uint32_t e = 0; e < n; e++) vrg_item_entry_post(c, e);
    s.nnz = 0;
    for (uint32_t l = 0; l < c.L; l++) {
        const uint32_t o = vrg_delta_off(c) + l;
        uint32_t a = c.dIn[o], b = c.dOut[o], d = c.dConv[o];
        if (a | b | d) {
            uint32_t i = s.nnz++;
            c.nz_lev[i] = l; c.nz_val[i] = c.lev[l]; c.nz_cin[i] = a; c.nz_cout[i] = b; c.nz_cconv[i] = d;
            c.hout[l] += (int32_t)d;                    // included voxels join the outer region
            c.dIn[o] = c.dOut[o] = c.dConv[o] = 0;
        }
    }
    s.use_tab = c.L <= n;
    if (s.use_tab)
        for (uint32_t l = 0; l < c.L; l++) vrg_corrections(c, c.lev[l], c.tabC[3 * (size_t)l], c.tabC[3 * (size_t)l + 1], c.tabC[3 * (size_t)l + 2]);
    s.ncnt = 3 * n;
    uint32_t tot = exclusive_scan(c.scan, s.ncnt);
    uint32_t b0 = vrg_slot_B0(s, 0);
    s.ni_new = (b0 < s.ncnt) ? c.scan[b0] : tot;
    s.nb_new = tot;
    if (tot > c.bcap) { s.error = 1; s.done = -1; return; }
    for (uint32_t e = 0; e < n; e++) vrg_item_scatter_entry(c, e);
    for (uint32_t r = 0; r < s.nf; r++) for (uint32_t k = 0; k < 27; k++) vrg_item_scatter_promo(c, r, k);
    for (uint32_t i = 0; i < s.nfresh; i++) vrg_exact_serial(c, nxt, c.fresh[i]);
    // iterNum += 1 (:117) and the trace record of this update() call
    s.ni = s.ni_new; s.no = s.nb_new - s.ni_new; s.iter++;
    if ((uint32_t)s.iter < c.trace_cap) {
        VrgTrace& t = c.trace[s.iter];
        t.nflip = s.nf; t.nseg = c.inc[VC_NIN]; t.n_in = c.inc[VC_NIN]; t.n_out = c.inc[VC_NOUT]; t.ni = s.ni; t.no = s.no;
    }
    s.nf = 0; s.npend = 0; s.nmk = 0; s.nfresh = 0;
    if (s.error) s.done = -1;
}

void be_events_collect(VrgEvents*, long long) {}

void be_recount_hist(const VrgCtx& c, int par, int32_t* rin, int32_t* rout) {
    for_real_voxels(c, [&](uint32_t idx, int, int, int) {
        uint8_t b = c.lab[0][idx];
        if (b & VB_X) return;
        uint32_t lev = vrg_level_of(c, (double)c.I[idx]);
        if (b & VB_S) rin[lev]++; else rout[lev]++;
    });
}

uint32_t be_collect_segmented(const VrgCtx& c, int par, uint64_t* stamps, uint32_t* idxs, uint32_t cap) {
    uint32_t n = 0;
    for_real_voxels(c, [&](uint32_t idx, int, int, int) {
        if (c.lab[0][idx] & VB_S) { if (n < cap) { stamps[n] = c.stamp[idx]; idxs[n] = idx; } n++; }
    });
    return n;
}
