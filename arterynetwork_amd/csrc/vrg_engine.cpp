// vrg_engine.cpp - handle management and the C-ABI entry points (include/vrg.h) on top of a backend.
// Compiled with hipcc into libvrg_hip.so (backend vrg_device.hip).  tests/hostmodel compiles the same
// file with VRG_API_PREFIX=vrgm_ against the sequential test backend.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vrg.h"
#include "vrg_backend.h"
#include "vrg_items.h"

#ifndef VRG_API_PREFIX
#define VRG_API_PREFIX vrg_
#endif
#define VRG_CAT2(a, b) a##b
#define VRG_CAT(a, b) VRG_CAT2(a, b)
#define API(name) VRG_CAT(VRG_API_PREFIX, name)

struct vrg_handle {
    VrgCtx c;
    std::string err;
    std::vector<void*> owned;
    int device = 0;
    bool have_vol = false, have_lab = false, inited = false;
    int variant = 0, batch = 8, storage16 = 0, dense_off = 0;
    uint16_t* lev16_buf = nullptr;
    uint64_t band_capacity = 0;
    VrgEvents ev{0, 0.0, 0};
    std::chrono::steady_clock::time_point t0;
    int64_t V = 0;
    uint8_t* lab_base[2] = {nullptr, nullptr};
    vrg_reduce_fn reduce_fn = nullptr;
    void* reduce_user = nullptr;
};

extern "C" void API(destroy)(vrg_handle* h);

namespace {

int fail(vrg_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

template <class T> T* alloc(vrg_handle* h, size_t n) {
    void* p = be_alloc(std::max<size_t>(n, 1) * sizeof(T));
    if (p) h->owned.push_back(p);
    return (T*)p;
}

VrgDense get_dense(vrg_handle* h) {     // region sizes as the band side keeps them + the sums of the last dense pass
    VrgDense d; be_download(&d, h->c.dn, sizeof(d));
    int64_t n[2]; be_download(n, h->c.inc, sizeof(n));
    d.n_in = (double)n[0]; d.n_out = (double)n[1];
    return d;
}
VrgState get_state(vrg_handle* h) { VrgState s; be_download(&s, h->c.st, sizeof(s)); return s; }
void put_state(vrg_handle* h, const VrgState& s) { be_upload(h->c.st, &s, sizeof(s)); }

void idx_to_xyz(const VrgCtx& c, uint32_t idx, int64_t* out) {
    int x, y, z; vrg_coords(c, idx, x, y, z);
    out[0] = x; out[1] = y; out[2] = z;
}

int check_state_error(vrg_handle* h, const VrgState& s) {
    if (const char* be = be_last_error()) return fail(h, VRG_E_INTERNAL, be);
    if (s.error == 1) return fail(h, VRG_E_CAPACITY, "band capacity exceeded; raise option band_capacity");
    if (s.error == 2) return fail(h, VRG_E_CAPACITY, "flip capacity exceeded; raise option band_capacity");
    if (s.error == 7) return fail(h, VRG_E_CAPACITY, "class-change list capacity exceeded; raise option band_capacity");
    if (s.error) return fail(h, VRG_E_INTERNAL, "internal consistency check failed (code " + std::to_string(s.error) + ")");
    return VRG_OK;
}

}  // namespace

extern "C" {

int API(create)(int64_t nx, int64_t ny, int64_t nz, int device, vrg_handle** out) {
    if (!out) return VRG_E_ARG;
    *out = nullptr;
    if (nx < 1 || ny < 1 || nz < 1) return VRG_E_ARG;
    int64_t PX = (nx + 2 + 15) / 16 * 16, PY = ny + 4, PZ = nz + 4;
    if ((double)PX * (double)PY * (double)PZ >= 4294967040.0) return VRG_E_ARG;   // 32-bit voxel indices
    if (be_set_device(device) != 0) return VRG_E_NOGPU;
    vrg_handle* h = new vrg_handle();
    h->device = device;
    std::memset(&h->c, 0, sizeof(VrgCtx));
    VrgCtx& c = h->c;
    c.nx = (int32_t)nx; c.ny = (int32_t)ny; c.nz = (int32_t)nz;
    c.PX = (int32_t)PX; c.PY = (int32_t)PY; c.PZ = (int32_t)PZ;
    c.PV = (uint32_t)(PX * PY * PZ);
    c.z0 = 0; c.z1 = (int32_t)nz;
    h->V = nx * ny * nz;
    const size_t PVu = ((size_t)c.PV + 1023) / 1024 * 1024;   // dense arrays end on a whole 1024-voxel unit
    c.I = alloc<float>(h, PVu);
    c.clsb[0] = alloc<uint32_t>(h, PVu / 16); c.clsb[1] = alloc<uint32_t>(h, PVu / 16);
    c.nchg = alloc<uint32_t>(h, 32);
    c.vent = alloc<uint32_t>(h, PVu);
    // 16 guard bytes in front: voxel (0,0,0)'s 2-ring reaches 2 bytes before the padded array
    h->lab_base[0] = alloc<uint8_t>(h, (size_t)c.PV + 32);
    h->lab_base[1] = alloc<uint8_t>(h, (size_t)c.PV + 32);
    c.lab[0] = h->lab_base[0] ? h->lab_base[0] + 16 : nullptr;
    c.lab[1] = h->lab_base[1] ? h->lab_base[1] + 16 : nullptr;
    c.stamp = alloc<uint64_t>(h, c.PV);
    c.st = alloc<VrgState>(h, 1);
    c.dn = alloc<VrgDense>(h, 16);                   // own allocation: written by the dense kernel only
    c.counters = alloc<uint32_t>(h, 64);
    c.dn_part = alloc<VrgDense>(h, 16);
    c.inc = alloc<int64_t>(h, 32); c.dctl = alloc<int64_t>(h, 32);   // one allocation each: written from different streams
    c.world = 1;
    if (!c.I || !c.lab[0] || !c.lab[1] || !c.stamp || !c.st || !c.dn || !c.counters || !c.dn_part || !c.inc || !c.dctl || !c.clsb[0] || !c.clsb[1] || !c.nchg || !c.vent) { API(destroy)(h); return VRG_E_MEM; }
    be_fill(c.inc, 0, 32 * sizeof(int64_t)); be_fill(c.dctl, 0, 32 * sizeof(int64_t));
    be_fill((void*)c.I, 0, PVu * 4);
    if (c.clsb[0]) be_fill(c.clsb[0], 0, PVu / 4);
    if (c.clsb[1]) be_fill(c.clsb[1], 0, PVu / 4);
    if (c.nchg) be_fill(c.nchg, 0, 32 * sizeof(uint32_t));
    be_fill(h->lab_base[0], VB_OOB, (size_t)c.PV + 32);
    be_fill(h->lab_base[1], VB_OOB, (size_t)c.PV + 32);
    be_fill(c.st, 0, sizeof(VrgState));
    be_fill(c.dn, 0, sizeof(VrgDense));
    be_fill(c.dn_part, 0, sizeof(VrgDense));
    be_fill(c.counters, 0, 64 * sizeof(uint32_t));
    *out = h;
    return VRG_OK;
}

void API(destroy)(vrg_handle* h) {
    if (!h) return;
    be_sync();
    for (void* p : h->owned) be_free(p);
    delete h;
}

const char* API(last_error)(const vrg_handle* h) { return h ? h->err.c_str() : "null handle"; }

int API(set_option)(vrg_handle* h, const char* name, int64_t value) {
    if (!h || !name) return VRG_E_ARG;
    std::string n(name);
    if (n == "band_capacity") { if (h->inited || value < 1) return fail(h, VRG_E_STATE, "band_capacity must be set before vrg_init"); h->band_capacity = (uint64_t)value; }
    else if (n == "sweep_variant") h->variant = (int)value;
    else if (n == "events") h->ev.enabled = value != 0;
    else if (n == "dense_off") h->dense_off = value != 0;   // measurement aid: band chain alone; re-initialise afterwards
    else if (n == "batch") h->batch = (int)std::max<int64_t>(1, value);
    else if (n == "sweep_blocks" || n == "prio_mode" || n == "graph") be_set_tuning(name, value);
    else if (n == "storage16") { if (h->inited) return fail(h, VRG_E_STATE, "storage16 must be set before vrg_init"); h->storage16 = value != 0; }
    else return fail(h, VRG_E_ARG, "unknown option " + n);
    return VRG_OK;
}

int API(set_volume)(vrg_handle* h, const void* data, int dtype, const int64_t st[3]) {
    if (!h || !data || !st || dtype < VRG_U8 || dtype > VRG_F64) return fail(h, VRG_E_ARG, "set_volume: bad argument");
    int inexact = 0;
    int rc = be_pack_volume(h->c, (float*)h->c.I, data, dtype, st, &inexact);
    if (rc) return fail(h, VRG_E_ARG, "set_volume: unsupported strides");
    if (inexact) return fail(h, VRG_E_INEXACT, "set_volume: intensities are not exactly representable in fp32");
    h->have_vol = true; h->inited = false;
    h->c.lev = nullptr;                              // distinct-value table is rebuilt by the next vrg_init
    return VRG_OK;
}

int API(set_labels)(vrg_handle* h, const void* labels, int dtype, const int64_t st[3]) {
    if (!h || !labels || !st || dtype < VRG_U8 || dtype > VRG_F64) return fail(h, VRG_E_ARG, "set_labels: bad argument");
    be_fill(h->lab_base[0], VB_OOB, (size_t)h->c.PV + 32);
    be_fill(h->lab_base[1], VB_OOB, (size_t)h->c.PV + 32);
    int bad = 0;
    int rc = be_pack_labels(h->c, h->c.lab[0], labels, dtype, st, &bad);
    if (rc) return fail(h, VRG_E_ARG, "set_labels: unsupported strides");
    if (bad) return fail(h, VRG_E_ARG, "set_labels: valueMap must contain only 0 (seed), 3 (outside), 4 (excluded)");
    h->have_lab = true; h->inited = false;
    return VRG_OK;
}

int API(init)(vrg_handle* h, double H) {
    if (!h) return VRG_E_ARG;
    if (!h->have_vol || !h->have_lab) return fail(h, VRG_E_STATE, "vrg_init: set_volume and set_labels first");
    if (h->inited) return fail(h, VRG_E_STATE, "vrg_init: already initialised; set_labels again to restart");
    VrgCtx& c = h->c;
    h->t0 = std::chrono::steady_clock::now();       // start_time (:38)
    c.H = H;
    c.A = std::pow(2.0 * M_PI, -0.5);               // A = (2*np.pi)**(-0.5) (:7)
    // levels
    double* lev = nullptr; uint32_t L = 0;
    if (c.lev) { /* re-init on the same volume: keep */ lev = (double*)c.lev; L = c.L; }
    else {
        if (be_build_levels(c, &lev, &L)) return fail(h, VRG_E_MEM, "vrg_init: level table");
        h->owned.push_back(lev);
        c.lev = lev; c.L = L;
        c.hin = alloc<int32_t>(h, L); c.hout = alloc<int32_t>(h, L);
        c.dIn = alloc<uint32_t>(h, 2 * (size_t)L); c.dOut = alloc<uint32_t>(h, 2 * (size_t)L); c.dConv = alloc<uint32_t>(h, 2 * (size_t)L);
        c.nz_lev = alloc<uint32_t>(h, L); c.nz_val = alloc<double>(h, L);
        c.nz_cin = alloc<uint32_t>(h, L); c.nz_cout = alloc<uint32_t>(h, L); c.nz_cconv = alloc<uint32_t>(h, L);
        c.tabC = alloc<double>(h, 3 * (size_t)L);
        c.lscan = alloc<uint32_t>(h, (size_t)L + 16);
        if (!c.hin || !c.hout || !c.dIn || !c.dOut || !c.dConv || !c.nz_lev || !c.nz_val || !c.nz_cin || !c.nz_cout || !c.nz_cconv || !c.tabC || !c.lscan)
            return fail(h, VRG_E_MEM, "vrg_init: level arrays");
    }
    c.lev16 = nullptr;
    if (h->storage16) {                             // 16-bit intensity storage: level indices + LDS value table
        if (L > 16384) return fail(h, VRG_E_ARG, "storage16: more than 16384 distinct intensity values");
        if (!h->lev16_buf) h->lev16_buf = alloc<uint16_t>(h, ((size_t)c.PV + 1023) / 1024 * 1024);
        if (!h->lev16_buf) return fail(h, VRG_E_MEM, "vrg_init: 16-bit level volume");
        be_build_lev16(c, h->lev16_buf);
        c.lev16 = h->lev16_buf;
    }
    be_fill(c.hin, 0, (size_t)L * 4); be_fill(c.hout, 0, (size_t)L * 4);
    be_fill(c.dIn, 0, (size_t)L * 8); be_fill(c.dOut, 0, (size_t)L * 8); be_fill(c.dConv, 0, (size_t)L * 8);
    // band storage
    if (!c.b_idx[0]) {
        uint64_t V = (uint64_t)h->V;
        uint64_t cap = h->band_capacity ? h->band_capacity : (V <= (32u << 20) ? V : std::max<uint64_t>(32u << 20, V / 8));
        cap = std::min<uint64_t>(std::max<uint64_t>(cap, 64), V);
        c.bcap = (uint32_t)cap; c.fcap = c.bcap;
        for (int p = 0; p < 2; p++) {
            c.b_idx[p] = alloc<uint32_t>(h, c.bcap); c.b_lev[p] = alloc<uint32_t>(h, c.bcap);
            c.b_ip[p] = alloc<double>(h, c.bcap); c.b_op[p] = alloc<double>(h, c.bcap); c.b_pend[p] = alloc<uint8_t>(h, c.bcap);
            if (c.b_pend[p]) be_fill(c.b_pend[p], 0, c.bcap);
            if (!c.b_idx[p] || !c.b_lev[p] || !c.b_ip[p] || !c.b_op[p] || !c.b_pend[p]) return fail(h, VRG_E_MEM, "vrg_init: band arrays");
        }
        c.e_flag = alloc<uint8_t>(h, c.bcap); c.e_surv = alloc<uint8_t>(h, c.bcap); c.e_new = alloc<uint8_t>(h, c.bcap);
        c.e_res = alloc<uint8_t>(h, c.bcap); c.e_mask = alloc<uint32_t>(h, c.bcap);
        c.scan = alloc<uint32_t>(h, 3 * (size_t)c.bcap + 16);
        c.bsum = alloc<uint32_t>(h, 1024);
        c.flist = alloc<uint32_t>(h, c.fcap); c.fidx = alloc<uint32_t>(h, c.fcap);
        c.pend = alloc<uint32_t>(h, c.fcap); c.fresh = alloc<uint32_t>(h, c.bcap);
        c.init_key = alloc<uint64_t>(h, c.bcap); c.init_idx = alloc<uint32_t>(h, c.bcap);
        c.mcap = (uint32_t)std::min<uint64_t>(V, 0xffffffffull);
        c.mk_idx = alloc<uint32_t>(h, c.mcap); c.mk_new = alloc<uint8_t>(h, (size_t)c.mcap + 16);
        // class changes of one sweep: its flips + the excluded voxels it includes - a small fraction of the band in
        // practice; everything for small volumes, 2 x band when the caller sized the band explicitly
        c.ccap = (uint32_t)std::min<uint64_t>(V, h->band_capacity ? 2 * (uint64_t)c.bcap : std::max<uint64_t>(4u << 20, c.bcap / 4));
        for (int p = 0; p < 2; p++) {
            c.chg_dw[p] = alloc<uint32_t>(h, c.ccap); c.chg_x[p] = alloc<uint32_t>(h, c.ccap);
            if (!c.chg_dw[p] || !c.chg_x[p]) return fail(h, VRG_E_MEM, "vrg_init: class-change lists");
        }
        c.nstat = 4096;
        c.st_nin = alloc<int64_t>(h, c.nstat); c.st_nout = alloc<int64_t>(h, c.nstat);
        c.st_sin = alloc<double>(h, c.nstat); c.st_sout = alloc<double>(h, c.nstat);
        c.trace_cap = 1u << 16;
        c.trace = alloc<VrgTrace>(h, c.trace_cap);
        if (!c.e_flag || !c.e_surv || !c.e_new || !c.scan || !c.bsum || !c.e_res || !c.e_mask || !c.flist || !c.fidx || !c.pend || !c.fresh ||
            !c.init_key || !c.init_idx || !c.mk_idx || !c.mk_new || !c.st_nin || !c.st_nout || !c.st_sin || !c.st_sout || !c.trace)
            return fail(h, VRG_E_MEM, "vrg_init: work arrays");
    }
    VrgState s; std::memset(&s, 0, sizeof(s));
    s.iterMax = 0; s.maxSegmentSize = 0;
    put_state(h, s);
    be_init_band(c);
    s = get_state(h);
    if (s.error || (uint64_t)s.ninit_in + s.ninit_out > c.bcap) return fail(h, VRG_E_CAPACITY, "vrg_init: band capacity exceeded");
    if (s.nseed == 0) return fail(h, VRG_E_EMPTY, "vrg_init: valueMap has no seed (label 0) voxel");
    be_init_sort(c, s.ninit_in, s.ninit_out);
    s.ni = s.ninit_in; s.no = s.ninit_out; s.nfresh = s.ni + s.no;
    put_state(h, s);
    be_init_finish(c, h->reduce_fn, h->reduce_user);
    s = get_state(h);
    int rc = check_state_error(h, s);
    if (rc) return rc;
    h->inited = true;
    h->ev.ms_total = 0; h->ev.launches = 0;
    return VRG_OK;
}

int API(run)(vrg_handle* h, int64_t iterMax, int64_t maxSegmentSize, double maxSeconds, vrg_result* out) {
    if (!h) return VRG_E_ARG;
    if (!h->inited) return fail(h, VRG_E_STATE, "vrg_run: call vrg_init first");
    VrgCtx& c = h->c;
    if (iterMax < 0 || iterMax + 1 >= (int64_t)c.trace_cap) return fail(h, VRG_E_ARG, "vrg_run: iterMax out of range");
    VrgState s = get_state(h);
    int rc = check_state_error(h, s);
    if (rc) return rc;
    int32_t iter0 = s.iter;
    s.done = 0; s.time_up = 0; s.iterMax = (int32_t)iterMax; s.maxSegmentSize = maxSegmentSize;
    s.nf = 0; s.npend = 0; s.nmk = 0; s.nfresh = 0;      // counters of a trip that stopped before update()
    put_state(h, s);
    double ms0 = h->ev.ms_total; long long l0 = h->ev.launches;
    auto t_begin = std::chrono::steady_clock::now();
    for (;;) {
        int64_t remaining = iterMax - s.iter;
        int nb = (int)std::min<int64_t>(h->batch, std::max<int64_t>(remaining, 0) + 1);   // +1: the trip that sets the stop flag
        if (maxSeconds >= 0 && s.iter < iterMax) {   // wall-clock cap (:97): tested after the no-flip test, before update()
            double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - h->t0).count();
            if (el >= maxSeconds) { s.time_up = 1; put_state(h, s); nb = 1; }
        }
        int32_t before = s.iter;
        for (int i = 0; i < nb; i++) be_sweep_once(c, h->variant | (h->dense_off ? 4 : 0), &h->ev, h->reduce_fn, h->reduce_user);
        s = get_state(h);
        be_events_collect(&h->ev, s.iter - before);
        if (s.done || s.error) break;
    }
    be_sync();
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    int64_t dense_err = 0;                           // raised by the dense stream, possibly after the band side stopped
    be_download(&dense_err, c.dctl + VD_ERR, sizeof(dense_err));
    if (h->dense_off) { dense_err = 0; h->inited = false; }   // the dense pass sequence is broken on purpose: init again
    if (dense_err) s.error = (int32_t)dense_err;
    rc = check_state_error(h, s);
    if (rc) return rc;
    if (out) {
        out->stop_reason = s.done; out->iter_num = s.iter + 1; out->sweeps = s.iter - iter0;
        VrgDense d = get_dense(h);
        out->nseg = (int64_t)d.n_in; out->n_in = (int64_t)d.n_in; out->n_out = (int64_t)d.n_out; out->ni = s.ni; out->no = s.no;
        out->sum_in = d.sum_in; out->sum_out = d.sum_out; out->seconds = secs;
        out->sweep_kernel_ms = h->ev.ms_total - ms0; out->sweep_launches = h->ev.launches - l0;
    }
    return VRG_OK;
}

int API(get_labels)(vrg_handle* h, void* outp, int dtype, const int64_t st[3]) {
    if (!h || !outp || !st || dtype < VRG_U8 || dtype > VRG_F64) return fail(h, VRG_E_ARG, "get_labels: bad argument");
    if (be_unpack_labels(h->c, h->c.lab[0], outp, dtype, st)) return fail(h, VRG_E_ARG, "get_labels: unsupported strides");
    return VRG_OK;
}

int API(get_segmented)(vrg_handle* h, int64_t* coords, int64_t cap, int64_t* n) {
    if (!h || !n) return VRG_E_ARG;
    if (!h->inited) return fail(h, VRG_E_STATE, "get_segmented: not initialised");
    int64_t nseg = (int64_t)get_dense(h).n_in;
    *n = nseg;
    if (!coords) return VRG_OK;
    if (cap < nseg) return fail(h, VRG_E_ARG, "get_segmented: buffer too small");
    std::vector<uint64_t> stamps((size_t)nseg + 1);
    std::vector<uint32_t> idxs((size_t)nseg + 1);
    uint32_t got = be_collect_segmented(h->c, 0, stamps.data(), idxs.data(), (uint32_t)nseg);
    if ((int64_t)got != nseg) return fail(h, VRG_E_INTERNAL, "get_segmented: count mismatch");
    std::vector<uint32_t> order(got);
    for (uint32_t i = 0; i < got; i++) order[i] = i;
    // list order of segmentedList: seeds in np.where order (:44), then appended per applied flip-in (:200)
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return stamps[a] < stamps[b]; });
    for (uint32_t i = 0; i < got; i++) idx_to_xyz(h->c, idxs[order[i]], coords + 3 * (size_t)i);
    return VRG_OK;
}

int API(get_band)(vrg_handle* h, int which, int64_t* coords, double* ip, double* op, int64_t cap, int64_t* n) {
    if (!h || !n) return VRG_E_ARG;
    if (!h->inited) return fail(h, VRG_E_STATE, "get_band: not initialised");
    VrgState s = get_state(h);
    int par = s.iter & 1;
    uint32_t cnt = which ? s.no : s.ni, off = which ? s.ni : 0;
    *n = cnt;
    if (!coords && !ip && !op) return VRG_OK;
    if (cap < cnt) return fail(h, VRG_E_ARG, "get_band: buffer too small");
    if (coords) {
        std::vector<uint32_t> idx(cnt + 1);
        be_download(idx.data(), h->c.b_idx[par] + off, (size_t)cnt * 4);
        for (uint32_t i = 0; i < cnt; i++) idx_to_xyz(h->c, idx[i], coords + 3 * (size_t)i);
    }
    if (ip) be_download(ip, h->c.b_ip[par] + off, (size_t)cnt * 8);
    if (op) be_download(op, h->c.b_op[par] + off, (size_t)cnt * 8);
    return VRG_OK;
}

int API(get_trace)(vrg_handle* h, vrg_trace_rec* outp, int64_t cap, int64_t* n) {
    if (!h || !n) return VRG_E_ARG;
    if (!h->inited) return fail(h, VRG_E_STATE, "get_trace: not initialised");
    VrgState s = get_state(h);
    *n = s.iter + 1;
    if (!outp) return VRG_OK;
    if (cap < *n) return fail(h, VRG_E_ARG, "get_trace: buffer too small");
    static_assert(sizeof(vrg_trace_rec) == sizeof(VrgTrace), "trace record layout");
    be_download(outp, h->c.trace, (size_t)(*n) * sizeof(VrgTrace));
    return VRG_OK;
}

int API(get_levels)(vrg_handle* h, double* values, int32_t* hin, int32_t* hout, int32_t* rin, int32_t* rout,
                    int64_t cap, int64_t* n) {
    if (!h || !n) return VRG_E_ARG;
    if (!h->inited) return fail(h, VRG_E_STATE, "get_levels: not initialised");
    uint32_t L = h->c.L;
    *n = L;
    if (!values && !hin && !hout && !rin && !rout) return VRG_OK;
    if (cap < L) return fail(h, VRG_E_ARG, "get_levels: buffer too small");
    if (values) be_download(values, h->c.lev, (size_t)L * 8);
    if (hin) be_download(hin, h->c.hin, (size_t)L * 4);
    if (hout) be_download(hout, h->c.hout, (size_t)L * 4);
    if (rin && rout) {
        int32_t* di = (int32_t*)be_alloc((size_t)L * 4);
        int32_t* dout = (int32_t*)be_alloc((size_t)L * 4);
        if (!di || !dout) return fail(h, VRG_E_MEM, "get_levels");
        be_fill(di, 0, (size_t)L * 4); be_fill(dout, 0, (size_t)L * 4);
        be_recount_hist(h->c, 0, di, dout);
        be_download(rin, di, (size_t)L * 4); be_download(rout, dout, (size_t)L * 4);
        be_free(di); be_free(dout);
    }
    return VRG_OK;
}

int API(set_slab)(vrg_handle* h, int64_t z0, int64_t z1) {
    if (!h) return VRG_E_ARG;
    if (z0 < 0 || z1 > h->c.nz || z0 >= z1) return fail(h, VRG_E_ARG, "set_slab: need 0 <= z0 < z1 <= nz");
    if (h->inited) return fail(h, VRG_E_STATE, "set_slab: call before vrg_init");
    h->c.z0 = (int32_t)z0; h->c.z1 = (int32_t)z1;
    return VRG_OK;
}

int API(comm_unique_id)(void* id128) { return (id128 && be_comm_unique_id(id128) == 0) ? VRG_OK : VRG_E_INTERNAL; }

int API(comm_init)(vrg_handle* h, int nranks, int rank, const void* id128) {
    if (!h || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, VRG_E_ARG, "comm_init: bad argument");
    if (be_comm_init(nranks, rank, id128) != 0) return fail(h, VRG_E_INTERNAL, "comm_init: RCCL communicator could not be created");
    h->c.world = nranks;
    return VRG_OK;
}

int API(set_reduce_callback)(vrg_handle* h, vrg_reduce_fn fn, void* user) {
    if (!h) return VRG_E_ARG;
    h->reduce_fn = fn; h->reduce_user = user;
    if (fn) h->c.world = 2;                          // partials go through the reduction (any value > 1)
    return VRG_OK;
}

}  // extern "C"
