// vrg_engine.cpp - handle management and the C-ABI entry points (include/vrg.h) on top of a backend.
// Compiled with hipcc into libvrg_hip.so (backend vrg_device.hip).  tests/hostmodel compiles the same
// file with VRG_API_PREFIX=vrgm_ against the sequential test backend.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
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

// leader / follower replication (vrg_repl.h)
struct VrgRepl {
    int nranks = 0, rank = 0, leader_verifies = 1;      // nranks 0: off
    int transport = 0;                                  // TR_*
    vrg_bcast_fn bcast = nullptr; vrg_allsum_fn allsum = nullptr; void* user = nullptr;
    uint32_t cap = 1u << 20, swcap = 258;               // records / sweep headers a batch buffer holds
    uint8_t* buf[2] = {nullptr, nullptr};               // leader: the two batch buffers it fills in turn; follower: its two staging buffers
    size_t buf_bytes = 0;
    uint64_t seq = 0;                                   // batches published (leader) / taken (follower) since the handle was created
    uint8_t* host = nullptr; size_t host_bytes = 0;     // callback transport: host copy of a batch (page-locked)
    uint8_t* ctl = nullptr;                             // ipc transport: the leader's control block (its own allocation on the leader, mapped on a follower)
    uint8_t* peer_buf[2] = {nullptr, nullptr};          // ... and, on a follower, the leader's two batch buffers, mapped
    bool ctl_mapped = false;
    uint64_t sum_round = 0;
    bool stream = true;                                 // option "repl_stream": the log travels sweep by sweep (0: once per batch, as up to round 5)
    bool open = false;                                  // leader: a batch is opened and not yet closed
    uint32_t sent_sw = 0, sent_rec = 0;                 // leader (rccl / callback): sweeps / records of the open batch that have travelled
    uint64_t chunk_seq = 0;                             // chunks sent since the handle was created
    uint8_t* chunk_dev = nullptr;                       // (a small device scratch area)
    VrgLogChunk* chunk_ring = nullptr; uint32_t chunk_slot = 0;   // rccl: the chunk structs, a ring of 256 in page-locked host memory
    uint64_t* host_ready = nullptr;                     // rccl / callback: the progress word in page-locked HOST memory - the band chain stores it there (a posted write), the leader's
                                                        // host thread reads it without a single HIP call (polling a device word with small copies slowed the chain 4x: every copy is a kernel + cache flush)
    int64_t chunk_min = -1;                             // option "repl_chunk": sweeps a chunk waits for while its batch runs (-1: 8 over RCCL - three broadcasts per chunk -, 1 through callbacks)
    int64_t fault = 0;                                  // option "repl_fault" (tests): n > 0 - the leader fails on the host side when it opens its n-th batch; n < 0 - a follower cannot use its |n|-th chunk
    int32_t failed = 0;                                 // follower: the log could not be used (the run goes on taking chunks, and fails at its end on every rank)
    long long batches = 0, records = 0, sweeps = 0, verified = 0, last_verified = 0, chunks = 0;   // diagnostics
};

struct vrg_handle {
    VrgCtx c;
    VrgRepl repl;
    VrgBackend* be = nullptr;
    std::string err;
    std::vector<void*> owned;
    int device = 0;
    bool have_vol = false, have_lab = false, inited = false;
    bool sync_mode = false;              // trips are driven one at a time from the host (many flips per sweep)
    uint8_t* pool_block = nullptr; uint8_t* marks_block = nullptr;   // the two families of arrays sized by demand: one allocation each (size_pool / size_marks)
    bool probe = false;                  // the next batch is the first after vrg_init: one trip, to learn the sweep's size
    int verify_every = 1;                // option "verify_every"
    int64_t bin_above = 2048;            // option "bin_above": level tables larger than this evaluate their exact densities through bins
    uint32_t nb_alloc = 0;
    int fused = 1;                       // option "fused": sweeps with few flips run update() as ONE launch (k_sweep)
    bool fuse_mode = false;              // ... and the trips being enqueued now are of that kind
    int variant = 0, batch = 8, storage16 = 0, dense_off = 0;
    uint16_t* lev16_buf = nullptr;
    uint32_t* lidx_buf = nullptr; bool lidx_valid = false;   // per-voxel level index of a large level table (VrgCtx::lidx)
    float* I32 = nullptr; double* I64 = nullptr;
    uint64_t band_capacity = 0;
    uint64_t cap_floor = 1u << 16;       // smallest pool / marked-list capacity (tests lower it to exercise the growth paths)
    VrgEvents ev{0, 0.0, 0, 0, 0.0, 0};
    std::chrono::steady_clock::time_point t0;
    int64_t V = 0;
    size_t PVu = 0;
    uint8_t* lab_base[2] = {nullptr, nullptr};
    vrg_reduce_fn reduce_fn = nullptr;
    void* reduce_user = nullptr;
    long long bails[6] = {0, 0, 0, 0, 0, 0};   // how often a trip came back, by VBAIL_* reason
    long long fused_trips = 0;
    long long sync_trips = 0;
    long long data_nonzero = 0;          // np.count_nonzero(dataArray), counted while the volume was packed
};

extern "C" void API(destroy)(vrg_handle* h);

namespace {

int fail(vrg_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

template <class T> T* alloc(vrg_handle* h, size_t n) {
    void* p = be_alloc(h->be, std::max<size_t>(n, 1) * sizeof(T));
    if (p) h->owned.push_back(p);
    return (T*)p;
}
void release(vrg_handle* h, void* p) {
    if (!p) return;
    auto it = std::find(h->owned.begin(), h->owned.end(), p);
    if (it != h->owned.end()) h->owned.erase(it);
    be_free(h->be, p);
}
uint64_t pow2_at_least(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }

VrgDense get_dense(vrg_handle* h) {     // region sizes as the band side keeps them + the sums of the last dense pass
    VrgDense d; be_download(h->be, &d, h->c.dn, sizeof(d));
    int64_t n[2]; be_download(h->be, n, h->c.inc, sizeof(n));
    d.n_in = (double)n[0]; d.n_out = (double)n[1];
    return d;
}
// (every read of the state also tells the backend how large the pool is: it sizes k_band's grid by it)
VrgState get_state(vrg_handle* h) {
    VrgState s; be_download(h->be, &s, h->c.st, sizeof(s));
    be_set_tuning(h->be, "band_hint", s.np);
    be_set_tuning(h->be, "iter_hint", s.iter);
    be_set_tuning(h->be, "flip_hint", s.last_nf);
    be_set_tuning(h->be, "direct_hint", !vrg_tab_pays(h->c.L, s.ni + s.no));
    return s;
}
void put_state(vrg_handle* h, const VrgState& s) {
    be_upload(h->be, h->c.st, &s, sizeof(s));
    const int64_t stop = (s.done || s.bail) ? 1 : 0;       // the dense side's copy of "stopped / handed back"
    be_upload(h->be, h->c.gate + VG_STOP, &stop, sizeof(stop));
}

void idx_to_xyz(const VrgCtx& c, uint32_t idx, int64_t* out) {
    int x, y, z; vrg_coords(c, idx, x, y, z);
    out[0] = x; out[1] = y; out[2] = z;
}

int check_state_error(vrg_handle* h, const VrgState& s) {
    if (const char* be = be_last_error(h->be)) { std::string m = be; be_clear_error(h->be); return fail(h, VRG_E_INTERNAL, m); }
    if (s.error == 1) return fail(h, VRG_E_CAPACITY, "band pool capacity exceeded");
    if (s.error == 2) return fail(h, VRG_E_CAPACITY, "flip list capacity exceeded");
    if (s.error == 4 || s.error == 7) return fail(h, VRG_E_CAPACITY, "marked-voxel list capacity exceeded");
    if (s.error == 11) return fail(h, VRG_E_CAPACITY, "change log capacity exceeded");
    if (s.error == 12) return fail(h, VRG_E_INTERNAL, "replication: a rank's labels have drifted from the leader's change log (code 12)");
    if (s.error == 13) return fail(h, VRG_E_INTERNAL, "replication: another rank of the group failed (code 13)");
    if (s.error) return fail(h, VRG_E_INTERNAL, "internal consistency check failed (code " + std::to_string(s.error) + ")");
    return VRG_OK;
}

// ---- arrays sized by demand ------------------------------------------------------------------------------
// Each family of arrays - the band pool with the flip arrays, the marked-voxel lists - lives in ONE allocation that is carved up: growing a family is one
// allocation, its copies, one synchronisation and one free (an allocation or a free costs 0.1-0.2 ms and a free synchronises the device: 25 arrays grown one
// by one were 5 ms in the middle of a run).
struct Carve {                                       // (base null: a dry run that only adds up the bytes - no arithmetic on a null pointer)
    uintptr_t base; size_t off = 0;
    explicit Carve(uint8_t* b) : base(reinterpret_cast<uintptr_t>(b)) {}
    template <class T> T* take(size_t n) { off = (off + 255) & ~(size_t)255; T* p = base ? reinterpret_cast<T*>(base + off) : nullptr; off += std::max<size_t>(n, 1) * sizeof(T); return p; }
};
struct PoolArrays {
    uint32_t* p_idx; uint32_t* p_lev; double* p_ip; double* p_op; float* p_err; uint64_t* p_key; uint8_t* p_flag; uint32_t* freel;
    uint32_t* flist; uint64_t* f_key; uint32_t* fr_idx; uint32_t* fr_lev; uint32_t* f_slot; uint32_t* f_idx; uint32_t* f_lev; uint8_t* f_res;
    uint32_t* pend; uint32_t* slow; uint32_t* rk_part; uint32_t* fresh; uint64_t* init_key; uint32_t* init_idx;
    size_t carve(uint8_t* base, size_t cap) {
        Carve k(base);
        p_idx = k.take<uint32_t>(cap); p_lev = k.take<uint32_t>(cap); p_ip = k.take<double>(cap); p_op = k.take<double>(cap); p_err = k.take<float>(cap);
        p_key = k.take<uint64_t>(cap); p_flag = k.take<uint8_t>(cap); freel = k.take<uint32_t>(cap);
        flist = k.take<uint32_t>(cap); f_key = k.take<uint64_t>(cap); fr_idx = k.take<uint32_t>(cap); fr_lev = k.take<uint32_t>(cap); f_slot = k.take<uint32_t>(cap);
        f_idx = k.take<uint32_t>(cap); f_lev = k.take<uint32_t>(cap); f_res = k.take<uint8_t>(cap); pend = k.take<uint32_t>(cap); slow = k.take<uint32_t>(cap);
        rk_part = k.take<uint32_t>(cap); fresh = k.take<uint32_t>(cap); init_key = k.take<uint64_t>(cap); init_idx = k.take<uint32_t>(cap);
        return k.off + 256;
    }
};
// band pool + flip arrays (capacity bcap = fcap, a power of two); contents of the first `keep` slots survive
bool size_pool(vrg_handle* h, uint64_t want, uint32_t keep, uint32_t keep_free) {
    VrgCtx& c = h->c;
    uint64_t cap = pow2_at_least(std::max<uint64_t>(want, h->cap_floor));
    if (cap > 0x80000000ull) return false;
    PoolArrays n;
    const size_t bytes = n.carve(nullptr, (size_t)cap);
    uint8_t* blk = alloc<uint8_t>(h, bytes);
    if (!blk) return false;
    n.carve(blk, (size_t)cap);
    if (c.p_idx && keep) {
        be_copy(h->be, n.p_idx, c.p_idx, (size_t)keep * 4); be_copy(h->be, n.p_lev, c.p_lev, (size_t)keep * 4); be_copy(h->be, n.p_ip, c.p_ip, (size_t)keep * 8);
        be_copy(h->be, n.p_op, c.p_op, (size_t)keep * 8); be_copy(h->be, n.p_err, c.p_err, (size_t)keep * 4); be_copy(h->be, n.p_key, c.p_key, (size_t)keep * 8);
        be_copy(h->be, n.p_flag, c.p_flag, (size_t)keep); be_copy(h->be, n.fresh, c.fresh, (size_t)keep * 4);
    }
    if (c.freel && keep_free) be_copy(h->be, n.freel, c.freel, (size_t)keep_free * 4);
    if (cap > keep) be_fill(h->be, n.p_flag + keep, 0, cap - keep);
    be_fill(h->be, n.rk_part, 0, cap * sizeof(uint32_t));
    be_sync(h->be);
    if (h->pool_block) release(h, h->pool_block);
    h->pool_block = blk;
    c.p_idx = n.p_idx; c.p_lev = n.p_lev; c.p_ip = n.p_ip; c.p_op = n.p_op; c.p_err = n.p_err; c.p_key = n.p_key; c.p_flag = n.p_flag; c.freel = n.freel;
    c.flist = n.flist; c.f_key = n.f_key; c.fr_idx = n.fr_idx; c.fr_lev = n.fr_lev; c.f_slot = n.f_slot; c.f_idx = n.f_idx; c.f_lev = n.f_lev; c.f_res = n.f_res;
    c.pend = n.pend; c.slow = n.slow; c.rk_part = n.rk_part; c.fresh = n.fresh; c.init_key = n.init_key; c.init_idx = n.init_idx;
    c.bcap = (uint32_t)cap; c.fcap = c.bcap;
    return true;
}
// marked-voxel list, class-change lists, dead list (capacity mcap); the change lists' contents survive
bool size_marks(vrg_handle* h, uint64_t want, bool keep) {
    VrgCtx& c = h->c;
    uint64_t cap = pow2_at_least(std::max<uint64_t>(want, h->cap_floor));
    if (cap > 0x80000000ull) return false;
    const size_t k = keep ? c.mcap : 0;
    uint32_t* mk_idx; uint8_t* mk_new; uint8_t* mk_old; uint32_t* dead; uint32_t* dw[2]; uint32_t* x[2];
    auto carve = [&](uint8_t* base) -> size_t {
        Carve kk(base);
        mk_idx = kk.take<uint32_t>(cap); mk_new = kk.take<uint8_t>(cap + 16); mk_old = kk.take<uint8_t>(cap + 16); dead = kk.take<uint32_t>(cap);
        for (int p = 0; p < 2; p++) { dw[p] = kk.take<uint32_t>(cap); x[p] = kk.take<uint32_t>(cap); }
        return kk.off + 256;
    };
    const size_t bytes = carve(nullptr);
    uint8_t* blk = alloc<uint8_t>(h, bytes);
    if (!blk) return false;
    carve(blk);
    for (int p = 0; p < 2 && k; p++) { be_copy(h->be, dw[p], c.chg_dw[p], k * 4); be_copy(h->be, x[p], c.chg_x[p], k * 4); }
    be_sync(h->be);
    if (h->marks_block) release(h, h->marks_block);
    h->marks_block = blk;
    c.mk_idx = mk_idx; c.mk_new = mk_new; c.mk_old = mk_old; c.dead = dead;
    for (int p = 0; p < 2; p++) { c.chg_dw[p] = dw[p]; c.chg_x[p] = x[p]; }
    c.mcap = (uint32_t)cap;
    return true;
}

#include "vrg_repl.h"

// vrg_run on a follower: take the leader's batches until the one that ends the run; then the collective finish
int run_follower(vrg_handle* h, const VrgState& s0, vrg_result* out) {
    const VrgCtx& c = h->c;
    const auto t_begin = std::chrono::steady_clock::now();
    const double ms0 = h->ev.ms_total; const long long l0 = h->ev.launches;
    VrgLogBatch last; std::memset(&last, 0, sizeof(last));
    h->repl.failed = 0;
    if (check_state_error(h, s0)) h->repl.failed = 13;  // (a follower that is in error already still takes the log - the run is collective - and fails at its end)
    int rc = repl_follow(h, last);
    if (rc) return rc;
    be_sync(h->be);
    be_events_collect(h->be, &h->ev, 1ll << 60);
    if (h->repl.failed) {                               // this rank could not use the log: the group's closing all-reduce still needs it - every rank then fails
        const std::string why = h->err;
        int32_t error = h->repl.failed;
        rc = repl_finish(h, s0.iter, last.iter, last.n_in, last.n_out, error);
        h->inited = false;                              // (labels half applied: the handle has to be set up again)
        return rc ? rc : fail(h, VRG_E_INTERNAL, why.empty() ? "replication: this rank could not use the leader's change log" : why);
    }
    // the state a leader's run would have left: what the results (trace length, `segmented`, sizes) are read from
    VrgState s = get_state(h);
    s.iter = last.iter; s.done = last.stop_reason; s.ni = last.ni; s.no = last.no; s.ties = last.ties; s.near_ties = last.near_ties;
    s.ties_filed = s.ties; s.near_filed = s.near_ties; s.error = last.error;
    put_state(h, s);
    int64_t sizes[2] = {last.n_in, last.n_out};
    be_upload(h->be, c.inc, sizes, sizeof(sizes));
    const int64_t k = last.iter;
    be_upload(h->be, c.dctl + VD_SEQ, &k, 8); be_upload(h->be, c.dctl + VD_RSEQ, &k, 8); be_upload(h->be, c.gate + VG_REQ, &k, 8);
    int32_t error = last.error;
    rc = repl_finish(h, s0.iter, last.iter, last.n_in, last.n_out, error);
    if (rc) return rc;
    if (error != s.error) { s.error = error; put_state(h, s); }
    rc = check_state_error(h, s);
    if (rc) return rc;
    if (out) {
        std::memset(out, 0, sizeof(*out));
        out->stop_reason = last.stop_reason; out->iter_num = last.iter + 1; out->sweeps = last.iter - s0.iter;
        out->nseg = last.n_in; out->n_in = last.n_in; out->n_out = last.n_out; out->ni = last.ni; out->no = last.no;
        if (last.iter > s0.iter) { VrgTrace t; be_download(h->be, &t, c.trace + last.iter, sizeof(t)); out->sum_in = t.sum_in; out->sum_out = t.sum_out; }
        out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        out->sweep_kernel_ms = h->ev.ms_total - ms0; out->sweep_launches = h->ev.launches - l0;
        out->ties = (int64_t)(uint32_t)(last.ties - s0.ties); out->near_ties = (int64_t)(uint32_t)(last.near_ties - s0.near_ties);
    }
    return VRG_OK;
}

}  // namespace

extern "C" {

int API(create)(int64_t nx, int64_t ny, int64_t nz, int device, vrg_handle** out) {
    if (!out) return VRG_E_ARG;
    *out = nullptr;
    if (nx < 1 || ny < 1 || nz < 1) return VRG_E_ARG;
    int64_t PX = (nx + 2 + 15) / 16 * 16, PY = ny + 4, PZ = nz + 4;
    if ((double)PX * (double)PY * (double)PZ >= 4294960000.0) return VRG_E_ARG;   // 32-bit voxel indices, with room to round up
    VrgBackend* be = be_create(device);
    if (!be) return VRG_E_NOGPU;
    vrg_handle* h = new vrg_handle();
    h->be = be;
    h->device = device;
    std::memset(&h->c, 0, sizeof(VrgCtx));
    VrgCtx& c = h->c;
    c.nx = (int32_t)nx; c.ny = (int32_t)ny; c.nz = (int32_t)nz;
    c.PX = (int32_t)PX; c.PY = (int32_t)PY; c.PZ = (int32_t)PZ;
    c.PV = (uint32_t)(PX * PY * PZ);
    c.z0 = 0; c.z1 = (int32_t)nz;
    h->V = nx * ny * nz;
    const size_t PVu = ((size_t)c.PV + 1023) / 1024 * 1024;   // dense arrays end on a whole 1024-voxel unit
    h->PVu = PVu;
    c.clsb[0] = alloc<uint32_t>(h, PVu / 16); c.clsb[1] = alloc<uint32_t>(h, PVu / 16);
    c.nchg = alloc<uint32_t>(h, 32);
    c.ubits = alloc<uint32_t>(h, PVu / 1024 / 32 + 64);
    c.unew[0] = alloc<uint32_t>(h, PVu / 1024 / 32 + 64); c.unew[1] = alloc<uint32_t>(h, PVu / 1024 / 32 + 64);
    c.ulist = alloc<uint32_t>(h, PVu / 1024 + 64);
    c.uctl = alloc<uint32_t>(h, 64);
    c.vent = alloc<uint32_t>(h, PVu);
    // 16 guard bytes in front: voxel (0,0,0)'s 2-ring reaches 2 bytes before the padded array
    h->lab_base[0] = alloc<uint8_t>(h, (size_t)c.PV + 32);
    c.lab[0] = h->lab_base[0] ? h->lab_base[0] + 16 : nullptr;
    c.stamp = alloc<uint64_t>(h, c.PV);
    c.stb[0] = alloc<VrgState>(h, 1); c.stb[1] = alloc<VrgState>(h, 1);       // (fused trips swap them: vrg_items.h "open-ended sweeps")
    c.st = c.stg = c.stb[0]; c.st_other = c.stb[1]; c.lvl_par = -1;
    c.dn = alloc<VrgDense>(h, 16);                   // own allocation: written by the dense kernel only
    c.counters = alloc<uint32_t>(h, 64);
    c.dbg = alloc<uint64_t>(h, 64 + VRG_DBG_WG * VRG_DBG_PER);
    c.dn_part = alloc<VrgDense>(h, 16);
    c.dn_ring = alloc<VrgDense>(h, VRG_RING); c.exp_ring = alloc<int64_t>(h, 2 * VRG_RING);
    c.stage_in = alloc<VrgDense>(h, VRG_STAGE); c.stage_out = alloc<VrgDense>(h, VRG_STAGE);
    c.incb[0] = alloc<int64_t>(h, 32); c.incb[1] = alloc<int64_t>(h, 32); c.inc = c.inc_in = c.incb[0];
    c.dctl = alloc<int64_t>(h, 32);   // one allocation each: written from different streams
    c.gate = alloc<int64_t>(h, 32);
    c.fexp = alloc<int64_t>(h, 8);
    c.nstat = 4096;
    c.st_nin = alloc<int64_t>(h, c.nstat); c.st_nout = alloc<int64_t>(h, c.nstat);
    c.st_sin = alloc<double>(h, c.nstat); c.st_sout = alloc<double>(h, c.nstat);
    c.trace_cap = 1u << 16;
    c.trace = alloc<VrgTrace>(h, c.trace_cap);
    c.world = 1; c.ver_n = 1; c.ver_me = 0;
    if (!c.lab[0] || !c.stamp || !c.stb[0] || !c.stb[1] || !c.dn || !c.counters || !c.dn_part || !c.gate || !c.fexp || !c.dn_ring || !c.exp_ring || !c.stage_in || !c.stage_out || !c.incb[0] || !c.incb[1] || !c.dctl || !c.clsb[0] || !c.clsb[1] ||
        !c.nchg || !c.ubits || !c.unew[0] || !c.unew[1] || !c.ulist || !c.uctl || !c.vent || !c.st_nin || !c.st_nout || !c.st_sin || !c.st_sout || !c.trace) { API(destroy)(h); return VRG_E_MEM; }
    be_fill(be, c.incb[0], 0, 32 * sizeof(int64_t)); be_fill(be, c.incb[1], 0, 32 * sizeof(int64_t)); be_fill(be, c.dctl, 0, 32 * sizeof(int64_t)); be_fill(be, c.gate, 0, 32 * sizeof(int64_t));
    be_fill(be, c.clsb[0], 0, PVu / 4); be_fill(be, c.clsb[1], 0, PVu / 4);
    be_fill(be, c.nchg, 0, 32 * sizeof(uint32_t));
    be_fill(be, h->lab_base[0], VB_OOB, (size_t)c.PV + 32);
    be_fill(be, c.stb[0], 0, sizeof(VrgState)); be_fill(be, c.stb[1], 0, sizeof(VrgState));
    be_fill(be, c.dn, 0, sizeof(VrgDense));
    be_fill(be, c.dn_part, 0, sizeof(VrgDense));
    be_fill(be, c.dn_ring, 0, VRG_RING * sizeof(VrgDense)); be_fill(be, c.exp_ring, 0, 2 * VRG_RING * sizeof(int64_t));
    be_fill(be, c.stage_in, 0, VRG_STAGE * sizeof(VrgDense)); be_fill(be, c.stage_out, 0, VRG_STAGE * sizeof(VrgDense));
    be_fill(be, c.counters, 0, 64 * sizeof(uint32_t));
    if (c.dbg) be_fill(be, c.dbg, 0, (64 + VRG_DBG_WG * VRG_DBG_PER) * sizeof(uint64_t));
    *out = h;
    return VRG_OK;
}

void API(destroy)(vrg_handle* h) {
    if (!h) return;
    be_sync(h->be);
    if (h->repl.host) be_host_free(h->be, h->repl.host);
    if (h->repl.host_ready) be_host_free(h->be, h->repl.host_ready);
    if (h->repl.chunk_ring) be_host_free(h->be, h->repl.chunk_ring);
    if (h->repl.ctl_mapped) { be_ipc_close(h->be, h->repl.ctl); be_ipc_close(h->be, h->repl.peer_buf[0]); be_ipc_close(h->be, h->repl.peer_buf[1]); }
    for (void* p : h->owned) be_free(h->be, p);
    be_destroy(h->be);
    delete h;
}

const char* API(last_error)(const vrg_handle* h) { return h ? h->err.c_str() : "null handle"; }

int API(set_option)(vrg_handle* h, const char* name, int64_t value) {
    if (!h || !name) return VRG_E_ARG;
    std::string n(name);
    if (n == "band_capacity") { if (h->inited || value < 1) return fail(h, VRG_E_STATE, "band_capacity must be set before vrg_init"); h->band_capacity = (uint64_t)value; }
    else if (n == "capacity_floor") { if (h->c.p_idx || value < 1) return fail(h, VRG_E_STATE, "capacity_floor must be set before the first vrg_init"); h->cap_floor = (uint64_t)value; }
    else if (n == "sweep_variant") { if (h->inited) return fail(h, VRG_E_STATE, "sweep_variant must be set before vrg_init"); h->variant = (int)value; }
    else if (n == "events") h->ev.enabled = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 1 << 20);
    else if (n == "chain_events") h->ev.chain_enabled = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 1 << 20);
    else if (n == "dense_off") h->dense_off = value != 0;   // measurement aid: band chain alone; re-initialise afterwards
    else if (n == "batch") h->batch = (int)std::max<int64_t>(1, value);
    else if (n == "fused") h->fused = value != 0;
    else if (n == "bin_above") { if (value < 0) return fail(h, VRG_E_ARG, "bin_above: a number of levels >= 0"); h->bin_above = value; h->inited = false; }
    else if (n == "verify_every") { if (value < 0) return fail(h, VRG_E_ARG, "verify_every: 0 (never), 1 (every sweep: the default) or n > 1 (every n-th sweep)"); h->verify_every = (int)std::min<int64_t>(value, 1 << 20); be_set_tuning(h->be, name, value); }
    else if (n == "open_sweeps" || n == "mark_compact" || n == "band_blocks_max") be_set_tuning(h->be, name, value);
    else if (n == "sweep_blocks" || n == "prio_mode" || n == "small_flips" || n == "fuse_max" || n == "memo_above" || n == "serial_streams" || n == "skip_excluded" || n == "nt_loads" || n == "dense_pipe") be_set_tuning(h->be, name, value);
    else if (n == "storage16") h->storage16 = value != 0;      // takes effect at the next vrg_init
    else if (n == "repl_fault") h->repl.fault = value;             // tests: a host-side failure in the middle of a replicated run (every rank must return an error, none may hang)
    else if (n == "repl_chunk") h->repl.chunk_min = value;
    else if (n == "repl_stream") h->repl.stream = value != 0;      // any time between runs: 1 (default) the change log travels sweep by sweep; 0: once per batch of trips
    else if (n == "log_capacity") { if (h->repl.buf[0] || value < 1024 || value > 0x20000000ll) return fail(h, VRG_E_STATE, "log_capacity: 1024 .. 2^29 records, before the first vrg_run of a replicated handle"); h->repl.cap = (uint32_t)value; }
    else return fail(h, VRG_E_ARG, "unknown option " + n);
    return VRG_OK;
}

int API(set_volume)(vrg_handle* h, const void* data, int dtype, const int64_t st[3]) {
    if (!h || !data || !st || dtype < VRG_U8 || dtype > VRG_F64) return fail(h, VRG_E_ARG, "set_volume: bad argument");
    VrgCtx& c = h->c;
    // fp32 storage whenever every value is exactly representable (integer dtypes up to 16 bit, float32, and float64 /
    // wider integers that happen to be); float64 storage otherwise (the dense pass then streams 8 B per voxel)
    if (!h->I32) { h->I32 = alloc<float>(h, h->PVu); if (!h->I32) return fail(h, VRG_E_MEM, "set_volume: intensity volume"); be_fill(h->be, h->I32, 0, h->PVu * 4); }
    int inexact = 0;
    int rc = be_pack_volume(h->be, c, h->I32, nullptr, data, dtype, st, &inexact, &h->data_nonzero);
    if (rc) return fail(h, VRG_E_ARG, "set_volume: unsupported strides");
    if (inexact) {
        if (!h->I64) { h->I64 = alloc<double>(h, h->PVu); if (!h->I64) return fail(h, VRG_E_MEM, "set_volume: float64 intensity volume"); be_fill(h->be, h->I64, 0, h->PVu * 8); }
        rc = be_pack_volume(h->be, c, nullptr, h->I64, data, dtype, st, &inexact, &h->data_nonzero);
        if (rc) return fail(h, VRG_E_ARG, "set_volume: unsupported strides");
        c.I = nullptr; c.I64 = h->I64;
    } else { c.I = h->I32; c.I64 = nullptr; }
    h->have_vol = true; h->inited = false;
    if (c.lev) {                                     // distinct-value table and its arrays are rebuilt by the next vrg_init
        release(h, (void*)c.lev); c.lev = nullptr;
        if (c.lev_map) { release(h, (void*)c.lev_map); c.lev_map = nullptr; }
        if (c.ktab) { release(h, (void*)c.ktab); c.ktab = nullptr; }
        h->lidx_valid = false; c.lidx = nullptr;
        release(h, c.hin); release(h, c.hout); release(h, c.dIn); release(h, c.dOut); release(h, c.dConv); release(h, c.ltouch);
        release(h, c.nz_key); release(h, c.nz_val); release(h, c.nz_cin); release(h, c.nz_cout); release(h, c.nz_cconv); release(h, c.tabC);
        c.hin = c.hout = nullptr; c.dIn = c.dOut = c.dConv = c.ltouch = nullptr;
        for (int p = 0; p < 2; p++) { c.dInS[p] = nullptr; c.dOutS[p] = nullptr; c.dConvS[p] = nullptr; }
        c.nz_key = nullptr; c.nz_val = nullptr;
        c.nz_cin = c.nz_cout = c.nz_cconv = nullptr; c.tabC = nullptr; c.L = 0;
    }
    return VRG_OK;
}

int API(set_labels)(vrg_handle* h, const void* labels, int dtype, const int64_t st[3]) {
    if (!h || !labels || !st || dtype < VRG_U8 || dtype > VRG_F64) return fail(h, VRG_E_ARG, "set_labels: bad argument");
    be_sync(h->be);
    be_fill(h->be, h->lab_base[0], VB_OOB, (size_t)h->c.PV + 32);
    int bad = 0;
    int rc = be_pack_labels(h->be, h->c, h->c.lab[0], labels, dtype, st, &bad);
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
    VrgBackend* be = h->be;
    h->t0 = std::chrono::steady_clock::now();       // start_time (:38)
    c.H = H;
    c.A = std::pow(2.0 * M_PI, -0.5);               // A = (2*np.pi)**(-0.5) (:7)
    // levels
    if (!c.lev) {
        double* lev = nullptr; uint32_t L = 0;
        if (be_build_levels(be, c, &lev, &L)) return fail(h, VRG_E_MEM, "vrg_init: level table");
        h->owned.push_back(lev);
        const uint32_t zcap = (uint32_t)pow2_at_least(L);
        int32_t* hin = alloc<int32_t>(h, L); int32_t* hout = alloc<int32_t>(h, L);
        uint32_t* dIn = alloc<uint32_t>(h, 2 * (size_t)L); uint32_t* dOut = alloc<uint32_t>(h, 2 * (size_t)L); uint32_t* dConv = alloc<uint32_t>(h, 2 * (size_t)L);   // (two sets by sweep parity: open-ended sweeps)
        uint32_t* ltouch = alloc<uint32_t>(h, L);
        uint64_t* nz_key = alloc<uint64_t>(h, zcap); double* nz_val = alloc<double>(h, zcap);
        uint32_t* nz_cin = alloc<uint32_t>(h, zcap); uint32_t* nz_cout = alloc<uint32_t>(h, zcap); uint32_t* nz_cconv = alloc<uint32_t>(h, zcap);
        double* tabC = alloc<double>(h, 3 * (size_t)L);
        c.hin = hin; c.hout = hout; c.dIn = dIn; c.dOut = dOut; c.dConv = dConv; c.ltouch = ltouch;
        c.dInS[0] = dIn; c.dInS[1] = dIn ? dIn + L : nullptr; c.dOutS[0] = dOut; c.dOutS[1] = dOut ? dOut + L : nullptr; c.dConvS[0] = dConv; c.dConvS[1] = dConv ? dConv + L : nullptr;
        c.nz_key = nz_key; c.nz_val = nz_val; c.nz_cin = nz_cin; c.nz_cout = nz_cout; c.nz_cconv = nz_cconv; c.tabC = tabC;
        c.zcap = zcap; c.L = L;
        if (!hin || !hout || !dIn || !dOut || !dConv || !ltouch || !nz_key || !nz_val || !nz_cin || !nz_cout || !nz_cconv || !tabC) {
            release(h, lev);                         // c.lev stays null: the next vrg_init starts over
            return fail(h, VRG_E_MEM, "vrg_init: level arrays");
        }
        c.lev = lev;
        // integer-valued levels within a span of 65 536: the direct value -> level map (VrgCtx::lev_map)
        c.lev_map = nullptr; c.lev_min = 0;
        if (L >= 2 && L <= 65535) {
            double ends[2] = {0, 0};
            be_download(be, &ends[0], c.lev, sizeof(double)); be_download(be, &ends[1], c.lev + (L - 1), sizeof(double));
            const double span = ends[1] - ends[0] + 1.0;
            if (ends[0] == std::floor(ends[0]) && span >= 2.0 && span <= 65536.0) {
                uint16_t* map = alloc<uint16_t>(h, (size_t)span);
                if (map) {
                    if (be_build_lev_map(be, c, map, (uint32_t)span)) { c.lev_map = map; c.lev_min = ends[0]; }
                    else release(h, map);
                }
            }
        }
    }
    const uint32_t L = c.L;
    if (L <= (uint32_t)VRG_KTAB_LEVELS) {                // the kernel between every pair of levels (depends on H: rebuilt by every init)
        if (!c.ktab) c.ktab = alloc<double>(h, (size_t)L * L);
        if (!c.ktab) return fail(h, VRG_E_MEM, "vrg_init: kernel table");
        be_build_ktab(be, c, const_cast<double*>(c.ktab));
    }
    // large level tables: the exact densities (:252-255) through bin moments (vrg_items.h "binned exact densities"); the bin width
    // follows from H, so the bins are laid out by every init
    c.nb = 0;
    if ((int64_t)L > h->bin_above) {
        double ends[2] = {0, 0};
        be_download(be, &ends[0], c.lev, sizeof(double)); be_download(be, &ends[1], c.lev + (L - 1), sizeof(double));
        const double hh = VRG_BIN_THETA / std::sqrt(2.0 * VRG_BIN_T * H), nbd = std::floor((ends[1] - ends[0]) / (2.0 * hh)) + 1.0;
        if (H > 0 && nbd >= 1.0 && nbd <= 4194304.0) {    // (more bins than that - a range of > 70 000 kernel widths: the sums over the levels stay)
            const uint32_t nb = (uint32_t)nbd;
            if (nb > h->nb_alloc) {
                if (c.bm_in) release(h, c.bm_in);
                if (c.bm_out) release(h, c.bm_out);
                c.bm_in = alloc<int64_t>(h, (size_t)nb * (VRG_BIN_K + 1)); c.bm_out = alloc<int64_t>(h, (size_t)nb * (VRG_BIN_K + 1));
                h->nb_alloc = (c.bm_in && c.bm_out) ? nb : 0;
                if (!h->nb_alloc) return fail(h, VRG_E_MEM, "vrg_init: bin moments");
            }
            c.nb = nb; c.bin_lo = ends[0]; c.bin_h = hh;
        }
    }
    c.lev16 = nullptr; c.lidx = nullptr;
    if (!h->storage16 && L > (uint32_t)VRG_KTAB_LEVELS) {   // large level table: every voxel's level index once, instead of a search whenever a voxel enters the band
        if (!h->lidx_buf) h->lidx_buf = alloc<uint32_t>(h, h->PVu);
        if (!h->lidx_buf) return fail(h, VRG_E_MEM, "vrg_init: level-index volume");
        if (!h->lidx_valid) { be_build_lidx(be, c, h->lidx_buf); h->lidx_valid = true; }
        c.lidx = h->lidx_buf;
    }
    if (h->storage16) {                             // 16-bit intensity storage: level indices + LDS value table
        if (L > 16384) return fail(h, VRG_E_ARG, "storage16: more than 16384 distinct intensity values");
        if (!h->lev16_buf) h->lev16_buf = alloc<uint16_t>(h, h->PVu);
        if (!h->lev16_buf) return fail(h, VRG_E_MEM, "vrg_init: 16-bit level volume");
        be_build_lev16(be, c, h->lev16_buf);
        c.lev16 = h->lev16_buf;
    }
    c.lab[1] = nullptr;
    if (h->variant & 1) {                           // full-stencil check variant: scratch label volume
        if (!h->lab_base[1]) { h->lab_base[1] = alloc<uint8_t>(h, (size_t)c.PV + 32); if (!h->lab_base[1]) return fail(h, VRG_E_MEM, "vrg_init: scratch labels"); }
        c.lab[1] = h->lab_base[1] + 16;
        be_fill(be, h->lab_base[1], VB_OOB, (size_t)c.PV + 32);
    }
    be_fill(be, c.hin, 0, (size_t)L * 4); be_fill(be, c.hout, 0, (size_t)L * 4);
    be_fill(be, c.dIn, 0, (size_t)L * 8); be_fill(be, c.dOut, 0, (size_t)L * 8); be_fill(be, c.dConv, 0, (size_t)L * 8);
    be_fill(be, c.ltouch, 0, (size_t)L * 4);
    // band pool and work arrays: sized by demand (they grow when a trip reports that it needs more)
    // (unless the caller chose capacities: sized by the volume - a slot costs ~110 bytes, a marked-list entry ~26; re-allocating in the middle of a run costs a
    // host round trip, a copy and the rest of a batch, and a mask with many vessels grows its band by 10^4 entries per sweep.  880x880x640: 4 M slots = 0.45 GB of 288)
    const uint64_t auto_pool = (h->band_capacity || h->cap_floor != (1u << 16)) ? h->band_capacity : std::min<uint64_t>(4u << 20, (uint64_t)h->V / 32u);
    const uint64_t auto_marks = (h->band_capacity || h->cap_floor != (1u << 16)) ? 0 : std::min<uint64_t>(8u << 20, (uint64_t)h->V / 16u);
    if (!c.p_idx && !size_pool(h, auto_pool, 0, 0)) return fail(h, VRG_E_MEM, "vrg_init: band arrays");
    if (!c.mk_idx && !size_marks(h, auto_marks, false)) return fail(h, VRG_E_MEM, "vrg_init: work arrays");
    VrgState s;
    for (int attempt = 0;; attempt++) {
        std::memset(&s, 0, sizeof(s));
        put_state(h, s);
        be_init_band(be, c);
        s = get_state(h);
        const uint64_t need = (uint64_t)s.ninit_in + s.ninit_out;
        if (need <= c.bcap) break;
        // staging overflowed: the label pass is idempotent, so size the pool for what it counted and stage again
        if (attempt || !size_pool(h, 2 * need, 0, 0)) return fail(h, VRG_E_MEM, "vrg_init: band arrays");
    }
    if (s.nseed == 0) return fail(h, VRG_E_EMPTY, "vrg_init: valueMap has no seed (label 0) voxel");
    be_init_sort(be, c, s.ninit_in, s.ninit_out);
    s.ni = s.ninit_in; s.no = s.ninit_out; s.nfresh = s.ni + s.no; s.error = 0;
    put_state(h, s);
    be_fill(be, c.p_flag, 0, c.bcap);
    be_fill(be, c.ubits, 0, (h->PVu / 1024 / 32 + 64) * sizeof(uint32_t));     // rebuilt from the labels by be_init_finish
    be_fill(be, c.unew[0], 0, (h->PVu / 1024 / 32 + 64) * sizeof(uint32_t)); be_fill(be, c.unew[1], 0, (h->PVu / 1024 / 32 + 64) * sizeof(uint32_t));
    be_fill(be, c.uctl, 0, 64 * sizeof(uint32_t));
    be_init_finish(be, c, h->reduce_fn, h->reduce_user);
    s = get_state(h);
    int rc = check_state_error(h, s);
    if (rc) return rc;
    h->inited = true; h->sync_mode = false; h->fuse_mode = false; h->probe = true;
    h->ev.ms_total = 0; h->ev.launches = 0; h->ev.chain_ms_total = 0; h->ev.chain_launches = 0;
    return VRG_OK;
}

int API(run)(vrg_handle* h, int64_t iterMax, int64_t maxSegmentSize, double maxSeconds, vrg_result* out) {
    if (!h) return VRG_E_ARG;
    if (!h->inited) return fail(h, VRG_E_STATE, "vrg_run: call vrg_init first");
    VrgCtx& c = h->c;
    VrgBackend* be = h->be;
    if (iterMax < 0 || iterMax + 1 >= (int64_t)c.trace_cap) return fail(h, VRG_E_ARG, "vrg_run: iterMax out of range");
    VrgState s = get_state(h);
    VrgRepl& rp = h->repl;
    const bool replicated = rp.nranks > 0;
    int rc = VRG_OK;
    if (replicated) {                                   // (checks that come out the same on every rank: the same calls with the same arguments)
        if (rp.transport == TR_NONE && rp.nranks > 1) return fail(h, VRG_E_STATE, "vrg_run: replicated handle without a transport (vrg_repl_set_callbacks / vrg_repl_use_rccl / vrg_repl_ipc_*)");
        if (h->variant & 1) return fail(h, VRG_E_STATE, "vrg_run: the full-stencil check variant keeps no change log");
        if (!rp.buf[0] && !repl_alloc_buffers(h, rp.cap)) return fail(h, VRG_E_MEM, "vrg_run: change log buffers");
        if (rp.rank > 0) return run_follower(h, s, out);
    }
    int32_t iter0 = s.iter;
    // A replicated run is collective: once the followers wait for the log, a leader that cannot go on still has to end the run for them -
    // a final batch with the error set - and to join the closing all-reduce; every rank then returns an error instead of waiting for ever.
    auto abort_group = [&](int code) -> int {
        if (!replicated || rp.nranks <= 1 || rp.transport == TR_NONE) return code;
        const std::string why = h->err;
        be_sync(be);
        VrgState sf = get_state(h);
        if (rp.open || repl_open_batch(h, sf) == VRG_OK) {
            (void)repl_close_batch(h, sf, true, 13);
            int32_t e = 13;
            (void)repl_finish(h, iter0, sf.iter, 0, 0, e);
        }
        h->err = why;
        return code;
    };
    rc = check_state_error(h, s);
    if (rc) return abort_group(rc);
    const uint32_t ties0 = s.ties, near0 = s.near_ties;
    s.done = 0; s.time_up = 0; s.bail = 0; s.iterMax = (int32_t)iterMax; s.maxSegmentSize = maxSegmentSize;
    s.nf = 0; s.npend = 0; s.nmk = 0;                    // counters of a trip that stopped before update()
    put_state(h, s);
    double ms0 = h->ev.ms_total; long long l0 = h->ev.launches;
    double cms0 = h->ev.chain_ms_total; long long cl0 = h->ev.chain_launches;
    auto t_begin = std::chrono::steady_clock::now();
    const bool no_dense = h->dense_off || (replicated && !rp.leader_verifies);      // (a leader that counts nothing enqueues no dense pass at all)
    c.dense_none = (replicated && !rp.leader_verifies) ? 1 : 0;
    const int base_flags = ((h->variant & 1) ? VRG_SWEEP_FULL : 0) | (no_dense ? VRG_SWEEP_NODENSE : 0);
    const uint32_t small = be_small_flip_limit(be), fuse_max = be_fuse_limit(be, c);
    const bool can_fuse = h->fused && !(base_flags & VRG_SWEEP_FULL) && be_fuse_ok(be, c);
    // (a run starts fused when the sweep before - if any - had few flips; the switch is made with the streams idle)
    if (can_fuse && !h->fuse_mode && !h->sync_mode && 2 * (uint64_t)s.last_nf <= fuse_max) { be_fuse_enter(be, c); h->fuse_mode = true; get_state(h); }
    if (!can_fuse) h->fuse_mode = false;
    for (;;) {
        const bool sync = h->sync_mode || (base_flags & VRG_SWEEP_FULL) || be_wants_sync(be, c);
        const bool fuse = h->fuse_mode && !sync;
        int64_t remaining = iterMax - s.iter;
        int nb = sync ? 1 : (int)std::min<int64_t>(h->batch, std::max<int64_t>(remaining, 0) + 1);   // +1: the trip that sets the stop flag
        if (h->probe) nb = 1;                            // (the first trip after vrg_init: nobody knows how many flips the first sweep lists - a mask of many vessels, 10^5 - and a trip
                                                         // that is handed back takes the rest of its batch with it)
        if (replicated) { nb = std::min<int>(nb, (int)rp.swcap - 2); rc = repl_open_batch(h, s); if (rc) return abort_group(rc); }   // (the batch's change log: one buffer)
        if (replicated && rp.fault > 0 && (int64_t)rp.seq + 1 == rp.fault) return abort_group(fail(h, VRG_E_MEM, "vrg_run: injected host-side failure (option repl_fault)"));
        if (maxSeconds >= 0 && s.iter < iterMax) {   // wall-clock cap (:97): tested after the no-flip test, before update()
            double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - h->t0).count();
            if (el >= maxSeconds) { s.time_up = 1; put_state(h, s); nb = 1; }
        }
        int32_t before = s.iter;
        be_sweep_batch(be, c, base_flags | (sync ? VRG_SWEEP_SYNC : 0) | (fuse ? VRG_SWEEP_FUSED : 0), nb, &h->ev, h->reduce_fn, h->reduce_user);
        if (replicated) {                                // while the trips run, what the band chain publishes of their log goes out (rccl / callback; on ipc the followers look themselves)
            do {
                rc = repl_pump(h);
                if (rc < 0) return abort_group(rc);
                if (rc == 0 && be_band_busy(be)) std::this_thread::sleep_for(std::chrono::microseconds(15));
            } while (be_band_busy(be));
        }
        if (sync) h->sync_trips++;
        if (fuse) h->fused_trips += nb;
        s = get_state(h);
        be_events_collect(be, &h->ev, s.iter - before);
        if (replicated) {                                // the batch is complete: its header, and whatever of it has not travelled yet (its last sweep at least)
            rc = repl_close_batch(h, s, s.done || s.error, 0);
            if (rc) return abort_group(rc);
        }
        if (s.done || s.error) break;
        if (s.bail) {                                // the trip was handed back untouched: make room / change mode, do it again
            const uint64_t nf = s.nf;
            h->bails[std::min(s.bail, 5)]++;
            h->probe = false;
            // The rest of the batch was enqueued behind the trip that came back: its k_gate + recount pairs may still sit
            // in the dense stream.  They have to run out while the device's stop word (gate[VG_STOP]) is still set -
            // put_state below clears it; a leftover gate would then wait for the NEXT sweep's request and shift which
            // launch counts which sweep (and, on Z-slabs, let ranks pack different numbers of recounts into one all-reduce).
            be_sync(be);
            if (s.bail == VBAIL_FLIPS) {
                // more flips than the trip's launches were sized for: up to the device-resident limit the trip is simply enqueued again
                // with launches sized for them (the chip-wide ordering kernels); beyond it the host drives the trips
                if (nf <= small) be_set_tuning(be, "flip_hint_min", (long long)nf);
                else h->sync_mode = true;
            }
            else if (s.bail == VBAIL_FUSE) h->fuse_mode = false;
            else if (s.bail == VBAIL_LOG) { rc = repl_log_full(h, s); if (rc) return abort_group(rc); }
            else if (s.bail == VBAIL_MARKS) {
                if (nf * 125u > 0x3fffffffull || !size_marks(h, 2 * nf * 125u, true)) return abort_group(fail(h, VRG_E_MEM, "vrg_run: marked-voxel arrays"));
            } else {
                if (!size_pool(h, 2 * ((uint64_t)s.np + nf * 27u), s.np, s.nfree)) return abort_group(fail(h, VRG_E_MEM, "vrg_run: band arrays"));
            }
            // (whatever the reason, the trip's flip count is known now: everything that count implies is settled in THIS round trip - the kind of trip, the arrays'
            // sizes - instead of one hand-back per discovery: a first sweep of 2*10^5 flips used to come back four times, each time with the rest of its batch to drain)
            if (nf > fuse_max) h->fuse_mode = false;
            if (nf > small) h->sync_mode = true;
            else if (nf > 0 && !h->sync_mode) be_set_tuning(be, "flip_hint_min", (long long)nf);
            if (!(base_flags & VRG_SWEEP_FULL) && nf * 125u > c.mcap && nf * 125u <= 0x3fffffffull) {       // (counted as what it is: the marked-voxel arrays grew)
                if (s.bail != VBAIL_MARKS) h->bails[VBAIL_MARKS]++;
                if (!size_marks(h, nf * 125u + nf * 32u, true)) return abort_group(fail(h, VRG_E_MEM, "vrg_run: marked-voxel arrays"));
            }
            if ((uint64_t)s.np + nf * 27u > c.bcap) {
                if (s.bail != VBAIL_POOL) h->bails[VBAIL_POOL]++;
                if (!size_pool(h, 2 * ((uint64_t)s.np + nf * 27u), s.np, s.nfree)) return abort_group(fail(h, VRG_E_MEM, "vrg_run: band arrays"));
            }
            s.bail = 0; s.nf = 0;
            s.ties = s.ties_filed; s.near_ties = s.near_filed;   // the trip's sign tests are made again: count them once
            put_state(h, s);
            continue;
        }
        h->probe = false;
        if (h->sync_mode && 2 * (uint64_t)s.last_nf <= small) h->sync_mode = false;   // the flips fit one workgroup again
        if (can_fuse && !h->fuse_mode && !h->sync_mode && 2 * (uint64_t)s.last_nf <= fuse_max) {   // ... or one fused launch
            be_sync(be);
            be_fuse_enter(be, c); h->fuse_mode = true;
            get_state(h);                                // (k_band's grid is sized for corrections evaluated entry by entry)
        }
    }
    if (!no_dense) be_dense_flush(be, c, h->reduce_fn, h->reduce_user);   // Z-slabs: close the passes still waiting for their all-reduce
    be_sync(be);
    // passes were left out (option verify_every): the run's last sweep is counted after all, so that the sizes kept by
    // increments never leave a run unchecked
    if (!replicated && !h->dense_off && h->verify_every != 1 && s.iter > iter0 && !s.error) be_verify_last(be, c, h->reduce_fn, h->reduce_user);
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    int64_t dense_err = 0;                           // raised by the dense stream, possibly after the band side stopped
    be_download(be, &dense_err, c.dctl + VD_ERR, sizeof(dense_err));
    if (h->dense_off) { dense_err = 0; h->inited = false; }   // the dense pass sequence is broken on purpose: init again
    if (dense_err) s.error = (int32_t)dense_err;
    if (replicated) {                                // the sums the verifiers filed, every rank's status (collective)
        int64_t sizes[2]; be_download(be, sizes, c.inc, sizeof(sizes));
        rc = repl_finish(h, iter0, s.iter, sizes[0], sizes[1], s.error);
        if (rc) return rc;
        secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    }
    rc = check_state_error(h, s);
    if (rc) return rc;
    if (out) {
        out->stop_reason = s.done; out->iter_num = s.iter + 1; out->sweeps = s.iter - iter0;
        VrgDense d = get_dense(h);
        out->nseg = (int64_t)d.n_in; out->n_in = (int64_t)d.n_in; out->n_out = (int64_t)d.n_out; out->ni = s.ni; out->no = s.no;
        out->sum_in = d.sum_in; out->sum_out = d.sum_out; out->seconds = secs;
        if (replicated && s.iter > iter0) { VrgTrace t; be_download(be, &t, c.trace + s.iter, sizeof(t)); out->sum_in = t.sum_in; out->sum_out = t.sum_out; }
        out->sweep_kernel_ms = h->ev.ms_total - ms0; out->sweep_launches = h->ev.launches - l0;
        out->chain_kernel_ms = h->ev.chain_ms_total - cms0; out->chain_launches = h->ev.chain_launches - cl0;
        out->ties = (int64_t)(uint32_t)(s.ties - ties0); out->near_ties = (int64_t)(uint32_t)(s.near_ties - near0);
    }
    return VRG_OK;
}

int API(get_labels)(vrg_handle* h, void* outp, int dtype, const int64_t st[3]) {
    if (!h || !outp || !st || dtype < VRG_U8 || dtype > VRG_F64) return fail(h, VRG_E_ARG, "get_labels: bad argument");
    if (be_unpack_labels(h->be, h->c, h->c.lab[0], outp, dtype, st, 0)) return fail(h, VRG_E_ARG, "get_labels: unsupported strides");
    return VRG_OK;
}

int API(get_segmented_map)(vrg_handle* h, void* outp, int dtype, const int64_t st[3]) {
    if (!h || !outp || !st || dtype < VRG_U8 || dtype > VRG_F64) return fail(h, VRG_E_ARG, "get_segmented_map: bad argument");
    if (be_unpack_labels(h->be, h->c, h->c.lab[0], outp, dtype, st, 1)) return fail(h, VRG_E_ARG, "get_segmented_map: unsupported strides");
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
    uint32_t got = be_collect_segmented(h->be, h->c, stamps.data(), idxs.data(), (uint32_t)nseg);
    if ((int64_t)got != nseg) return fail(h, VRG_E_INTERNAL, "get_segmented: count mismatch");
    std::vector<uint32_t> order(got);
    for (uint32_t i = 0; i < got; i++) order[i] = i;
    // list order of segmentedList: seeds in np.where order (:44), then appended per applied flip-in (:200)
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return stamps[a] < stamps[b]; });
    for (uint32_t i = 0; i < got; i++) idx_to_xyz(h->c, idxs[order[i]], coords + 3 * (size_t)i);
    return VRG_OK;
}

// innerBndList (which = 0) / outerBndList (1) in the reference's list order: the pool's live slots of that list, by key
int API(get_band)(vrg_handle* h, int which, int64_t* coords, double* ip, double* op, int64_t cap, int64_t* n) {
    if (!h || !n) return VRG_E_ARG;
    if (!h->inited) return fail(h, VRG_E_STATE, "get_band: not initialised");
    VrgState s = get_state(h);
    uint32_t cnt = which ? s.no : s.ni;
    *n = cnt;
    if (!coords && !ip && !op) return VRG_OK;
    if (cap < cnt) return fail(h, VRG_E_ARG, "get_band: buffer too small");
    const uint32_t np = s.np;
    std::vector<uint8_t> flag(np + 1); std::vector<uint64_t> key(np + 1);
    be_download(h->be, flag.data(), h->c.p_flag, np); be_download(h->be, key.data(), h->c.p_key, (size_t)np * 8);
    std::vector<uint32_t> sel; sel.reserve(cnt);
    const uint8_t want = (uint8_t)(PF_ALIVE | (which ? 0 : PF_INNER));
    for (uint32_t i = 0; i < np; i++) if ((flag[i] & (PF_ALIVE | PF_INNER)) == want) sel.push_back(i);
    if (sel.size() != cnt) return fail(h, VRG_E_INTERNAL, "get_band: list length differs from the pool's live slots");
    std::sort(sel.begin(), sel.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    std::vector<uint32_t> idx(np + 1); std::vector<double> v(np + 1);
    if (coords) {
        be_download(h->be, idx.data(), h->c.p_idx, (size_t)np * 4);
        for (uint32_t i = 0; i < cnt; i++) idx_to_xyz(h->c, idx[sel[i]], coords + 3 * (size_t)i);
    }
    if (ip) { be_download(h->be, v.data(), h->c.p_ip, (size_t)np * 8); for (uint32_t i = 0; i < cnt; i++) ip[i] = v[sel[i]]; }
    if (op) { be_download(h->be, v.data(), h->c.p_op, (size_t)np * 8); for (uint32_t i = 0; i < cnt; i++) op[i] = v[sel[i]]; }
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
    be_sync(h->be);                                  // the dense stream files the intensity sums
    be_download(h->be, outp, h->c.trace, (size_t)(*n) * sizeof(VrgTrace));
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
    if (values) be_download(h->be, values, h->c.lev, (size_t)L * 8);
    if (hin) be_download(h->be, hin, h->c.hin, (size_t)L * 4);
    if (hout) be_download(h->be, hout, h->c.hout, (size_t)L * 4);
    if (rin && rout) {
        int32_t* di = (int32_t*)be_alloc(h->be, (size_t)L * 4);
        int32_t* dout = (int32_t*)be_alloc(h->be, (size_t)L * 4);
        if (!di || !dout) return fail(h, VRG_E_MEM, "get_levels");
        be_fill(h->be, di, 0, (size_t)L * 4); be_fill(h->be, dout, 0, (size_t)L * 4);
        be_recount_hist(h->be, h->c, di, dout);
        be_download(h->be, rin, di, (size_t)L * 4); be_download(h->be, rout, dout, (size_t)L * 4);
        const long long bad = be_check_bins(h->be, h->c, di, dout);     // (with bins: the moments against ones built from the recount - integers: exactly equal)
        be_free(h->be, di); be_free(h->be, dout);
        if (bad) return fail(h, VRG_E_INTERNAL, "get_levels: " + std::to_string(bad) + " bin moment word(s) differ from the dense recount");
    }
    return VRG_OK;
}

// how the run went (diagnostics): out[0..3] = trips handed back {-, flips, marks, pool}, out[4] = host-driven trips,
// out[5] = pool capacity, out[6] = marked-list capacity, out[7] = pool slots in use; with cap >= 9 also
// out[8] = bytes one dense pass requests from memory with the current labels (0 before init); with cap >= 14 also how the
// dense pass is launched: out[9] = non-temporal loads, out[10] = storage (0 fp32, 1 u16 level index, 2 f64), out[11] =
// workgroups, out[12] = skip_excluded, out[13] = listed units
int API(get_stats)(vrg_handle* h, int64_t* outp, int64_t cap) {
    if (!h || !outp || cap < 8) return VRG_E_ARG;
    for (int i = 0; i < 4; i++) outp[i] = h->bails[i];
    outp[4] = h->sync_trips; outp[5] = h->c.bcap; outp[6] = h->c.mcap;
    outp[7] = h->inited ? get_state(h).np : 0;
    if (cap >= 9) {
        outp[8] = 0;
        if (h->inited) { be_sync(h->be); outp[8] = (int64_t)be_dense_bytes(h->be, h->c); }
    }
    if (cap >= 17) { outp[15] = h->fused_trips; outp[16] = h->bails[4]; }
    if (cap >= 18) outp[17] = h->inited ? h->c.nb : 0;
    if (cap >= 19) outp[18] = be_memo_trips(h->be);
    if (cap >= 20) outp[19] = h->data_nonzero;
    if (cap >= 23) outp[22] = h->inited ? be_slow_flips(h->be, h->c) : 0;
    // what a large level table costs in device memory: the bin moments (two classes x nb_alloc bins x 9 words) and the per-voxel level index
    if (cap >= 22) {
        outp[20] = (int64_t)h->nb_alloc * (VRG_BIN_K + 1) * 8 * 2;
        outp[21] = (h->lidx_buf ? (int64_t)h->PVu * 4 : 0) + (h->lev16_buf ? (int64_t)h->PVu * 2 : 0);
    }
    if (cap >= 14) {
        int64_t di[5] = {0, 0, 0, 0, 0};
        uint32_t uc[2] = {0, 0};
        if (h->inited) { be_dense_info(h->be, h->c, di); be_download(h->be, uc, h->c.uctl, sizeof(uc)); }
        outp[9] = di[0]; outp[10] = di[1]; outp[11] = di[2]; outp[12] = di[3]; outp[13] = uc[0];
        if (cap >= 15) outp[14] = di[4];
    }
    return VRG_OK;
}

// diagnostic build (-DVRG_STAMPS): the 64 in-kernel time stamps of the last sweep (100-MHz ticks); zeros otherwise
int API(debug_stamps)(vrg_handle* h, uint64_t* out64) {
    if (!h || !out64 || !h->c.dbg) return VRG_E_ARG;
    be_sync(h->be);
    be_download(h->be, out64, h->c.dbg, 64 * sizeof(uint64_t));
    return VRG_OK;
}
// ... and the per-workgroup stamps of k_mark_relabel's last launch: VRG_DBG_PER words for each of VRG_DBG_WG workgroups (tools/mark_stamps.py)
int API(debug_stamps_wide)(vrg_handle* h, uint64_t* out, int64_t cap) {
    if (!h || !out || !h->c.dbg || cap < (int64_t)(VRG_DBG_WG * VRG_DBG_PER)) return VRG_E_ARG;
    be_sync(h->be);
    be_download(h->be, out, h->c.dbg + 64, (size_t)VRG_DBG_WG * VRG_DBG_PER * sizeof(uint64_t));
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
    if (be_comm_init(h->be, nranks, rank, id128) != 0) {
        const char* m = be_last_error(h->be);
        std::string msg = m ? m : "comm_init: RCCL communicator could not be created";
        be_clear_error(h->be);
        return fail(h, VRG_E_INTERNAL, msg);
    }
    if (!h->repl.nranks) h->c.world = nranks;         // (a rank of a leader / follower group counts whole volumes: it is not a Z-slab rank, whatever carries its log)
    return VRG_OK;
}

int API(repl_init)(vrg_handle* h, int nranks, int rank, int leader_verifies) {
    if (!h || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return fail(h, VRG_E_ARG, "repl_init: need 1 <= nranks <= 64, 0 <= rank < nranks");
    if (h->inited) return fail(h, VRG_E_STATE, "repl_init: call before vrg_init");
    if (h->reduce_fn || h->c.world > 1) return fail(h, VRG_E_STATE, "repl_init: the handle is already a Z-slab rank");
    VrgRepl& r = h->repl;
    r.nranks = nranks; r.rank = rank; r.leader_verifies = (leader_verifies || nranks == 1) ? 1 : 0;
    if (nranks == 1 && !leader_verifies) r.leader_verifies = 0;      // (measurement aid: a leader alone that counts nothing; its last sweep is still counted when a run ends)
    h->c.ver_n = std::max(1, repl_verifiers(r));
    h->c.ver_me = rank == 0 ? (r.leader_verifies ? 0 : -1) : -1;     // (a follower's own dense stream stays idle: it counts in its apply stream)
    be_set_tuning(h->be, "repl", 1);
    return VRG_OK;
}

int API(repl_set_callbacks)(vrg_handle* h, vrg_bcast_fn bcast, vrg_allsum_fn allsum, void* user) {
    if (!h || !bcast || !allsum) return fail(h, VRG_E_ARG, "repl_set_callbacks: both callbacks are needed");
    if (!h->repl.nranks) return fail(h, VRG_E_STATE, "repl_set_callbacks: vrg_repl_init first");
    h->repl.bcast = bcast; h->repl.allsum = allsum; h->repl.user = user; h->repl.transport = TR_CALLBACK;
    h->c.world = 1;                                   // (a vrg_comm_init before this call may have marked the handle as a Z-slab rank)
    return VRG_OK;
}

int API(repl_use_rccl)(vrg_handle* h) {
    if (!h) return VRG_E_ARG;
    if (!h->repl.nranks) return fail(h, VRG_E_STATE, "repl_use_rccl: vrg_repl_init first");
    double probe = 0.0;                               // (is there a communicator?  a one-word all-reduce says so - collective like everything here)
    double* d = alloc<double>(h, 1);
    if (!d) return fail(h, VRG_E_MEM, "repl_use_rccl");
    be_upload(h->be, d, &probe, 8);
    const int rc = be_repl_allsum(h->be, d, 1);
    be_repl_wait(h->be);
    release(h, d);
    if (rc) { be_clear_error(h->be); return fail(h, VRG_E_STATE, "repl_use_rccl: no RCCL communicator (vrg_comm_init first)"); }
    h->c.world = 1;                                   // (vrg_comm_init marks the handle as a Z-slab rank: it is not - every rank counts whole volumes)
    h->repl.transport = TR_RCCL;
    return VRG_OK;
}

// blob: [ctl handle 64][buf0 handle 64][buf1 handle 64][cap u32][swcap u32]
int API(repl_ipc_export)(vrg_handle* h, void* blob, int64_t cap, int64_t* bytes) {
    if (!h || !blob || !bytes || cap < 256) return fail(h, VRG_E_ARG, "repl_ipc_export: need a blob of at least 256 bytes");
    VrgRepl& r = h->repl;
    if (!r.nranks || r.rank != 0) return fail(h, VRG_E_STATE, "repl_ipc_export: the leader (rank 0) of a replicated handle exports");
    if (!r.buf[0] && !repl_alloc_buffers(h, r.cap)) return fail(h, VRG_E_MEM, "repl_ipc_export: change log buffers");
    if (!repl_alloc_ctl(h)) return fail(h, VRG_E_MEM, "repl_ipc_export: control block");
    uint8_t* o = (uint8_t*)blob;
    std::memset(o, 0, 256);
    if (be_ipc_export(h->be, r.ctl, o) || be_ipc_export(h->be, r.buf[0], o + 64) || be_ipc_export(h->be, r.buf[1], o + 128))
        return fail(h, VRG_E_INTERNAL, "repl_ipc_export: hipIpcGetMemHandle failed");
    std::memcpy(o + 192, &r.cap, 4); std::memcpy(o + 196, &r.swcap, 4);
    *bytes = 256;
    r.transport = TR_IPC; h->c.world = 1;
    return VRG_OK;
}

int API(repl_ipc_import)(vrg_handle* h, const void* blob, int64_t bytes) {
    if (!h || !blob || bytes < 256) return fail(h, VRG_E_ARG, "repl_ipc_import: the leader's 256-byte blob");
    VrgRepl& r = h->repl;
    if (!r.nranks || r.rank == 0) return fail(h, VRG_E_STATE, "repl_ipc_import: a follower (rank > 0) of a replicated handle imports");
    if (r.ctl_mapped) return fail(h, VRG_E_STATE, "repl_ipc_import: already mapped");
    const uint8_t* o = (const uint8_t*)blob;
    uint32_t cap = 0, swcap = 0; std::memcpy(&cap, o + 192, 4); std::memcpy(&swcap, o + 196, 4);
    if (swcap != r.swcap) return fail(h, VRG_E_ARG, "repl_ipc_import: the leader's log layout differs from this library's");
    r.ctl = (uint8_t*)be_ipc_open(h->be, o);
    r.peer_buf[0] = (uint8_t*)be_ipc_open(h->be, o + 64); r.peer_buf[1] = (uint8_t*)be_ipc_open(h->be, o + 128);
    if (!r.ctl || !r.peer_buf[0] || !r.peer_buf[1]) return fail(h, VRG_E_INTERNAL, "repl_ipc_import: hipIpcOpenMemHandle failed");
    r.ctl_mapped = true;
    if (cap != r.cap) { if (r.buf[0]) return fail(h, VRG_E_STATE, "repl_ipc_import: staging buffers already sized"); r.cap = cap; }
    r.transport = TR_IPC; h->c.world = 1;
    return VRG_OK;
}

int API(repl_stats)(vrg_handle* h, int64_t* out, int64_t cap) {
    if (!h || !out || cap < 8) return VRG_E_ARG;
    const VrgRepl& r = h->repl;
    out[0] = r.batches; out[1] = r.records; out[2] = r.sweeps; out[3] = r.verified; out[4] = r.last_verified; out[5] = r.transport;
    out[6] = r.nranks ? repl_verifiers(r) : 0; out[7] = r.nranks ? repl_my_slot(r) : 0;
    if (cap >= 9) out[8] = r.chunks;
    return VRG_OK;
}

int API(set_reduce_callback)(vrg_handle* h, vrg_reduce_fn fn, void* user) {
    if (!h) return VRG_E_ARG;
    h->reduce_fn = fn; h->reduce_user = user;
    if (fn) h->c.world = 2;                          // partials go through the reduction (any value > 1)
    return VRG_OK;
}

}  // extern "C"
