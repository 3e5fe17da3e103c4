// vrg_repl.h - leader / follower replication of a run over several GPUs; part of
// vrg_engine.cpp's translation unit (it uses the handle and the helpers defined there).
//
// WHY NOT Z-SLABS WITH A HALO PLANE.  update() (variationalRegionGrowing.py:156-259) is a chain of ~8 dependent round trips over a few
// thousand band entries and a few dozen flips - 25-30 us on one GPU - and its flip order (:48, :88, :111) is GLOBAL: a label slab
// per GPU would put three dependent xGMI exchanges (flip ranks, the skip rule's fix-point :198, the 2-ring inclusion :177-179)
// inside that chain, on every GPU, every sweep.  What does cost O(V) per sweep is the reference's dense recount (:113-116, :249-250).
// So the work is cut by ROLE and the recount by TIME:
//   rank 0, the LEADER, runs the band chain exactly as one GPU does and logs what every sweep did to the labels: one 16-byte
//       record per place of the sweep's marked list (~100 KB per sweep), one header with the sweep's trace record and the region
//       sizes it produced;
//   every other rank, a FOLLOWER, holds the intensities, the labels and the class bits, applies the log sweep by sweep (a record
//       whose `old` byte is not what the follower holds is an error) and COUNTS the sweeps assigned to it - round robin over the
//       verifiers - with the very dense pass one GPU runs (same kernel, same unit list, same workgroups: the same sums bit for bit),
//       against the sizes in the sweep's header.  With N - 1 verifiers each counts every (N-1)-th sweep over the WHOLE volume at the
//       full-volume pass's efficiency (0.77 of peak at 880x880x640 against 0.57 for an 80-plane slab + a gate per sweep).
//   Small groups (N <= 4) let the leader count its share too (leader_verifies): its chain then runs beside a pass, as on one GPU.
// The log moves once per batch of trips (option "batch"), off the decisions' path, through one of three transports:
//   callback  host buffers through a caller-supplied broadcast (tests: torch.distributed / gloo; any other fabric)
//   rccl      ncclBroadcast of the batch buffer on the leader's transport stream (vrg_comm_init's communicator)
//   ipc       the followers map the leader's batch buffers (hipIpc) and copy them out themselves - device to device, over xGMI
//             between GPUs; a `ready` / `ack` pair of counters in the leader's control block orders it
// At the end of a run every rank holds the same labels, `segmented` order and trace; the intensity sums each verifier filed are
// exchanged with one small all-reduce.
#pragma once

enum { TR_NONE = 0, TR_CALLBACK = 1, TR_RCCL = 2, TR_IPC = 3 };
enum { IPC_READY = 0, IPC_ACK = 8, IPC_SUM_READY = 8 + 64, IPC_SUM_IN = 8 + 64 + 8, IPC_WORDS = 8 + 64 + 8 + 64 };   // 64-bit words of the control block
constexpr size_t IPC_SUMCAP = 3 * 65536 + 16;         // doubles a rank contributes to an all-reduce (3 per sweep of the run + status)

static size_t repl_hb_bytes(uint32_t swcap) { return (sizeof(VrgLogBatch) + (size_t)swcap * sizeof(VrgLogSweep) + 255) / 256 * 256; }
static VrgLogSweep* repl_sw_of(uint8_t* buf) { return reinterpret_cast<VrgLogSweep*>(buf + sizeof(VrgLogBatch)); }
static VrgLogRec* repl_rec_of(const VrgRepl& r, uint8_t* buf) { return reinterpret_cast<VrgLogRec*>(buf + repl_hb_bytes(r.swcap)); }
static int repl_verifiers(const VrgRepl& r) { return r.leader_verifies ? r.nranks : r.nranks - 1; }
static int repl_my_slot(const VrgRepl& r) { return r.leader_verifies ? r.rank : r.rank - 1; }     // (-1: a leader that counts nothing)

static bool repl_alloc_buffers(vrg_handle* h, uint32_t cap) {
    VrgRepl& r = h->repl;
    const size_t bytes = repl_hb_bytes(r.swcap) + (size_t)cap * sizeof(VrgLogRec);
    for (int j = 0; j < 2; j++) {
        uint8_t* p = alloc<uint8_t>(h, bytes);
        if (!p) return false;
        be_fill(h->be, p, 0, repl_hb_bytes(r.swcap));
        if (r.buf[j]) release(h, r.buf[j]);
        r.buf[j] = p;
    }
    r.cap = cap; r.buf_bytes = bytes;
    return true;
}

static uint8_t* repl_host(vrg_handle* h, size_t bytes) {       // the callback transport's page-locked buffer, at least `bytes` long
    VrgRepl& r = h->repl;
    if (bytes > r.host_bytes) {
        if (r.host) be_host_free(h->be, r.host);
        r.host_bytes = pow2_at_least(std::max<size_t>(bytes, 1u << 16));
        r.host = (uint8_t*)be_host_alloc(h->be, r.host_bytes);
        if (!r.host) r.host_bytes = 0;
    }
    return r.host;
}

// bounded polling of a 64-bit counter in device memory (possibly another process's, mapped): true once *word >= want
static bool repl_poll(vrg_handle* h, const uint8_t* base, size_t word, uint64_t want, double timeout_s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        uint64_t v = 0;
        be_repl_copy(h->be, &v, base + 8 * word, 8);
        if (v >= want) return true;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
        if (spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// ---- the leader's side -------------------------------------------------------------------------------------------------------
// before a batch of trips is enqueued: its buffer is free again, the launches' context points at it
static int repl_open_batch(vrg_handle* h, const VrgState& s) {
    VrgRepl& r = h->repl;
    const uint64_t n = r.seq + 1;                      // the batch about to be written
    uint8_t* buf = r.buf[(n - 1) & 1];
    if (r.transport == TR_RCCL) be_repl_wait(h->be);   // (batch n - 2's broadcast has read it)
    if (r.transport == TR_IPC && n > 2)
        for (int q = 1; q < r.nranks; q++)
            if (!repl_poll(h, r.ctl, IPC_ACK + q, n - 2, 120.0)) return fail(h, VRG_E_INTERNAL, "replication: rank " + std::to_string(q) + " did not take batch " + std::to_string(n - 2) + " of the change log");
    VrgCtx& c = h->c;
    c.log_rec = repl_rec_of(r, buf); c.log_sw = repl_sw_of(buf); c.log_cap = r.cap; c.log_swcap = r.swcap;
    c.log_pos0 = s.log_pos; c.log_nsw0 = s.log_nsw;
    return VRG_OK;
}
// after the batch (the band stream is idle: the engine has read the state): its header
static int repl_close_batch(vrg_handle* h, const VrgState& s, bool final, VrgLogBatch& hb) {
    VrgRepl& r = h->repl;
    const VrgCtx& c = h->c;
    std::memset(&hb, 0, sizeof(hb));
    hb.seq = ++r.seq; hb.nsw = s.log_nsw - c.log_nsw0; hb.nrec = s.log_pos - c.log_pos0;
    hb.final = final ? 1 : 0; hb.stop_reason = s.done; hb.iter = s.iter; hb.error = s.error;
    int64_t sizes[2]; be_download(h->be, sizes, c.inc, sizeof(sizes));
    hb.n_in = sizes[0]; hb.n_out = sizes[1]; hb.ni = s.ni; hb.no = s.no; hb.ties = s.ties; hb.near_ties = s.near_ties;
    if (hb.nsw > r.swcap || hb.nrec > r.cap) return fail(h, VRG_E_INTERNAL, "replication: the batch overran its change log");
    r.batches++; r.records += hb.nrec; r.sweeps += hb.nsw;
    return VRG_OK;
}
// ... and off it goes - while the NEXT batch's trips already run: everything here uses the transport stream (the band stream is busy)
static int repl_send(vrg_handle* h, const VrgLogBatch& hb) {
    VrgRepl& r = h->repl;
    const uint64_t n = hb.seq;
    uint8_t* buf = r.buf[(n - 1) & 1];
    be_repl_copy(h->be, buf, &hb, sizeof(hb));
    const size_t hbb = repl_hb_bytes(r.swcap), rb = (size_t)hb.nrec * sizeof(VrgLogRec);
    switch (r.transport) {
    case TR_CALLBACK: {
        uint8_t* hp = repl_host(h, hbb + rb);
        if (!hp) return fail(h, VRG_E_MEM, "replication: host buffer");
        be_repl_copy(h->be, hp, buf, hbb);
        r.bcast(hp, (int64_t)hbb, 0, r.user);
        if (rb) { be_repl_copy(h->be, hp + hbb, buf + hbb, rb); r.bcast(hp + hbb, (int64_t)rb, 0, r.user); }
        break; }
    case TR_RCCL:
        if (be_repl_bcast(h->be, buf, hbb, 0) || (rb && be_repl_bcast(h->be, buf + hbb, rb, 0))) return fail(h, VRG_E_INTERNAL, "replication: RCCL broadcast failed");
        break;
    case TR_IPC:
        be_repl_copy(h->be, r.ctl + 8 * IPC_READY, &n, 8);
        break;
    default: break;
    }
    return VRG_OK;
}
// the trip was handed back because its records would not fit the batch's log: an empty log that is still too small grows
static int repl_log_full(vrg_handle* h, const VrgState& s) {
    VrgRepl& r = h->repl;
    const uint64_t need = (uint64_t)s.nf * 125u;
    if (need <= r.cap) return VRG_OK;                  // (the batch held other sweeps: the next one starts with an empty log)
    if (r.transport == TR_IPC) return fail(h, VRG_E_CAPACITY, "replication: a sweep of " + std::to_string(s.nf) + " flips does not fit the change log (option log_capacity, before vrg_repl_init)");
    if (r.transport == TR_RCCL) be_repl_wait(h->be);
    if (need > 0x20000000ull || !repl_alloc_buffers(h, (uint32_t)pow2_at_least(2 * need))) return fail(h, VRG_E_MEM, "replication: change log");
    return VRG_OK;
}

// ---- a follower's run: apply every batch, count the sweeps that are this rank's --------------------------------------------------
static int repl_receive(vrg_handle* h, VrgLogBatch& hb, std::vector<uint8_t>& hblock, uint8_t*& staging) {
    VrgRepl& r = h->repl;
    const uint64_t n = r.seq + 1;
    const size_t hbb = repl_hb_bytes(r.swcap);
    staging = r.buf[(n - 1) & 1];
    be_follow_wait(h->be, (int)((n - 1) & 1));         // (the kernels that read this staging buffer two batches ago are done)
    hblock.resize(hbb);
    auto grow_for = [&](uint32_t nrec) -> bool {        // a batch larger than the staging buffers: both grow (idle first: kernels may still read them)
        if (nrec <= r.cap) return true;
        be_sync(h->be);
        if (!repl_alloc_buffers(h, (uint32_t)pow2_at_least(nrec))) return false;
        staging = r.buf[(n - 1) & 1];
        return true;
    };
    switch (r.transport) {
    case TR_CALLBACK: {
        r.bcast(hblock.data(), (int64_t)hbb, 0, r.user);
        std::memcpy(&hb, hblock.data(), sizeof(hb));
        if (!grow_for(hb.nrec)) return fail(h, VRG_E_MEM, "replication: staging buffers");
        const size_t rb = (size_t)hb.nrec * sizeof(VrgLogRec);
        if (rb) {
            uint8_t* hp = repl_host(h, rb);
            if (!hp) return fail(h, VRG_E_MEM, "replication: host buffer");
            r.bcast(hp, (int64_t)rb, 0, r.user); be_repl_copy(h->be, staging + hbb, hp, rb);
        }
        break; }
    case TR_RCCL: {
        if (be_repl_bcast(h->be, staging, hbb, 0)) return fail(h, VRG_E_INTERNAL, "replication: RCCL broadcast failed");
        be_repl_wait(h->be);
        be_repl_copy(h->be, hblock.data(), staging, hbb);
        std::memcpy(&hb, hblock.data(), sizeof(hb));
        uint8_t* first = staging;
        if (!grow_for(hb.nrec)) return fail(h, VRG_E_MEM, "replication: staging buffers");
        (void)first;
        const size_t rb = (size_t)hb.nrec * sizeof(VrgLogRec);
        if (rb) { if (be_repl_bcast(h->be, staging + hbb, rb, 0)) return fail(h, VRG_E_INTERNAL, "replication: RCCL broadcast failed"); be_repl_wait(h->be); }
        break; }
    case TR_IPC: {
        if (!repl_poll(h, r.ctl, IPC_READY, n, 300.0)) return fail(h, VRG_E_INTERNAL, "replication: the leader did not publish batch " + std::to_string(n) + " of the change log");
        const uint8_t* src = r.peer_buf[(n - 1) & 1];
        be_repl_copy(h->be, hblock.data(), src, hbb);
        std::memcpy(&hb, hblock.data(), sizeof(hb));
        if (hb.nrec > r.cap) return fail(h, VRG_E_CAPACITY, "replication: the leader's batch does not fit this rank's staging buffer");
        const size_t rb = (size_t)hb.nrec * sizeof(VrgLogRec);
        if (rb) be_repl_copy(h->be, staging + hbb, src + hbb, rb);
        be_repl_copy(h->be, r.ctl + 8 * (IPC_ACK + r.rank), &n, 8);     // (copied out: the leader may write that buffer again)
        break; }
    default: return fail(h, VRG_E_STATE, "replication: no transport set");
    }
    if (hb.seq != n) return fail(h, VRG_E_INTERNAL, "replication: batch " + std::to_string(hb.seq) + " arrived where batch " + std::to_string(n) + " was due");
    r.seq = n;
    r.batches++; r.records += hb.nrec; r.sweeps += hb.nsw;
    return VRG_OK;
}

static int repl_follow(vrg_handle* h, VrgLogBatch& last) {
    VrgRepl& r = h->repl;
    const VrgCtx& c = h->c;
    std::vector<uint8_t> hblock;
    const int nver = repl_verifiers(r), me = repl_my_slot(r);
    for (;;) {
        VrgLogBatch hb; uint8_t* staging = nullptr;
        int rc = repl_receive(h, hb, hblock, staging);
        if (rc) return rc;
        const VrgLogSweep* sw = reinterpret_cast<const VrgLogSweep*>(hblock.data() + sizeof(VrgLogBatch));
        VrgLogRec* recs = repl_rec_of(r, staging);
        // the sweeps up to the next one this rank counts go in one step (the sweeps between two counts: N - 2 of them)
        uint32_t first = 0;
        for (uint32_t i = 0; i < hb.nsw; i++) {
            if ((uint64_t)sw[i].rec0 + sw[i].nrec > hb.nrec) return fail(h, VRG_E_INTERNAL, "replication: a sweep's records lie outside its batch");
            const bool mine = !vrg_dense_skipped((int64_t)sw[i].sweep, h->verify_every, nver, me);
            if (!mine && i + 1 < hb.nsw) continue;
            be_follow_apply(h->be, c, recs, sw + first, (int)(i + 1 - first), mine ? 1 : 0);
            if (mine) { be_follow_count(h->be, c, &h->ev); r.verified++; r.last_verified = sw[i].sweep; }
            first = i + 1;
        }
        be_follow_mark(h->be, (int)((hb.seq - 1) & 1));
        last = hb;
        if (hb.final) break;
    }
    return VRG_OK;
}

// ---- sums over the ranks (trace sums and status at the end of a run) ----------------------------------------------------------------
static int repl_allsum(vrg_handle* h, std::vector<double>& v) {
    VrgRepl& r = h->repl;
    if (r.nranks <= 1) return VRG_OK;
    switch (r.transport) {
    case TR_CALLBACK: r.allsum(v.data(), (int64_t)v.size(), r.user); return VRG_OK;
    case TR_RCCL: {
        double* d = alloc<double>(h, v.size());
        if (!d) return fail(h, VRG_E_MEM, "replication: all-reduce buffer");
        be_upload(h->be, d, v.data(), v.size() * 8);
        const int rc = be_repl_allsum(h->be, d, v.size());
        be_repl_wait(h->be);
        be_download(h->be, v.data(), d, v.size() * 8);
        release(h, d);
        return rc ? fail(h, VRG_E_INTERNAL, "replication: RCCL all-reduce failed") : VRG_OK; }
    case TR_IPC: {
        if (v.size() > IPC_SUMCAP) return fail(h, VRG_E_ARG, "replication: too many sweeps in one run for the all-reduce area");
        const uint64_t round = ++r.sum_round;
        uint8_t* area = r.ctl + 8 * IPC_WORDS;
        if (r.rank != 0) {
            be_upload(h->be, area + (size_t)r.rank * IPC_SUMCAP * 8, v.data(), v.size() * 8);
            be_upload(h->be, r.ctl + 8 * (IPC_SUM_IN + r.rank), &round, 8);
            if (!repl_poll(h, r.ctl, IPC_SUM_READY, round, 300.0)) return fail(h, VRG_E_INTERNAL, "replication: the leader did not finish the all-reduce");
            be_download(h->be, v.data(), area, v.size() * 8);
        } else {
            std::vector<double> part(v.size());
            for (int q = 1; q < r.nranks; q++) {       // (in rank order: the same sums on every run)
                if (!repl_poll(h, r.ctl, IPC_SUM_IN + q, round, 300.0)) return fail(h, VRG_E_INTERNAL, "replication: rank " + std::to_string(q) + " did not contribute to the all-reduce");
                be_download(h->be, part.data(), area + (size_t)q * IPC_SUMCAP * 8, v.size() * 8);
                for (size_t i = 0; i < v.size(); i++) v[i] += part[i];
            }
            be_upload(h->be, area, v.data(), v.size() * 8);
            be_upload(h->be, r.ctl + 8 * IPC_SUM_READY, &round, 8);
        }
        return VRG_OK; }
    default: return fail(h, VRG_E_STATE, "replication: no transport set");
    }
}

// every rank, when the run's sweeps are applied everywhere: the last sweep counted if nobody has (option verify_every), then the
// intensity sums each verifier filed and every rank's status, summed over the ranks; the trace records of sweeps (iter0, iter] patched
static int repl_finish(vrg_handle* h, int32_t iter0, int32_t iter, int64_t n_in, int64_t n_out, int32_t& error) {
    VrgRepl& r = h->repl;
    const VrgCtx& c = h->c;
    const int nver = repl_verifiers(r);
    if (iter > iter0 && !error && (nver < 1 || vrg_dense_skipped(iter, h->verify_every, 1, 0))) {
        // left out by verify_every (or a leader alone that counts nothing): the first verifier counts it after all, so that no run returns unchecked
        if (r.rank == ((r.leader_verifies || r.nranks == 1) ? 0 : 1)) {
            if (r.rank == 0) be_verify_last(h->be, c, nullptr, nullptr);
            else {
                VrgLogSweep w; std::memset(&w, 0, sizeof(w));
                VrgTrace t; be_download(h->be, &t, c.trace + iter, sizeof(t));
                w.nflip = t.nflip; w.nseg = t.nseg; w.n_in = n_in; w.n_out = n_out; w.ni = t.ni; w.no = t.no; w.ties = t.ties; w.near_ties = t.near_ties; w.sweep = (uint32_t)iter;
                be_follow_apply(h->be, c, nullptr, &w, 1, 1);             // (no records: the sweep is applied; its trace record again, the unit list, what to reproduce)
                be_follow_count(h->be, c, nullptr);
            }
        }
    }
    be_sync(h->be);
    int64_t derr = 0; be_download(h->be, &derr, c.dctl + VD_ERR, sizeof(derr));
    if (derr == 12) {
        int64_t f[8]; be_download(h->be, f, c.fexp, sizeof(f));
        std::fprintf(stderr, "[vrg rank %d] sweep %lld: voxel %lld holds label byte %lld, the log says %lld -> %lld\n", r.rank, (long long)f[3], (long long)f[4], (long long)f[5], (long long)f[6], (long long)f[7]);
    }
    if (derr == 5 && r.rank > 0) {                    // (say which sweep, for whoever reads the log of this rank)
        int64_t f[8]; be_download(h->be, f, c.fexp, sizeof(f));
        std::fprintf(stderr, "[vrg rank %d] sweep %lld: counted n_in %lld n_out %lld, the leader filed %lld / %lld\n", r.rank, (long long)f[3], (long long)f[4], (long long)f[5], (long long)f[6], (long long)f[7]);
    }
    const size_t ns = (size_t)std::max(0, iter - iter0);
    std::vector<VrgTrace> tr(ns + 1);
    if (ns) be_download(h->be, tr.data(), c.trace + iter0 + 1, ns * sizeof(VrgTrace));
    std::vector<double> v(3 * ns + 8, 0.0);
    for (size_t i = 0; i < ns; i++)
        if (tr[i].sum_in == tr[i].sum_in) { v[3 * i] = tr[i].sum_in; v[3 * i + 1] = tr[i].sum_out; v[3 * i + 2] = 1.0; }     // (NaN: not counted here)
    v[3 * ns] = (error || derr) ? 1.0 : 0.0;           // ranks in error
    v[3 * ns + 1] = derr == 5 ? 1.0 : 0.0;             // ... whose count disagreed with the leader's sizes
    v[3 * ns + 2] = derr == 12 ? 1.0 : 0.0;            // ... whose labels had drifted from the log's
    v[3 * ns + 3] = (double)iter;                      // (N x iter: every rank ended on the same sweep)
    int rc = repl_allsum(h, v);
    if (rc) return rc;
    for (size_t i = 0; i < ns; i++) {
        if (v[3 * i + 2] > 1.5) return fail(h, VRG_E_INTERNAL, "replication: sweep " + std::to_string(iter0 + 1 + (int)i) + " was counted by more than one rank");
        if (v[3 * i + 2] > 0.5) { tr[i].sum_in = v[3 * i]; tr[i].sum_out = v[3 * i + 1]; }
        else { tr[i].sum_in = std::nan(""); tr[i].sum_out = std::nan(""); }
    }
    if (ns) be_upload(h->be, c.trace + iter0 + 1, tr.data(), ns * sizeof(VrgTrace));
    if (v[3 * ns + 3] != (double)iter * r.nranks) return fail(h, VRG_E_INTERNAL, "replication: the ranks ended on different sweeps");
    if (derr == 5 || v[3 * ns + 1] > 0.5) error = 5;
    else if (derr == 12 || v[3 * ns + 2] > 0.5) error = 12;
    else if (v[3 * ns] > 0.5 && !error) error = 13;     // another rank failed
    return VRG_OK;
}
