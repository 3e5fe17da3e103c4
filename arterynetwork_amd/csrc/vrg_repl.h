// vrg_repl.h - leader / follower replication of a run over several GPUs; part of
// vrg_engine.cpp's translation unit (it uses the handle and the helpers defined there).
//
// WHY NOT Z-SLABS WITH A HALO PLANE.  update() (variationalRegionGrowing.py:156-259) is a chain of ~8 dependent round trips over a few
// thousand band entries and a few dozen flips - 25-30 us on one GPU - and its flip order (:48, :88, :111) is GLOBAL: a label slab
// per GPU would put three dependent xGMI exchanges (flip ranks, the skip rule's fix-point :198, the 2-ring inclusion :177-179)
// inside that chain, on every GPU, every sweep.  What does cost O(V) per sweep is the reference's dense recount (:113-116, :249-250).
// So the work is cut by ROLE and the recount by TIME:
//   rank 0, the LEADER, runs the band chain exactly as one GPU does and logs what every sweep did to the labels: one 16-byte
//       record per label byte that changes, one header with the sweep's trace record and the region sizes it produced;
//   every other rank, a FOLLOWER, holds the intensities, the labels and the class bits, applies the log sweep by sweep (a record
//       whose `old` byte is not what the follower holds is an error) and COUNTS the sweeps assigned to it - round robin over the
//       verifiers - with the very dense pass one GPU runs (same kernel, same unit list, same workgroups: the same sums bit for bit),
//       against the sizes in the sweep's header.  With N - 1 verifiers each counts every (N-1)-th sweep over the WHOLE volume at the
//       full-volume pass's efficiency (0.77 of peak at 880x880x640 against 0.57 for an 80-plane slab + a gate per sweep).
//   Small groups (N <= 3) let the leader count its share too (leader_verifies): its chain then runs beside a pass, as on one GPU.
//
// THE LOG TRAVELS SWEEP BY SWEEP (round 6; it used to move once per batch of trips, so a follower started a batch late and the run ended
// a batch late).  The trips are still enqueued in batches (option "batch") into one of two buffers, but the leader's band chain PUBLISHES
// how far the batch's log is complete in a 64-bit progress word (VrgCtx::log_ready, vrg_log_publish): the first kernel of trip k+1's update()
// (k_sweep, k_order, k_trip_open) does it for sweep k - its records and its header were written by kernels that have ended.  Only the last
// sweep of a batch waits for the host, which closes the batch (VrgLogBatch) when its trips are done.  Three transports:
//   ipc       the followers map the leader's batch buffers and control block (hipIpc) and poll the progress word themselves; what is new
//             they copy out - headers to the host, records device to device, over xGMI between GPUs; an `ack` word per follower tells
//             the leader when a buffer may be written again
//   rccl      the progress word lives in page-locked HOST memory: the band chain posts it there, the leader's host thread reads it without
//             touching the GPU and broadcasts what is new - eight sweeps or more - as a CHUNK (VrgLogChunk, the sweep headers, the
//             records: ncclBroadcast on the transport stream, straight out of the batch buffer)
//   callback  the same chunks through a caller-supplied broadcast of host buffers (tests: torch.distributed / gloo; any other fabric)
// A follower's lag is one poll + one copy, not one batch.  At the end of a run every rank holds the same labels, `segmented` order and
// trace; the intensity sums each verifier filed are exchanged with one small all-reduce.
//
// A replicated vrg_run is COLLECTIVE, and so are its failures: a leader that fails on the host side (allocation, transport, a state in
// error) still closes and publishes a final batch with the error set and joins the closing all-reduce; a follower that meets a log it
// cannot use records its error, keeps taking chunks until the final one, and joins it too - every rank returns, none waits for ever.
#pragma once

enum { TR_NONE = 0, TR_CALLBACK = 1, TR_RCCL = 2, TR_IPC = 3 };
// 64-bit words of the control block (leader's device memory): batches closed; the open batch's progress word; per-rank acks; the all-reduce area's flags
enum { IPC_READY = 0, IPC_SW = 1, IPC_ACK = 8, IPC_SUM_READY = 8 + 64, IPC_SUM_IN = 8 + 64 + 8, IPC_WORDS = 8 + 64 + 8 + 64 };
constexpr size_t IPC_SUMCAP = 3 * 65536 + 16;         // doubles a rank contributes to an all-reduce (3 per sweep of the run + status)

static size_t repl_hb_bytes(uint32_t swcap) { return (sizeof(VrgLogBatch) + (size_t)swcap * sizeof(VrgLogSweep) + 255) / 256 * 256; }
static VrgLogSweep* repl_sw_of(uint8_t* buf) { return reinterpret_cast<VrgLogSweep*>(buf + sizeof(VrgLogBatch)); }
static VrgLogRec* repl_rec_of(const VrgRepl& r, uint8_t* buf) { return reinterpret_cast<VrgLogRec*>(buf + repl_hb_bytes(r.swcap)); }
static int repl_verifiers(const VrgRepl& r) { return r.leader_verifies ? r.nranks : r.nranks - 1; }
static int repl_my_slot(const VrgRepl& r) { return r.leader_verifies ? r.rank : r.rank - 1; }     // (-1: a leader that counts nothing)

static bool repl_alloc_ctl(vrg_handle* h) {            // the leader's control block (every transport: the progress word lives there)
    VrgRepl& r = h->repl;
    if (r.ctl) return true;
    const size_t cb = 8 * IPC_WORDS + (size_t)std::max(1, r.nranks) * IPC_SUMCAP * 8;
    r.ctl = alloc<uint8_t>(h, cb);
    if (!r.ctl) return false;
    be_fill(h->be, r.ctl, 0, cb);
    be_sync(h->be);
    return true;
}
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
    if (!r.chunk_dev) { r.chunk_dev = alloc<uint8_t>(h, 256); if (!r.chunk_dev) return false; }
    r.cap = cap; r.buf_bytes = bytes;
    if (r.rank == 0 && !repl_alloc_ctl(h)) return false;
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

// rccl: the next slot of the chunk structs' ring in page-locked host memory (the broadcasts read / write it directly; a slot is reused 256 chunks later -
// the leader waits for its transport stream whenever it opens a batch)
static VrgLogChunk* repl_chunk_slot(vrg_handle* h) {
    VrgRepl& r = h->repl;
    if (!r.chunk_ring) { r.chunk_ring = (VrgLogChunk*)be_host_alloc(h->be, 256 * sizeof(VrgLogChunk)); if (!r.chunk_ring) return nullptr; }
    return r.chunk_ring + (r.chunk_slot++ & 255u);
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
// the open batch's progress as the progress word `w` states it for batch n: false when the word still belongs to another batch
static bool repl_progress_of(uint64_t w, uint64_t n, uint32_t& nsw, uint32_t& nrec) {
    if ((w >> 42) != (n & ((1ull << VRG_LOG_SEQ_BITS) - 1ull))) return false;
    nsw = (uint32_t)(w >> 32) & ((1u << VRG_LOG_SW_BITS) - 1u); nrec = (uint32_t)w;
    return true;
}

// ---- the leader's side -------------------------------------------------------------------------------------------------------
// before a batch of trips is enqueued: its buffer is free again, the launches' context points at it
static int repl_open_batch(vrg_handle* h, const VrgState& s) {
    VrgRepl& r = h->repl;
    const uint64_t n = r.seq + 1;                      // the batch about to be written
    uint8_t* buf = r.buf[(n - 1) & 1];
    if (r.transport == TR_RCCL) be_repl_wait(h->be);   // (batch n - 2's broadcasts have read it)
    if (r.transport == TR_IPC && n > 2)
        for (int q = 1; q < r.nranks; q++)
            if (!repl_poll(h, r.ctl, IPC_ACK + q, n - 2, 120.0)) return fail(h, VRG_E_INTERNAL, "replication: rank " + std::to_string(q) + " did not take batch " + std::to_string(n - 2) + " of the change log");
    VrgCtx& c = h->c;
    c.log_rec = repl_rec_of(r, buf); c.log_sw = repl_sw_of(buf); c.log_cap = r.cap; c.log_swcap = r.swcap;
    c.log_pos0 = s.log_pos; c.log_nsw0 = s.log_nsw;
    // where the band chain publishes the batch's progress: the control block (device memory the followers have mapped) on ipc; page-locked
    // host memory otherwise - this rank's host thread polls it while the batch runs, and must not touch the GPU to do so
    c.log_ready = nullptr;
    if (r.stream && r.transport == TR_IPC && r.ctl) c.log_ready = reinterpret_cast<uint64_t*>(r.ctl) + IPC_SW;
    else if (r.stream && (r.transport == TR_RCCL || r.transport == TR_CALLBACK)) {
        if (!r.host_ready) { r.host_ready = (uint64_t*)be_host_alloc(h->be, 64); if (r.host_ready) std::memset(r.host_ready, 0, 64); }
        c.log_ready = r.host_ready;                    // (null: no streaming - the batch travels when it closes)
    }
    c.log_seq = (uint32_t)(n & ((1ull << VRG_LOG_SEQ_BITS) - 1ull));
    r.sent_sw = 0; r.sent_rec = 0; r.open = true;
    return VRG_OK;
}
// one chunk on its way (rccl / callback): the struct, its sweep headers, its records - everything on the transport stream
static int repl_send_chunk(vrg_handle* h, uint64_t n, uint32_t nsw, uint32_t nrec, const VrgLogBatch* hb) {
    VrgRepl& r = h->repl;
    uint8_t* buf = r.buf[(n - 1) & 1];
    VrgLogChunk ch; std::memset(&ch, 0, sizeof(ch));
    ch.seq = ++r.chunk_seq; ch.batch = n; ch.sw0 = r.sent_sw; ch.nsw = nsw - r.sent_sw; ch.rec0 = r.sent_rec; ch.nrec = nrec - r.sent_rec; ch.cap = r.cap;
    if (hb) { ch.closed = 1; ch.hb = *hb; }
    const size_t so = sizeof(VrgLogBatch) + (size_t)ch.sw0 * sizeof(VrgLogSweep), sb = (size_t)ch.nsw * sizeof(VrgLogSweep);
    const size_t ro = repl_hb_bytes(r.swcap) + (size_t)ch.rec0 * sizeof(VrgLogRec), rb = (size_t)ch.nrec * sizeof(VrgLogRec);
    if (r.transport == TR_CALLBACK) {
        uint8_t* hp = repl_host(h, std::max(sizeof(ch), std::max(sb, rb)));
        if (!hp) return fail(h, VRG_E_MEM, "replication: host buffer");
        std::memcpy(hp, &ch, sizeof(ch)); r.bcast(hp, (int64_t)sizeof(ch), 0, r.user);
        if (sb) { be_repl_copy(h->be, hp, buf + so, sb); r.bcast(hp, (int64_t)sb, 0, r.user); }
        if (rb) { be_repl_copy(h->be, hp, buf + ro, rb); r.bcast(hp, (int64_t)rb, 0, r.user); }
    } else if (r.transport == TR_RCCL) {
        // (the chunk struct travels out of page-locked host memory, a ring of slots: no copy kernel, no wait - every kernel the transport puts on this GPU
        // while the batch runs delays the band chain; measured on one GPU, 880x880x640: a chunk cost the chain ~40 us with a host-to-device copy and a wait in it)
        VrgLogChunk* slot = repl_chunk_slot(h);
        if (!slot) return fail(h, VRG_E_MEM, "replication: chunk ring");
        *slot = ch;
        if (be_repl_bcast(h->be, slot, sizeof(ch), 0) || (sb && be_repl_bcast(h->be, buf + so, sb, 0)) || (rb && be_repl_bcast(h->be, buf + ro, rb, 0)))
            return fail(h, VRG_E_INTERNAL, "replication: RCCL broadcast failed");
    }
    r.sent_sw = nsw; r.sent_rec = nrec; r.chunks++;
    if (r.transport == TR_RCCL && (r.chunks & 127) == 0) be_repl_wait(h->be);     // (the ring: never more than half of it in flight)
    return VRG_OK;
}
// while the batch's trips run (rccl / callback; on ipc the followers look for themselves): what the band chain has published since the
// last look goes out.  Returns 1 when something was sent, 0 when not, < 0 on failure.
static int repl_pump(vrg_handle* h) {
    VrgRepl& r = h->repl;
    if (!r.stream || !r.open || !r.host_ready || (r.transport != TR_RCCL && r.transport != TR_CALLBACK)) return 0;
    const uint64_t w = *reinterpret_cast<volatile uint64_t*>(r.host_ready);
    uint32_t nsw = 0, nrec = 0;
    const uint32_t least = r.chunk_min > 0 ? (uint32_t)r.chunk_min : (r.transport == TR_RCCL ? 8u : 1u);
    if (!repl_progress_of(w, r.seq + 1, nsw, nrec) || nsw < r.sent_sw + least) return 0;
    if (nsw > r.swcap || nrec > r.cap || nrec < r.sent_rec) return fail(h, VRG_E_INTERNAL, "replication: the change log's progress word is out of range");
    const int rc = repl_send_chunk(h, r.seq + 1, nsw, nrec, nullptr);
    return rc ? rc : 1;
}
// after the batch (the band stream is idle: the engine has read the state): its header, and off goes what has not travelled yet
static int repl_close_batch(vrg_handle* h, const VrgState& s, bool final, int32_t error) {
    VrgRepl& r = h->repl;
    const VrgCtx& c = h->c;
    VrgLogBatch hb; std::memset(&hb, 0, sizeof(hb));
    hb.seq = ++r.seq; hb.nsw = s.log_nsw - c.log_nsw0; hb.nrec = s.log_pos - c.log_pos0;
    hb.final = final ? 1 : 0; hb.stop_reason = s.done; hb.iter = s.iter; hb.error = error ? error : s.error;
    int64_t sizes[2]; be_download(h->be, sizes, c.inc, sizeof(sizes));
    hb.n_in = sizes[0]; hb.n_out = sizes[1]; hb.ni = s.ni; hb.no = s.no; hb.ties = s.ties; hb.near_ties = s.near_ties;
    r.open = false;
    const bool overrun = hb.nsw > r.swcap || hb.nrec > r.cap || hb.nsw < r.sent_sw || hb.nrec < r.sent_rec;
    if (overrun) {                                     // (the followers still have to be told: the run ends here, in error)
        hb.nsw = r.sent_sw; hb.nrec = r.sent_rec; hb.final = 1; if (!hb.error) hb.error = 13;
    }
    r.batches++; r.records += hb.nrec; r.sweeps += hb.nsw;
    const uint64_t n = hb.seq;
    uint8_t* buf = r.buf[(n - 1) & 1];
    int rc = VRG_OK;
    if (r.transport == TR_IPC) {
        be_repl_copy(h->be, buf, &hb, sizeof(hb));
        be_repl_copy(h->be, r.ctl + 8 * IPC_READY, &n, 8);
    } else rc = repl_send_chunk(h, n, hb.nsw, hb.nrec, &hb);
    if (!rc && overrun) rc = fail(h, VRG_E_INTERNAL, "replication: the batch overran its change log");
    return rc;
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

// ---- a follower's run: apply every sweep as it arrives, count the sweeps that are this rank's -------------------------------------
// the next chunk of batch n: its sweep headers into hdrs[sw_done ..] (host), its records into the staging buffer at their places
static int repl_next_chunk(vrg_handle* h, uint64_t n, uint32_t sw_done, uint32_t rec_done, uint8_t*& staging, std::vector<VrgLogSweep>& hdrs, VrgLogChunk& ch) {
    VrgRepl& r = h->repl;
    const size_t hbb = repl_hb_bytes(r.swcap);
    std::memset(&ch, 0, sizeof(ch));
    auto grow_for = [&](uint32_t cap) -> bool {         // the leader's buffers have grown: so do both staging buffers (idle first: kernels may still read them)
        if (cap <= r.cap) return true;
        if (sw_done || rec_done) return false;          // (only ever between two batches)
        be_sync(h->be);
        if (!repl_alloc_buffers(h, cap)) return false;
        staging = r.buf[(n - 1) & 1];
        return true;
    };
    if (r.transport == TR_IPC) {
        const uint8_t* src = r.peer_buf[(n - 1) & 1];
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; spins++) {
            uint64_t w[2] = {0, 0};
            be_repl_copy(h->be, w, r.ctl + 8 * IPC_READY, 16);
            uint32_t nsw = 0, nrec = 0;
            if (w[0] >= n) {                            // the batch is closed: its header says what there is
                be_repl_copy(h->be, &ch.hb, src, sizeof(VrgLogBatch));
                ch.closed = 1; nsw = ch.hb.nsw; nrec = ch.hb.nrec;
            } else if (!r.stream || !repl_progress_of(w[1], n, nsw, nrec)) { nsw = sw_done; nrec = rec_done; }
            if (nsw > r.swcap || nrec > r.cap || nsw < sw_done || nrec < rec_done) return fail(h, VRG_E_INTERNAL, "replication: the leader's change log is out of range");
            if (ch.closed || nsw > sw_done) { ch.batch = n; ch.sw0 = sw_done; ch.nsw = nsw - sw_done; ch.rec0 = rec_done; ch.nrec = nrec - rec_done; ch.cap = r.cap; break; }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 300.0) return fail(h, VRG_E_INTERNAL, "replication: the leader did not publish batch " + std::to_string(n) + " of the change log");
            if (spins > 256) std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        if (ch.nsw) be_repl_copy(h->be, hdrs.data() + ch.sw0, src + sizeof(VrgLogBatch) + (size_t)ch.sw0 * sizeof(VrgLogSweep), (size_t)ch.nsw * sizeof(VrgLogSweep));
        if (ch.nrec) be_repl_copy(h->be, staging + hbb + (size_t)ch.rec0 * sizeof(VrgLogRec), src + hbb + (size_t)ch.rec0 * sizeof(VrgLogRec), (size_t)ch.nrec * sizeof(VrgLogRec));
        if (ch.closed) be_repl_copy(h->be, r.ctl + 8 * (IPC_ACK + r.rank), &n, 8);     // (copied out: the leader may write that buffer again)
        r.chunks++;
        return VRG_OK;
    }
    if (r.transport != TR_CALLBACK && r.transport != TR_RCCL) return fail(h, VRG_E_STATE, "replication: no transport set");
    if (r.transport == TR_CALLBACK) r.bcast(&ch, (int64_t)sizeof(ch), 0, r.user);
    else {                                              // (a follower receives into device memory and copies out: only the ROOT's source is host memory - nothing here runs beside a band chain)
        if (be_repl_bcast(h->be, r.chunk_dev, sizeof(ch), 0)) return fail(h, VRG_E_INTERNAL, "replication: RCCL broadcast failed");
        be_repl_wait(h->be);
        be_repl_copy(h->be, &ch, r.chunk_dev, sizeof(ch));
    }
    // (sizes first: whatever else is wrong with the chunk, the broadcasts that follow it have to be matched)
    const size_t sb = (size_t)ch.nsw * sizeof(VrgLogSweep), rb = (size_t)ch.nrec * sizeof(VrgLogRec);
    bool ok = ch.batch == n && ch.sw0 == sw_done && ch.rec0 == rec_done && (uint64_t)ch.sw0 + ch.nsw <= r.swcap && grow_for(ch.cap) && (uint64_t)ch.rec0 + ch.nrec <= r.cap;
    std::vector<uint8_t> sink;                          // (a chunk this rank cannot place is still received)
    if (r.transport == TR_CALLBACK) {
        if (sb) { if (ok) r.bcast(hdrs.data() + ch.sw0, (int64_t)sb, 0, r.user); else { sink.resize(sb); r.bcast(sink.data(), (int64_t)sb, 0, r.user); } }
        if (rb) {
            uint8_t* hp = ok ? repl_host(h, rb) : nullptr;
            if (!hp) { ok = false; sink.resize(rb); hp = sink.data(); }
            r.bcast(hp, (int64_t)rb, 0, r.user);
            if (ok) be_repl_copy(h->be, staging + hbb + (size_t)ch.rec0 * sizeof(VrgLogRec), hp, rb);
        }
    } else {
        uint8_t* scratch = nullptr;
        if (!ok && (sb || rb)) { scratch = alloc<uint8_t>(h, std::max(sb, rb)); if (!scratch) return fail(h, VRG_E_MEM, "replication: receive buffer"); }
        uint8_t* sdst = ok ? staging + sizeof(VrgLogBatch) + (size_t)ch.sw0 * sizeof(VrgLogSweep) : scratch;
        uint8_t* rdst = ok ? staging + hbb + (size_t)ch.rec0 * sizeof(VrgLogRec) : scratch;
        if ((sb && be_repl_bcast(h->be, sdst, sb, 0)) || (rb && be_repl_bcast(h->be, rdst, rb, 0))) return fail(h, VRG_E_INTERNAL, "replication: RCCL broadcast failed");
        be_repl_wait(h->be);
        if (ok && sb) be_repl_copy(h->be, hdrs.data() + ch.sw0, sdst, sb);
        if (scratch) release(h, scratch);
    }
    r.chunks++;
    if (!ok) { h->err = "replication: chunk " + std::to_string(ch.seq) + " of the change log does not continue batch " + std::to_string(n) + " where this rank stands (or its staging buffers could not grow)"; return 1; }
    return VRG_OK;
}

static int repl_follow(vrg_handle* h, VrgLogBatch& last) {
    VrgRepl& r = h->repl;
    const VrgCtx& c = h->c;
    std::vector<VrgLogSweep> hdrs(r.swcap + 1);
    const int nver = repl_verifiers(r), me = repl_my_slot(r);
    for (;;) {                                          // batches
        const uint64_t n = r.seq + 1;
        uint8_t* staging = r.buf[(n - 1) & 1];
        be_follow_wait(h->be, (int)((n - 1) & 1));      // (the kernels that read this staging buffer two batches ago are done)
        uint32_t sw_done = 0, rec_done = 0, first = 0;
        VrgLogBatch hb; std::memset(&hb, 0, sizeof(hb));
        for (;;) {                                      // chunks of the batch
            VrgLogChunk ch;
            const int rc = repl_next_chunk(h, n, sw_done, rec_done, staging, hdrs, ch);
            if (rc < 0) return rc;                      // (the transport itself failed: nothing more can arrive)
            if (r.fault < 0 && r.chunks == -r.fault && !r.failed) { r.failed = 13; h->err = "replication: injected failure of this rank (option repl_fault)"; }
            if (rc > 0 && !r.failed) r.failed = 13;     // a log this rank cannot use: it keeps taking chunks - the run is collective - and reports at the end
            if (!r.failed) {
                VrgLogRec* recs = repl_rec_of(r, staging);
                // the sweeps up to the next one this rank counts go in one step (the sweeps between two counts: N - 2 of them)
                for (uint32_t i = sw_done; i < sw_done + ch.nsw; i++) {
                    if ((uint64_t)hdrs[i].rec0 + hdrs[i].nrec > (uint64_t)rec_done + ch.nrec) { r.failed = 13; h->err = "replication: a sweep's records lie outside what has arrived"; break; }
                    if (vrg_dense_skipped((int64_t)hdrs[i].sweep, h->verify_every, nver, me)) continue;
                    be_follow_apply(h->be, c, recs, hdrs.data() + first, (int)(i + 1 - first), 1);
                    be_follow_count(h->be, c, &h->ev); r.verified++; r.last_verified = hdrs[i].sweep;
                    first = i + 1;
                }
            }
            sw_done += ch.nsw; rec_done += ch.nrec;
            if (!ch.closed) continue;
            hb = ch.hb;
            if (!r.failed && (hb.seq != n || hb.nsw != sw_done || hb.nrec != rec_done)) { r.failed = 13; h->err = "replication: batch " + std::to_string(hb.seq) + " closed where batch " + std::to_string(n) + " was due, or with other contents than arrived"; }
            if (!r.failed && first < sw_done) be_follow_apply(h->be, c, repl_rec_of(r, staging), hdrs.data() + first, (int)(sw_done - first), 0);
            break;
        }
        be_follow_mark(h->be, (int)((n - 1) & 1));
        r.seq = n;
        r.batches++; r.records += rec_done; r.sweeps += sw_done;
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
    const bool any_error = v[3 * ns] > 0.5;
    for (size_t i = 0; i < ns && !any_error; i++) {
        if (v[3 * i + 2] > 1.5) return fail(h, VRG_E_INTERNAL, "replication: sweep " + std::to_string(iter0 + 1 + (int)i) + " was counted by more than one rank");
        if (v[3 * i + 2] > 0.5) { tr[i].sum_in = v[3 * i]; tr[i].sum_out = v[3 * i + 1]; }
        else { tr[i].sum_in = std::nan(""); tr[i].sum_out = std::nan(""); }
    }
    if (ns && !any_error) be_upload(h->be, c.trace + iter0 + 1, tr.data(), ns * sizeof(VrgTrace));
    if (derr == 5 || v[3 * ns + 1] > 0.5) error = 5;
    else if (derr == 12 || v[3 * ns + 2] > 0.5) error = 12;
    else if (any_error && !error) error = 13;          // another rank failed
    else if (!error && v[3 * ns + 3] != (double)iter * r.nranks) return fail(h, VRG_E_INTERNAL, "replication: the ranks ended on different sweeps");
    return VRG_OK;
}
