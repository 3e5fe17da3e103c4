// vrg_backend.h - what vrg_engine.cpp needs from an execution backend.
//
// The product backend is vrg_device.hip (HIP kernels on gfx950).  tests/hostmodel/ implements the
// same interface with sequential loops over the very same item functions (vrg_items.h) so that the
// parallel restatement of the reference's sequential update() can be checked against the oracle
// without a GPU; that model is test infrastructure and is never linked into the product library.
//
// Everything a backend keeps between calls - device, streams, events, RCCL communicator, first error - lives
// in its VrgBackend object, one per handle: two handles (two volumes, two devices, two host threads) share nothing.
#pragma once
#include <stddef.h>
#include "vrg_types.h"

struct VrgBackend;             // opaque: defined by the backend

struct VrgEvents {            // optional HIP-event timing of the dense sweep launches
    int enabled;              // 0: off; n > 0: every n-th trip of a batch is timed
    double ms_total;
    long long launches;
    // ... and of the band chain of a trip (start of k_band to end of k_close, on the band stream, i.e. BESIDE the dense pass)
    int chain_enabled;
    double chain_ms_total;
    long long chain_launches;
};

VrgBackend* be_create(int device);                             // nullptr: no usable device
void be_destroy(VrgBackend* b);
void be_set_tuning(VrgBackend* b, const char* name, long long value);   // kernel launch knobs ("sweep_blocks", "prio_mode")
void* be_alloc(VrgBackend* b, size_t bytes);
void be_free(VrgBackend* b, void* p);
void be_fill(VrgBackend* b, void* p, int byte, size_t bytes);
void be_upload(VrgBackend* b, void* dst, const void* src, size_t bytes);     // src may be host or device memory
void be_download(VrgBackend* b, void* dst, const void* src, size_t bytes);   // dst may be host or device memory
void be_copy(VrgBackend* b, void* dst, const void* src, size_t bytes);       // device to device, stream-ordered
void be_sync(VrgBackend* b);
bool be_band_busy(VrgBackend* b);              // the band stream still has work enqueued (never blocks)
const char* be_last_error(VrgBackend* b);      // first backend (HIP / RCCL) failure of this handle, or nullptr
void be_clear_error(VrgBackend* b);

// repack caller arrays ([x][y][z] with element strides) into / out of the padded device layout; dstI (fp32) or dstI64
// *inexact: some value is not representable in fp32 (only written when dstI is given)
// *nonzero (may be null): the number of non-zero values (np.count_nonzero(dataArray), the reference's closing message)
int be_pack_volume(VrgBackend* b, const VrgCtx& c, float* dstI, double* dstI64, const void* src, int dtype, const int64_t st[3], int* inexact, long long* nonzero);
int be_pack_labels(VrgBackend* b, const VrgCtx& c, uint8_t* dst, const void* src, int dtype, const int64_t st[3], int* bad);
// what: 0 = labels 0..4, 1 = segmentedMap (1 where the label is 0 or 1)
int be_unpack_labels(VrgBackend* b, const VrgCtx& c, const uint8_t* lab, void* dst, int dtype, const int64_t st[3], int what);

// sorted distinct intensity values; allocates *lev (backend memory), returns the count in *L
int be_build_levels(VrgBackend* b, const VrgCtx& c, double** lev, uint32_t* L);

// map[v - lev[0]] = level index for a level table of integers (true), or false when a level is not an integer; `map` holds
// lev[L-1] - lev[0] + 1 entries
bool be_build_lev_map(VrgBackend* b, const VrgCtx& c, uint16_t* map, uint32_t span);
// ktab[a * L + b] = vrg_kern(lev[b] - lev[a]) for every pair of levels (L <= VRG_KTAB_LEVELS; c.H, c.A set)
void be_build_ktab(VrgBackend* b, const VrgCtx& c, double* ktab);
// binned exact densities (c.nb, c.bin_lo, c.bin_h, c.bm_in / c.bm_out set): the bins' moments from c.hin / c.hout; and - a
// verification aid - how many moment words differ from the ones built from the class histograms rin / rout (device arrays)
void be_build_bins(VrgBackend* b, const VrgCtx& c);
long long be_check_bins(VrgBackend* b, const VrgCtx& c, const int32_t* rin, const int32_t* rout);
// 16-bit storage: level index of every voxel (after be_build_levels), padded layout
void be_build_lev16(VrgBackend* b, const VrgCtx& c, uint16_t* dst);
void be_build_lidx(VrgBackend* b, const VrgCtx& c, uint32_t* dst);    // large level tables: the level index of every voxel as 32 bits (c.lidx must be null during the call)

// init mode (:129-155): labels by morphology + staged band entries; then order them and finish
void be_init_band(VrgBackend* b, const VrgCtx& c);
void be_init_sort(VrgBackend* b, const VrgCtx& c, uint32_t n_in, uint32_t n_out);   // keys -> p_idx in list order
typedef void (*be_reduce_fn)(double v[4], void* user);
void be_init_finish(VrgBackend* b, const VrgCtx& c, be_reduce_fn cb, void* user);   // levels of entries, histograms, exact densities, region stats

// One trip through the while-loop body (:58-117); a no-op once st->done or st->bail is set.
//   flags: VRG_SWEEP_FULL   the relabel stencil runs over every voxel instead of the marked ones (check variant)
//          VRG_SWEEP_NODENSE  the dense recount is not launched (measurement aid)
//          VRG_SWEEP_SYNC   host-driven trip: the backend synchronises after the decisions, may grow nothing itself,
//                           but runs update() with device-wide kernels and sorts - any number of flips
// Without VRG_SWEEP_SYNC the trip is two (VRG_SWEEP_FUSED: k_band, k_sweep) or four launches (k_band, k_order, k_mark_relabel,
// k_close) and is enqueued without synchronising; a trip its kind cannot take is handed back untouched through st->bail.
//          VRG_SWEEP_FUSED  update() as ONE launch (k_sweep, vrg_items.h "fused sweep"): sweeps with at most be_fuse_limit()
//                           flips; a sweep with more is handed back (VBAIL_FUSE) and the engine repeats the trip unfused
enum { VRG_SWEEP_FULL = 1, VRG_SWEEP_NODENSE = 4, VRG_SWEEP_SYNC = 8, VRG_SWEEP_FUSED = 16 };
// (c is the engine's context: a fused trip swaps the two state buffers - c.st / c.stg say where the state is afterwards.  first / last:
// the trip's place in its batch - the last fused trip of a batch closes its sweep itself, so that the host only ever reads closed states;
// the others may be open-ended, vrg_items.h "open-ended sweeps")
void be_sweep_once(VrgBackend* b, VrgCtx& c, int flags, VrgEvents* ev, be_reduce_fn cb, void* user, bool first, bool last);
// n trips in a row (what the engine enqueues between two looks at the state)
void be_sweep_batch(VrgBackend* b, VrgCtx& c, int flags, int n, VrgEvents* ev, be_reduce_fn cb, void* user);
// Z-slabs: all-reduce and close the dense passes whose slab sums are still waiting (they are reduced a few sweeps at a
// time); collective - every rank calls it at the same point.  The engine calls it before it reads results.
void be_dense_flush(VrgBackend* b, const VrgCtx& c, be_reduce_fn cb, void* user);
// option verify_every != 1 (be_set_tuning): at the end of a run - streams idle, passes closed - count the labels of the last
// sweep after all and compare with the sizes kept by increments (a mismatch: dctl[VD_ERR] = 5); collective over the ranks
void be_verify_last(VrgBackend* b, const VrgCtx& c, be_reduce_fn cb, void* user);
// true: this volume's trips should be host-driven (VRG_SWEEP_SYNC) from the start - its level table is so large that
// the exact densities of new band entries are spread over the whole chip, which the host has to size
bool be_wants_sync(VrgBackend* b, const VrgCtx& c);
// flips one workgroup takes on (k_order); more -> VBAIL_FLIPS
uint32_t be_small_flip_limit(VrgBackend* b);
// fused trips: flips per sweep they take (more -> VBAIL_FUSE), whether this volume can run them at all, and what has to
// happen when the engine switches to them after trips of another kind (the stream is idle then)
uint32_t be_fuse_limit(VrgBackend* b, const VrgCtx& c);
bool be_fuse_ok(VrgBackend* b, const VrgCtx& c);
void be_fuse_enter(VrgBackend* b, const VrgCtx& c);

// RCCL communicator for the per-sweep all-reduce of the slab statistics (device backend only)
int be_comm_unique_id(void* id128);
int be_comm_init(VrgBackend* b, int nranks, int rank, const void* id128);

// ---- leader / follower replication (vrg_repl.h): what a FOLLOWER does with the leader's change log ------------------------------------
// apply the nrec records of sweep k (device array `recs`, this handle's device) to this handle's labels, class bits, unit bitmap and
// stamps, in stream order; `hdr` (host memory) is the sweep's header: its trace record is filed.  A record whose `old` byte is not what
// this handle holds raises dctl[VD_ERR] = 12.
// (n consecutive sweeps, headers hdr[0..n), their records at recs + hdr[i].rec0; count_last: the last of them is counted next -
// be_follow_count - so the unit list is brought up to date and its expected sizes are filed.)  The label bytes / stamps are written on a
// stream of their own, beside a dense pass; the class bits - all a pass reads - on the stream the passes run on.
void be_follow_apply(VrgBackend* b, const VrgCtx& c, const VrgLogRec* recs, const VrgLogSweep* hdr, int n, int count_last);
// ... and count the sweep be_follow_apply announced (every voxel: this handle's dense pass) against the sizes the leader filed
// for it; a mismatch raises dctl[VD_ERR] = 5; the intensity sums go into the sweep's trace record.  ev: optional HIP-event timing of the pass.
void be_follow_count(VrgBackend* b, const VrgCtx& c, VrgEvents* ev);
// a follower's two staging buffers: mark = everything enqueued so far (the kernels that read buffer `slot`); wait = that has finished
void be_follow_mark(VrgBackend* b, int slot);
void be_follow_wait(VrgBackend* b, int slot);
// transports of the log between the ranks' devices.  RCCL (device backend, after be_comm_init): broadcast from rank `root` / sum over
// the ranks, enqueued on the backend's transport stream; be_repl_wait waits for that stream.  Return 0, or -1: not available.
int be_repl_bcast(VrgBackend* b, void* dev_buf, size_t bytes, int root);
int be_repl_allsum(VrgBackend* b, double* dev_buf, size_t n);
void be_repl_wait(VrgBackend* b);
// a copy (any direction: host, device, another process's mapped device memory) on the transport stream, waited for - a follower's band
// stream is busy applying and counting the batch before while the next one arrives
void be_repl_copy(VrgBackend* b, void* dst, const void* src, size_t bytes);
// page-locked host memory (the callback transport's copy of a batch: copies to and from the device run at the link's speed)
void* be_host_alloc(VrgBackend* b, size_t bytes);
void be_host_free(VrgBackend* b, void* p);
// inter-process handles of device allocations (ranks on one node: the follower copies straight out of the leader's buffers)
int be_ipc_export(VrgBackend* b, void* dev_ptr, void* handle64);
void* be_ipc_open(VrgBackend* b, const void* handle64);        // nullptr: failed
void be_ipc_close(VrgBackend* b, void* mapped);

// fold the HIP-event pairs recorded since the last call into ev (only the first n_valid were real sweeps)
void be_events_collect(VrgBackend* b, VrgEvents* ev, long long n_valid);

// dense recount of the class histograms (verification aid)
void be_recount_hist(VrgBackend* b, const VrgCtx& c, int32_t* rin, int32_t* rout);
// bytes one dense pass requests from memory with the current labels: class words + the 128-byte intensity lines that
// hold an included voxel (all lines of the slab with option skip_excluded = 0); measurement aid for the roofline
uint64_t be_dense_bytes(VrgBackend* b, const VrgCtx& c);
void be_dense_info(VrgBackend* b, const VrgCtx& c, int64_t out[5]);
long long be_slow_flips(VrgBackend* b, const VrgCtx& c);   // four-launch trips of thousands of flips: flips k_mark_compact left to the general kernel (an excluded voxel within their 5x5x5 cube), since the handle was created
long long be_memo_trips(VrgBackend* b);      // fused trips that kept the per-level memo (a launch of their own behind the sweep; large bands)   // {nt loads, storage mode, workgroups, skip_excluded, k_recount_pipe} of the dense pass
// (stamp, idx) of every segmented voxel, unordered; returns the count
uint32_t be_collect_segmented(VrgBackend* b, const VrgCtx& c, uint64_t* stamps, uint32_t* idxs, uint32_t cap);
