/*
 * vrg.h - C-ABI of libvrg_hip.so: the MI355X-native variational region growing sweep.
 *
 * The reference has no FFI layer for this path: its boundary is the Python function
 *     variationalRegionGrowing(dataArray, valueMap, H=2.25, maxSegmentSize=5000)
 *         -> (segmented, segmentedMap, valueMap)          Code/variationalRegionGrowing.py:10-37
 * The Python mirror of that function (arterynetwork_amd/variationalRegionGrowing.py) binds the entry
 * points below with ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 * Each entry point names the reference statements it replaces.
 *
 * Conventions: every function returns 0 on success or a negative VRG_E_* code; the message is
 * available from vrg_last_error().  No exceptions or host-language objects cross the boundary; the
 * library never keeps a caller pointer after a call returns.  Pointers may be host or device
 * pointers (the copy kind is inferred).  A handle is bound to one GPU and is not thread-safe.
 * Arrays are addressed as [x][y][z] with explicit element strides, so both numpy C order and the
 * Fortran order nibabel returns are accepted without a host-side copy.
 */
#ifndef VRG_H
#define VRG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vrg_handle vrg_handle;

enum { VRG_U8 = 0, VRG_I16 = 1, VRG_U16 = 2, VRG_I32 = 3, VRG_I64 = 4, VRG_F32 = 5, VRG_F64 = 6 };

enum {
    VRG_OK = 0,
    VRG_E_ARG = -1,        /* bad argument (shape, dtype, strides, label outside {0,3,4}, ...) */
    VRG_E_NOGPU = -2,      /* no usable HIP device: the product has no CPU fallback */
    VRG_E_MEM = -3,        /* allocation failed */
    VRG_E_STATE = -4,      /* call order (e.g. run before init) */
    VRG_E_EMPTY = -5,      /* empty seed set: the reference raises at :48 */
    VRG_E_INEXACT = -6,    /* intensities not exactly representable in the fp32 device volume */
    VRG_E_CAPACITY = -7,   /* band / flip capacity exceeded (raise with vrg_set_option) */
    VRG_E_INTERNAL = -8
};

/* stop reasons of the driver loop (variationalRegionGrowing.py:91-104, :118-121) */
enum { VRG_RUNNING = 0, VRG_STOP_CONVERGED = 1, VRG_STOP_TIME = 2, VRG_STOP_SIZE = 3, VRG_STOP_ITERMAX = 4 };

typedef struct {
    int32_t stop_reason;      /* VRG_STOP_* (VRG_RUNNING if iterMax sweeps of this call were used up) */
    int32_t iter_num;         /* the reference's iterNum at return: 1 + incremental sweeps applied */
    int64_t sweeps;           /* incremental update() sweeps applied by this call */
    int64_t nseg, n_in, n_out;/* len(segmented), innerSize, outerSize (:51-52, :115-116) */
    int64_t ni, no;           /* len(innerBnd), len(outerBnd) */
    double sum_in, sum_out;   /* sum of intensities over the inner / outer region */
    double seconds;           /* wall time of this call's sweeps (device synchronised) */
    double sweep_kernel_ms;   /* HIP-event time summed over this call's dense sweep launches (option "events") */
    int64_t sweep_launches;
    double chain_kernel_ms;   /* HIP-event time from the start of k_band to the end of k_close, summed over the trips option
                                 "chain_events" selected: the band chain of a sweep as it runs BESIDE the dense pass */
    int64_t chain_launches;
    int64_t ties;             /* sign tests (:87) of this call whose two sides agreed to a relative 1e-11, or that divided by an
                                 empty region: the reference decides those by the rounding of np.sum's pairwise order, which
                                 no regrouped summation reproduces - labels are bit-exact unless ties > 0.  With binned exact
                                 densities (more than "bin_above" distinct values) every band entry also carries the absolute error
                                 bound of its last exact evaluation (2e-8 of the sums, proved) through all later corrections, and a
                                 sign test that error could turn is counted here too: the statement holds for every level table */
    int64_t near_ties;        /* ... whose relative margin was below 2e-5 (twice the float tolerance of 1e-5): a heuristic
                                 indicator of decisions the reference's float32 arithmetic (float32 dataArray under numpy 2) or a
                                 differently ordered / binned summation could make differently.  near_ties == 0 does not
                                 certify label agreement; it says no decision was closer than that */
} vrg_result;

typedef struct {              /* one record per update() call; index 0 = init mode (:129-155) */
    int64_t nflip, nseg, n_in, n_out, ni, no;
    double sum_in, sum_out;
    int64_t ties, near_ties;  /* of the sign tests that produced this sweep's flip list (see vrg_result) */
} vrg_trace_rec;

/* Create a handle for an nx*ny*nz volume on HIP device `device`. */
int vrg_create(int64_t nx, int64_t ny, int64_t nz, int device, vrg_handle** out);
void vrg_destroy(vrg_handle* h);
const char* vrg_last_error(const vrg_handle* h);

/* Options (value 0/1 unless noted; before vrg_init unless "any time"):
 *   "band_capacity"  slots reserved for the narrow band (default: by the volume - one slot per 32 voxels, at least 65536, at most 4 M; the
 *                    arrays grow by themselves when a sweep needs more, so this only saves the re-allocations)
 *   "capacity_floor" smallest capacity of the pool and of the marked-voxel arrays (default 65536; tests lower it)
 *   "storage16"      (takes effect at the next vrg_init) keep intensities as 16-bit level indices (needs <= 16384 distinct values): the dense
 *                    pass streams 2 B instead of 4 B per voxel; results are bit-identical
 *   "sweep_variant"  0 = relabel only the marked voxels (default), 1 = check variant that runs the
 *                    label stencil on every voxel (slow; must give the same state)
 *   "events"         any time; n > 0: time the dense pass of every n-th sweep of a batch with HIP events
 *                    (vrg_result.sweep_kernel_ms / sweep_launches; an event pair costs the dense stream a few us)
 *   "chain_events"   any time; n > 0: time the band chain (k_band's start to k_close's end, band stream) of every n-th
 *                    trip of a batch with HIP events (vrg_result.chain_kernel_ms / chain_launches)
 *   "batch"          any time; sweeps enqueued between host checks of the stop flag (default 8)
 *   "small_flips"    any time; flips per sweep up to which update() stays on the device as the four-launch chain, without
 *                    a host synchronisation (default and maximum 65536; above 512 flips its ordering step runs chip-wide);
 *                    sweeps with more are driven from the host with device-wide kernels and rocPRIM sorts
 *   "serial_streams" any time; the host orders the band and dense streams (a synchronisation per sweep) instead of the
 *                    kernels waiting for each other on the device - for tools that run one kernel at a time
 *                    (rocprofv3 --pmc), under which a device-side wait could never end
 *   "skip_excluded"  any time; 1 (default): the dense pass does not fetch the intensities of runs of excluded voxels
 *                    (label 4, as the reference's dataArray[mask] gathers :249-250 never touch them); 0: it streams
 *                    the whole slab.  Same sums bit for bit.
 *   "nt_loads"       any time; -1 (default): the dense pass uses non-temporal loads when it fetches more than the
 *                    Infinity Cache can keep (about 300 MB per pass) and ordinary loads below that; 0 / 1 force one
 *   "verify_every"   any time; n = 1 (default): the dense pass over every voxel - the reference's recount of innerSize /
 *                    outerSize (:113-116), here a CHECK of the sizes the decisions read (kept by increments) and the source
 *                    of the trace's intensity sums - runs after every sweep; n > 1: after every n-th sweep; 0: never.
 *                    Labels, lists and densities do not depend on it; the trace's sum_in / sum_out are NaN for sweeps whose
 *                    pass was left out; the LAST sweep of a vrg_run call is counted when the call ends whatever n is
 *                    (a mismatch surfaces as VRG_E_INTERNAL), so no run returns unchecked.  The handle stays valid.
 *   "fused"          any time; 1 (default): a sweep with at most 128 flips runs update() (:156-259) as ONE launch (k_sweep);
 *                    0: always the four-launch chain (k_order, k_mark_relabel, k_close).  Same results.
 *   "bin_above"      before vrg_init; level tables (distinct intensity values) larger than this evaluate the exact densities of
 *                    new band entries (:252-255) through intensity bins - proved relative error 2e-8, DESIGN.md section 4 -
 *                    instead of summing over every level (default 2048; a huge value switches the bins off)
 *   "memo_above"     any time; band entries above which a fused trip keeps the per-level memo of the density corrections
 *                    (a launch of its own, k_memo; default 32768 - below that every entry sums its correction itself)
 *   "dense_off"      any time; measurement aid: the dense recount is not launched (the band chain alone);
 *                    the handle has to be initialised again afterwards
 *   "sweep_blocks", "prio_mode"
 *                    any time; launch tuning knobs of the dense pass / the two streams (0 = automatic) */
int vrg_set_option(vrg_handle* h, const char* name, int64_t value);

/* dataArray (:16): any VRG_* dtype.  Kept as fp32 when every value is exactly representable in fp32 (integer
 * volumes, float32 volumes), as float64 otherwise (the dense pass then streams 8 B per voxel); the arithmetic is
 * float64 either way.  A device pointer must not be written by other streams any more when the call is made (the
 * library reads it on its own stream). */
int vrg_set_volume(vrg_handle* h, const void* data, int dtype, const int64_t strides_xyz[3]);
/* valueMap on entry (:18-21): labels must be in {0 seed, 3 outside, 4 excluded}. */
int vrg_set_labels(vrg_handle* h, const void* labels, int dtype, const int64_t strides_xyz[3]);

/* Seeds, segmentedMap, init-mode update() and region sizes (:38-52, :129-155). */
int vrg_init(vrg_handle* h, double H);

/* The while loop (:56-117): decide flips (:79-88), stop tests (:91-104), incremental update()
 * (:156-259), region recount (:113-116) - until a stop test fires or `iterMax` sweeps in total have
 * been applied since vrg_init.  maxSeconds < 0 disables the wall-clock cap (:97).  May be called
 * again with a larger iterMax to continue. */
int vrg_run(vrg_handle* h, int64_t iterMax, int64_t maxSegmentSize, double maxSeconds, vrg_result* out);

/* valueMap on return (:33-36), written with the reference's label values 0..4. */
int vrg_get_labels(vrg_handle* h, void* out, int dtype, const int64_t strides_xyz[3]);
/* segmentedMap (:31-32): 1 where the label is 0 or 1, else 0, written like vrg_get_labels writes (any dtype, host or device memory);
 * `segmented` (:29-30) in the reference's list order. */
int vrg_get_segmented_map(vrg_handle* h, void* out, int dtype, const int64_t strides_xyz[3]);
int vrg_get_segmented(vrg_handle* h, int64_t* coords_xyz, int64_t cap, int64_t* n);
/* innerBnd (which = 0) / outerBnd (1) in list order with innerProb / outerProb at those voxels. */
int vrg_get_band(vrg_handle* h, int which, int64_t* coords_xyz, double* inner_prob, double* outer_prob,
                 int64_t cap, int64_t* n);
int vrg_get_trace(vrg_handle* h, vrg_trace_rec* out, int64_t cap, int64_t* n);
/* Verification aid: class histograms over the sorted distinct intensity values, recounted densely
 * from the labels (:149-150 / :249-250) next to the incrementally maintained ones. */
int vrg_get_levels(vrg_handle* h, double* values, int32_t* hist_in, int32_t* hist_out,
                   int32_t* recount_in, int32_t* recount_out, int64_t cap, int64_t* n);

/* Diagnostics of the handle's runs so far: out[0..3] = trips handed back to the host {unused, too many flips for
 * one workgroup, marked-voxel arrays grown, band pool grown}, out[4] = host-driven trips, out[5] = band pool
 * capacity, out[6] = marked-list capacity, out[7] = pool slots in use.  cap >= 8.  With cap >= 9 also out[8] =
 * the bytes one dense pass requests from memory with the current labels (class words + the 128-byte intensity lines
 * that hold an included voxel; every line of the slab with option skip_excluded = 0) - the roofline's numerator.
 * With cap >= 14 also how the dense pass is launched: out[9] = 1 for non-temporal loads, out[10] = intensity storage
 * (0 fp32, 1 u16 level index, 2 float64, 3 u16 level index with the value table of the dense pass held as doubles: up to 4096 levels), out[11] = workgroups, out[12] = skip_excluded, out[13] = units on its list.  With cap >= 15 also out[14] = 1 when the
 * pass is the two-trips-deep kernel k_recount_pipe (option dense_pipe; fp32 storage with skip_excluded), 0 for k_recount_bits.
 * With cap >= 20 also out[19] = the number of non-zero values of dataArray (np.count_nonzero, the reference's closing message :95), counted when the volume was set.
 * With cap >= 22 also what a large level table costs in device memory: out[20] = bytes of the bin moments (L > 2048 distinct values: up to 4 194 304 bins x 9
 * 64-bit words x 2 classes = 604 MB), out[21] = bytes of the per-voxel level index (4 B per voxel; 2 B with 16-bit storage).
 * With cap >= 23 also out[22] = flips of sweeps with thousands of flips that the compact relabel kernel left to the general one (an excluded voxel
 * within two voxels of the flip), since the handle was created. */
int vrg_get_stats(vrg_handle* h, int64_t* out, int64_t cap);

/* Diagnostic builds only (compiled with -DVRG_STAMPS, tools/chain_stamps.py): 64 in-kernel time stamps (100-MHz ticks) of
 * the band chain's last sweep; all zero in the product build. */
int vrg_debug_stamps(vrg_handle* h, uint64_t* out64);
/* ... and, per workgroup of k_mark_relabel's last launch (the four-launch trip's relabel kernel), 32 words for each of 1024 workgroups:
 * entry, state loaded, end of rounds 1..10, filing started / ended, exit, hardware id, phases of rounds 0 and 2 (tools/mark_stamps.py); cap >= 32768 words. */
int vrg_debug_stamps_wide(vrg_handle* h, uint64_t* out, int64_t cap);

/* ---- multi-GPU (one process per GPU) --------------------------------------------------
 * Every rank holds the label volume and applies the O(band) relabel identically (it is deterministic),
 * so no label halo has to travel; the O(V) per-sweep work - the dense region recount - is cut into
 * Z-slabs: this handle recounts planes [z0, z1) only and the partial region statistics are summed over
 * the ranks once per sweep, either by RCCL on the dense stream (vrg_comm_init) or by a host callback.  The
 * decisions never wait for that sum: region sizes are kept by increments on every rank and the summed recount
 * only has to reproduce them (a mismatch surfaces as VRG_E_INTERNAL). */
int vrg_set_slab(vrg_handle* h, int64_t z0, int64_t z1);
/* RCCL: rank 0 obtains a 128-byte id and the caller broadcasts it (e.g. torch.distributed); every rank
 * then calls vrg_comm_init (collective). */
int vrg_comm_unique_id(void* id128);
int vrg_comm_init(vrg_handle* h, int nranks, int rank, const void* id128);
/* Host-side alternative: fn turns the local partials {n_in, n_out, sum_in, sum_out} into global totals
 * in place; called once per sweep (and once by vrg_init).  Costs a host synchronisation per sweep. */
typedef void (*vrg_reduce_fn)(double partial_to_total[4], void* user);
int vrg_set_reduce_callback(vrg_handle* h, vrg_reduce_fn fn, void* user);

/* ---- multi-GPU, leader / follower replication (one process per GPU) ---------------------------------
 * Rank 0 (the leader) runs the whole band chain - decisions, update(), densities (:58-117) - exactly as one GPU does and
 * logs what every sweep did to the labels; the other ranks (followers) hold the intensities and the labels, apply the log
 * and COUNT the sweeps assigned to them - the reference's dense recount of innerSize / outerSize (:113-116), here the check
 * of the sizes the decisions read - round robin, each over the whole volume with the very pass one GPU runs.  The
 * leader counts a share too when leader_verifies != 0 (small groups).  vrg_create / vrg_set_volume / vrg_set_labels / vrg_init
 * / vrg_run are then COLLECTIVE: every rank makes the same calls with the same arguments (use maxSeconds < 0 or expect the
 * leader's clock to decide).  After vrg_run every rank holds the same labels, `segmented` order, trace and vrg_result; the
 * band (vrg_get_band) lives on the leader only.  A count that disagrees, or a rank whose labels have drifted from the log,
 * fails the run on every rank (VRG_E_INTERNAL).  Call vrg_repl_init before vrg_init, then choose ONE transport for the log. */
int vrg_repl_init(vrg_handle* h, int nranks, int rank, int leader_verifies);
/* Options of a replicated handle (vrg_set_option): "repl_stream" 1 (default) - the log travels sweep by sweep while a batch of trips
 * runs (a follower lags by a poll and a copy); 0 - once per batch.  "log_capacity" - records a batch buffer holds (before the first
 * vrg_run).  "repl_fault" - tests only: n > 0 makes the leader fail on the host side when it opens its n-th batch, n < 0 makes a
 * follower unable to use its |n|-th chunk; either way EVERY rank's vrg_run returns an error (the run is collective: a failing
 * rank still ends the run for the others and joins the closing all-reduce). */
/* transport 1 - host callbacks (any fabric; the CPU tests use torch.distributed / gloo): bcast broadcasts `bytes` bytes of
 * `buf` (host memory) from rank `root` to every rank, allsum sums n doubles over the ranks in place; both blocking, collective */
typedef void (*vrg_bcast_fn)(void* buf, int64_t bytes, int root, void* user);
typedef void (*vrg_allsum_fn)(double* buf, int64_t n, void* user);
int vrg_repl_set_callbacks(vrg_handle* h, vrg_bcast_fn bcast, vrg_allsum_fn allsum, void* user);
/* transport 2 - RCCL: the communicator of vrg_comm_init carries the log (ncclBroadcast on a stream of its own) */
int vrg_repl_use_rccl(vrg_handle* h);
/* transport 3 - hipIpc between the ranks of one node: the leader exports its log buffers (a blob of *bytes <= cap bytes, cap >= 256)
 * that the caller hands to every follower, which maps them and copies each batch out itself - device to device, over xGMI
 * between GPUs; the leader never waits for a follower unless one falls two batches behind */
int vrg_repl_ipc_export(vrg_handle* h, void* blob, int64_t cap, int64_t* bytes);
int vrg_repl_ipc_import(vrg_handle* h, const void* blob, int64_t bytes);
/* diagnostics: out[0] = batches published / taken, out[1] = log records, out[2] = sweeps in them, out[3] = sweeps this rank counted,
 * out[4] = the last of them, out[5] = transport (1 callback, 2 rccl, 3 ipc), out[6] = verifiers, out[7] = this rank's place among them (-1: none);
 * with cap >= 9 also out[8] = chunks of the log sent / taken (the log travels sweep by sweep: several chunks per batch of trips) */
int vrg_repl_stats(vrg_handle* h, int64_t* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
