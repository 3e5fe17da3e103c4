// vrg_types.h - data layout shared by the HIP kernels, the C-ABI host code and the test host model.
//
// Volume layout in HBM (all dense volumes share it): x fastest, z slowest, every axis padded so
// that stencil code never tests bounds:
//     idx(x,y,z) = ((z+2)*PY + (y+2))*PX + x,   PX = roundup(nx+2,16), PY = ny+4, PZ = nz+4
// Padding label bytes hold VB_OOB forever; x = -1,-2 wrap onto the previous row's padding.
//
// Label byte (internal encoding of the reference's valueMap, variationalRegionGrowing.py:21):
//     bit0 S  segmented (labels 0,1)          bit3 L  listed in this sweep's flip list (:88)
//     bit1 B  in the narrow band (labels 1,2)  bit4 P  flip will be applied (flip-outs always; flip-ins
//     bit2 X  excluded (label 4)                        per the skip rule, see vrg_items.h)
//     bit5 OOB padding / outside the volume    bit7 M  marked for this sweep's relabel stencil
//     label 0 = S, 1 = S|B, 2 = B, 3 = 0, 4 = X
//
// The narrow band (innerBndList / outerBndList of the reference, :147-148, :257-258) is an UNORDERED POOL of
// slots; a voxel keeps its slot for as long as it stays in the band.  The reference's list order - which decides
// the order in which flips are applied - is carried by a 64-bit key per slot (vrg_items.h, "list order"), so the
// lists are never rebuilt: per sweep only the few flips are sorted by key.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/vrg.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VRG_HD __host__ __device__ __forceinline__
#else
#define VRG_HD inline
#endif

enum : uint8_t {
    VB_S = 1, VB_B = 2, VB_X = 4, VB_L = 8, VB_P = 16, VB_OOB = 32, VB_M = 128,
    VB_LABEL = 7
};

// band slot flags
enum : uint8_t { PF_ALIVE = 1, PF_INNER = 2, PF_PEND = 4 };   // PEND: densities still to be computed exactly (:252-255)

// stop reasons VRG_STOP_* (variationalRegionGrowing.py:91-104,118-121) come from include/vrg.h

// f_res codes written by the relabel for every listed flip
enum : uint8_t { FR_FINAL = 3, FR_FRESH = 4, FR_WRITTEN = 8 };

// why the one-workgroup sweep kernel handed a trip back to the host (VrgState::bail); nothing was modified
enum { VBAIL_FLIPS = 1, VBAIL_MARKS = 2, VBAIL_POOL = 3, VBAIL_FUSE = 4, VBAIL_LOG = 5 };   // FUSE: more flips than the fused sweep kernel takes (k_sweep); LOG: the batch's change log is full

// The fused sweep (k_sweep, vrg_items.h "fused sweep"): update() of a sweep with at most VRG_FUSE_MAX flips as ONE launch, one
// flip per workgroup of VRG_FUSE_THREADS threads (thread p < 125 = place p of the flip's 5x5x5 cube, thread t < 81 = row t of
// its 9x9x9 label tile, thread t < nf = flip t of the sweep's list).
enum { VRG_KTAB_LEVELS = 2048 };
// binned exact densities: Taylor order, and the bound on |H * d * delta| (d = distance of the bin centre) the bin width is chosen for
enum { VRG_BIN_K = 8 };
#define VRG_BIN_THETA 0.5
#define VRG_BIN_T 745.2          // exp(-T) is 0.0 in double arithmetic beyond this: a bin further than sqrt(2T/H) contributes exactly nothing
#define VRG_BIN_SCALE 1073741824.0   // 2^30
enum { VRG_FUSE_MAX = 128, VRG_FUSE_THREADS = 128, VRG_FUSE_LEVELS = 2048, VRG_FUSE_PLACES = 125 };
// ... on a LARGE level table (more than VRG_FUSE_LEVELS values; needs the per-voxel level index VrgCtx::lidx): the touched levels
// are listed by their first toucher and sorted by whoever closes the sweep - in LDS, VRG_FUSE_KEYS of them at most; a voxel
// touches one level, a flip owns at most 125 voxels: VRG_FUSE_MAX_BIG flips can never list more
enum { VRG_FUSE_KEYS = 8192, VRG_FUSE_MAX_BIG = VRG_FUSE_KEYS / VRG_FUSE_PLACES };      // (a power of two: sorted in place)

struct VrgTrace {            // one record per update() call (0 = init)
    int64_t nflip, nseg, n_in, n_out, ni, no;
    double sum_in, sum_out;  // sum of intensities over the inner / outer regions
    int64_t ties, near_ties; // sign tests (:87) behind this sweep's flip list that were exact ties / near ties (VRG_TIE_*)
};

// Sign tests whose outcome the reference's own rounding decides (vrg_decide_core): |inner/innerSize - outer/outerSize|
// <= VRG_TIE_REL * max(|.|, |.|) - an exact mathematical tie (proportional class histograms of an integer volume), which
// np.sum's pairwise rounding decides in the reference (:87) - or an empty region (x / 0).  Near ties (<= VRG_TIE_NEAR_REL)
// are an INDICATOR, not a certificate: decisions inside the float tolerance of north_star (1e-5) - where the reference's
// float32 arithmetic (float32 dataArray under numpy 2: densities measured up to 9.7e-6 away, profiles/
// r03_float32_divergence_gpu.json) or this library's binned evaluation of the exact densities (<= 2e-8, vrg_items.h
// "binned mode") could decide differently.  The threshold is twice that tolerance.
#define VRG_TIE_REL 1e-11
// binned exact densities (large level tables): proved relative error of an evaluation (vrg_items.h "binned exact densities": 1.46e-8 truncation + fixed-point
// rounding), rounded up.  The ABSOLUTE error of an entry's two densities stays with it through every later correction (VrgCtx::p_err), and a sign test
// it could turn is counted as a tie - so "labels are bit-exact unless ties > 0" holds with bins too
#define VRG_BIN_REL_ERR 2.0e-8
#define VRG_TIE_NEAR_REL 2e-5

// device-resident scalars; every kernel reads them at entry (no host round trip per sweep)
struct VrgState {
    int32_t iter;        // incremental sweeps applied so far (= reference iterNum - 1)
    int32_t done;        // stop reason, 0 while running
    int32_t iterMax;
    int32_t time_up;     // host: wall-clock cap reached (:97) - the next trip only decides and stops
    int32_t bail;        // VBAIL_*: the trip has to be redone by the host-driven path / with larger arrays
    int64_t maxSegmentSize;
    uint32_t ni, no;     // lengths of innerBndList / outerBndList
    uint32_t np;         // pool slots in use (high-water mark; dead slots below it sit on the free list)
    uint32_t nfree;      // free list length
    uint32_t last_nf;    // ... of the sweep applied last
    uint32_t npend;      // flip-ins waiting in the skip-rule fix-point
    uint32_t nfresh;     // slots that (re-)entered the band this sweep: exact densities due
    uint32_t nfx;        // ... of the sweep just closed, computed by the next trip's first kernel (k_band)
    uint32_t nmk;        // voxels marked for the relabel stencil this sweep
    uint32_t nnz;        // distinct intensity levels touched by this sweep's density corrections
    uint32_t nalloc;     // slots handed out this sweep
    uint32_t ndead;      // slots that left the band this sweep
    int32_t d_ni, d_no;  // list length changes of this sweep
    uint32_t ninit_in, ninit_out, nseed;
    int32_t corr;        // the corrections of the sweep just applied (:236-247) are still to be added to the surviving
                         // entries: the next k_band does it on its way through the pool
    int32_t use_tab;     // ... from the per-level memo tabC instead of entry by entry
    int32_t tab_ok;      // decided when update() opens: fewer intensity levels than band entries, a memo pays
    uint32_t ties_filed, near_filed;   // ... as of the last trace record
    // the fused sweep defers what nothing on the band side waits for to the NEXT trip's k_band (which reads no labels):
    int32_t apply_pending;             // the label bytes of sweep `iter` are still to be written (+ class bits, the request for its dense pass)
    uint32_t ap_n;                     // ... places of the marked list to look at (k_sweep files voxel (flip r, place p) at r * 125 + p; VRG_NONE = nothing there)
    uint32_t fr_base, fr_n;            // ... and the dead slots still to be put onto the free list: freel[fr_base + j] = dead[j], j < fr_n
    int32_t d_nin, d_nout;             // region size changes of the sweep in progress (k_sweep adds them up; the closing thread moves them into VrgCtx::inc)
    uint32_t nvisit;                   // listed flips the fused stencil visited (must equal nf)
    uint32_t nnz_new;                  // fused sweep on a large level table: levels listed so far by the sweep in progress (nnz still says what the
                                       // sweep before left for this trip's k_band)
    // the change log (leader / follower replication, VrgCtx::log_rec): records and sweep headers written since init - monotone
    // counters; a launch knows where its batch's buffer starts (VrgCtx::log_pos0 / log_nsw0)
    uint32_t log_pos, log_nsw;
    int32_t wide;                      // four-launch trip: more flips than one workgroup orders in LDS - k_rank_wide / k_prepass_wide / k_fix_wide do k_order's work chip-wide
    int32_t open;                      // an OPEN-ENDED fused sweep has run on this state and nobody has closed it yet (vrg_items.h "open-ended sweeps"): the next trip's
                                       // k_band derives the closed state - every workgroup for itself - and one of its threads files it
    uint32_t log_n;                    // fused sweep in progress: records its workgroups have reserved so far (atomic count; log_pos itself moves when the sweep closes)
    // ---- the LIVE words, in a cache line of their own.  k_band's decisions bump them with device-scope atomics while ONE thread of the same
    // kernel files the rest of the state with plain stores (open-ended sweeps): a line that takes both is written back whole by the storing
    // CU's L2 and the atomics' results are lost - seen as lost flips under the interleaving campaign.  So: no line of the state ever takes
    // plain stores and atomics of different workgroups in one kernel; these four never share a line with anything that is filed.
    // (explicit padding, not alignas: an over-aligned member makes every kernel's by-value copy of the state an over-aligned stack object - k_band
    // then ran with 128 bytes of scratch and k_sweep with 188 instead of 140 VGPRs: +2 us per kernel.  The buffers themselves are 256-byte aligned.)
    uint32_t pad_live[21];
    uint32_t nf;                       // listed flips of the sweep being processed (atomic count)
    uint32_t ties, near_ties;          // tie / near-tie sign tests since init (atomic counts; see VRG_TIE_REL)
    int32_t error;                     // capacity overflow etc. (written through)
    uint32_t live_pad[28];
};
static_assert(sizeof(VrgState) % 128 == 0 && offsetof(VrgState, nf) % 128 == 0, "the live words have a cache line of their own");

// ---- the change log: what a sweep did to the label volume, for the ranks that do not run the band chain themselves ------------------
// One record per place of the sweep's marked list (a fused sweep: 125 places per flip, most of them VRG_NONE), written where the
// label byte is written (the apply step of every kind of trip), 16 bytes = one store; one header per sweep, written by whoever
// declares the sweep's labels in place.  A batch of trips writes into one buffer: [VrgLogBatch][VrgLogSweep x log_swcap][VrgLogRec x log_cap].
struct VrgLogRec {
    uint32_t idx;                      // voxel (padded layout), or VRG_NONE: nothing at this place
    uint32_t rank;                     // a voxel that becomes segmented: its rank in the sweep's flip list (stamp = sweep << 32 | rank: the list order of `segmented`, :200)
    uint8_t old, nw; uint16_t pad;     // label byte before / after (clean: no L / P / mark bits)
    uint32_t pad2;
};
struct VrgLogSweep {                   // = the sweep's trace record + where its records are
    int64_t nflip, nseg, n_in, n_out, ni, no, ties, near_ties;   // (VrgTrace without the sums; n_in / n_out: what the sweep's dense pass must reproduce)
    uint32_t sweep, nrec;              // sweep number (1-based, = VrgState::iter after it); records
    uint32_t rec0, pad;                // first record, relative to the batch buffer's records
};
// A batch's sweeps travel while the batch is still running (per-sweep streaming): the leader's band chain publishes how far the batch's log
// is complete - sweeps whose header is filed and whose records have been written by a kernel that has ENDED - in one 64-bit progress word
// (VrgCtx::log_ready): batch number (22 bits) << 42 | sweeps (10 bits) << 32 | records (32 bits); on the hipIpc transport the followers poll
// that word themselves, on the others the leader's host does and sends what is new as a CHUNK: this struct, then the chunk's sweep headers,
// then its records.  The chunk that completes a batch carries the batch header.
enum { VRG_LOG_SEQ_BITS = 22, VRG_LOG_SW_BITS = 10 };
VRG_HD uint64_t vrg_log_progress(uint64_t seq, uint32_t nsw, uint32_t nrec) { return ((seq & ((1ull << VRG_LOG_SEQ_BITS) - 1ull)) << 42) | ((uint64_t)(nsw & ((1u << VRG_LOG_SW_BITS) - 1u)) << 32) | (uint64_t)nrec; }
struct VrgLogBatch {                   // written by the leader's host when the batch is complete
    uint64_t seq;                      // batch number, from 1
    uint32_t nsw, nrec;                // sweeps and records in it
    int32_t final;                     // the run has ended with this batch ...
    int32_t stop_reason, iter, error;  // ... this way (VrgState::done, iter, error)
    int64_t n_in, n_out;               // region sizes at the end of the batch
    uint32_t ni, no;
    uint32_t ties, near_ties;          // VrgState counters at the end of the batch
    uint32_t pad[2];
};
struct VrgLogChunk {                   // sweeps [sw0, sw0 + nsw) and records [rec0, rec0 + nrec) of batch `batch` (places inside the batch's buffer)
    uint64_t seq, batch;               // chunk number since the handle was created (from 1); the batch it belongs to
    uint32_t sw0, nsw, rec0, nrec;
    int32_t closed;                    // the batch is complete with this chunk: hb is its header
    uint32_t cap;                      // records the leader's batch buffers hold (a follower's staging buffers follow at the start of a batch)
    uint32_t pad[4];
    VrgLogBatch hb;
};
static_assert(sizeof(VrgLogBatch) == 72 && sizeof(VrgLogChunk) == 128, "change log message layout");


// results of the dense recount; written by the dense stream only (own allocation, own cache lines).
// The dense pass of sweep k runs on its own stream while the band kernels already prepare sweep k+1: the region
// SIZES the next decisions need (:81-82, :101) are integers and follow exactly from the labels the sparse relabel
// changed (VrgCtx::inc); the dense pass recounts them from all voxels, must agree (else error 5), and supplies the
// intensity sums.
struct VrgDense {            // all four as double so one all-reduce sums them over the Z-slabs (counts < 2^53: exact)
    double n_in, n_out;      // region sizes (:51-52, :115-116)
    double sum_in, sum_out;  // sums of intensities over the two regions
};

enum { VC_NIN = 0, VC_NOUT = 1 };                  // VrgCtx::inc
// VrgCtx::gate - what the dense side polls while it waits for the band side (a cache line nothing else touches: the
// decisions read VrgCtx::inc all the time): VG_REQ sweep number of the last label write; VG_STOP the run has stopped or the
// trip was handed back
enum { VG_REQ = 0, VG_STOP = 1 };
// VrgCtx::dctl - VD_RSEQ: dense recounts done since init (recount k reads class copy k & 1); VD_SEQ: passes closed, i.e.
// cross-checked against the incremental sizes and filed in the trace (on one GPU the recount closes its own pass; with
// Z-slabs the partial sums of several recounts are all-reduced together, so VD_SEQ trails VD_RSEQ); VD_ERR: a pass
// disagreed with the incremental sizes; VD_NST: entries of the staged all-reduce
// VD_GO: written by the gate in front of a recount - 1: count the sweep now; 0: nothing to count (the run has stopped) or the
// sweep's pass is left out (option verify_every)
enum { VD_SEQ = 0, VD_ERR = 1, VD_RSEQ = 2, VD_NST = 3, VD_GO = 4 };
enum { UC_N = 0, UC_GEN = 16, UC_GEN_STRIDE = 16 };       // VrgCtx::uctl: list length; units newly listed by the sweeps of parity p at UC_GEN + p * UC_GEN_STRIDE (own cache lines)
enum { VRG_RING = 64, VRG_STAGE = 16 };
// diagnostic build (-DVRG_STAMPS): per-workgroup time stamps of one kernel behind the chain's 64 (VrgCtx::dbg): VRG_DBG_PER words for each of VRG_DBG_WG workgroups
enum { VRG_DBG_WG = 1024, VRG_DBG_PER = 32 };           // sweeps a recount result / expected size is kept for; slab sums per all-reduce

struct VrgCtx {
    int32_t nx, ny, nz;
    int32_t PX, PY, PZ;
    int32_t z0, z1;            // Z-slab [z0, z1) this device recounts (whole volume on one GPU)
    uint32_t PV;               // PX*PY*PZ
    double H, A;               // kernel A*exp(-0.5*H*d^2) (:7,:10)
    const float* I;            // intensities, padded layout (fp32-exact volumes)
    const double* I64;         // ... or float64, when the volume has values fp32 cannot hold (then I is null)
    const uint16_t* lev16;     // optional 16-bit storage: level index per voxel (same layout); the dense pass then
                               // streams 2 B instead of 4 B of intensity per voxel (values come from an LDS table)
    // large level tables (L > VRG_KTAB_LEVELS) without 16-bit storage: the level index of every voxel (same layout, built by
    // vrg_init together with the class histograms) - the band kernels need a voxel's level when it enters the band or the outer
    // region, and a search through a table of millions of values is 20+ dependent loads in the middle of the relabel stencil
    const uint32_t* lidx;
    uint8_t* lab[2];           // lab[0]: label bytes, updated in place; lab[1]: scratch of the full-stencil check variant
    // class bits: what the dense pass needs of a label - inner (S) / outer (not S, not excluded) - 2 bits per voxel;
    // lane l of a wave owns dword l of each 1024-voxel unit (its 16 voxels 256*j + 4*l + b), so a unit is one coalesced
    // 256-B request.  TWO copies: the dense pass of sweep k reads clsb[k & 1] while the label write of sweep k+1 already
    // changes clsb[(k+1) & 1]; each copy therefore receives the class changes of two sweeps at a time - its own and,
    // from the change list of the sweep before (chg_*[(k-1) & 1]), the one it sat out.
    uint32_t* clsb[2];
    // Which 1024-voxel units hold a voxel of class != 0 at all (one bit per unit, 32 units per word).  Excluded voxels
    // (label 4) only ever turn into outer ones (:166-168, :177-179) and padding never changes, so bits are only ever SET:
    // one copy serves both class copies (a unit listed too early is read as all-excluded and adds nothing).  The dense
    // pass visits only the listed units: the 54 % of the bench volume outside the brain mask cost it one bit per KiB.
    uint32_t* ubits;
    // ... and the units a sweep lists for the first time, by the sweep's parity: the dense stream's gate merges unew[k & 1] into
    // ubits when it prepares pass k - the labels of sweep k are in place then, and sweep k + 2 cannot have started writing
    // (it waits for pass k) - so pass k's list holds exactly the units listed up to sweep k, whatever the timing: which
    // wave sums which unit, and with it the rounding of the intensity sums, is the same in every run
    uint32_t* unew[2];
    // The listed units of this device's slab (whole units only) in ascending order: what the dense pass walks, all its
    // waves in formation (trip t of wave w takes entries (t * nwaves + w) * UNITS ...).  Rebuilt from the bitmap by the
    // dense stream's gate kernel whenever the sweep listed a new unit (uctl[UC_GEN + parity]): always sorted, so the
    // order of the sums does not depend on which thread listed a unit first.
    uint32_t* ulist;
    uint32_t* uctl;            // UC_*: list length, generation the list was built at | generation of the bitmap (own cache line)
    uint32_t mcap;             // capacity of the marked-voxel list and of the class-change lists
    uint32_t* chg_dw[2];       // per sweep parity: dword index ...
    uint32_t* chg_x[2];        // ... and xor mask of every class change that sweep made
    uint32_t* nchg;            // their lengths (2 counters, own allocation)
    uint32_t* mk_idx;          // marked voxels of this sweep ...
    uint8_t* mk_new;           // ... and their bytes after it
    uint8_t* mk_old;           // ... and before it (the class of the old byte is what the apply step compares)
    uint64_t* stamp;           // (sweep<<32 | flip rank) of a voxel's last listing; seeds: lex index
    // intensity levels: sorted distinct values and per-class histograms (:149-150, :249-250)
    uint32_t L;
    const double* lev;
    // integer-valued volumes whose values span at most 65 536 (what scanners deliver: 12-bit, 16-bit): value -> level index
    // directly, lev_map[v - lev_min] - ONE load where the binary search through `lev` makes log2(L) dependent ones; null else
    const uint16_t* lev_map;
    double lev_min;
    int32_t* hin;
    int32_t* hout;
    uint32_t* dIn;             // per-level counts of innerAdded / outerAdded / addedPoints (:232-235); zero between sweeps
    uint32_t* dOut;
    uint32_t* dConv;
    uint32_t* ltouch;          // per level: already on this sweep's touched list
    uint32_t zcap;             // capacity of the touched-level list (a power of two >= L)
    uint64_t* nz_key;          // touched levels as sort keys (unordered until the sweep's level sort)
    double* nz_val;            // ... in ascending order: their values and counts
    uint32_t* nz_cin; uint32_t* nz_cout; uint32_t* nz_cconv;
    double* tabC;              // per-level memo of the three corrections (3*L), see VrgState::use_tab
    // small level tables (L <= VRG_KTAB_LEVELS): the kernel between every pair of levels, ktab[a * L + b] = A*exp(-0.5*H*(lev[b]-lev[a])^2),
    // built by vrg_init with the very expression the sweeps evaluate - a correction then costs a load where it cost an exp
    const double* ktab;
    // LARGE level tables (L > option bin_above; vrg_items.h "binned exact densities"): the class histograms also as MOMENTS over
    // uniform intensity bins - per bin and class VRG_BIN_K + 1 sums of w * (delta / h)^k over the bin's voxels (delta = value - bin
    // centre, h = half the bin width, w = exp(-0.5*H*delta^2)), kept as 64-bit FIXED-POINT integers (scale 2^30) so that atomic
    // updates commute exactly: bit-reproducible whatever order the voxels arrive in, and a voxel that leaves a class takes out
    // exactly what it put in.  nb = 0: no bins (the exact densities sum over the levels).
    uint32_t nb;
    double bin_lo, bin_h;      // lower edge of bin 0; half width (bin b covers [bin_lo + 2*h*b, bin_lo + 2*h*(b+1)))
    int64_t* bm_in; int64_t* bm_out;   // [nb][VRG_BIN_K + 1]
    // band pool, SoA; capacity bcap slots
    uint32_t bcap;
    uint32_t* p_idx;           // voxel
    uint32_t* p_lev;           // level index of its intensity
    double* p_ip;              // innerProb / outerProb (:132-133)
    double* p_op;
    float* p_err;              // bound on the absolute error p_ip / p_op carry since their exact evaluation (binned densities: VRG_BIN_REL_ERR of the sums; 0 without bins)
    uint64_t* p_key;           // list-order key (vrg_items.h)
    uint8_t* p_flag;           // PF_*
    uint32_t* vent;            // per voxel: slot of the band entry sitting there (valid while the B bit is set)
    uint32_t* freel;           // free slots (stack)
    uint32_t* dead;            // slots that died this sweep (moved onto the free list when the sweep closes)
    // this sweep's flips
    uint32_t fcap;             // a power of two
    uint32_t* flist;           // the listed flips as k_band appended them, unordered: slot ...
    uint64_t* f_key;           // ... sort key (list bit | list-order key) ...
    uint32_t* fr_idx;          // ... voxel ...
    uint32_t* fr_lev;          // ... and intensity level (so that ordering the flips needs no look-up through the slot)
    uint32_t* f_slot;          // flip list in the reference's order (:88): slot ...
    uint32_t* f_idx;           // ... voxel ...
    uint32_t* f_lev;           // ... and intensity level of flip r
    uint8_t* f_res;            // FR_* result of flip r
    uint32_t* pend;            // ranks of the flip-ins in the skip-rule fix-point
    uint32_t* slow;            // four-launch trips with thousands of flips: the flips (ranks) k_mark_compact leaves to k_mark_relabel - an excluded voxel within their 5x5x5 cube (count: counters[48])
    uint32_t* rk_part;         // chip-wide ordering (k_rank_wide): the number of smaller keys found so far, per listed flip; all zero between sweeps
    uint32_t* fresh;           // slots needing exact densities
    // dense statistics partials (one slot per sweep workgroup)
    uint32_t nstat;
    int64_t* st_nin; int64_t* st_nout; double* st_sin; double* st_sout;
    // init scratch
    uint64_t* init_key; uint32_t* init_idx;
    // Fused trips keep the state in TWO buffers and swap them every trip: k_band reads stb[x] (what the sweep before left), decides into
    // stb[x ^ 1] - the flip counter and the tie counters are live there while the closed state is being filed beside them - and k_sweep runs on
    // stb[x ^ 1].  Nobody ever files a state into the buffer other workgroups of the same kernel are still reading.
    VrgState* stb[2];
    VrgState* st_other;        // k_sweep: the buffer the NEXT trip's k_band will decide into (its live counters are set up here)
    // the per-level counters in two sets, by the parity of the sweep that fills them: an open-ended sweep's set is read by the next trip's
    // k_band while nobody can zero it; the sweep after that does (dIn / dOut / dConv point at the set a kernel works on; the other trips: set 0)
    uint32_t* dInS[2]; uint32_t* dOutS[2]; uint32_t* dConvS[2];
    int32_t lvl_par;           // k_band: the set an open-ended sweep before this trip has filled (-1: the sweep before was closed by its own kernel)
    VrgState* st;              // the state as the item functions READ it (a kernel may point it at a snapshot of its own)
    VrgState* stg;             // ... and where it lives in global memory: every atomic and store to a state word goes here
    VrgDense* dn;              // global region statistics of the last closed pass (sum over all Z-slabs)
    VrgDense* dn_part;         // init: this device's slab partials
    VrgDense* dn_ring;         // [VRG_RING] this device's slab partials of recount k at k % VRG_RING
    int64_t* exp_ring;         // [2 * VRG_RING] region sizes after sweep k at k % VRG_RING: what dense pass k must reproduce
    VrgDense* stage_in;        // [VRG_STAGE] partials of the recounts not yet closed, packed for ONE all-reduce ...
    VrgDense* stage_out;       // ... and their totals
    // (the region sizes live beside the state and swap with it: incb[j] belongs to stb[j].  `inc` = the sizes of the buffer a kernel files
    // into / works on (stg), `inc_in` = those of the buffer it reads (st) - k_band after an open-ended sweep reads inc_in + the sweep's
    // increments in every workgroup while ONE thread files the new sizes into inc: never into what the others still read)
    int64_t* incb[2]; int64_t* inc_in;
    int64_t* inc;              // band side (own cache line): region sizes kept by increments as labels are applied -
                               // what the decisions and stop tests read - and the sweep number of the last apply
    int64_t* dctl;             // dense side (own cache line): dense passes closed since init
    int64_t* gate;             // band -> dense hand-off words (own cache line), VG_*
    int32_t world;             // number of slabs / ranks (1: dn is written directly)
    // per-launch modes of the batched kernels (a kernel gets its own copy of this struct):
    int32_t lvl_scan;          // 1: the levels a sweep touched are found by scanning the per-level counters when it closes
                               //    (small level tables); vrg_note_level then only counts - no list-building atomics
                               // 2: (fused sweep, large level table) listed by the first toucher, counted in VrgState::nnz_new
    int32_t lev_fast;          // 1: a voxel's level index is cheap here (16-bit storage, or the level table in LDS): a flip's
                               //    level is looked up from its intensity instead of fetched through its rank
    // leader / follower replication (vrg_engine.cpp "replication"): the change log of the batch of trips being enqueued (null: no log)
    VrgLogRec* log_rec; VrgLogSweep* log_sw;
    uint32_t log_cap, log_swcap;       // capacities of the two arrays
    uint32_t log_pos0, log_nsw0;       // VrgState::log_pos / log_nsw when the batch's buffer was opened
    uint64_t* log_ready;               // the batch's progress word (vrg_log_progress): written (system scope) by the first kernel of the next trip's update(), when the sweep's header and records are complete (null: nobody streams)
    uint32_t log_seq;                  // ... and the batch's number
    // which dense passes this handle counts: sweep k is counted by verifier ((k / every) - 1) % ver_n (vrg_dense_skipped); ver_me = this
    // handle's place among the verifiers, -1: it counts none.  One GPU: ver_n = 1, ver_me = 0.
    int32_t ver_n, ver_me;
    int64_t* fexp;             // a follower's next count: {sweep, n_in, n_out} it has to reproduce
    int32_t dense_none;        // no dense pass is enqueued at all (a leader that verifies nothing): the band side keeps the pass counters in step itself
    uint32_t* counters;        // arrival tickets of the last-workgroup reductions (zero between launches)
    // four-launch trips: where the relabel kernels' workgroups RESERVE their stretches of the sweep's lists - VrgState::nalloc / ndead / nfresh / nmk / d_ni / d_no kept
    // outside the state, on cache lines of their own: [0] nalloc | ndead << 32, [16] nfresh | nmk << 32 (one returning 64-bit add each), [32] d_ni, [48] d_no (32-bit).
    // Same-address atomics execute one after the other at the memory side (~10-17 ns each): six per workgroup on ONE line were 26-78 us of k_mark_relabel at 12 900
    // flips per sweep (profiles/NOTES_r05.md).  k_close reads them beside the state and zeroes them when it closes the sweep.  null: the state's own words (every other kind of trip).
    uint64_t* rsv;
    uint64_t* dbg;             // diagnostic build only (-DVRG_STAMPS): in-kernel time stamps of the band chain, see tools/chain_stamps.py
    VrgTrace* trace;
    uint32_t trace_cap;
};
