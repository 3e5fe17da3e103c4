// vrg_backend.h - what vrg_engine.cpp needs from an execution backend.
//
// The product backend is vrg_device.hip (HIP kernels on gfx950).  tests/hostmodel/ implements the
// same interface with sequential loops over the very same item functions (vrg_items.h) so that the
// parallel restatement of the reference's sequential update() can be checked against the oracle
// without a GPU; that model is test infrastructure and is never linked into the product library.
#pragma once
#include <stddef.h>
#include "vrg_types.h"

struct VrgEvents {            // optional HIP-event timing of the dense sweep launches
    int enabled;
    double ms_total;
    long long launches;
};

int be_set_device(int device);
void be_set_tuning(const char* name, long long value);   // kernel launch knobs ("sweep_blocks")                 // 0 ok, <0 no usable device
void* be_alloc(size_t bytes);
void be_free(void* p);
void be_fill(void* p, int byte, size_t bytes);
void be_upload(void* dst, const void* src, size_t bytes);     // src may be host or device memory
void be_download(void* dst, const void* src, size_t bytes);   // dst may be host or device memory
void be_sync();
const char* be_last_error();                   // first backend (HIP) failure since start-up, or nullptr

// repack caller arrays ([x][y][z] with element strides) into / out of the padded device layout
int be_pack_volume(const VrgCtx& c, float* dstI, const void* src, int dtype, const int64_t st[3], int* inexact);
int be_pack_labels(const VrgCtx& c, uint8_t* dst, const void* src, int dtype, const int64_t st[3], int* bad);
int be_unpack_labels(const VrgCtx& c, const uint8_t* lab, void* dst, int dtype, const int64_t st[3]);

// sorted distinct intensity values; allocates *lev (backend memory), returns the count in *L
int be_build_levels(const VrgCtx& c, double** lev, uint32_t* L);

// 16-bit storage: level index of every voxel (after be_build_levels), padded layout
void be_build_lev16(const VrgCtx& c, uint16_t* dst);

// init mode (:129-155): labels by morphology + staged band entries; then order them and finish
void be_init_band(const VrgCtx& c);
void be_init_sort(const VrgCtx& c, uint32_t n_in, uint32_t n_out);   // keys -> b_idx[0] in list order
typedef void (*be_reduce_fn)(double v[4], void* user);
void be_init_finish(const VrgCtx& c, be_reduce_fn cb, void* user);   // levels of entries, histograms, exact densities, region stats

// one trip through the while-loop body (:58-117); a no-op once st->done is set
void be_sweep_once(const VrgCtx& c, int variant, VrgEvents* ev, be_reduce_fn cb, void* user);

// RCCL communicator for the per-sweep all-reduce of the slab statistics (device backend only)
int be_comm_unique_id(void* id128);
int be_comm_init(int nranks, int rank, const void* id128);

// fold the HIP-event pairs recorded since the last call into ev (only the first n_valid were real sweeps)
void be_events_collect(VrgEvents* ev, long long n_valid);

// dense recount of the class histograms (verification aid)
void be_recount_hist(const VrgCtx& c, int par, int32_t* rin, int32_t* rout);
// (stamp, idx) of every segmented voxel, unordered; returns the count
uint32_t be_collect_segmented(const VrgCtx& c, int par, uint64_t* stamps, uint32_t* idxs, uint32_t cap);
