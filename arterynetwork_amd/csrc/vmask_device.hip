// vmask_device.hip - HIP kernels behind include/vmask.h: exact Euclidean distance transform,
// connected-component labelling, and the stage-1 vessel-mask pipeline.
//
// Layout: dense C order [n0][n1][n2] (i2 fastest), 32-bit voxel indices.
//
// EDT: Meijster's exact integer algorithm (the squared distance is an integer, so any exact method -
//   scipy uses a Voronoi feature transform - yields the same value; out = sqrt in float64).
//   Phase 1 along axis 0 (two scans), then the lower-envelope pass along axis 1 and along axis 2,
//   one thread per line with the wave's lanes on neighbouring lines (coalesced), the envelope stack in registers /
//   LDS / a chunked spill area (k_edt_envelope); for axis 2 the volume is transposed i1 <-> i2 in 64 x 64 LDS tiles
//   before and after (the second of them takes the root: float64 out, no pass of its own).  The scans and transposes
//   run at 4.3-5 TB/s; the envelope pass is bound by instruction issue.
// Labelling: union-find with compare-and-swap linking (root = smallest raster index of the component), one
//   union pass over the 3/9/13 forward neighbours, path flattening, component sizes by atomics, and
//   raster-order numbering = number of roots before a root (what skimage / scipy number by), from a root bitmap and an
//   exclusive scan over its words' popcounts.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/vmask.h"
#include "../../include/vrg.h"

namespace {

constexpr int TPB = 256;
constexpr int32_t EDT_INF = 1 << 29;
std::string g_err;

#define VM_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); return VRG_E_INTERNAL; } } while (0)

struct Dims { int32_t n0, n1, n2; };

int grid_for(uint64_t n) { return (int)std::min<uint64_t>(65535u * 16u, (n + TPB - 1) / TPB); }

// ---------------------------------------------------------------- EDT
// phase 1: squared distance to the nearest zero voxel along axis 0; one thread per VEC neighbouring (i1,i2) columns (VEC = 4
// when the row length allows aligned 4-voxel accesses: 256 B of mask and 1 KB of G per wave and row).  The downward scan
// leaves its distance in F as 16 bits (65535 = no zero above; anything >= 32768 ends as EDT_INF anyway), the upward scan
// reads that back - it is 0 exactly where the mask is, so the mask is not read again - and writes min(down, up)^2:
// 9 bytes per voxel (round 3: 14), the loads of 32 / 16 rows in flight together (round 3: one row at a time, each waited for;
// with 4 columns per thread a 880x880x640 volume has only ~2 waves per SIMD: the depth of the batches is what fills the bus).
#ifndef EDT_AX0_DOWN
#define EDT_AX0_DOWN 32
#define EDT_AX0_UP 16
#endif
template <int VEC> struct Ax0;
template <> struct Ax0<1> { using M = uint8_t; using F = uint16_t; using G = int32_t; };
template <> struct Ax0<4> { using M = uchar4; using F = ushort4; using G = int4; };
template <int VEC>
__global__ void __launch_bounds__(TPB) k_edt_axis0(const uint8_t* __restrict__ mask, uint16_t* __restrict__ F, int32_t* __restrict__ G, Dims d) {
    using MT = typename Ax0<VEC>::M; using FT = typename Ax0<VEC>::F; using GT = typename Ax0<VEC>::G;
    const uint32_t ncol = (uint32_t)d.n1 * (uint32_t)d.n2 / VEC;           // (columns in units of VEC)
    const MT* __restrict__ mk = reinterpret_cast<const MT*>(mask);
    FT* __restrict__ fo = reinterpret_cast<FT*>(F);
    GT* __restrict__ go = reinterpret_cast<GT*>(G);
    for (uint32_t col = blockIdx.x * blockDim.x + threadIdx.x; col < ncol; col += gridDim.x * blockDim.x) {
        uint32_t dist[VEC];
        auto down = [&](const MT& m, int32_t i) {
            uint8_t mb[VEC]; uint16_t fb[VEC];
            __builtin_memcpy(mb, &m, VEC);
#pragma unroll
            for (int v = 0; v < VEC; v++) { dist[v] = mb[v] ? min(dist[v] + 1u, 65535u) : 0u; fb[v] = (uint16_t)dist[v]; }
            FT fv; __builtin_memcpy(&fv, fb, 2 * VEC);
            fo[(size_t)i * ncol + col] = fv;
        };
        auto up = [&](const FT& f, int32_t i) {
            uint16_t fb[VEC]; int32_t gb[VEC];
            __builtin_memcpy(fb, &f, 2 * VEC);
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                dist[v] = fb[v] ? min(dist[v] + 1u, 65535u) : 0u;
                const uint32_t g = min((uint32_t)fb[v], dist[v]);
                gb[v] = g >= 32768u ? EDT_INF : (int32_t)(g * g);
            }
            GT gv; __builtin_memcpy(&gv, gb, 4 * VEC);
            go[(size_t)i * ncol + col] = gv;
        };
        // (whole batches without a branch inside - a branch between the requests and their uses makes the compiler move every
        // request down to its use, one waited-for load at a time -, the last rows one by one)
#pragma unroll
        for (int v = 0; v < VEC; v++) dist[v] = 65535u;
        int32_t i = 0;
        for (; i + EDT_AX0_DOWN <= d.n0; i += EDT_AX0_DOWN) {               // downward
            MT mv[EDT_AX0_DOWN];
#pragma unroll
            for (int k = 0; k < EDT_AX0_DOWN; k++) mv[k] = mk[(size_t)(i + k) * ncol + col];
#pragma unroll
            for (int k = 0; k < EDT_AX0_DOWN; k++) down(mv[k], i + k);
        }
        for (; i < d.n0; i++) down(mk[(size_t)i * ncol + col], i);
#pragma unroll
        for (int v = 0; v < VEC; v++) dist[v] = 65535u;
        i = d.n0 - 1;
        for (; i - EDT_AX0_UP + 1 >= 0; i -= EDT_AX0_UP) {                  // upward
            FT fv[EDT_AX0_UP];
#pragma unroll
            for (int k = 0; k < EDT_AX0_UP; k++) fv[k] = fo[(size_t)(i - k) * ncol + col];
#pragma unroll
            for (int k = 0; k < EDT_AX0_UP; k++) up(fv[k], i - k);
        }
        for (; i >= 0; i--) up(fo[(size_t)i * ncol + col], i);
    }
}

// floor(a / b) for b > 0, |a| < 2^52 (here |a| < 2^34, b < 2^17): the double quotient of a non-multiple lies at least 1/b
// away from an integer - more than its rounding error while |a| < 2^52 -, so the floor of the rounded quotient is exact.
// (A 64-bit integer division is a ~150-instruction routine on this chip, one per step of every line.)
__device__ __forceinline__ long long floordiv(long long a, long long b) {
    return (long long)floor((double)a / (double)b);
}
// the same for volumes whose squared diagonal is below EDT_INF, in 32-bit arithmetic (every square and sum stays under 2^30 there,
// every factor under 2^24: the full-rate 24-bit multiplier serves; the kernel is bound by its instruction count): a float
// quotient from the hardware reciprocal (1 ulp; a correctly rounded one is a 12-instruction sequence) - a, the reciprocal and
// the product each off by at most 2^-23 relative, so the quotient by less than 2^-6 while it is below 2^15: its truncation is
// the floor or one beside it, set right by one multiplication.  A quotient of 2^15 or more only has to come out >= m (the
// entry is then not pushed).  No branch.
__device__ __forceinline__ int32_t floordiv(int32_t a, int32_t b) {
    const float qf = (float)a * __builtin_amdgcn_rcpf((float)b);
    const bool big = fabsf(qf) >= 32768.f;
    int32_t q = big ? 0 : (int32_t)qf;
    const int32_t r = a - __mul24(q, b);
    q += (int32_t)(r >= b) - (int32_t)(r < 0);
    return big ? (qf > 0 ? (1 << 20) : -(1 << 20)) : q;
}
// d * d + g
__device__ __forceinline__ int32_t sq_add(int32_t d, int32_t g) { return __mul24(d, d) + g; }
__device__ __forceinline__ long long sq_add(long long d, long long g) { return d * d + g; }
__device__ __forceinline__ int32_t mul_(int32_t a, int32_t b) { return __mul24(a, b); }
__device__ __forceinline__ long long mul_(long long a, long long b) { return a * b; }

// Meijster phase 2 along axis 1: Gout(u) = min_i (u-i)^2 + Gin(i), one thread per line, the 64 lanes of a wave on 64
// neighbouring lines of ONE i0 slab (i2 consecutive: every row access is one 256-byte request, at a wave-uniform base
// plus a 32-bit offset - no 64-bit address arithmetic per access; lanes past the end of a row repeat its last line).
// The lower envelope is a per-line stack of (site | start << 16, G(site)).  Its top sits in registers and its topmost
// <= EDT_RING entries in LDS (a ring per thread, [slot][thread], one 8-byte access per entry: conflict-free).  Deeper entries
// live in a spill area in global memory, one contiguous
// region per line (padded to EDT_CHUNK entries), and move between the two in aligned chunks of EDT_CHUNK entries =
// 64 bytes: a full ring sheds its oldest chunk, a pop below the ring brings one back (leaving room for as many pushes
// before the next move).  The line's values are fetched EDT_AHEAD rows ahead of the scan.
// Most entries never get that far (round 3): an entry (site s, start t, value v = (t-s)^2 + G(s) at its start) can only be
// popped by a later site u' with v > (t-u')^2 + G(u') - impossible once (u-t)^2 >= v for the scan position u, as G >= 0
// and u' >= u.  Such an entry is FINAL, and so is everything below it (a stack pops from the top).  So before each batch
// of EDT_AHEAD sites the bottom of the stack is written out for good - entry i's rows [t_i, t_(i+1)) as soon as entry
// i+1 is final - and leaves the ring at that end.  A background voxel (G = 0, t = s) is final at once: in a vessel mask,
// where nearly every voxel is background, the stack never outgrows the ring, rows leave a batch behind the scan (all lanes
// of a wave the same rows: whole 256-byte requests), and the spill area - 3.4 GB written and read back per pass at
// 880x880x640 in round 2, 2.7 x the pass's floor - is touched only inside large solid regions, whose entries stay poppable
// for long.  What was spilled there is written out from the spill area once it is final (the oldest chunk first, all lanes
// of the wave together), so that the ring's bottom can leave again behind it.
// Measured (profiles/r03_mask_pmc.csv, brain-sized ellipsoid): 10.7 -> 6.8 GB and 4.53 -> 3.07 ms per pass; the pass is
// then bound by its ~150 VALU instructions per site (SQ counters: VALU busy 0.6), not by HBM.  Round 4 cut those: hardware
// reciprocal and 24-bit multiplies, the top's value at its start kept in a register, unsigned ring slots, uniform bases.
// (Round 1: every push wrote and every pop read global memory in the data's layout, 4 bytes at a time at addresses
// that differ from lane to lane, and every step waited for its own load - 12.4 ms per pass at 880x880x640.)
// (Round 4, per mask at 880x880x640: a 12-entry ring - 26 instead of 20 waves per CU - is 4-8 % faster where the stack stays
// shallow, a tube, and 8-15 % slower inside the solid ellipsoid, more spills; 32 entries - 10 waves per CU - 1.4-1.6 x slower)
#ifndef EDT_RING_N            // (-D overrides: tuning experiments only)
#define EDT_RING_N 16
#define EDT_CHUNK_N 8
#endif
constexpr int EDT_RING = EDT_RING_N;
constexpr int EDT_CHUNK = EDT_CHUNK_N;
#ifndef EDT_AHEAD_N
#define EDT_AHEAD_N 8
#endif
#ifndef EDT_TPB_N
#define EDT_TPB_N 64
#endif
constexpr int EDT_AHEAD = EDT_AHEAD_N;
constexpr int EDT_TPB = EDT_TPB_N;
__host__ __device__ inline uint32_t edt_chunks(int32_t n2) { return ((uint32_t)n2 + 63u) / 64u; }     // 64-line groups per i0 slab
template <typename I>      // int32_t for volumes whose squared diagonal is below EDT_INF (edt_squared), else long long
__global__ void __launch_bounds__(EDT_TPB) k_edt_envelope(const int32_t* __restrict__ Gin, int32_t* __restrict__ Gout, uint2* __restrict__ SP, Dims d) {
    __shared__ uint2 ring[EDT_RING][EDT_TPB];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t cpl = edt_chunks(d.n2), nitems = (uint32_t)d.n0 * cpl;
    const int32_t m = d.n1;
    const size_t mp = ((size_t)m + EDT_CHUNK - 1) / EDT_CHUNK * EDT_CHUNK;
    const uint32_t stride = (uint32_t)d.n2;
    for (uint32_t item0 = blockIdx.x * (EDT_TPB / 64) + (tid >> 6); item0 < nitems; item0 += gridDim.x * (EDT_TPB / 64)) {
        const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item0);      // (the same in every lane of the wave)
        const uint32_t i0 = item / cpl, i2 = min((item - i0 * cpl) * 64u + lane, (uint32_t)d.n2 - 1u);
        const int32_t* __restrict__ gin = Gin + (size_t)i0 * d.n1 * d.n2;
        int32_t* __restrict__ gout = Gout + (size_t)i0 * d.n1 * d.n2;
        uint2* __restrict__ spill = SP + ((size_t)i0 * d.n2 + i2) * mp;
#define AT(u) (i2 + (uint32_t)(u) * stride)
#define SLOT(i) ((uint32_t)(i) % (uint32_t)EDT_RING)
        // entries [low, q] of the stack are in the ring (entry i in slot i % EDT_RING), entries [eb, low) in the spill
        // area (low - eb a multiple of EDT_CHUNK), entries [0, eb) are final and written: rows [0, ue) of the output
        int32_t q = 0, low = 0, eb = 0, ue = 0;
        I ts = 0, tt = 0, tg = gin[AT(0)], tv = tg;      // top of the stack: site, start, G(site), its value at its start
        I dn_t = 0, dn_v = 0;                            // while low > eb: start of entry eb + EDT_CHUNK (the one above the oldest spilled chunk) and its value there
        ring[0][tid] = make_uint2(0u, (uint32_t)tg);
        auto pop = [&]() {                                        // q was decremented and is >= eb: its entry becomes the top
            if (q < low) {                                        // (q == low - 1: the chunk below the ring comes back)
                low -= EDT_CHUNK;
                uint2 e[EDT_CHUNK];
#pragma unroll
                for (int i = 0; i < EDT_CHUNK; i++) e[i] = spill[low + i];
#pragma unroll
                for (int i = 0; i < EDT_CHUNK; i++) ring[SLOT(low + i)][tid] = e[i];
            }
            const uint2 p = ring[SLOT(q)][tid];
            ts = p.x & 0xffffu; tt = p.x >> 16; tg = (int32_t)p.y; tv = sq_add(tt - ts, tg);
        };
        auto rows_out = [&](I s0, I g0, I r0, I r1) {             // rows [r0, r1) of the output from entry (site s0, G g0)
            for (I r = r0; r < r1; r++) {
                const I v = sq_add(r - s0, g0);
                gout[AT(r)] = v >= EDT_INF ? EDT_INF : (int32_t)v;
            }
        };
        auto fetch = [&](int32_t (&gv)[EDT_AHEAD], int32_t u0) {  // the line's values of a batch (rows past the end repeat the last one)
#pragma unroll
            for (int k = 0; k < EDT_AHEAD; k++) gv[k] = gin[AT(min(u0 + k, m - 1))];
        };
        auto batch = [&](const int32_t u0, const int32_t (&gv)[EDT_AHEAD]) {
            // (once per batch of EDT_AHEAD sites: everything below is about sites >= u0)
            for (;;) {                                            // final entries that had to be spilled: the oldest chunk [eb, eb + EDT_CHUNK)
                const I dd = u0 - dn_t;                           // ... is final when the entry above it (start dn_t, value dn_v there) is
                const bool has = low > eb, fin = has && dd >= 0 && dn_v <= mul_(dd, dd);
                // (the lanes of a wave - neighbouring lines - write their chunks out TOGETHER: a lane on its own would leave
                // 4 bytes in each of its rows' cache lines long before or after its neighbours do, every one a write of its own)
                const unsigned long long wh = __ballot(has), wf = __ballot(fin);
                if (!wh || wf != wh) break;
                if (!has) continue;
                uint2 e[EDT_CHUNK];
#pragma unroll
                for (int i = 0; i < EDT_CHUNK; i++) e[i] = spill[eb + i];
#pragma unroll
                for (int i = 0; i < EDT_CHUNK; i++)
                    rows_out(e[i].x & 0xffffu, (int32_t)e[i].y, e[i].x >> 16, i + 1 < EDT_CHUNK ? (I)(e[(i + 1) % EDT_CHUNK].x >> 16) : dn_t);
                ue = (int32_t)dn_t; eb += EDT_CHUNK;
                if (low > eb) {                                   // the entry above the next chunk: spilled itself, or the bottom of the ring
                    const uint2 n = eb + EDT_CHUNK < low ? spill[eb + EDT_CHUNK] : ring[SLOT(low)][tid];
                    const I sn = n.x & 0xffffu;
                    dn_t = n.x >> 16; dn_v = sq_add(dn_t - sn, (I)(int32_t)n.y);
                }
            }
            if (low == eb && low < q) {                       // the final bottom of the stack leaves (nothing of it is in the spill area)
                uint2 p0 = ring[SLOT(low)][tid];
                for (;;) {
                    const uint2 p1 = ring[SLOT(low + 1)][tid];
                    const I s1 = p1.x & 0xffffu, t1 = p1.x >> 16, v1 = sq_add(t1 - s1, (I)(int32_t)p1.y);
                    const I dd = u0 - t1;
                    if (!(dd >= 0 && v1 <= mul_(dd, dd))) break;   // (else entry low + 1 can still be popped: entry low may become the top again)
                    rows_out(p0.x & 0xffffu, (int32_t)p0.y, p0.x >> 16, t1);
                    ue = (int32_t)t1; low++; eb++;
                    p0 = p1;
                    if (low >= q) break;
                }
            }
#pragma unroll
            for (int k = 0; k < EDT_AHEAD; k++) {
                const int32_t u = u0 + k;
                if (u >= m) break;
                const I Gu = gv[k];
                while (q >= eb) {
                    if (tv <= sq_add(tt - u, Gu)) break;
                    if (--q >= eb) pop();
                }
                if (q < eb) { q = 0; low = 0; ts = u; tt = 0; tg = Gu; tv = sq_add((I)u, Gu); ring[0][tid] = make_uint2((uint32_t)u, (uint32_t)(int32_t)Gu); }   // (eb == 0: a final entry is never popped)
                else {
                    const I w = 1 + floordiv(mul_((I)u + ts, (I)u - ts) + Gu - tg, (I)2 * (u - ts));
                    if (w < m) {                                  // w >= 1 here: the top still wins at its own start
                        q++;
                        if (q - low >= EDT_RING) {                // the ring is full: its oldest chunk moves to the spill area
#pragma unroll
                            for (int i = 0; i < EDT_CHUNK; i++) spill[low + i] = ring[SLOT(low + i)][tid];
                            low += EDT_CHUNK;
                            if (low - eb == EDT_CHUNK) {          // the first spilled chunk: the entry above it is the ring's bottom now
                                const uint2 pn = ring[SLOT(low)][tid];
                                const I sn = pn.x & 0xffffu;
                                dn_t = pn.x >> 16; dn_v = sq_add(dn_t - sn, (I)(int32_t)pn.y);
                            }
                        }
                        ts = u; tt = w; tg = Gu; tv = sq_add(w - u, Gu);
                        ring[SLOT(q)][tid] = make_uint2((uint32_t)u | ((uint32_t)w << 16), (uint32_t)(int32_t)Gu);
                    }
                }
            }
        };
        // (Requesting a batch's values one or two batches ahead - the memory counter retires loads and stores in order, so a load
        // used in its own batch also waits for the row stores issued after it - was measured in round 4: 24 more registers, a
        // wave less per SIMD, 5 % slower.  The waves beside this one cover the wait.)
        for (int32_t u0 = 1; u0 < m; u0 += EDT_AHEAD) { int32_t gv[EDT_AHEAD]; fetch(gv, u0); batch(u0, gv); }
        for (int32_t u = m - 1; u >= ue; u--) {
            const I v = sq_add(u - ts, tg);
            gout[AT(u)] = v >= EDT_INF ? EDT_INF : (int32_t)v;
            if (u == tt && --q >= eb) pop();
        }
#undef AT
#undef SLOT
    }
}
// out[i0][b][a] = in[i0][a][b] for every i0-slab (na x nb -> nb x na), 64 x 64 tiles through LDS: both sides coalesced
// (OUT = double: the squared distances leave as distances, scipy's float64 - no pass of its own for the root)
template <typename OUT>
__global__ void __launch_bounds__(256) k_transpose12(const int32_t* __restrict__ in, OUT* __restrict__ out, int32_t n0, int32_t na, int32_t nb) {
    __shared__ int32_t tile[64][65];
    const uint32_t ta = (na + 63) / 64, tb = (nb + 63) / 64;
    const uint64_t ntiles = (uint64_t)n0 * ta * tb;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;     // 64 x 4
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t i0 = (uint32_t)(t / ((uint64_t)ta * tb)), r = (uint32_t)(t % ((uint64_t)ta * tb));
        const int a0 = (int)(r / tb) * 64, b0 = (int)(r % tb) * 64;
        const size_t slab = (size_t)i0 * na * nb;
        for (int j = ty; j < 64; j += 4) {
            int a = a0 + j, b = b0 + tx;
            if (a < na && b < nb) tile[j][tx] = in[slab + (size_t)a * nb + b];
        }
        __syncthreads();
        for (int j = ty; j < 64; j += 4) {
            int b = b0 + j, a = a0 + tx;
            if (a < na && b < nb) {
                if constexpr (sizeof(OUT) == 8) out[slab + (size_t)b * na + a] = sqrt((double)tile[tx][j]);
                else out[slab + (size_t)b * na + a] = tile[tx][j];
            }
        }
        __syncthreads();
    }
}

// squared EDT of a device-resident mask into G (device).  Scratch: one more int32 volume and the envelope pass's
// spill area (8 bytes per voxel, lines padded to EDT_CHUNK entries).
// The lower-envelope pass runs one thread per line with the lanes of a wave on neighbouring lines, which is
// coalesced only when the lines are NOT along the fastest axis: the axis-2 pass therefore runs on the i1 <-> i2
// transposed volume (two tiled transposes, ~4 GB of traffic each at 880x880x640, instead of a 20x slower pass).
// (dist != nullptr: the distances themselves, float64, into dist - G is scratch then)
int edt_squared(const uint8_t* dmask, Dims d, int32_t* G, double* dist = nullptr) {
    const size_t V = (size_t)d.n0 * d.n1 * d.n2;
    auto padded = [](size_t m) { return (m + EDT_CHUNK - 1) / EDT_CHUNK * EDT_CHUNK; };
    const size_t spill_entries = std::max((size_t)d.n0 * d.n2 * padded((size_t)d.n1), (size_t)d.n0 * d.n1 * padded((size_t)d.n2));
    int32_t* G2 = nullptr; uint2* SP = nullptr;
    VM_TRY(hipMalloc(&G2, V * 4));
    if (hipMalloc(&SP, spill_entries * sizeof(uint2)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(G2); g_err = "out of device memory (EDT scratch)"; return VRG_E_MEM; }
    // (every finite squared distance, and so every square and sum of the envelope pass, below EDT_INF = 2^29)
    const bool small = (int64_t)d.n0 * d.n0 + (int64_t)d.n1 * d.n1 + (int64_t)d.n2 * d.n2 < (int64_t)EDT_INF;
    // (G2 holds the downward scan's 16-bit distances first)
    if (((size_t)d.n1 * d.n2) % 4 == 0 && (reinterpret_cast<uintptr_t>(dmask) & 3u) == 0) k_edt_axis0<4><<<grid_for((uint64_t)d.n1 * d.n2 / 4), TPB>>>(dmask, reinterpret_cast<uint16_t*>(G2), G, d);
    else k_edt_axis0<1><<<grid_for((uint64_t)d.n1 * d.n2), TPB>>>(dmask, reinterpret_cast<uint16_t*>(G2), G, d);
    auto egrid = [](Dims e) { return (int)std::min<uint64_t>(65535u * 16u, ((uint64_t)e.n0 * edt_chunks(e.n2) * 64u + EDT_TPB - 1) / EDT_TPB); };
    if (small) k_edt_envelope<int32_t><<<egrid(d), EDT_TPB>>>(G, G2, SP, d);
    else k_edt_envelope<long long><<<egrid(d), EDT_TPB>>>(G, G2, SP, d);
    const int tgrid = (int)std::min<uint64_t>(65535u * 4u, (uint64_t)d.n0 * ((d.n1 + 63) / 64) * ((d.n2 + 63) / 64));
    k_transpose12<int32_t><<<tgrid, 256>>>(G2, G, d.n0, d.n1, d.n2);        // G = [n0][n2][n1]
    Dims dt = {d.n0, d.n2, d.n1};
    if (small) k_edt_envelope<int32_t><<<egrid(dt), EDT_TPB>>>(G, G2, SP, dt);
    else k_edt_envelope<long long><<<egrid(dt), EDT_TPB>>>(G, G2, SP, dt);
    if (dist) k_transpose12<double><<<tgrid, 256>>>(G2, dist, d.n0, d.n2, d.n1);
    else k_transpose12<int32_t><<<tgrid, 256>>>(G2, G, d.n0, d.n2, d.n1);   // back to [n0][n1][n2]
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(G2); (void)hipFree(SP);
    if (e != hipSuccess) { g_err = std::string("EDT kernels: ") + hipGetErrorString(e); return VRG_E_INTERNAL; }
    return VRG_OK;
}

// ---------------------------------------------------------------- connected components
// parent pointers are read at agent scope (L2), never from a CU's possibly stale L1 line
__device__ __forceinline__ int32_t cc_ld(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t cc_find(int32_t* parent, int32_t x) {
    int32_t p = cc_ld(parent + x);
    while (p != x) { int32_t g = cc_ld(parent + p); if (g != p) parent[x] = g; x = p; p = g; }   // path halving
    return x;
}
__device__ __forceinline__ void cc_union(int32_t* parent, int32_t a, int32_t b) {
    for (;;) {
        a = cc_find(parent, a); b = cc_find(parent, b);
        if (a == b) return;
        if (a < b) { int32_t t = a; a = b; b = t; }              // link the larger root under the smaller ...
        int32_t old = atomicCAS(&parent[a], a, b);               // ... only while it still is a root
        if (old == a) return;
        a = old;                                                 // it was linked meanwhile: continue from its parent
    }
}
// (vec: vol is 4-byte aligned - four voxels per thread and turn: 256 bytes of voxels and 1 KB of parents per wave)
__global__ void __launch_bounds__(TPB) k_cc_init(const uint8_t* __restrict__ vol, int32_t* __restrict__ parent, uint32_t V, int vec) {
    uint32_t done = 0;
    if (vec) {
        const uchar4* __restrict__ v4 = reinterpret_cast<const uchar4*>(vol);
        int4* __restrict__ p4 = reinterpret_cast<int4*>(parent);
        const uint32_t nq = V / 4u;
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) {
            const uchar4 b = v4[i];
            const int32_t j = (int32_t)(4u * i);
            p4[i] = make_int4(b.x ? j : -1, b.y ? j + 1 : -1, b.z ? j + 2 : -1, b.w ? j + 3 : -1);
        }
        done = nq * 4u;
    }
    for (uint32_t i = done + blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x)
        parent[i] = vol[i] ? (int32_t)i : -1;
}
__global__ void k_cc_union(int32_t* __restrict__ parent, Dims d, int connectivity) {
    const uint32_t V = (uint32_t)d.n0 * d.n1 * d.n2;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
        if (parent[i] < 0) continue;
        int32_t i2 = i % d.n2, i1 = (i / d.n2) % d.n1, i0 = i / ((uint32_t)d.n2 * d.n1);
        for (int k = 14; k < 27; k++) {                          // forward half of the 3x3x3 neighbourhood
            int a = k / 9 - 1, b = (k / 3) % 3 - 1, c = k % 3 - 1;
            if ((a != 0) + (b != 0) + (c != 0) > connectivity) continue;
            int32_t j0 = i0 + a, j1 = i1 + b, j2 = i2 + c;
            if (j0 < 0 || j1 < 0 || j2 < 0 || j0 >= d.n0 || j1 >= d.n1 || j2 >= d.n2) continue;
            uint32_t j = ((uint32_t)j0 * d.n1 + j1) * d.n2 + j2;
            if (parent[j] >= 0) cc_union(parent, (int32_t)i, (int32_t)j);
        }
    }
}
// roots go to their own array and the walk is read-only: a path-halving store racing with another
// thread's final store would otherwise leave a non-root in the flattened array.  Which voxels ARE roots goes into a bitmap,
// one 64-bit word per wave (a ballot - the wave's 64 voxels are consecutive and 64-aligned: blocks of 256 threads, strides
// that are multiples of 256), instead of a 4-byte flag per voxel.
__global__ void __launch_bounds__(TPB) k_cc_flatten(const int32_t* __restrict__ parent, int32_t* __restrict__ root, unsigned long long* __restrict__ rootbits, uint32_t V) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
        int32_t x = parent[i];
        if (x >= 0) for (;;) { int32_t p = parent[x]; if (p == x) break; x = p; }
        root[i] = x;                                           // (-1: background)
        const unsigned long long w = __ballot(x == (int32_t)i);
        if ((threadIdx.x & 63u) == 0u) rootbits[i >> 6] = w;   // (lane 0 is active whenever a lane of its wave is)
    }
}
// Component number - 1 of the root at voxel r = the roots before it in raster order (what skimage / scipy number by): the
// roots before its 64-voxel word (off: an exclusive scan over the words' popcounts - V / 64 elements instead of the V-element
// scan of a flag array, which was the slowest kernel of the labelling, 1.5 ms at 880x880x640, and 8 bytes per voxel of
// scratch) + the roots below it inside the word.  Both arrays together are 12 bytes per 64 voxels: cache-resident.
struct CcRank { const unsigned long long* bits; const uint32_t* off; };
__device__ __forceinline__ uint32_t cc_rank(CcRank k, uint32_t r) {
    return k.off[r >> 6] + (uint32_t)__popcll(k.bits[r >> 6] & ((1ull << (r & 63u)) - 1ull));
}
struct PopcU64 { __host__ __device__ uint32_t operator()(unsigned long long w) const { return (uint32_t)__builtin_popcountll(w); } };
__global__ void __launch_bounds__(TPB) k_cc_sizes(const int32_t* __restrict__ root, CcRank rk, unsigned long long* __restrict__ sizes, uint32_t V) {
    const int4* __restrict__ r4 = reinterpret_cast<const int4*>(root);      // (root is this library's own array: hipMalloc-aligned)
    const uint32_t nq = V / 4u;
    auto count = [&](int32_t r) { if (r >= 0) atomicAdd(&sizes[cc_rank(rk, (uint32_t)r)], 1ull); };
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) { const int4 r = r4[i]; count(r.x); count(r.y); count(r.z); count(r.w); }
    for (uint32_t i = nq * 4u + blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) count(root[i]);
}
// (vec: four voxels per thread and turn - the caller's array, when it is a device pointer, need not be 16-byte aligned)
__global__ void __launch_bounds__(TPB) k_cc_labels(const int32_t* __restrict__ root, CcRank rk, int32_t* __restrict__ labels, uint32_t V, int vec) {
    const int4* __restrict__ r4 = reinterpret_cast<const int4*>(root);
    int4* __restrict__ l4 = reinterpret_cast<int4*>(labels);
    const uint32_t nq = vec ? V / 4u : 0u;
    auto lab = [&](int32_t r) { return r >= 0 ? (int32_t)cc_rank(rk, (uint32_t)r) + 1 : 0; };
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) {
        const int4 r = r4[i];
        l4[i] = make_int4(lab(r.x), lab(r.y), lab(r.z), lab(r.w));
    }
    for (uint32_t i = nq * 4u + blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) labels[i] = lab(root[i]);
}
__global__ void __launch_bounds__(TPB) k_cc_filter(const int32_t* __restrict__ root, CcRank rk, const unsigned long long* __restrict__ sizes,
                                                   unsigned long long min_size, uint8_t* __restrict__ out, unsigned long long* kept, uint32_t V, int vec) {
    unsigned long long local = 0;
    const int4* __restrict__ r4 = reinterpret_cast<const int4*>(root);
    uchar4* __restrict__ o4 = reinterpret_cast<uchar4*>(out);
    const uint32_t nq = vec ? V / 4u : 0u;
    auto keep = [&](int32_t r) -> uint8_t { return (r >= 0 && sizes[cc_rank(rk, (uint32_t)r)] > min_size) ? 1 : 0; };    // labelSize <= 150 is removed (:197-199)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) {
        const int4 r = r4[i];
        const uchar4 k = make_uchar4(keep(r.x), keep(r.y), keep(r.z), keep(r.w));
        o4[i] = k; local += (unsigned)k.x + k.y + k.z + k.w;
    }
    for (uint32_t i = nq * 4u + blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) { const uint8_t k = keep(root[i]); out[i] = k; local += k; }
    if (local) atomicAdd(kept, local);
}

struct CC { int32_t* parent = nullptr; int32_t* root = nullptr; unsigned long long* bits = nullptr; uint32_t* off = nullptr; unsigned long long* sizes = nullptr; void* tmp = nullptr; uint32_t ncomp = 0;
            CcRank rank() const { return CcRank{bits, off}; } };

void cc_free(CC& c) { (void)hipFree(c.parent); (void)hipFree(c.root); (void)hipFree(c.bits); (void)hipFree(c.off); (void)hipFree(c.sizes); (void)hipFree(c.tmp); }

int cc_run(const uint8_t* dvol, Dims d, int connectivity, CC& c) {
    uint32_t V = (uint32_t)d.n0 * d.n1 * d.n2;
    const size_t nw = ((size_t)V + 63) / 64;
    VM_TRY(hipMalloc(&c.parent, (size_t)V * 4)); VM_TRY(hipMalloc(&c.root, (size_t)V * 4)); VM_TRY(hipMalloc(&c.bits, nw * 8)); VM_TRY(hipMalloc(&c.off, nw * 4));
    k_cc_init<<<grid_for(((uint64_t)V + 3) / 4), TPB>>>(dvol, c.parent, V, (reinterpret_cast<uintptr_t>(dvol) & 3u) == 0 ? 1 : 0);
    k_cc_union<<<grid_for(V), TPB>>>(c.parent, d, connectivity);
    k_cc_flatten<<<grid_for(V), TPB>>>(c.parent, c.root, c.bits, V);
    auto counts = rocprim::make_transform_iterator(c.bits, PopcU64());
    size_t tb = 0;
    VM_TRY(rocprim::exclusive_scan(nullptr, tb, counts, c.off, 0u, nw, rocprim::plus<uint32_t>()));
    VM_TRY(hipMalloc(&c.tmp, tb));
    VM_TRY(rocprim::exclusive_scan(c.tmp, tb, counts, c.off, 0u, nw, rocprim::plus<uint32_t>()));
    uint32_t last_off = 0; unsigned long long last_bits = 0;
    VM_TRY(hipMemcpy(&last_off, c.off + (nw - 1), 4, hipMemcpyDeviceToHost));
    VM_TRY(hipMemcpy(&last_bits, c.bits + (nw - 1), 8, hipMemcpyDeviceToHost));
    c.ncomp = last_off + (uint32_t)__builtin_popcountll(last_bits);
    VM_TRY(hipMalloc(&c.sizes, ((size_t)c.ncomp + 1) * 8));
    VM_TRY(hipMemset(c.sizes, 0, ((size_t)c.ncomp + 1) * 8));
    k_cc_sizes<<<grid_for(((uint64_t)V + 3) / 4), TPB>>>(c.root, c.rank(), c.sizes, V);
    VM_TRY(hipGetLastError());
    return VRG_OK;
}

// ---------------------------------------------------------------- thresholds (:187-191)
// (vec: v is 16-byte aligned - whole 16-byte loads, four of them in flight per thread; one 4-byte load per turn of the loop ran
// at 2.2 TB/s)
template <class T> __global__ void __launch_bounds__(TPB) k_minmax(const T* __restrict__ v, size_t n, T* out /*[2]*/, int vec) {
    constexpr int W = 16 / (int)sizeof(T);
    T lo = v[0], hi = v[0];
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    auto acc = [&](const uint4& q) {
        T x[W]; __builtin_memcpy(x, &q, 16);
#pragma unroll
        for (int k = 0; k < W; k++) { lo = x[k] < lo ? x[k] : lo; hi = x[k] > hi ? x[k] : hi; }
    };
    size_t done = 0;
    if (vec) {
        const uint4* __restrict__ q = reinterpret_cast<const uint4*>(v);
        const size_t nq = n / W;
        size_t i = tid;
        for (; i + 3 * nth < nq; i += 4 * nth) { const uint4 a = q[i], b = q[i + nth], c = q[i + 2 * nth], d = q[i + 3 * nth]; acc(a); acc(b); acc(c); acc(d); }
        for (; i < nq; i += nth) acc(q[i]);
        done = nq * W;
    }
    for (size_t i = done + tid; i < n; i += nth) { T x = v[i]; lo = x < lo ? x : lo; hi = x > hi ? x : hi; }
    for (int o = 32; o > 0; o >>= 1) { T a = __shfl_xor(lo, o, 64), b = __shfl_xor(hi, o, 64); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
    __shared__ T sl[4], sh[4];
    if ((threadIdx.x & 63) == 0) { sl[threadIdx.x >> 6] = lo; sh[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { lo = sl[w] < lo ? sl[w] : lo; hi = sh[w] > hi ? sh[w] : hi; }
        out[2 + 2 * blockIdx.x] = lo; out[3 + 2 * blockIdx.x] = hi;
    }
}
template <class T> __global__ void k_threshold(const T* __restrict__ v, const int32_t* __restrict__ G, double edt_max, T thr1, T thr2,
                                               uint8_t* __restrict__ fg, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T x = v[i];
        if (sqrt((double)G[i]) <= edt_max && x <= thr1) x = 0;     // :187-189
        if (x <= thr2) x = 0;                                       // :190-191
        fg[i] = x != 0 ? 1 : 0;                                     // :194
    }
}

bool is_dev(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}
// device-resident copy of an input (or the pointer itself if it already is one)
template <class T> int stage(const T* src, size_t n, const T** d, void** owned) {
    *owned = nullptr;
    if (is_dev(src)) { *d = src; return VRG_OK; }
    VM_TRY(hipMalloc(owned, n * sizeof(T)));
    VM_TRY(hipMemcpy(*owned, src, n * sizeof(T), hipMemcpyHostToDevice));
    *d = (const T*)*owned;
    return VRG_OK;
}
template <class T> int deliver(T* dst, const T* dsrc, size_t n) {
    VM_TRY(hipMemcpy(dst, dsrc, n * sizeof(T), hipMemcpyDefault));
    return VRG_OK;
}
int check(int device, int64_t n0, int64_t n1, int64_t n2, Dims& d) {
    if (n0 < 1 || n1 < 1 || n2 < 1 || (double)n0 * n1 * n2 >= 2147483000.0 || n0 > 32000 || n1 > 32000 || n2 > 32000) { g_err = "shape out of range"; return VRG_E_ARG; }
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || device < 0 || device >= cnt) { (void)hipGetLastError(); g_err = "no usable HIP device (no CPU fallback)"; return VRG_E_NOGPU; }
    VM_TRY(hipSetDevice(device));
    d.n0 = (int32_t)n0; d.n1 = (int32_t)n1; d.n2 = (int32_t)n2;
    return VRG_OK;
}

template <class T> int vessel_mask_impl(const uint8_t* dbrain, const T* dves, Dims d, double edt_max, double frac1, double frac2,
                                        int64_t min_size, uint8_t* dout, int64_t* kept) {
    size_t V = (size_t)d.n0 * d.n1 * d.n2;
    int32_t* G = nullptr;
    VM_TRY(hipMalloc(&G, V * 4));
    int rc = edt_squared(dbrain, d, G);                            // distance_transform_edt(brainVolumeMask) :183
    if (rc) { (void)hipFree(G); return rc; }
    const int nb = 1024;
    T* mm = nullptr;
    VM_TRY(hipMalloc(&mm, (2 + 2 * nb) * sizeof(T)));
    k_minmax<T><<<nb, TPB>>>(dves, V, mm, (reinterpret_cast<uintptr_t>(dves) & 15u) == 0 ? 1 : 0);
    std::vector<T> h(2 + 2 * nb);
    VM_TRY(hipMemcpy(h.data(), mm, h.size() * sizeof(T), hipMemcpyDeviceToHost));
    (void)hipFree(mm);
    T lo = h[2], hi = h[3];                                        // np.amin / np.amax :187
    for (int b = 1; b < nb; b++) { lo = std::min(lo, h[2 + 2 * b]); hi = std::max(hi, h[3 + 2 * b]); }
    // numpy scalar arithmetic in the volume's dtype (float32 stays float32 under NEP 50)
    T thr1 = lo + (T)frac1 * (hi - lo), thr2 = lo + (T)frac2 * (hi - lo);
    uint8_t* fg = nullptr;
    VM_TRY(hipMalloc(&fg, V));
    k_threshold<T><<<grid_for(V), TPB>>>(dves, G, edt_max, thr1, thr2, fg, V);
    (void)hipFree(G);
    CC c;
    rc = cc_run(fg, d, 3, c);                                      // labelVolume(..., maxHop=3) :195
    if (rc) { cc_free(c); (void)hipFree(fg); return rc; }
    unsigned long long* dk = nullptr;
    VM_TRY(hipMalloc(&dk, 8)); VM_TRY(hipMemset(dk, 0, 8));
    k_cc_filter<<<grid_for(((uint64_t)V + 3) / 4), TPB>>>(c.root, c.rank(), c.sizes, (unsigned long long)min_size, dout, dk, (uint32_t)V, (reinterpret_cast<uintptr_t>(dout) & 3u) == 0 ? 1 : 0);
    unsigned long long k = 0;
    VM_TRY(hipMemcpy(&k, dk, 8, hipMemcpyDeviceToHost));
    if (kept) *kept = (int64_t)k;
    (void)hipFree(dk); (void)hipFree(fg); cc_free(c);
    return VRG_OK;
}

}  // namespace

extern "C" {

const char* vmask_last_error(void) { return g_err.c_str(); }

int vmask_edt(int device, const uint8_t* mask, int64_t n0, int64_t n1, int64_t n2, double* out) {
    Dims d;
    if (!mask || !out) { g_err = "null pointer"; return VRG_E_ARG; }
    int rc = check(device, n0, n1, n2, d);
    if (rc) return rc;
    size_t V = (size_t)n0 * n1 * n2;
    const uint8_t* dm; void* own;
    rc = stage(mask, V, &dm, &own);
    if (rc) return rc;
    int32_t* G = nullptr; double* dout = nullptr;
    VM_TRY(hipMalloc(&G, V * 4));
    const bool od = is_dev(out);
    if (od) dout = out;
    else if (hipMalloc(&dout, V * 8) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(G); if (own) (void)hipFree(own); g_err = "out of device memory (EDT output)"; return VRG_E_MEM; }
    rc = edt_squared(dm, d, G, dout);                                  // (the last transpose writes the roots: synchronised inside)
    if (!rc && !od) rc = deliver(out, (const double*)dout, V);
    if (!od) (void)hipFree(dout);
    (void)hipFree(G); if (own) (void)hipFree(own);
    return rc;
}

int vmask_label(int device, const uint8_t* volume, int64_t n0, int64_t n1, int64_t n2, int connectivity,
                int32_t* labels, int64_t* sizes, int64_t cap, int64_t* n) {
    Dims d;
    if (!volume || !labels || connectivity < 1 || connectivity > 3) { g_err = "bad argument"; return VRG_E_ARG; }
    int rc = check(device, n0, n1, n2, d);
    if (rc) return rc;
    size_t V = (size_t)n0 * n1 * n2;
    const uint8_t* dv; void* own;
    rc = stage(volume, V, &dv, &own);
    if (rc) return rc;
    CC c;
    rc = cc_run(dv, d, connectivity, c);
    if (!rc) {
        bool od = is_dev(labels);
        int32_t* dl = labels;
        if (!od) VM_TRY(hipMalloc(&dl, V * 4));
        k_cc_labels<<<grid_for(((uint64_t)V + 3) / 4), TPB>>>(c.root, c.rank(), dl, (uint32_t)V, (reinterpret_cast<uintptr_t>(dl) & 15u) == 0 ? 1 : 0);
        if (!od) { rc = deliver(labels, (const int32_t*)dl, V); (void)hipFree(dl); }
        else VM_TRY(hipDeviceSynchronize());
        if (n) *n = c.ncomp;
        if (!rc && sizes) {
            if (cap < (int64_t)c.ncomp) { g_err = "sizes buffer too small"; rc = VRG_E_ARG; }
            else if (c.ncomp) VM_TRY(hipMemcpy(sizes, c.sizes, (size_t)c.ncomp * 8, hipMemcpyDeviceToHost));
        }
    }
    cc_free(c); if (own) (void)hipFree(own);
    return rc;
}

int vmask_vessel_mask(int device, const uint8_t* brainMask, const void* vesselness, int dtype, int64_t n0, int64_t n1, int64_t n2,
                      double edt_max, double frac1, double frac2, int64_t min_size, uint8_t* out, int64_t* kept) {
    Dims d;
    if (!brainMask || !vesselness || !out || (dtype != VRG_F32 && dtype != VRG_F64)) { g_err = "bad argument"; return VRG_E_ARG; }
    int rc = check(device, n0, n1, n2, d);
    if (rc) return rc;
    size_t V = (size_t)n0 * n1 * n2;
    const uint8_t* db; void* ownb;
    rc = stage(brainMask, V, &db, &ownb);
    if (rc) return rc;
    bool od = is_dev(out);
    uint8_t* dout = out;
    if (!od) VM_TRY(hipMalloc(&dout, V));
    void* ownv = nullptr;
    if (dtype == VRG_F32) { const float* dv; rc = stage((const float*)vesselness, V, &dv, &ownv); if (!rc) rc = vessel_mask_impl<float>(db, dv, d, edt_max, frac1, frac2, min_size, dout, kept); }
    else { const double* dv; rc = stage((const double*)vesselness, V, &dv, &ownv); if (!rc) rc = vessel_mask_impl<double>(db, dv, d, edt_max, frac1, frac2, min_size, dout, kept); }
    if (!rc && !od) rc = deliver(out, (const uint8_t*)dout, V);
    if (!od) (void)hipFree(dout);
    if (ownb) (void)hipFree(ownb); if (ownv) (void)hipFree(ownv);
    return rc;
}

}  // extern "C"
