/*
 * vmask.h - C-ABI (in libvrg_hip.so) of the voxel passes on either side of the VRG stage
 * (SURVEY.md section 8 rows f2-f4): what Code/generateVesselVolume.py and its consumers do with
 * scipy / scikit-image on the CPU, as HIP kernels on MI355X.
 *
 *   vmask_edt          scipy.ndimage.distance_transform_edt(mask)       generateVesselVolume.py:183,
 *                                                                       manualCorrectionGUI.py:248 (vessel radii)
 *   vmask_label        skimage.measure.label(volume, return_num=True, connectivity=maxHop) + np.bincount
 *                                                                       generateVesselVolume.py:107-136 (labelVolume),
 *                                                                       skeletonization.py:108
 *   vmask_vessel_mask  the threshold / component-size pipeline of       generateVesselVolume.py:187-199
 *
 * All arrays are dense C-order [n0][n1][n2] (the caller's own axis order; numbering of components
 * follows that raster order exactly as skimage / scipy do).  Pointers may be host or device pointers.
 * Return 0 on success, negative VRG_E_* codes of vrg.h otherwise.
 */
#ifndef VMASK_H
#define VMASK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* out[v] = Euclidean distance (unit sampling) from v to the nearest voxel with mask == 0; 0 where mask == 0.
 * mask: uint8.  out: float64, like scipy returns. */
int vmask_edt(int device, const uint8_t* mask, int64_t n0, int64_t n1, int64_t n2, double* out);

/* Connected components of volume != 0 with skimage's `connectivity` 1 (6), 2 (18) or 3 (26 neighbours).
 * labels: int32, 0 = background, components 1..n numbered in raster order of their first voxel.
 * sizes (optional, capacity cap): voxel count of component k at sizes[k-1]; *n receives the component count. */
int vmask_label(int device, const uint8_t* volume, int64_t n0, int64_t n1, int64_t n2, int connectivity,
                int32_t* labels, int64_t* sizes, int64_t cap, int64_t* n);

/* generateVesselVolume.py:187-199 in one call:
 *   lo = min(vesselness), hi = max(vesselness)
 *   v2 = vesselness;  v2[(edt(brainMask) <= edt_max) & (v2 <= lo + frac1*(hi-lo))] = 0      (:187-189)
 *   v2[v2 <= lo + frac2*(hi-lo)] = 0                                                          (:190-191)
 *   v2 = (v2 != 0);  drop 26-connected components with size <= min_size                       (:194-199)
 * vesselness: float32 or float64 (dtype VRG_F32 / VRG_F64).  out: uint8 0/1.  *kept receives the voxel count. */
int vmask_vessel_mask(int device, const uint8_t* brainMask, const void* vesselness, int dtype,
                      int64_t n0, int64_t n1, int64_t n2, double edt_max, double frac1, double frac2,
                      int64_t min_size, uint8_t* out, int64_t* kept);

const char* vmask_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
