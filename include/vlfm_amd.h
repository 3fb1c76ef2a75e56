/*
 * vlfm_amd.h -- C ABI of libvlfm_amd.so: the MI355X (gfx950) implementation of VLFM's per-step
 * perception + mapping hot path.
 *
 * The reference has no FFI for this path: its boundary is the Python class API of
 * vlfm.mapping.{ValueMap,ObstacleMap} and vlfm.vlm.*Client.  The functions below are what a ctypes
 * binding inside those classes calls (INTEGRATION.md shows the stub); each one cites the reference
 * lines it replaces.  Conventions:
 *   - extern "C", plain pointers + sizes, no ownership transfer, no hidden allocation in step calls
 *   - pointers prefixed d_ are DEVICE pointers (HBM of the current HIP device), h_ are HOST pointers
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream)
 *   - return 0 on success, <0 = vlfm_status (mapped to the reference's Python exceptions by the host layer)
 *   - all kernels are batched over `n` environment slots; maps are [n_envs][S][S] resident in HBM
 */
#ifndef VLFM_AMD_H
#define VLFM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    VLFM_OK = 0,
    VLFM_ERR_INVALID = -1,       /* bad argument (null pointer, non-positive size, unsupported shape) */
    VLFM_ERR_OUTSIDE_MAP = -2,   /* camera / waypoint cell outside the map: AssertionError "Pixel location is
                                    outside the image." (vlfm/utils/img_utils.py:43, :235-237) */
    VLFM_ERR_INDEX = -3,         /* obstacle scatter hit a cell >= S: IndexError (vlfm/mapping/obstacle_map.py:101,
                                    caught by vlfm/policy/base_objectnav_policy.py:157-162) */
    VLFM_ERR_HIP = -4,           /* a HIP runtime call failed (see vlfm_last_error) */
    VLFM_ERR_CAPACITY = -5       /* a caller-provided scratch/output capacity was too small */
} vlfm_status;

/* fusion modes of ValueMap._fuse_new_data (vlfm/mapping/value_map.py:377-429) */
enum { VLFM_FUSE_DEFAULT = 0, VLFM_FUSE_REPLACE = 1, VLFM_FUSE_EQUAL_WEIGHTING = 2 };

const char* vlfm_last_error(void);
int vlfm_abi_version(void);

/* Optional per-kernel timing: when enabled every kernel launch of this library is bracketed by hipEvents on its
 * launch stream.  kernel_name is the device function's name (e.g. "depth_ingest_kernel"). */
int vlfm_profile_enable(int on);
int vlfm_profile_read(const char* kernel_name, double* mean_ms, int* launches);

/* ---------------------------------------------------------------------------------------------
 * Per-observation pose parameters of one value-map update (64 bytes, uploaded to HBM by the caller).
 * Filled on the host by vlfm_value_map_pose_params(); consumed by vlfm_value_map_update_batched().
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    double inv_affine[6]; /* dst->src 2x3 matrix cv::warpAffine derives from getRotationMatrix2D
                             (vlfm/utils/img_utils.py:23-26) */
    int32_t row0, col0;   /* map cell of the template's top-left corner (may be negative; clipped on device)
                             = camera cell - T/2, camera cell by truncation (vlfm/mapping/value_map.py:309-313) */
    int32_t env;          /* environment slot the observation belongs to */
    int32_t reserved;
} vlfm_vm_pose;

/* Host: restates ValueMap._localize_new_data's scalar prologue (value_map.py:297-313) + rotate_image's matrix
 * (img_utils.py:23-25) for n observations.  h_tf: [n][16] row-major camera->episodic transforms (f64).
 * h_yaw: [n] yaw angles if the caller already evaluated extract_yaw (geometry_utils.py:145-159) -- a Python host
 * passes numpy.arctan2's result so that the angle is the reference's to the last bit -- or NULL (libm atan2).
 * h_env: [n] env slot per observation (NULL = 0..n-1).  Returns VLFM_ERR_OUTSIDE_MAP if a camera cell falls
 * outside [0,S) (place_img_in_img's assertion, img_utils.py:43); *bad_index then holds the observation. */
int vlfm_value_map_pose_params(const double* h_tf, const double* h_yaw, const int32_t* h_env, int n, int map_size,
                               int pixels_per_meter, int template_size, vlfm_vm_pose* h_out, int* bad_index);

/* Host: the unmasked confidence table of ValueMap._get_confidence_mask (value_map.py:337-351) for a T x T
 * template, T = 2*int(max_depth*ppm)+1, plus the 16.16 fixed-point sector polygon cv2.ellipse rasterises for
 * ValueMap._get_blank_cone_mask (value_map.py:321-335).  h_conf: [T*T] f32.  h_poly_xy: [2*cap] int64.
 * Returns T (>0) or a negative status. */
int vlfm_cone_template_host(double fov, double max_depth, int pixels_per_meter, double min_confidence,
                            float* h_conf, int conf_capacity, int64_t* h_poly_xy, int poly_capacity,
                            int* n_poly);

/* Host: tan(linspace(-fov/2, fov/2, W)) in f64 (value_map.py:237,242). */
int vlfm_tan_table_host(double fov, int width, double* h_out);

/* Device: d_template[T*T] = inside(sector polygon) ? d_conf[T*T] : 0.  One workgroup. */
int vlfm_cone_template_build(const float* d_conf, const int64_t* d_poly_xy, int n_poly, int template_size,
                             float* d_template, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Depth ingest: ONE pass over each depth image feeding both maps.
 *   (a) column max  -> d_colmax[n][W]      (np.max(depth, axis=0), value_map.py:234)
 *   (b) obstacle scatter -> d_obstacle[env][S][S] u8 (unproject, transform, height band, rint cell, store 1;
 *       obstacle_map.py:92-101 + geometry_utils.py:205-236 + base_map.py:44-46)          [optional]
 * d_depth: [n][H][W] f32 in [0,1].
 * d_colmax_keys [n][W] u32: column maxima as order-preserving keys (0 = -inf).  The buffer must be zero when the call
 * is made: allocate it zeroed once; vlfm_value_map_update_batched consumes the keys and writes the zeros back, so a
 * steady ingest -> update cadence needs no memset.  d_status [n] is sticky: the kernel only ever writes
 * VLFM_ERR_INDEX into it; the caller zeroes it after reading.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    double tf[12];        /* first three rows of the camera->episodic 4x4 (row-major) */
    float depth_scale;    /* f32(max_depth - min_depth) */
    float depth_offset;   /* f32(min_depth) */
    float depth_max;      /* f32(max_depth) */
    float reserved0;
    double fx, fy;
    double min_height, max_height;
    int32_t env;          /* obstacle-map slot */
    int32_t scatter;      /* 0: column max only (update_obstacles=False) */
} vlfm_ingest_params;     /* 152 bytes */

int vlfm_depth_ingest_batched(const float* d_depth, int n, int height, int width,
                              const vlfm_ingest_params* d_params,
                              uint32_t* d_colmax_keys /* [n][W] or NULL */,
                              uint8_t* d_obstacle /* [n_envs][S][S] or NULL */, int map_size, int pixels_per_meter,
                              int32_t* d_status /* [n] out: 0 ok, VLFM_ERR_INDEX if a point fell off the map */,
                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * ValueMap.update_map for n observations (value_map.py:100-128 = :221-260 + :288-319 + :357-429).
 *   d_colmax_keys [n][W]     column-max keys from depth ingest (consumed: reset to 0 by this call)
 *   d_tan      [W]           f64 tan table (vlfm_tan_table_host)
 *   d_template [T*T]         f32 masked confidence template
 *   d_pose     [n]           vlfm_vm_pose
 *   d_values   [n][C]        f64 values (BLIP-2 cosines)
 *   d_conf     [n_envs][S][S]      f32 confidence maps   (BaseMap._map)
 *   d_value    [n_envs][S][S][C]   f32 value maps        (ValueMap._value_map)
 *   d_explored [n_envs][S][S] u8 or NULL: ObstacleMap.explored_area when the value map was built with
 *              obstacle_map=... (value_map.py:369-375); NULL = Habitat default (windowed update, exact).
 * With d_explored the full-map zeroing is done by vlfm_value_map_mask_unexplored_batched (call it first).
 * ------------------------------------------------------------------------------------------- */
int vlfm_value_map_update_batched(uint32_t* d_colmax_keys, int width, const double* d_tan,
                                  const float* d_template, int template_size,
                                  const vlfm_vm_pose* d_pose, const double* d_values, int n,
                                  float* d_conf, float* d_value, int map_size, int channels, int pixels_per_meter,
                                  double min_depth, double max_depth,
                                  int use_max_confidence, int fusion_type,
                                  const uint8_t* d_explored,
                                  int32_t* d_vertices /* scratch [n][width+2][2] int32 */, void* stream);

/* Full-map half of _fuse_new_data when an obstacle map is attached (value_map.py:369-375):
 * conf = value = 0 wherever explored == 0, for the n listed env slots.  Streaming, HBM-bound. */
int vlfm_value_map_mask_unexplored_batched(const int32_t* d_env /* [n] or NULL = 0..n-1 */, int n,
                                           const uint8_t* d_explored, float* d_conf, float* d_value,
                                           int map_size, int channels, void* stream);

/* ValueMap.sort_waypoints scoring (value_map.py:146-187 + img_utils.py:213-266): per waypoint and channel the
 * median of the positive cells inside the radius disc, -1 if none.
 *   d_cells  [m][3] int32: (env, row, col) of each waypoint (host computes them by truncation, value_map.py:164-167)
 *   d_disc   [(2r+1)] int32 half-widths per disc row (cv2.circle raster, produced by vlfm_disc_rows_host)
 *   d_out    [m][C] f32 medians;  d_order [m] int32 = stable descending order within each env (C==1 only)
 */
int vlfm_disc_rows_host(int radius, int32_t* h_halfwidth /* [2r+1] */);
int vlfm_value_map_sort_waypoints_batched(const float* d_value, int map_size, int channels,
                                          const int32_t* d_cells, int m, int radius, const int32_t* d_disc,
                                          float* d_out, void* stream);


/* ---------------------------------------------------------------------------------------------
 * VLM-side kernels (vlfm/vlm/blip2itm.py:37-54; LAVIS eval transform + ITC head [ext])
 * ------------------------------------------------------------------------------------------- */

/* Host: Pillow's precompute_coeffs + normalize_coeffs_8bpc (bicubic, a=-0.5, antialiased) for one axis.
 * h_bounds [out_size][2] = (first tap, tap count); h_kk [out_size][ksize] 22-bit fixed-point taps. */
int vlfm_resample_coeffs_host(int in_size, int out_size, int32_t* h_bounds, int32_t* h_kk, int kk_capacity,
                              int* ksize_out);

/* Device: d_rgb [n][H][W][3] u8 -> d_out [n][3][out][out] (out_dtype 0=f32, 1=f16, 2=bf16):
 * PIL.Image.resize((out,out), BICUBIC) (two 8-bit passes, d_tmp [n][H][out][3] u8 scratch) -> /255 -> (x-mean)/std. */
int vlfm_preprocess_rgb_batched(const uint8_t* d_rgb, int n, int height, int width, int out_size,
                                const int32_t* d_hbounds, const int32_t* d_hk, int hksize,
                                const int32_t* d_vbounds, const int32_t* d_vk, int vksize,
                                const float* h_mean3, const float* h_std3, uint8_t* d_tmp, void* d_out,
                                int out_dtype, void* stream);

/* Device: ITC head epilogue.  d_proj [B][NQ][P] f32 = vision_proj(Q-Former query outputs) (the 768->256 GEMM itself is
 * a plain library GEMM), d_text [B][P] L2-normalised text features -> d_out [B] = max_q <normalize(proj_q), text>.
 * One wavefront per (image, query): L2 norm + dot by lane-strided loads and a 64-lane shuffle reduction, then the
 * max over queries.  NQ <= 64. */
int vlfm_itc_head_batched(const float* d_proj, int batch, int n_query, int proj_dim,
                          const float* d_text, float* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VLFM_AMD_H */
