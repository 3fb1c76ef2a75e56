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

#include <stddef.h>
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

/* Optional per-kernel timing: on = 1 brackets every kernel launch of this library with hipEvents on its launch stream,
 * on = n > 1 every n-th launch of each kernel (sampling: a timed dispatch costs host time and keeps the runtime's
 * completion thread awake), 0 switches it off.  kernel_name is the device function's name (e.g. "depth_ingest_kernel"). */
int vlfm_profile_enable(int on);
int vlfm_profile_read(const char* kernel_name, double* mean_ms, int* launches);

/* How the CURRENT device's host waits (hipStreamSynchronize / hipEventSynchronize, also PyTorch's) behave: blocking != 0
 * selects hipDeviceScheduleBlockingSync (the waiting thread sleeps on the completion interrupt), 0 restores the
 * runtime's default (hipDeviceScheduleAuto: a busy wait that holds one host core for as long as the GPU runs).  With one
 * rank per GPU and ~100 ms of queued perception work per step the busy wait is what fills the node's CPU quota; the
 * reference has no counterpart (its per-step waits are blocking socket reads, vlfm/vlm/server_wrapper.py:78-109). */
int vlfm_host_wait_mode(int blocking);

/* ---------------------------------------------------------------------------------------------
 * Per-observation pose parameters of one value-map update (64 bytes, uploaded to HBM by the caller).
 * Filled on the host by vlfm_value_map_pose_params(); consumed by vlfm_value_map_update_fused_batched().
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

/* Device: d_template[T*T] = inside(sector polygon) ? d_conf[T*T] : 0, and d_template_bits [T][ceil(T/32)] = the bit
 * plane (d_template > 0).  One workgroup; once per (fov, max_depth), like the reference's cache (value_map.py:37). */
int vlfm_cone_template_build(const float* d_conf, const int64_t* d_poly_xy, int n_poly, int template_size,
                             float* d_template, uint32_t* d_template_bits, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Depth ingest: ONE pass over each depth image feeding both maps.
 *   (a) column max  -> d_colmax[n][W]      (np.max(depth, axis=0), value_map.py:234)
 *   (b) obstacle scatter -> d_obstacle[env][S][stride] bit-packed (unproject, transform, height band, rint cell, set bit;
 *       obstacle_map.py:92-101 + geometry_utils.py:205-236 + base_map.py:44-46)          [optional]
 * d_depth: [n][H][W] f32 in [0,1].
 * d_colmax_keys [n][W] u32: column maxima as order-preserving keys (0 = -inf).  The buffer must be zero when the call
 * is made: allocate it zeroed once; vlfm_value_map_update_fused_batched consumes the keys and writes the zeros back, so a
 * steady ingest -> update cadence needs no memset.  d_status [n][2] is sticky: the kernel only ever sets flags in it;
 * the caller zeroes it after reading.
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
    int32_t scatter;      /* bit 0: scatter obstacles (0 = column max only, update_obstacles=False);
                             bit 1: every zero depth texel is a filled hole (hole_area_thresh == -1, obstacle_map.py:87-89);
                             bit 2: zero texels are left to vlfm_depth_scatter_holes_batched (their fate depends on
                                    fill_small_holes, which needs the whole image first) */
} vlfm_ingest_params;     /* 152 bytes */

/* Undo journal of the speculative single depth pass (fill_small_holes mode).  The pass places every NON-zero texel before
 * fill_small_holes has run; a valid texel that lies INSIDE a filled hole contour (an "island": the reference's
 * drawContours(.., -1) covers it, it becomes 1.0 and is dropped, img_utils.py:385-388) must not count.  The pass
 * therefore records the cell id (row * S + col) of every obstacle bit it is the FIRST to set; for island frames
 * vlfm_fill_small_holes_batched clears exactly those bits and vlfm_depth_scatter_holes_batched re-places the valid
 * texels outside the filled area.  d_count must be zero on entry (allocate zeroed; vlfm_fill_small_holes_batched
 * resets it -- a caller whose vlfm_depth_ingest_batched was NOT followed by vlfm_fill_small_holes_batched, e.g. after an
 * error in between, zeroes d_count itself before the next ingest, or a later island frame would undo that step's bits).
 * In this mode a texel that falls off the map is only NOTED by the ingest (status word 1, bit 1): whether the reference
 * would have scattered it is decided by fill_small_holes, which promotes the note to VLFM_ERR_INDEX in status word 0
 * unless the frame has islands (their surviving texels are placed, and judged, again by the hole scatter).
 * Observations of one call must belong to distinct environment slots when a journal is used. */
typedef struct {
    uint32_t* d_cells;    /* [n][capacity] */
    int32_t* d_count;     /* [n] */
    int32_t capacity;     /* entries per observation; (2*ceil(reach*ppm)+3)^2 is always enough */
    int32_t reserved;
} vlfm_scatter_journal;   /* 24 bytes, HOST struct holding device pointers */

int vlfm_depth_ingest_batched(const float* d_depth, int n, int height, int width,
                              const vlfm_ingest_params* d_params,
                              uint32_t* d_colmax_keys /* [n][W] or NULL */,
                              uint32_t* d_obstacle /* [n_envs][S][ceil(S/32)] bit-packed or NULL */, int map_size, int pixels_per_meter,
                              int32_t* d_status /* [n][2] sticky: [0] VLFM_ERR_INDEX if a point fell off the map,
                                                   [1] bit 0: the image holds a zero (invalid) depth texel; bit 1: the
                                                   speculative pass (journal mode) hit a cell off the map */,
                              uint32_t* d_hole_bits /* OUT [n][H][ceil(W/32)] bit plane of (depth == 0), or NULL */,
                              const uint32_t* d_filled_bits /* IN  [n][H][ceil(W/32)] texels fill_small_holes set to
                                                               1.0 (vlfm_fill_small_holes_batched), or NULL */,
                              const vlfm_scatter_journal* journal /* host pointer or NULL (see above) */,
                              void* stream);

/* Diagnostic: the depth scatter divides by the focal lengths with a hoisted reciprocal and one FMA correction step (same bits
 * as the IEEE division, a quarter of its instructions).  Counts into d_mismatches[0] (caller zeroes it) the numerators for
 * which that quotient differs from __ddiv_rn(numerator, divisor): must stay 0. */
int vlfm_selftest_div_exact(const double* d_numerators, int n, double divisor, int32_t* d_mismatches, void* stream);

/* ---------------------------------------------------------------------------------------------
 * fill_small_holes (vlfm/utils/img_utils.py:361-390) for n depth images, on the bit plane (depth == 0) produced by
 * vlfm_depth_ingest_batched: cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE) = every outer AND hole border, in OpenCV's
 * order; each border whose cv2.contourArea is < area_thresh is drawn filled (cv2.drawContours(.., 1, -1)) into
 * d_filled_bits.  One workgroup per image; images whose status word [1] is 0 (no zero texel) cost one early exit and
 * get an all-zero d_filled_bits only if they had holes before (see d_dirty).
 *   d_status   [n][2] the ingest status (word 1: bit 0 = image has zeros, bit 1 = speculative off-map hit; consumed here)
 *   d_scratch  vlfm_hole_scratch_bytes(n, H, W, cap_pts, cap_contours) bytes
 *   d_counts   [n][4] int32: (contours traced, contours filled, overflow flags: bit 0 contour scratch, bit 1 journal,
 *              bit 0 image had zero texels | bit 1 island frame = valid texels inside a filled contour)
 *   d_params / d_obstacle / map_size / journal: only with a journal (else NULL / NULL / 0 / NULL): the env slot of each
 *              observation, the obstacle planes and the undo journal of the speculative pass
 * Single depth pass (what ObstacleMapBatch does): vlfm_depth_ingest_batched with scatter bits 0|2 and a journal places
 * every NON-zero texel and writes d_hole_bits; this call decides which zeros become 1.0 and, for island frames, takes the
 * speculative pass's new bits back; vlfm_depth_scatter_holes_batched then places the zeros that survived (depth 0 ->
 * z = min_depth) from the two bit planes and -- island frames only, which is the one case that reads d_depth again --
 * the valid texels outside the filled area.
 * (Alternative without a journal: vlfm_depth_ingest_batched with scatter bit 0 clear, this call, then a second
 * vlfm_depth_ingest_batched with d_filled_bits and d_colmax_keys = NULL.)
 * ------------------------------------------------------------------------------------------- */
size_t vlfm_hole_scratch_bytes(int n, int height, int width, int cap_pts, int cap_contours);
int vlfm_fill_small_holes_batched(const uint32_t* d_hole_bits, const int32_t* d_status, int n, int height, int width,
                                  double area_thresh, void* d_scratch, size_t scratch_bytes, int cap_pts,
                                  int cap_contours, uint32_t* d_filled_bits, int32_t* d_counts,
                                  const vlfm_ingest_params* d_params, uint32_t* d_obstacle, int map_size,
                                  const vlfm_scatter_journal* journal, void* stream);
int vlfm_depth_scatter_holes_batched(const vlfm_ingest_params* d_params, int n, int height, int width,
                                     const uint32_t* d_hole_bits, const uint32_t* d_filled_bits,
                                     const int32_t* d_hole_counts /* d_counts of vlfm_fill_small_holes_batched */,
                                     uint32_t* d_obstacle, int map_size, int pixels_per_meter, int32_t* d_status,
                                     const float* d_depth /* [n][H][W]: read for island frames only; NULL = never */,
                                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * ValueMap.update_map for n observations (value_map.py:100-128 = :221-260 + :288-319 + :357-429) in ONE launch: every workgroup
 * rasterises the depth-profile polygon of its observation into LDS itself, so no visibility plane goes through HBM and no scratch
 * is needed.
 *   d_colmax_keys [n][W]     column-max keys from depth ingest (consumed: reset to 0 by this call)
 *   d_tan      [W]           f64 tan table (vlfm_tan_table_host)
 *   d_template [T*T]         f32 masked confidence template;  d_template_bits [T][ceil(T/32)] its (> 0) bit plane
 *   d_pose     [n]           vlfm_vm_pose
 *   d_values   [n][C]        f64 values (BLIP-2 cosines)
 *   d_conf     [n_envs][S][S]      f32 confidence maps   (BaseMap._map)
 *   d_value    [n_envs][S][S][C]   f64 value maps        (ValueMap._value_map).  f64 because the reference's array IS f64
 *              after the first weighted fuse (`values` is an f64 ndarray; value_map.py:423 re-binds `_value_map` to the
 *              f64 result) and stays f64; in the modes where it stays f32 (use_max_confidence :406, fusion "replace"
 *              :381-384) the stored doubles are f32-representable.  Either way the array equals the reference's bit for bit.
 *   d_explored_bits [n_envs][S][ceil(S/32)] bit-packed ObstacleMap.explored_area when the value map was built with
 *              obstacle_map=... (value_map.py:369-375), or NULL = Habitat default (windowed update, exact).
 *   d_written_bits [n_envs][S][ceil(S/32)] bit plane, zero at reset, owned by the caller and maintained by this call:
 *              bit = the cell has received a confidence.  Required with d_explored_bits (NULL otherwise): the full-map
 *              half of _fuse_new_data (value_map.py:369-375: conf = value = 0 wherever explored == 0) then clears exactly
 *              the cells in (written & ~explored) -- every other cell is zero already.  Observations of one call must
 *              belong to distinct environment slots.
 *   d_counters [n] int32, zero on entry and zero again on exit (hand-over of the column-max key buffer).
 *   d_conf_quadrant [(T/2+1)^2] f32: rows/cols >= T/2 of the UNMASKED confidence table of vlfm_cone_template_host (the
 *              table depends on |row - T/2| and |col - T/2| only, value_map.py:343-351).  Staged in LDS so that the rotated
 *              template taps are LDS reads (required).
 * (Rounds 1-5 also exported the three-launch form this replaced -- mask_unexplored + visible_mask + fuse; removed in round 6.)
 * ------------------------------------------------------------------------------------------- */
int vlfm_value_map_update_fused_batched(uint32_t* d_colmax_keys, int width, const double* d_tan,
                                        const float* d_template, const uint32_t* d_template_bits, int template_size,
                                        const vlfm_vm_pose* d_pose, const double* d_values, int n,
                                        float* d_conf, double* d_value, int map_size, int channels, int pixels_per_meter,
                                        double min_depth, double max_depth, int use_max_confidence, int fusion_type,
                                        const uint32_t* d_explored_bits, uint32_t* d_written_bits, int32_t* d_counters,
                                        const float* d_conf_quadrant, void* stream);

/* ValueMap.sort_waypoints scoring (value_map.py:146-187 + img_utils.py:213-266): per waypoint and channel the
 * median of the positive cells inside the radius disc, -1 if none.
 *   d_cells  [m][3] int32: (env, row, col) of each waypoint (host computes them by truncation, value_map.py:164-167)
 *   d_disc   [(2r+1)] int32 half-widths per disc row (cv2.circle raster, produced by vlfm_disc_rows_host)
 *   value_is_f32  non-zero when the reference's `_value_map` is still an f32 array in this map's mode (use_max_confidence
 *            or fusion "replace", or no update yet): np.median then averages the two middle elements of an even count in
 *            f32; zero = f64 arithmetic (the default weighted mode)
 *   d_out    [m][C] f64 medians (np.median's result in the array's dtype, widened), -1 where the disc holds no positive cell
 */
int vlfm_disc_rows_host(int radius, int32_t* h_halfwidth /* [2r+1] */);
int vlfm_value_map_sort_waypoints_batched(const double* d_value, int map_size, int channels,
                                          const int32_t* d_cells, int m, int radius, const int32_t* d_disc,
                                          int value_is_f32, double* d_out, void* stream);


/* ---------------------------------------------------------------------------------------------
 * VLM-side kernels (vlfm/vlm/blip2itm.py:37-54; LAVIS eval transform + ITC head [ext])
 * ------------------------------------------------------------------------------------------- */

/* Host: Pillow's precompute_coeffs + normalize_coeffs_8bpc (bicubic, a=-0.5, antialiased) for one axis.
 * h_bounds [out_size][2] = (first tap, tap count); h_kk [out_size][ksize] 22-bit fixed-point taps. */
int vlfm_resample_coeffs_host(int in_size, int out_size, int32_t* h_bounds, int32_t* h_kk, int kk_capacity,
                              int* ksize_out);

/* Same for Pillow's BICUBIC (filter 0) or BILINEAR (filter 1) kernels. */
int vlfm_resample_coeffs_filter_host(int in_size, int out_size, int filter, int32_t* h_bounds, int32_t* h_kk,
                                     int kk_capacity, int* ksize_out);

/* Device: MobileSAM.segment_bbox preprocessing (vlfm/vlm/sam.py:54 -> SamPredictor.set_image [ext]): PIL BILINEAR resize
 * to (out_h, out_w) (longest side 1024), (x - mean) / std on the 0..255 scale, zero padding to pad x pad:
 * d_rgb [n][H][W][3] u8 -> d_out [n][3][pad][pad] f32.  d_tmp: [n][H][out_w][3] u8 scratch. */
int vlfm_preprocess_sam_batched(const uint8_t* d_rgb, int n, int height, int width, int out_h, int out_w,
                                const int32_t* d_hbounds, const int32_t* d_hk, int hksize,
                                const int32_t* d_vbounds, const int32_t* d_vk, int vksize,
                                const float* h_mean3, const float* h_std3, int pad_size, uint8_t* d_tmp, float* d_out,
                                void* stream);

/* Device: d_rgb [n][H][W][3] u8 -> d_out (out_dtype 0=f32, 1=f16, 2=bf16):
 * PIL.Image.resize((out,out), BICUBIC) (two 8-bit passes, d_tmp [n][H][out][3] u8 scratch) -> /255 -> (x-mean)/std.
 * patch_size == 0: d_out is [n][3][out][out] (ToTensor layout).  patch_size == P > 0 (P divides out): d_out is
 * [n][(out/P)^2][3*P*P], the im2col rows of a stride-P patch-embedding convolution (same values, GEMM-ready). */
int vlfm_preprocess_rgb_batched(const uint8_t* d_rgb, int n, int height, int width, int out_size,
                                const int32_t* d_hbounds, const int32_t* d_hk, int hksize,
                                const int32_t* d_vbounds, const int32_t* d_vk, int vksize,
                                const float* h_mean3, const float* h_std3, uint8_t* d_tmp, void* d_out,
                                int out_dtype, int patch_size, void* stream);

/* Device: ITC head epilogue.  d_proj [B][NQ][P] f32 = vision_proj(Q-Former query outputs) (the 768->256 GEMM itself is
 * a plain library GEMM), d_text [B][P] L2-normalised text features -> d_out [B] = max_q <normalize(proj_q), text>.
 * One wavefront per (image, query): L2 norm + dot by lane-strided loads and a 64-lane shuffle reduction, then the
 * max over queries.  NQ <= 64. */
int vlfm_itc_head_batched(const float* d_proj, int batch, int n_query, int proj_dim,
                          const float* d_text, float* d_out, void* stream);

/* y = LayerNorm(x + c) * gamma + beta over the last dimension of f16 rows [rows][dim] (f32 statistics); c = f32 [dim]
 * per-channel constant or NULL.  Used by the ViT-g blocks of BLIP-2 (blip2itm.py): the projection / fc2 GEMMs accumulate
 * straight into the residual stream and their bias vectors, summed per layer on the host once, enter through c -- the
 * two residual-add kernels per block disappear.  dim % 8 == 0, dim <= 2048; y may alias x. */
int vlfm_layernorm_bias_f16(const void* d_x, const float* d_channel_bias, const void* d_gamma, const void* d_beta,
                            void* d_y, int rows, int dim, float eps, void* stream);

/* Self-attention of the ViT-g blocks: d_qkv [batch][tokens][3][heads][head_dim] f16 (the qkv GEMM's output as it is),
 * d_out [batch][tokens][heads][head_dim] f16 = softmax(q k^T * scale) v per (image, head), ready for the projection
 * GEMM.  Specialised for tokens == 257 and head_dim == 88 (ViT-g's native head width: the qkv and projection GEMMs keep
 * their original sizes); anything else returns VLFM_ERR_INVALID and the caller uses the library attention.  A persistent
 * kernel: one workgroup per CU walks its (image, head) items with K / V / Q arriving by LDS-DMA under the previous
 * item's MFMAs (150 KB of LDS), v_mfma_f32_32x32x16_f16, f32 softmax, the CLS query on the VALU. */
int vlfm_vit_attention_f16(const void* d_qkv, void* d_out, int batch, int tokens, int heads, int head_dim, float scale,
                           void* stream);

/* C[M][N] = epilogue(X[M][K] . W[N][K]^T + bias[N]): f16 operands and result, f32 accumulation on the matrix cores
 * (hand-written MFMA kernels, csrc/gemm_f16.hip: 256 x 256 x 64 tiles, LDS-DMA staging, 8-phase schedule).  epilogue 0 = bias
 * only, 1 = bias + EXACT (erf) GELU on the f32 accumulator -- the fc1 + GELU of the ViT-g MLP inside the forward of
 * blip2itm.py:52 in one pass (hipBLASLt's fused GELU is the tanh approximation) --, 2 (ABI 6) = accumulate: C += X . W^T + bias,
 * summed in f32, the residual-stream GEMMs (projection, fc2).  K % 64 == 0, N % 8 == 0, each operand below 4 GB; d_bias may be
 * NULL. */
int vlfm_gemm_f16_nt(const void* d_x, const void* d_w, const void* d_bias, void* d_c, int m, int n, int k, int epilogue,
                     void* stream);

/* Diagnostic, host only (no GPU needed): the order in which vlfm_gemm_f16_nt's 8-phase kernels walk the 256 x 256 tiles of an
 * m x n result -- list position i -> (m-tile, n-tile) in out[2 i], out[2 i + 1]; position i runs on XCD i % 8, the persistent kernel's
 * workgroup w takes positions w, w + grid, ...  Returns the number of tiles (negative: error).  group_m <= 0: the default. */
int vlfm_gemm_f16_tile_order(int m, int n, int group_m, int* out, int capacity_pairs);
/* ... and the persistent kernel's work items for `grid` workgroups: item i -> (list position of its tile, n-half 0 / 1 or -1 = the whole
 * tile) in out[2 i], out[2 i + 1]; the tiles of a ragged last round are split into their two n-halves when that round is at most half
 * full.  Returns the number of items. */
int vlfm_gemm_f16_work_items(int m, int n, int grid, int* out, int capacity_pairs);

/* C[M][N] = act(X[M][K] . W[N][K]^T + bias[N]) + residual[M][N]: f32 operands, f32 accumulation, f32 result on the matrix cores
 * (csrc/gemm_f32.hip) -- the Linear layers of GroundingDINO, which the reference runs in fp32 (vlfm/vlm/grounding_dino.py:38-74,
 * groundingdino's build_model [ext]), and of MobileSAM's TinyViT (vlfm/vlm/sam.py:40-57).
 *   activation  0 none, 1 ReLU, 2 exact (erf) GELU; d_bias / d_residual may be NULL (d_residual may alias d_c)
 *   precision   0 = v_mfma_f32_32x32x2_f32: bit for bit a k-ordered f32 fma chain (157 TFLOP/s peak)
 *               1 = every operand as hi + 2^-11 lo' in two f16 (representation error <= 2^-24 |a|, f32's own unit roundoff), three
 *                   f16 MFMAs per product block, f32 accumulation: f32-grade results at 5.3x the matrix rate.  Needs d_w_hi / d_w_lo
 *                   = vlfm_split_f32_to_f16_pair(d_w) (once per layer) and d_overflow, a device int that the kernel ORs with 1 when
 *                   an |operand| >= 65504 (f16's range) was met: the result of that call is then invalid and the caller must
 *                   repeat it with precision 0 (vlfm_amd/vlm/ops.py:LinearF32 does, at its next host synchronisation point)
 *   K % 32 == 0; X, W, C, residual and the split planes 16-byte aligned; M and N tails are handled. */
int vlfm_split_f32_to_f16_pair(const float* d_src, void* d_hi, void* d_lo, long long count, int* d_overflow, void* stream);
int vlfm_gemm_f32_nt(const float* d_x, const float* d_w, const float* d_bias, const float* d_residual, float* d_c, int m, int n,
                     int k, int activation, int precision, const void* d_w_hi, const void* d_w_lo, int* d_overflow, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Detector-side kernels (vlfm/vlm/yolov7.py:50-110, vlfm/vlm/grounding_dino.py:38-74)
 * ------------------------------------------------------------------------------------------- */

/* The memory-bound glue of TinyViT's windowed-attention blocks (MobileSAM's image encoder behind vlfm/vlm/sam.py:54; TinyViTBlock of
 * mobile_sam/modeling/tiny_vit_sam.py [ext]) on NHWC f32 rows (csrc/sam_ops.hip):
 *   vlfm_layernorm_rows_f32     LayerNorm over `channels` of [batch][height][width] rows; window > 0 writes the rows of the
 *                               zero-padded image in window order [batch * nWy * nWx][window^2][channels] (pad + window partition +
 *                               attn.norm in one pass; padded positions receive beta = LayerNorm(0))
 *   vlfm_window_reverse_add_f32 d_x[b][y][x][:] += d_windows[row of (b, y, x) in window order][:]  (reverse + crop + residual add)
 *   vlfm_dwconv3x3_nhwc_f32     the block's depthwise 3x3 `local_conv` (stride 1, padding 1) + folded-BatchNorm bias; d_w9c [9][channels]
 * channels % 4 == 0 everywhere. */
int vlfm_layernorm_rows_f32(const float* d_x, const float* d_gamma, const float* d_beta, float* d_out, int batch, int height,
                            int width, int channels, int window, float eps, void* stream);
int vlfm_window_reverse_add_f32(float* d_x, const float* d_windows, int batch, int height, int width, int channels, int window,
                                void* stream);
int vlfm_dwconv3x3_nhwc_f32(const float* d_x, const float* d_w9c, const float* d_bias, float* d_out, int batch, int height, int width,
                            int channels, void* stream);
/* TinyViT's window attention (head width 32; `Attention` of tiny_vit_sam.py [ext]): out = softmax(scale * q k^T + bias) v per
 * (window, head).  d_qkv [windows * tokens][heads][q | k | v][32] f32 (the qkv Linear's output), d_bias_t [heads][tokens][tokens] =
 * the additive bias TRANSPOSED (bias_t[h][j][i] = bias[h][i][j]), d_out [windows * tokens][heads * 32].  tokens <= 256. */
int vlfm_window_attention_f32(const float* d_qkv, const float* d_bias_t, float* d_out, long long windows, int tokens, int heads,
                              float scale, void* stream);
/* Swin variants (the GroundingDINO backbone, vlfm/vlm/grounding_dino.py:38-74; transformers' SwinLayer [ext]): windows of the zero-padded
 * image ROLLED by -shift along both axes (0 <= shift < window); pad_zero: padded positions receive 0 (Swin pads AFTER the norm) instead of
 * beta; d_mask_t [windows_per_image][tokens][tokens] = the shifted-window attention mask, added to window w's scores as
 * d_mask_t[w % windows_per_image] (NULL = none). */
int vlfm_layernorm_rows_shifted_f32(const float* d_x, const float* d_gamma, const float* d_beta, float* d_out, int batch, int height,
                                    int width, int channels, int window, float eps, int shift, int pad_zero, void* stream);
int vlfm_window_reverse_add_shifted_f32(float* d_x, const float* d_windows, int batch, int height, int width, int channels, int window,
                                        int shift, void* stream);
int vlfm_window_attention_masked_f32(const float* d_qkv, const float* d_bias_t, const float* d_mask_t, int windows_per_image, float* d_out,
                                     long long windows, int tokens, int heads, float scale, void* stream);

/* One convolution of the yolov7-e6e graph the reference runs in fp16 (vlfm/vlm/yolov7.py:35-48,89), BatchNorm folded:
 * out = act(conv(x, w) + bias) as an implicit GEMM on the matrix cores (csrc/conv_nhwc.hip), NHWC f16, f32 accumulation.
 *   d_x    [batch][height][width] pixels, x_pix_stride elements apart; the first `cin` channels of a pixel are read
 *   d_w    [cout][ksize][ksize][cin] (the channels_last memory of the [cout][cin][k][k] filter); d_bias [cout] or NULL
 *   d_out  [batch][Ho][Wo] pixels, out_pix_stride elements apart; `cout` channels are written (Ho = (height + 2 pad -
 *          ksize) / stride + 1, pad = ksize / 2) -- so a layer can write straight into a concatenation buffer
 *   d_zero at least 16 zero bytes in device memory (what rows in the zero padding fetch)
 *   act    0 = none, 1 = SiLU
 * ksize 1 or 3, stride 1 or 2, cin % 64 == 0, cout % 8 == 0, both pixel strides % 8 == 0, tensors < 2^31 elements;
 * anything else returns VLFM_ERR_INVALID (the caller keeps such a layer on the framework's convolution). */
/* 2x2 / stride-2 max pooling of a contiguous NHWC f16 tensor (DownC's pooling branch in the same graph): even height and width,
 * channels % 8 == 0. */
int vlfm_maxpool2x2_nhwc_f16(const void* d_x, void* d_out, int batch, int height, int width, int channels, void* stream);
int vlfm_conv_nhwc_tile(int pixels, int cin, int cout, int ksize, int* tile_pixels, int* tile_channels);  /* host: the tile shape chosen */
int vlfm_conv_nhwc_f16(const void* d_x, const void* d_w, const void* d_bias, void* d_out, const void* d_zero, int batch,
                       int height, int width, int cin, int cout, int ksize, int stride, int x_pix_stride,
                       int out_pix_stride, int act, void* stream);

/* Host: cv::computeResizeAreaTab for one axis (cv2.resize INTER_AREA, shrinking or identity).  Returns the tap count
 * used (<= ktaps_capacity) or a negative status.  h_first/h_count [dsize], h_w [dsize][ktaps_capacity]. */
int vlfm_resize_area_tab_host(int ssize, int dsize, int32_t* h_first, int32_t* h_count, float* h_w, int ktaps_capacity);

/* Device: YOLOv7.predict preprocessing (yolov7.py:70-83): d_rgb [n][H][W][3] u8 -> cv2.resize(.., (out_w, out_h),
 * INTER_AREA) -> CHW -> /255 in out_dtype (0 = f32, 1 = f16): d_out [n][3][out_h][out_w]. */
int vlfm_resize_area_batched(const uint8_t* d_rgb, int n, int height, int width, int out_h, int out_w,
                             const int32_t* d_xfirst, const int32_t* d_xcount, const float* d_xw, int xktaps,
                             const int32_t* d_yfirst, const int32_t* d_ycount, const float* d_yw, int yktaps,
                             void* d_out, int out_dtype, void* stream);

/* Device: torchvision to_tensor + normalize (grounding_dino.py:52-54): d_rgb [n][H][W][3] u8 -> d_out [n][3][H][W] f32. */
int vlfm_to_tensor_normalize_batched(const uint8_t* d_rgb, int n, int height, int width, const float* h_mean3,
                                     const float* h_std3, float* d_out, void* stream);

/* Device: torchvision.ops.nms (yolov7 non_max_suppression, yolov7.py:91-99).  d_boxes_xyxy [N][4] f32, d_order [n] int32 =
 * candidate indices sorted by descending score; d_keep [max_keep] receives the kept indices in that order, d_num_keep
 * their count.  d_scratch: vlfm_nms_scratch_bytes(n).  The whole reduction runs on the device (no host sync). */
size_t vlfm_nms_scratch_bytes(int n);
int vlfm_nms(const float* d_boxes_xyxy, const int32_t* d_order, int n, float iou_threshold, void* d_scratch,
             size_t scratch_bytes, int32_t* d_keep, int32_t* d_num_keep, int max_keep, void* stream);

/* Device: multi-scale deformable attention sampling of GroundingDINO (replaces the CUDA extension of the reference's
 * groundingdino package / its grid_sample fallback, vlfm/vlm/grounding_dino.py:14,61).  d_value [B][total][heads][D] f32,
 * d_spatial_shapes [L][2] (H,W) int32, d_level_start [L] int32, d_sampling_loc [B][Q][heads][L][P][2] in [0,1],
 * d_attn_weight [B][Q][heads][L][P] -> d_out [B][Q][heads*D].  Bilinear, zero padding, align_corners = False. */
int vlfm_ms_deform_attn(const float* d_value, const int32_t* d_spatial_shapes, const int32_t* d_level_start,
                        const float* d_sampling_loc, const float* d_attn_weight, int batch, int n_query, int n_heads,
                        int head_dim, int n_levels, int n_points, int total_len, float* d_out, void* stream);
/* The same with the softmax and the sampling-location arithmetic of GroundingDinoMultiscaleDeformableAttention.forward [ext] inside the
 * kernel: d_offsets_logits [B][Q][heads*L*P*2 + heads*L*P] = the raw output of the sampling_offsets | attention_weights Linears
 * (offsets [h][l][p][2], then logits [h][l][p]), d_reference [B][Q][L][ref_coords], ref_coords 2 (points: + offset / (W_l, H_l)) or 4
 * (boxes: + offset / P * size * 0.5).  8 heads of width 32. */
int vlfm_ms_deform_attn_fused(const float* d_value, const int32_t* d_spatial_shapes, const int32_t* d_level_start,
                              const float* d_offsets_logits, const float* d_reference, int batch, int n_query, int n_heads,
                              int head_dim, int n_levels, int n_points, int ref_coords, int total_len, float* d_out, void* stream);

/* Device: depthwise 3x3 convolution, padding 1, stride 1 or 2, NCHW f32: y = conv(x, w[C][1][3][3]) + bias[C] (NULL = none),
 * followed by the exact (erf) GELU when gelu != 0.  The depthwise convolutions of MobileSAM's TinyViT encoder behind
 * vlfm/vlm/sam.py:54 (MBConv conv2, local_conv, patch merging), with their folded BatchNorm as the bias.  width and the
 * output width must be multiples of 4. */
int vlfm_dwconv3x3_f32(const float* d_x, const float* d_w, const float* d_bias, float* d_y, int n, int channels, int height,
                       int width, int stride, int gelu, void* stream);

/* Device, in place: y[n][c][:] = act(y[n][c][:] + bias[c]) on an NCHW tensor of hw elements per plane; dtype 0 = f32, 1 = f16
 * (bias has the tensor's type); act 0 = none, 1 = exact (erf) GELU, 2 = SiLU.  The bias + activation behind every convolution
 * of the detector / MobileSAM encoders once their BatchNorm is folded into the weights (yolov7.py:70-99, sam.py:54). */
int vlfm_bias_act_nchw(void* d_y, const void* d_bias, int n, int channels, long long hw, int dtype, int act, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ObjectPointCloudMap._extract_object_cloud (vlfm/mapping/object_point_cloud_map.py:150-170,186-212)
 * ------------------------------------------------------------------------------------------- */

/* cv2.erode(mask, None, iterations) -> valid_depth -> get_point_cloud over the eroded mask, in np.where order.
 * d_depth [H][W] f32, d_mask [H][W] u8 (non-zero = object).  d_cloud [capacity][3] f64 (z, -x, -y); *d_count = number of
 * masked pixels (may exceed capacity: only the first `capacity` are written).  d_scratch:
 * vlfm_object_cloud_scratch_bytes(H, W). */
size_t vlfm_object_cloud_scratch_bytes(int height, int width);
int vlfm_object_cloud_extract(const float* d_depth, const uint8_t* d_mask, int height, int width, int erosion_iterations,
                              double min_depth, double max_depth, double fx, double fy, void* d_scratch, double* d_cloud,
                              int capacity, int32_t* d_count, void* stream);

/* open3d cluster_dbscan(eps, min_points) + "largest non-noise cluster" (object_point_cloud_map.py:186-212) for n <= 8192
 * f64 points.  d_labels [n] receives the cluster root index of every point (INT32_MAX = noise), d_keep the indices of the
 * largest cluster in ascending order, d_num_keep their count (0 = only noise). */
size_t vlfm_dbscan_scratch_bytes(int n);
int vlfm_dbscan_largest_cluster(const double* d_points, int n, double eps, int min_points, void* d_scratch,
                                size_t scratch_bytes, int32_t* d_labels, int32_t* d_keep, int32_t* d_num_keep, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ObstacleMap planes are bit-packed: 1 bit per cell, row stride ceil(cols/32) u32 words, bit x&31 of word x>>5.
 * ------------------------------------------------------------------------------------------- */
int vlfm_bits_pack(const uint8_t* d_src, uint32_t* d_dst, int planes, int rows, int cols, void* stream);
int vlfm_bits_unpack(const uint32_t* d_src, uint8_t* d_dst, int planes, int rows, int cols, void* stream);

/* cv2.dilate(img, ones((kernel_h, kernel_w))) on packed planes (odd sizes, anchor centre, border ignored)
 * (obstacle_map.py:105-109,125,159-163). */
int vlfm_bits_dilate(const uint32_t* d_src, uint32_t* d_dst, int planes, int rows, int cols, int kernel_w,
                     int kernel_h, void* stream);

/* cv2.findContours(img, RETR_EXTERNAL, method) (obstacle_map.py:128-132) on packed planes, one wavefront per plane.
 * method 1 = CHAIN_APPROX_NONE, 2 = CHAIN_APPROX_SIMPLE.  Outputs per plane: d_pts [cap_pts][2] (x,y) in DISCOVERY
 * order (OpenCV returns the reverse), d_starts/d_lens [cap_contours], d_counts [3] = (contours, points, overflow). */
int vlfm_find_contours_external(const uint32_t* d_img, int planes, int rows, int cols, int method,
                                uint32_t* d_scratch /* [2][planes][rows][stride] */, int32_t* d_pts, int cap_pts,
                                int32_t* d_starts, int32_t* d_lens, int cap_contours, int32_t* d_counts, void* stream);

/* The same result from ONE 1024-thread workgroup per plane (csrc/border_parallel.h: successor tables + list ranking on a padded
 * LDS copy) -- the form the obstacle-map kernels use on their windows.  The plane must fit the LDS window
 * (3 * (rows + 2) * ((ceil(cols/32) + 2) | 1) * 4 <= 144 KB and rows * ceil(cols/32) <= 65535), else VLFM_ERR_CAPACITY.
 * d_scratch: vlfm_find_contours_wg_scratch_bytes(...) bytes.  method | 0x100 keeps the follower's tables in global memory (the
 * fallback form of the map kernels, for windows whose tables do not fit the LDS behind the window planes); without the flag the
 * tables live in LDS and every border of the plane is ranked at once (the form the map kernels use when it fits).
 * vlfm_walk_path_counters: how the borders of all launches so far were traced -- h_out4 = (borders from the ranked tables, planes /
 * windows whose per-pixel tables were in global memory (large windows), borders walked by one lane although the ranked tables
 * existed, planes / windows with every table in LDS); reset != 0 zeroes them. */
size_t vlfm_find_contours_wg_scratch_bytes(int planes, int rows, int cols, int cap_pts);
int vlfm_walk_path_counters(long long* h_out4, int reset);
int vlfm_find_contours_external_wg(const uint32_t* d_img, int planes, int rows, int cols, int method, void* d_scratch,
                                   size_t scratch_bytes, int32_t* d_pts, int cap_pts, int32_t* d_starts, int32_t* d_lens,
                                   int cap_contours, int32_t* d_counts, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ObstacleMap.update_map, explore half (obstacle_map.py:105-169) for n environments.
 * ------------------------------------------------------------------------------------------- */
#define VLFM_FOG_MAX_POLY 80
typedef struct {
    int32_t env;            /* plane slot */
    int32_t ax, ay;         /* agent cell (x = col, y = row) = BaseMap._xy_to_px(tf[:2,3]) (obstacle_map.py:115-116) */
    int32_t radius;         /* int(max_line_len) = int(max_depth * ppm) */
    int32_t n_poly;         /* vertices in poly; 0 = this environment is skipped */
    int32_t reserved;
    double rot_c, rot_s;    /* cos/sin(-angle_cv2) as reveal_fog_of_war's get_two_farthest_points evaluates them [ext] */
    double line_len;        /* max_line_len * 1.05 */
    long long poly[2 * VLFM_FOG_MAX_POLY]; /* 16.16 fixed-point FOV sector polygon (cv2.ellipse), image coordinates */
} vlfm_fog_params;

/* Host: fills vlfm_fog_params for n environments.  h_agent_px [n][2] (x,y) cells; h_angle_cv2_deg [n] =
 * rad2deg(wrap_heading(yaw + pi/2)); h_rot_cs [n][2]; sector = cv2.ellipse(centre, (R,R), 0, a - fov/2, a + fov/2). */
int vlfm_fog_params_host(const int32_t* h_agent_px, const double* h_angle_cv2_deg, const double* h_rot_cs,
                         double fov_deg, double max_line_len, const int32_t* h_env, const int32_t* h_explore, int n,
                         vlfm_fog_params* h_out);

size_t vlfm_obstacle_scratch_bytes(int n_envs, int map_size, int cap_pts, int cap_contours);

/* Device pipeline: [navigable = ~dilate(obstacle, k x k); explored &= navigable] -> fog of war reveal -> explored
 * component selection -> frontier midpoints.
 *   d_obstacle/d_navigable/d_explored  [n_envs][S][stride] bit-packed planes
 *   d_bbox      [n_envs][4] int32 persistent (ymin, ymax, xmin, xmax) of everything ever revealed; reset value
 *               (S, -1, S, -1)
 *   d_frontiers [n_envs][cap_frontiers][2] f64 pixel coordinates (x, y) == ObstacleMap._frontiers_px
 *   d_counts    [n_envs][4] int32: (n frontiers, overflow flag, n contours, n chain points)
 *   d_windows   [n][12] int32 or NULL.  Per observation three inclusive cell windows (y0, y1, x0, x1; empty when y1 < y0):
 *               [0..3] where `navigable` has to be recomputed = the union of the reach windows (camera cell +- the largest
 *               distance a scattered texel can have, + 1) of every frame ingested into this slot since its last call with
 *               update_obstacles, grown by kernel_size / 2; [4..7] where `explored` has to be masked = the caller's mirror
 *               of d_bbox BEFORE this call (no explored bit exists outside it); [8..11] where the frontier stage's derived
 *               planes are refreshed = the windows [0..3] of every call since this slot's last call with explore, united
 *               with d_bbox AFTER this call's reveal (agent cell +- (fog_radius + 2), clipped) grown by 3.  The caller
 *               passes the whole map (0, S-1, 0, S-1) after a reset of the slot's planes, for a frame whose reach window
 *               leaves the map (NumPy's negative-index wrap, obstacle_map.py:101, lands on the far side) and for the first
 *               call on fresh scratch.  NULL = the reference's full-map passes (always valid).
 *               SKIPPED observations (d_prm[k].n_poly <= 0): the kernels neither mask nor refresh anything for them, so the
 *               caller must KEEP that observation's pending windows [0..3] / [8..11] in its own bookkeeping and hand them
 *               over again with the slot's next non-skipped call (ObstacleMapBatch._take_windows does); dropping them leaves
 *               stale derived planes and stale frontiers.
 *   window_blocks_navigable / window_blocks_prepare: launch sizes (256-word workgroups per observation) for the two windowed
 *               kernels = ceil(max over the batch of rows x 32-cell words of the window(s) / 256); the kernels stride, so
 *               any positive value is correct and 0 means "size for the full plane". */
int vlfm_obstacle_map_update_batched(const vlfm_fog_params* d_prm, int n, const uint32_t* d_obstacle,
                                     uint32_t* d_navigable, uint32_t* d_explored, int32_t* d_bbox, int n_envs,
                                     int map_size, int kernel_size, int fog_radius, double area_thresh_px,
                                     void* d_scratch, size_t scratch_bytes, int cap_pts, int cap_contours,
                                     double* d_frontiers, int cap_frontiers, int32_t* d_counts, int update_obstacles,
                                     int explore, const int32_t* d_windows, int window_blocks_navigable,
                                     int window_blocks_prepare, void* stream);

/* Debug/diagnostic: copies the per-environment status words of the last pipeline run to the host. */
int vlfm_obstacle_status(const void* d_scratch, int n_envs, int map_size, int cap_pts, int cap_contours,
                         int32_t* h_out);

#ifdef __cplusplus
}
#endif
#endif /* VLFM_AMD_H */
