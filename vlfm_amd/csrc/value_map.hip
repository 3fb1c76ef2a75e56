// value_map.hip -- gfx950 kernels for vlfm.mapping.ValueMap (reference: /root/reference/vlfm/mapping/value_map.py).
//
// Data layout in HBM (per GPU, n_envs environment slots resident for the whole episode):
//   conf  [n_envs][S][S]     f32   BaseMap._map of the ValueMap          (base_map.py:23)
//   value [n_envs][S][S][C]  f32   ValueMap._value_map, channel-last     (value_map.py:66)
// Per step and observation:  colmax [W] f32 (from depth ingest), pose (64 B), values [C] f64.
//
// Kernels (all HBM/LDS integer+float work, no MFMA):
//   cone_template_kernel        one-off: sector polygon -> masked confidence template [T][T]
//   value_map_update_kernel     one workgroup per observation:
//        phase 0  depth profile -> 642-vertex polygon (value_map.py:234-257)
//        phase 1  polygon -> LDS coverage bitmap            (cv2.drawContours fill, value_map.py:260)
//        phase 2  for each template pixel: inverse-affine bilinear tap of (template & ~coverage)  (rotate_image,
//                 img_utils.py:9-28), place at the camera cell (place_img_in_img, img_utils.py:31-61) and fuse
//                 straight into conf/value (value_map.py:357-429).  Cells whose new confidence is 0 are not touched:
//                 the reference's full-map arithmetic leaves them bit-identical (w1 == 1, w2 == 0).
//   mask_unexplored_kernel      streaming full-map pass for the obstacle_map-synchronised mode (value_map.py:369-375)
//   sort_waypoints_kernel       disc median per waypoint (value_map.py:146-187, img_utils.py:213-266)
//
// Index-producing float math uses explicit round-to-nearest intrinsics (__fmul_rn ...) so that no FMA contraction
// moves a truncation boundary.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vlfm_amd.h"
#include "profile.h"
#include "raster.h"
#include "status.h"

namespace vlfm {

// ------------------------------------------------------------------------------------------------ template build
__global__ __launch_bounds__(1024) void cone_template_kernel(const float* __restrict__ conf,
                                                             const long long* __restrict__ poly, int n_poly, int T,
                                                             float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int words = (T + 31) >> 5;
    LdsBitmap bm;
    bm.solid = reinterpret_cast<unsigned*>(smem);
    bm.parity = bm.solid + T * words;
    bm.rows = T; bm.cols = T; bm.words = words;
    for (int i = threadIdx.x; i < 2 * T * words; i += blockDim.x) bm.solid[i] = 0u;
    __syncthreads();
    // EllipseEx hands 16.16 vertices to CollectPolyEdges with shift = 16: y is rounded to a scanline, x stays 16.16.
    for (int i = threadIdx.x; i < n_poly; i += blockDim.x) {
        const int j = i == 0 ? n_poly - 1 : i - 1;
        const long long ax = poly[2 * j], ay = (poly[2 * j + 1] + (XY_ONE >> 1)) >> XY_SHIFT;
        const long long bx = poly[2 * i], by = (poly[2 * i + 1] + (XY_ONE >> 1)) >> XY_SHIFT;
        raster_edge(bm, ax, (int)ay, bx, (int)by);
    }
    __syncthreads();
    resolve_rows(bm, threadIdx.x, blockDim.x);
    __syncthreads();
    for (int i = threadIdx.x; i < T * T; i += blockDim.x) {
        const int y = i / T, x = i - y * T;
        out[i] = bm_test(bm.solid, words, y, x) ? conf[i] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------ update
struct UpdateArgs {
    unsigned* colmax;         // [n][W] order-preserving keys from depth ingest (consumed and reset to 0 here)
    const double* tan_tab;    // [W]
    const float* tmpl;        // [T][T]
    const vlfm_vm_pose* pose; // [n]
    int2* vertices;           // [n][W+2] scratch: profile polygon (x=col, y=row)
    const double* values;     // [n][C]
    float* conf;              // [n_envs][S][S]
    float* value;             // [n_envs][S][S][C]
    const unsigned char* explored;  // [n_envs][S][S] or null
    int W, T, S, C;
    float depth_scale, depth_offset;  // f32(max-min), f32(min)
    float ppm_f, half_t_f;            // f32(ppm), f32(T/2.0)
    double ppm_d, half_t_d;
    int use_max_conf, fusion;
};

__device__ inline float tap(const float* __restrict__ tmpl, const unsigned* cut, int words, int T, int y, int x) {
    if ((unsigned)x >= (unsigned)T || (unsigned)y >= (unsigned)T) return 0.0f;
    if (bm_test(cut, words, y, x)) return 0.0f;
    return tmpl[y * T + x];
}

// keys -> profile polygon (global scratch), and hand the key buffer back zeroed for the next ingest
__global__ __launch_bounds__(256) void depth_profile_kernel(UpdateArgs a) {
    const int obs = blockIdx.x, T = a.T, W = a.W;
    unsigned* cm = a.colmax + (size_t)obs * W;
    int2* vert = a.vertices + (size_t)obs * (W + 2);
    for (int i = threadIdx.x; i < W; i += blockDim.x) {
        const unsigned key = cm[i];
        cm[i] = 0u;
        const float raw = __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
        const float d = __fadd_rn(__fmul_rn(raw, a.depth_scale), a.depth_offset);        // f32 (value_map.py:234)
        const float xr = __fadd_rn(__fmul_rn(d, a.ppm_f), a.half_t_f);                   // f32 (:248)
        const double yl = __dadd_rn(__dmul_rn(__dmul_rn((double)d, a.tan_tab[i]), a.ppm_d), a.half_t_d);  // f64 (:242,249)
        vert[i + 1] = make_int2((int)(long long)yl, (int)(long long)xr);                 // astype(int): truncation
    }
    if (threadIdx.x == 0) {
        vert[0] = make_int2(0, T - 1);              // [0, last_col]         (:253)
        vert[W + 1] = make_int2(T - 1, T - 1);      // [last_row, last_col]  (:254)
    }
}

// grid = (row tiles, observations).  Every workgroup rasterises the (cheap) coverage bitmap of its observation into
// its own LDS and then fuses ROWS_PER_TILE template rows: ~13x more workgroups in flight than one-per-observation,
// which is what hides the L2/HBM latency of the tap + read-modify-write chain.
constexpr int ROWS_PER_TILE = 8;

template <int C_STATIC>
__global__ __launch_bounds__(512) void value_map_update_kernel(UpdateArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = a.T, W = a.W;
    const int words = (T + 31) >> 5;
    const int n_vert = W + 2;
    // LDS carve (all 16-byte aligned): bitmaps | affine tables
    unsigned* solid = reinterpret_cast<unsigned*>(smem);
    unsigned* parity = solid + T * words;
    int* adelta = reinterpret_cast<int*>(parity + T * words + ((4 - ((2 * T * words) & 3)) & 3));
    int* bdelta = adelta + T;
    int* X0 = bdelta + T;
    int* Y0 = X0 + T;

    const int obs = blockIdx.y;
    const int row_begin = blockIdx.x * ROWS_PER_TILE;
    const int row_end = min(row_begin + ROWS_PER_TILE, T);
    const vlfm_vm_pose pose = a.pose[obs];
    const int tid = threadIdx.x, nth = blockDim.x;

    LdsBitmap bm;
    bm.solid = solid; bm.parity = parity; bm.rows = T; bm.cols = T; bm.words = words;
    for (int i = tid; i < 2 * T * words; i += nth) solid[i] = 0u;
    // affine tables of cv::warpAffine (AB_BITS = 10, INTER_BITS = 5, round_delta = 16)
    for (int i = tid; i < T; i += nth) {
        adelta[i] = __double2int_rn(__dmul_rn(__dmul_rn(pose.inv_affine[0], (double)i), 1024.0));
        bdelta[i] = __double2int_rn(__dmul_rn(__dmul_rn(pose.inv_affine[3], (double)i), 1024.0));
        X0[i] = __double2int_rn(__dmul_rn(__dadd_rn(__dmul_rn(pose.inv_affine[1], (double)i), pose.inv_affine[2]), 1024.0)) + 16;
        Y0[i] = __double2int_rn(__dmul_rn(__dadd_rn(__dmul_rn(pose.inv_affine[4], (double)i), pose.inv_affine[5]), 1024.0)) + 16;
    }
    __syncthreads();

    // ---- phase 1: coverage of the "beyond the depth profile" polygon (value_map.py:253-260)
    const int2* vert = a.vertices + (size_t)obs * n_vert;
    for (int i = tid; i < n_vert; i += nth) {
        const int2 p0 = vert[i == 0 ? n_vert - 1 : i - 1], p1 = vert[i];
        raster_edge(bm, (long long)p0.x << XY_SHIFT, p0.y, (long long)p1.x << XY_SHIFT, p1.y);
    }
    __syncthreads();
    resolve_rows(bm, tid, nth);
    __syncthreads();

    // ---- phase 2: rotate + place + fuse
    const int S = a.S;
    const int C = C_STATIC > 0 ? C_STATIC : a.C;
    float* conf = a.conf + (size_t)pose.env * S * S;
    float* value = a.value + (size_t)pose.env * S * S * C;
    const unsigned char* explored = a.explored ? a.explored + (size_t)pose.env * S * S : nullptr;
    const double* vals = a.values + (size_t)obs * C;
    const float* __restrict__ tmpl = a.tmpl;

    // one wavefront per template row, lanes along x: coalesced map accesses, no integer division
    const int lane = tid & 63, wave = tid >> 6, n_waves = nth >> 6;
    for (int y = row_begin + wave; y < row_end; y += n_waves)
    for (int x = lane; x < T; x += 64) {
        const int mr = pose.row0 + y, mc = pose.col0 + x;
        if ((unsigned)mr >= (unsigned)S || (unsigned)mc >= (unsigned)S) continue;  // place_img_in_img clipping
        const int Xq = (X0[y] + adelta[x]) >> 5, Yq = (Y0[y] + bdelta[x]) >> 5;
        const int sx = Xq >> 5, sy = Yq >> 5;
        const int fxi = Xq & 31, fyi = Yq & 31;
        const float v0 = tap(tmpl, solid, words, T, sy, sx), v1 = tap(tmpl, solid, words, T, sy, sx + 1);
        const float v2 = tap(tmpl, solid, words, T, sy + 1, sx), v3 = tap(tmpl, solid, words, T, sy + 1, sx + 1);
        if (v0 == 0.0f && v1 == 0.0f && v2 == 0.0f && v3 == 0.0f) continue;
        // BilinearTab_f weights are products of k/32 in float (exact); accumulate in double like remapBilinear<double>
        const float fx = (float)fxi * 0.03125f, fy = (float)fyi * 0.03125f;
        const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
        const double nd = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)v0, (double)w0), __dmul_rn((double)v1, (double)w1)),
                                              __dmul_rn((double)v2, (double)w2)), __dmul_rn((double)v3, (double)w3));
        float nw = (float)nd;  // curr_map is f32 (value_map.py:316-317)
        if (nw == 0.0f) continue;
        const size_t cell = (size_t)mr * S + mc;
        if (explored && explored[cell] == 0) continue;  // new_map[explored == 0] = 0 (:373); old already zeroed
        float old = conf[cell];
        if (a.fusion == VLFM_FUSE_REPLACE) {  // (:377-385)
            conf[cell] = nw;
            for (int c = 0; c < C; c++) value[cell * C + c] = (float)vals[c];
            continue;
        }
        if (a.fusion == VLFM_FUSE_EQUAL_WEIGHTING) {  // (:386-391)
            if (old > 0.0f) old = 1.0f;
            nw = 1.0f;
        }
        if (nw < 0.35f && nw < old) continue;  // decision threshold (:398-399)
        if (a.use_max_conf) {                   // (:401-408)
            if (nw > old) {
                conf[cell] = nw;
                for (int c = 0; c < C; c++) value[cell * C + c] = (float)vals[c];
            }
            continue;
        }
        // weighted average (:414-424): weights in f32, value blend in f64 (values is an f64 ndarray), conf in f32
        const float den = __fadd_rn(old, nw);
        const float w_old = __fdiv_rn(old, den), w_new = __fdiv_rn(nw, den);
        for (int c = 0; c < C; c++) {
            const double v = __dadd_rn(__dmul_rn((double)value[cell * C + c], (double)w_old), __dmul_rn(vals[c], (double)w_new));
            value[cell * C + c] = (float)v;
        }
        conf[cell] = __fadd_rn(__fmul_rn(old, w_old), __fmul_rn(nw, w_new));
    }
}

// ------------------------------------------------------------------------------------------------ full-map mask
// conf = value = 0 where explored == 0.  16 cells per thread: one 16-B explored load, 16-B conf/value accesses.
// A group whose 16 cells are all explored costs 16 bytes; otherwise conf/value are read and only rewritten when a
// non-zero cell has to be cleared, so an idle (already clean) map costs reads only.
__device__ inline bool clear_unexplored4(float4& v, unsigned e4) {
    bool dirty = false;
    if ((e4 & 0x000000FFu) == 0 && v.x != 0.0f) { v.x = 0.0f; dirty = true; }
    if ((e4 & 0x0000FF00u) == 0 && v.y != 0.0f) { v.y = 0.0f; dirty = true; }
    if ((e4 & 0x00FF0000u) == 0 && v.z != 0.0f) { v.z = 0.0f; dirty = true; }
    if ((e4 & 0xFF000000u) == 0 && v.w != 0.0f) { v.w = 0.0f; dirty = true; }
    return dirty;
}
__device__ inline bool all_explored4(unsigned e4) {
    return (e4 & 0xFFu) && (e4 & 0xFF00u) && (e4 & 0xFF0000u) && (e4 & 0xFF000000u);
}

__global__ __launch_bounds__(256) void mask_unexplored_kernel(const int* __restrict__ env_ids, int S, int C,
                                                              const unsigned char* __restrict__ explored,
                                                              float* __restrict__ conf, float* __restrict__ value) {
    const int slot = env_ids ? env_ids[blockIdx.y] : (int)blockIdx.y;
    const size_t cells = (size_t)S * S;
    const unsigned char* ex = explored + (size_t)slot * cells;
    float* cf = conf + (size_t)slot * cells;
    float* vl = value + (size_t)slot * cells * C;
    const size_t n16 = cells / 16;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n16; g += (size_t)gridDim.x * blockDim.x) {
        const uint4 e = reinterpret_cast<const uint4*>(ex)[g];
        const unsigned ew[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (all_explored4(ew[q])) continue;
            const size_t quad = g * 4 + q;  // index of this float4 of cells
            float4 c4 = reinterpret_cast<float4*>(cf)[quad];
            if (clear_unexplored4(c4, ew[q])) reinterpret_cast<float4*>(cf)[quad] = c4;
            if (C == 1) {
                float4 v4 = reinterpret_cast<float4*>(vl)[quad];
                if (clear_unexplored4(v4, ew[q])) reinterpret_cast<float4*>(vl)[quad] = v4;
            } else {
                for (int k = 0; k < 4; k++) {
                    if (((ew[q] >> (8 * k)) & 0xFFu) != 0) continue;
                    for (int c = 0; c < C; c++) {
                        float* vp = vl + (quad * 4 + k) * C + c;
                        if (*vp != 0.0f) *vp = 0.0f;
                    }
                }
            }
        }
    }
    if (blockIdx.x == 0) {  // tail when S*S is not a multiple of 16
        for (size_t i = n16 * 16 + threadIdx.x; i < cells; i += blockDim.x) {
            if (ex[i] == 0) {
                cf[i] = 0.0f;
                for (int c = 0; c < C; c++) vl[i * C + c] = 0.0f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ sort_waypoints
// One workgroup (256 threads) per (waypoint, channel).  Gathers the positive cells of the disc into LDS, then finds
// the median by rank counting (n <= (2r+1)^2, r = 10 -> 441; O(n^2) compares spread over the workgroup).
__global__ __launch_bounds__(256) void sort_waypoints_kernel(const float* __restrict__ value, int S, int C,
                                                             const int* __restrict__ cells, int radius,
                                                             const int* __restrict__ disc_hw, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // dynamic LDS only (keeps the base 16-B aligned): [0] counter, [1..2] picked middles, [4..] gathered values
    int& n_pos = *reinterpret_cast<int*>(smem);
    float* picked = reinterpret_cast<float*>(smem) + 1;
    float* vals = reinterpret_cast<float*>(smem) + 4;
    const int wp = blockIdx.x, ch = blockIdx.y;
    const int env = cells[3 * wp], row = cells[3 * wp + 1], col = cells[3 * wp + 2];
    const int side = 2 * radius + 1;
    const int r0 = row - radius < 0 ? 0 : row - radius, c0 = col - radius < 0 ? 0 : col - radius;
    const int r1 = row + radius + 1 > S ? S : row + radius + 1, c1 = col + radius + 1 > S ? S : col + radius + 1;
    const int ch_h = r1 - r0, ch_w = c1 - c0;  // crop extents; the disc stays centred at (radius, radius) OF THE CROP
    if (threadIdx.x == 0) n_pos = 0;
    __syncthreads();
    const float* vm = value + (size_t)env * S * S * C;
    for (int i = threadIdx.x; i < side * side; i += blockDim.x) {
        const int dy = i / side, dx = i - dy * side;
        if (dy >= ch_h || dx >= ch_w) continue;
        const int hw = disc_hw[dy];
        const int off = dx - radius;
        if (off < -hw || off > hw) continue;
        const float v = vm[((size_t)(r0 + dy) * S + (c0 + dx)) * C + ch];
        if (v > 0.0f) vals[atomicAdd(&n_pos, 1)] = v;
    }
    __syncthreads();
    const int n = n_pos;
    if (n == 0) {
        if (threadIdx.x == 0) out[wp * C + ch] = -1.0f;
        return;
    }
    // rank of element i = #(v_j < v_i) + #(v_j == v_i, j < i): a permutation of 0..n-1 (gather order is arbitrary,
    // which is fine: equal values are interchangeable for a median)
    const int k_hi = n / 2, k_lo = (n - 1) / 2;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float vi = vals[i];
        int rank = 0;
        for (int j = 0; j < n; j++) {
            const float vj = vals[j];
            rank += (vj < vi) || (vj == vi && j < i);
        }
        if (rank == k_hi) picked[0] = vi;
        if (rank == k_lo) picked[1] = vi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // np.median: middle element, or the mean of the two middle elements for even n (f32 add, exact halving)
        out[wp * C + ch] = (n & 1) ? picked[0] : __fmul_rn(__fadd_rn(picked[1], picked[0]), 0.5f);
    }
}

}  // namespace vlfm

// ================================================================================================ C ABI
using namespace vlfm;

extern "C" int vlfm_cone_template_build(const float* d_conf, const int64_t* d_poly_xy, int n_poly, int template_size,
                                        float* d_template, void* stream) {
    if (!d_conf || !d_poly_xy || !d_template || n_poly < 3 || template_size <= 0) return fail(VLFM_ERR_INVALID, "cone_template_build: bad argument");
    const int T = template_size, words = (T + 31) >> 5;
    const size_t lds = (size_t)2 * T * words * sizeof(unsigned);
    if (lds > 160 * 1024) return fail(VLFM_ERR_CAPACITY, "cone_template_build: template too large for LDS");
    hipLaunchKernelGGL(cone_template_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, d_conf,
                       reinterpret_cast<const long long*>(d_poly_xy), n_poly, T, d_template);
    return check_launch("cone_template_kernel");
}

static size_t update_lds_bytes(int T) {
    const int words = (T + 31) >> 5;
    size_t bm_words = (size_t)2 * T * words;
    bm_words += (4 - (bm_words & 3)) & 3;
    return bm_words * 4 + (size_t)4 * T * sizeof(int);
}

extern "C" int vlfm_value_map_update_batched(uint32_t* d_colmax_keys, int width, const double* d_tan,
                                             const float* d_template, int template_size, const vlfm_vm_pose* d_pose,
                                             const double* d_values, int n, float* d_conf, float* d_value,
                                             int map_size, int channels, int pixels_per_meter, double min_depth,
                                             double max_depth, int use_max_confidence, int fusion_type,
                                             const uint8_t* d_explored, int32_t* d_vertices, void* stream) {
    if (n == 0) return VLFM_OK;
    if (!d_colmax_keys || !d_tan || !d_template || !d_pose || !d_values || !d_conf || !d_value || n < 0 || width <= 0 ||
        template_size <= 0 || map_size <= 0 || channels <= 0 || fusion_type < 0 || fusion_type > 2)
        return fail(VLFM_ERR_INVALID, "value_map_update_batched: bad argument");
    UpdateArgs a;
    a.colmax = reinterpret_cast<unsigned*>(d_colmax_keys); a.tan_tab = d_tan; a.tmpl = d_template; a.pose = d_pose; a.values = d_values;
    a.conf = d_conf; a.value = d_value; a.explored = d_explored;
    a.W = width; a.T = template_size; a.S = map_size; a.C = channels;
    // NumPy: f32 array (op) Python float -> the scalar is rounded to f32 first (value_map.py:234,248)
    a.depth_scale = (float)(max_depth - min_depth);
    a.depth_offset = (float)min_depth;
    a.ppm_f = (float)pixels_per_meter;
    a.half_t_f = (float)(template_size / 2.0);
    a.ppm_d = (double)pixels_per_meter;
    a.half_t_d = template_size / 2.0;
    a.use_max_conf = use_max_confidence; a.fusion = fusion_type;
    if (!d_vertices) return fail(VLFM_ERR_INVALID, "value_map_update_batched: d_vertices scratch is null");
    a.vertices = reinterpret_cast<int2*>(d_vertices);
    const size_t lds = update_lds_bytes(template_size);
    if (lds > 160 * 1024) return fail(VLFM_ERR_CAPACITY, "value_map_update_batched: template too large for LDS");
    {
        VLFM_TIMED("depth_profile_kernel", stream);
        hipLaunchKernelGGL(depth_profile_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, a);
    }
    const int tiles = (template_size + ROWS_PER_TILE - 1) / ROWS_PER_TILE;
    VLFM_TIMED("value_map_update_kernel", stream);
    if (channels == 1)
        hipLaunchKernelGGL(value_map_update_kernel<1>, dim3(tiles, n), dim3(512), lds, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(value_map_update_kernel<0>, dim3(tiles, n), dim3(512), lds, (hipStream_t)stream, a);
    return check_launch("value_map_update_kernel");
}

extern "C" int vlfm_value_map_mask_unexplored_batched(const int32_t* d_env, int n, const uint8_t* d_explored,
                                                      float* d_conf, float* d_value, int map_size, int channels,
                                                      void* stream) {
    if (n == 0) return VLFM_OK;
    if (!d_explored || !d_conf || !d_value || n < 0 || map_size <= 0 || channels <= 0)
        return fail(VLFM_ERR_INVALID, "mask_unexplored_batched: bad argument");
    if (((size_t)map_size * map_size) % 4 != 0) return fail(VLFM_ERR_INVALID, "mask_unexplored_batched: S*S must be a multiple of 4");
    const size_t n16 = (size_t)map_size * map_size / 16;
    int bx = (int)((n16 + 255) / 256);
    if (bx > 512) bx = 512;
    if (bx < 1) bx = 1;
    VLFM_TIMED("mask_unexplored_kernel", stream);
    hipLaunchKernelGGL(mask_unexplored_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, d_env, map_size,
                       channels, d_explored, d_conf, d_value);
    return check_launch("mask_unexplored_kernel");
}

extern "C" int vlfm_value_map_sort_waypoints_batched(const float* d_value, int map_size, int channels,
                                                     const int32_t* d_cells, int m, int radius, const int32_t* d_disc,
                                                     float* d_out, void* stream) {
    if (m == 0) return VLFM_OK;
    if (!d_value || !d_cells || !d_disc || !d_out || m < 0 || radius < 0 || channels <= 0)
        return fail(VLFM_ERR_INVALID, "sort_waypoints_batched: bad argument");
    const int side = 2 * radius + 1;
    const size_t lds = ((size_t)side * side + 4) * sizeof(float);
    if (lds > 150 * 1024) return fail(VLFM_ERR_CAPACITY, "sort_waypoints_batched: radius too large");
    VLFM_TIMED("sort_waypoints_kernel", stream);
    hipLaunchKernelGGL(sort_waypoints_kernel, dim3(m, channels), dim3(256), lds, (hipStream_t)stream, d_value,
                       map_size, channels, d_cells, radius, d_disc, d_out);
    return check_launch("sort_waypoints_kernel");
}
