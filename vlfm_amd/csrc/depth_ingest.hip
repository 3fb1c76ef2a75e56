// depth_ingest.hip -- the one pass over each 640x480 (or 1280x720) depth image that feeds BOTH maps (gfx950).
//
// The reference reads the depth image twice per step with ~10 MB of NumPy temporaries in between:
//   ValueMap._process_local_data   np.max(depth, axis=0)                         vlfm/mapping/value_map.py:234
//   ObstacleMap.update_map         scale, mask, unproject, transform, height band, rint, scatter
//                                  vlfm/mapping/obstacle_map.py:92-101, vlfm/utils/geometry_utils.py:205-236,
//                                  vlfm/mapping/base_map.py:44-46
// Here each depth texel is loaded from HBM exactly once (16 B per lane, coalesced), reduced into a per-column
// running maximum held in registers, and -- when obstacle scatter is requested -- unprojected in f64 and stored as a
// single bit (atomicOr) into the environment's bit-packed obstacle plane.  HBM-bound: 4*H*W bytes read per
// observation.  No MFMA: this is a reduction + scatter.
//
// Work decomposition: a workgroup owns ROWS_PER_BLOCK image rows; thread (cx, ry) owns one float4 column group and
// walks rows ry, ry+RY, ...; partial maxima are combined through LDS, then one atomicMax per column and workgroup
// (column maxima are stored as order-preserving unsigned keys so that atomicMax works for any sign).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vlfm_amd.h"
#include "profile.h"
#include "status.h"

namespace vlfm {

__device__ inline unsigned f32_key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ inline float key_f32(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

struct IngestArgs {
    const float* depth;             // [n][H][W]
    const vlfm_ingest_params* prm;  // [n]
    unsigned* colmax_keys;          // [n][W] (zero-initialised keys) or null
    unsigned* obstacle;             // [n_envs][S][stride] bit-packed, or null
    int* status;                    // [n][2]: (index error, saw zero depth)
    int H, W, W4, S, stride;
    int cols_per_block;             // float4 column groups handled by one workgroup in x
    int ry;                         // rows advanced per iteration (= blockDim.x / cols_per_block)
    int rows_per_block;
    double ppm;
};

__device__ inline void scatter_point(const IngestArgs& a, const vlfm_ingest_params& p, unsigned* grid, int obs,
                                     int u, int v, float d) {
    if (d == 0.0f) {
        // a hole in the depth image.  scatter bit 1 set: hole_area_thresh == -1 semantics (obstacle_map.py:87-89), every
        // zero becomes 1.0 and therefore falls outside max_depth.  Otherwise the caller must have filled small holes
        // already; we only report that zeros were present (status word 1) and use the texel as it is.
        if (p.scatter & 2) return;
        a.status[2 * obs + 1] = 1;
    }
    const float z = __fadd_rn(__fmul_rn(d, p.depth_scale), p.depth_offset);  // obstacle_map.py:92 (f32)
    if (!(z < p.depth_max)) return;                                          // :93
    // get_point_cloud (geometry_utils.py:230-234): int64 * f32 -> f64, then / fx
    const double zd = (double)z;
    const double xc = __ddiv_rn(__dmul_rn((double)(u - a.W / 2), zd), p.fx);
    const double yc = __ddiv_rn(__dmul_rn((double)(v - a.H / 2), zd), p.fy);
    const double c0 = zd, c1 = -xc, c2 = -yc;
    // transform_points (geometry_utils.py:207-213): tf @ [c;1], left to right, then divide by w
    const double* t = p.tf;
    const double X = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(t[0], c0), __dmul_rn(t[1], c1)), __dmul_rn(t[2], c2)), t[3]);
    const double Y = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(t[4], c0), __dmul_rn(t[5], c1)), __dmul_rn(t[6], c2)), t[7]);
    const double Z = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(t[8], c0), __dmul_rn(t[9], c1)), __dmul_rn(t[10], c2)), t[11]);
    // height band (obstacle_map.py:196-197), inclusive on both sides
    if (!(Z >= p.min_height && Z <= p.max_height)) return;
    // _xy_to_px (base_map.py:44-46): rint (half-even) of (y, x) * ppm + origin; first coordinate flipped about S
    const double half = (double)(a.S / 2);
    const double colf = (double)a.S - __dadd_rn(rint(__dmul_rn(Y, a.ppm)), half);
    const double rowf = __dadd_rn(rint(__dmul_rn(X, a.ppm)), half);
    long long row = (long long)rowf, col = (long long)colf;
    // NumPy fancy-index semantics (obstacle_map.py:101): [-S, -1] wraps, anything else outside raises IndexError
    if (row >= a.S || row < -a.S || col >= a.S || col < -a.S) {
        a.status[2 * obs] = VLFM_ERR_INDEX;
        return;
    }
    if (row < 0) row += a.S;
    if (col < 0) col += a.S;
    atomicOr(&grid[(size_t)row * a.stride + (col >> 5)], 1u << (col & 31));  // few in-band points: contention-free in practice
}

template <bool SCATTER>
__global__ __launch_bounds__(1024) void depth_ingest_kernel(IngestArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* part = reinterpret_cast<float4*>(smem);  // [ry][cols_per_block]
    const int obs = blockIdx.z;
    const int cx = threadIdx.x % a.cols_per_block, ry = threadIdx.x / a.cols_per_block;
    const int col4 = blockIdx.x * a.cols_per_block + cx;
    const int r_begin = blockIdx.y * a.rows_per_block;
    const int r_end = min(r_begin + a.rows_per_block, a.H);
    const bool live = col4 < a.W4 && ry < a.ry;
    const vlfm_ingest_params p = a.prm[obs];
    const float* img = a.depth + (size_t)obs * a.H * a.W;
    unsigned* grid = nullptr;
    if (SCATTER) grid = a.obstacle + (size_t)p.env * a.S * a.stride;
    const float ninf = -__builtin_huge_valf();
    float4 m = make_float4(ninf, ninf, ninf, ninf);
    if (live) {
        // four independent 16-byte loads in flight per lane before any use (latency hiding with few waves per CU)
        constexpr int UNROLL = 4;
        for (int r0 = r_begin + ry; r0 < r_end; r0 += UNROLL * a.ry) {
            float4 d[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; k++) {
                const int r = r0 + k * a.ry;
                d[k] = r < r_end ? reinterpret_cast<const float4*>(img + (size_t)r * a.W)[col4]
                                 : make_float4(ninf, ninf, ninf, ninf);
            }
#pragma unroll
            for (int k = 0; k < UNROLL; k++) {
                const int r = r0 + k * a.ry;
                m.x = fmaxf(m.x, d[k].x); m.y = fmaxf(m.y, d[k].y); m.z = fmaxf(m.z, d[k].z); m.w = fmaxf(m.w, d[k].w);
                if (SCATTER && (p.scatter & 1) && r < r_end) {
                    const int u = col4 * 4;
                    scatter_point(a, p, grid, obs, u + 0, r, d[k].x);
                    scatter_point(a, p, grid, obs, u + 1, r, d[k].y);
                    scatter_point(a, p, grid, obs, u + 2, r, d[k].z);
                    scatter_point(a, p, grid, obs, u + 3, r, d[k].w);
                }
            }
        }
    }
    if (a.colmax_keys == nullptr) return;
    if (ry < a.ry) part[ry * a.cols_per_block + cx] = m;
    __syncthreads();
    if (ry == 0 && col4 < a.W4) {
        for (int k = 1; k < a.ry; k++) {
            const float4 o = part[k * a.cols_per_block + cx];
            m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w);
        }
        unsigned* out = a.colmax_keys + (size_t)obs * a.W + col4 * 4;
        atomicMax(out + 0, f32_key(m.x));
        atomicMax(out + 1, f32_key(m.y));
        atomicMax(out + 2, f32_key(m.z));
        atomicMax(out + 3, f32_key(m.w));
    }
}


}  // namespace vlfm

using namespace vlfm;

extern "C" int vlfm_depth_ingest_batched(const float* d_depth, int n, int height, int width,
                                         const vlfm_ingest_params* d_params, uint32_t* d_colmax_keys, uint32_t* d_obstacle,
                                         int map_size, int pixels_per_meter, int32_t* d_status, void* stream) {
    if (n == 0) return VLFM_OK;
    if (!d_depth || !d_params || !d_status || n < 0 || height <= 0 || width <= 0)
        return fail(VLFM_ERR_INVALID, "depth_ingest_batched: bad argument");
    if (width % 4 != 0) return fail(VLFM_ERR_INVALID, "depth_ingest_batched: width must be a multiple of 4");
    if (!d_colmax_keys && !d_obstacle) return VLFM_OK;
    hipStream_t s = (hipStream_t)stream;
    IngestArgs a;
    a.depth = d_depth; a.prm = d_params; a.colmax_keys = reinterpret_cast<unsigned*>(d_colmax_keys);
    a.obstacle = d_obstacle; a.status = d_status;
    a.H = height; a.W = width; a.W4 = width / 4; a.S = map_size; a.stride = (map_size + 31) / 32;
    a.ppm = (double)pixels_per_meter;
    a.cols_per_block = a.W4 < 256 ? a.W4 : 256;
    a.ry = 640 / a.cols_per_block;
    if (a.ry < 1) a.ry = 1;
    if (a.ry > 8) a.ry = 8;
    const int threads = a.cols_per_block * a.ry;
    // enough workgroups to cover 256 CUs several times over, but at least 2 loads per thread
    int rows_per_block = 4 * a.ry;
    const int gx = (a.W4 + a.cols_per_block - 1) / a.cols_per_block;
    while ((long)n * gx * ((height + rows_per_block - 1) / rows_per_block) > 4096 && rows_per_block < height) rows_per_block *= 2;
    a.rows_per_block = rows_per_block;
    const int gy = (height + rows_per_block - 1) / rows_per_block;
    const size_t lds = (size_t)threads * sizeof(float4);
    {
        VLFM_TIMED("depth_ingest_kernel", s);
        if (d_obstacle)
            hipLaunchKernelGGL(depth_ingest_kernel<true>, dim3(gx, gy, n), dim3(threads), lds, s, a);
        else
            hipLaunchKernelGGL(depth_ingest_kernel<false>, dim3(gx, gy, n), dim3(threads), lds, s, a);
    }
    return check_launch("depth_ingest_kernel");
}
