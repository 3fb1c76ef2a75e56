// value_map.hip -- gfx950 kernels for vlfm.mapping.ValueMap (reference: /root/reference/vlfm/mapping/value_map.py).
//
// Data layout in HBM (per GPU, n_envs environment slots resident for the whole episode):
//   conf  [n_envs][S][S]     f32   BaseMap._map of the ValueMap          (base_map.py:23)
//   value [n_envs][S][S][C]  f64   ValueMap._value_map, channel-last     (value_map.py:66).  f64 because the reference's
//                                    array IS f64 after the first weighted fuse (`values` is an f64 ndarray, :423); in the
//                                    modes that keep it f32 (max-confidence :406, replace :381-384) the stored doubles are
//                                    f32-representable, so the map is the reference's to the last bit in every mode
// Per step and observation:  colmax [W] f32 (from depth ingest), pose (64 B), values [C] f64.
//
// Kernels (all HBM/LDS integer+float work, no MFMA):
//   cone_template_kernel        one-off: sector polygon -> masked confidence template [T][T]
//   value_map_update_fused_kernel   one launch per step (round 1's three-launch form -- mask_unexplored, visible_mask, fuse -- left
//        the tree in round 6):
//        phase 0  depth profile -> 642-vertex polygon (value_map.py:234-257)
//        phase 1  polygon -> LDS coverage bitmap            (cv2.drawContours fill, value_map.py:260)
//        phase 2  for each template pixel: inverse-affine bilinear tap of (template & ~coverage)  (rotate_image,
//                 img_utils.py:9-28), place at the camera cell (place_img_in_img, img_utils.py:31-61) and fuse
//                 straight into conf/value (value_map.py:357-429).  Cells whose new confidence is 0 are not touched:
//                 the reference's full-map arithmetic leaves them bit-identical (w1 == 1, w2 == 0).  With an obstacle map
//                 attached, the cells in (written & ~explored) of the window's rows are cleared first (value_map.py:369-375).
//   sort_waypoints_kernel       disc median per waypoint (value_map.py:146-187, img_utils.py:213-266)
//
// Index-producing float math uses explicit round-to-nearest intrinsics (__fmul_rn ...) so that no FMA contraction
// moves a truncation boundary.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/vlfm_amd.h"
#include "profile.h"
#include "raster.h"
#include "status.h"

namespace vlfm {

// ------------------------------------------------------------------------------------------------ template build
__global__ __launch_bounds__(1024) void cone_template_kernel(const float* __restrict__ conf,
                                                             const long long* __restrict__ poly, int n_poly, int T,
                                                             float* __restrict__ out, unsigned* __restrict__ out_bits) {
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
    // bit plane of (template > 0): what the per-observation visibility mask starts from
    for (int i = threadIdx.x; i < T * words; i += blockDim.x) {
        const int y = i / words, w = i - y * words;
        unsigned bits = 0u;
        for (int b = 0; b < 32; b++) {
            const int x = w * 32 + b;
            if (x < T && bm_test(bm.solid, words, y, x) && conf[y * T + x] > 0.0f) bits |= 1u << b;
        }
        out_bits[i] = bits;
    }
}

// ------------------------------------------------------------------------------------------------ update
struct UpdateArgs {
    unsigned* colmax;         // [n][W] order-preserving keys from depth ingest (consumed and reset to 0 here)
    const double* tan_tab;    // [W]
    const float* tmpl;        // [T][T]
    const unsigned* tmpl_bits;// [T][words]  template > 0
    const vlfm_vm_pose* pose; // [n]
    const double* values;     // [n][C]
    float* conf;              // [n_envs][S][S]
    double* value;            // [n_envs][S][S][C]
    const unsigned* explored; // [n_envs][S][ceil(S/32)] bit-packed ObstacleMap.explored_area, or null
    int W, T, S, C;
    float depth_scale, depth_offset;  // f32(max-min), f32(min)
    float ppm_f, half_t_f;            // f32(ppm), f32(T/2.0)
    double ppm_d, half_t_d;
    int use_max_conf, fusion;
};

constexpr int ROWS_PER_TILE = 8;
// The fusion arithmetic of one cell (value_map.py:377-429).  Returns false when the cell is left untouched.
template <int C>
__device__ inline bool fuse_cell(const UpdateArgs& a, float nw, float old, const double* oldv, const double* vals,
                                 float& conf_out, double* value_out) {
    if (a.fusion == VLFM_FUSE_REPLACE) {  // (:377-385)
        conf_out = nw;
        for (int c = 0; c < C; c++) value_out[c] = (double)(float)vals[c];   // assignment into the f32 array (:381-384)
        return true;
    }
    if (a.fusion == VLFM_FUSE_EQUAL_WEIGHTING) {  // (:386-391)
        if (old > 0.0f) old = 1.0f;
        nw = 1.0f;
    }
    if (nw < 0.35f && nw < old) return false;  // decision threshold (:398-399)
    if (a.use_max_conf) {                       // (:401-408)
        if (!(nw > old)) return false;
        conf_out = nw;
        for (int c = 0; c < C; c++) value_out[c] = (double)(float)vals[c];   // masked assignment, array stays f32 (:406)
        return true;
    }
    // weighted average (:414-424): weights in f32; the value blend is f64 * f64(w) + f64 * f64(w) and the RESULT STAYS f64
    // (`_value_map` is re-bound to the f64 product, :423) -- kept in f64 here too, so no rounding the reference does not do
    const float den = __fadd_rn(old, nw);
    const float w_old = __fdiv_rn(old, den), w_new = __fdiv_rn(nw, den);
    for (int c = 0; c < C; c++)
        value_out[c] = __dadd_rn(__dmul_rn(oldv[c], (double)w_old), __dmul_rn(vals[c], (double)w_new));
    conf_out = __fadd_rn(__fmul_rn(old, w_old), __fmul_rn(nw, w_new));
    return true;
}

// Optional phase timing of the single-launch update (compile with -DVLFM_PHASE_TIMING; tools/vm_phase_probe.py): thread 0
// of workgroup (0, 0) stamps the constant-rate 100 MHz counter at phase boundaries.  Zero cost otherwise.
#ifdef VLFM_PHASE_TIMING
__device__ long long g_vm_phase[16];
__device__ long long g_vm_span[2048][2];     // first and last phase stamp of EVERY workgroup (observation-major): the imbalance
#define VM_PHASE(k)                                                                                       \
    do {                                                                                                  \
        __syncthreads();                                                                                  \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_vm_phase[k] = wall_clock64();        \
        if (((k) == 0 || (k) == 7) && threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 2048)       \
            g_vm_span[blockIdx.y * gridDim.x + blockIdx.x][(k) == 7] = wall_clock64();                      \
    } while (0)
#define VM_STAMP(k)                                                                                       \
    do {                                                                                                  \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_vm_phase[k] = wall_clock64();        \
    } while (0)
#else
#define VM_PHASE(k) do {} while (0)
#define VM_STAMP(k) do {} while (0)
#endif

// The same tile with the template taps taken from LDS (single-launch kernel).  `quad` is the confidence table's
// (T/2+1)^2 quadrant: the table depends on |row - T/2| and |col - T/2| only (value_map.py:343-351), and a tap is read only
// where the visible bit -- which already contains the cone sector -- is set, so template[y][x] == quad[|y-T/2|][|x-T/2|] for
// every tap taken.  With the taps in LDS there is nothing to batch in the first phase: each row's new confidence is finished
// at once and only one float per row stays live, which keeps the kernel free of register spills at 16 wavefronts.
// Round 5: the cells a tile wants to fuse (new confidence != 0) are first COLLECTED into a per-workgroup LDS list -- (row, col,
// new confidence), 8 bytes -- and fused afterwards by fuse_list() with one lane per cell.  The tile sweep above was 14 dependent
// [LDS phase -> issue map reads -> wait -> fuse -> store] round trips per wavefront (7 passes x 2 half tiles) with three quarters
// of the lanes idle in each; at 256 observations, every CU waiting on the same memory system, those round trips were 26 of the
// kernel's 56 us (tools/vm_phase_probe.py).  The list form is one or two round trips with every lane busy.  When the list is full
// (more than `cap` cells: only a cone far wider than a camera's) a wavefront fuses its rows in place, the old way.
constexpr unsigned LIST_SENTINEL = 0xFFFFFFFFu;

template <int C_STATIC>
__device__ inline void fuse_tile_lds(const UpdateArgs& a, const vlfm_vm_pose& pose, const unsigned* vis, const float* quad,
                                     const int2* row_xy0, const int4 box, int row_begin, int tid, unsigned* written,
                                     uint2* list = nullptr, int* list_n = nullptr, int cap = 0) {
    const int T = a.T, S = a.S;
    const int words = (T + 31) >> 5;
    const int C = C_STATIC > 0 ? C_STATIC : a.C;
    float* conf = a.conf + (size_t)pose.env * S * S;
    double* value = a.value + (size_t)pose.env * S * S * C;
    const int ex_stride = (S + 31) >> 5;
    const unsigned* explored = a.explored ? a.explored + (size_t)pose.env * S * ex_stride : nullptr;
    const double* vals = a.values + (size_t)pose.reserved * C;
    const int lane = tid & 63, wave = tid >> 6;
    const int cq = T >> 1, Q = cq + 1;
    constexpr int R = ROWS_PER_TILE / 2;   // rows per batch: two batches per tile keep the batch's state in registers
    for (int x = wave * 64 + lane; x < T; x += 256) {
        if (x - lane > box.w || x - lane + 63 < box.z) continue;  // wave-uniform: segment outside the cone's columns
        const int mc = pose.col0 + x;
        const bool col_ok = (unsigned)mc < (unsigned)S;
        const int adelta = __double2int_rn(__dmul_rn(__dmul_rn(pose.inv_affine[0], (double)x), 1024.0));
        const int bdelta = __double2int_rn(__dmul_rn(__dmul_rn(pose.inv_affine[3], (double)x), 1024.0));
        VM_STAMP(8);
#pragma unroll 1
      for (int half = 0; half < 2; half++) {
        const int row_base = row_begin + half * R;
        float nw[R];
        // ---- phase A: source coordinates, visibility bits and template taps (all LDS), bilinear blend.  Two sweeps over
        // the tile's rows, WITHOUT per-lane branches: (1) every lane issues its four bit-plane reads at clamped addresses --
        // 32 independent LDS reads, one wait (a short-circuit `ok && test(...)` made the compiler emit one exec-masked
        // branch and one full LDS round trip per tap: 3.6 us of this phase's 4.3); (2) the taps + blend of a row run only
        // where the wavefront has a visible tap (ballot: most 64-column segments of a row miss the cone).  The row part of
        // cv::warpAffine's fixed-point source coordinate is the same for every lane: it comes from a per-workgroup table.
        int xq[R], yq[R];
        unsigned bits = 0u;
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int y = row_base + k;
            const int2 r0c = row_xy0[y < T ? y : 0];
            xq[k] = (r0c.x + adelta) >> 5; yq[k] = (r0c.y + bdelta) >> 5;
            const int sx = xq[k] >> 5, sy = yq[k] >> 5;
            const bool row_ok = y < T && (unsigned)(pose.row0 + y) < (unsigned)S && col_ok;
            const bool vx0 = (unsigned)sx < (unsigned)T, vx1 = (unsigned)(sx + 1) < (unsigned)T;
            const bool vy0 = (unsigned)sy < (unsigned)T, vy1 = (unsigned)(sy + 1) < (unsigned)T;
            const int r0 = vy0 ? sy * words : 0, r1 = vy1 ? (sy + 1) * words : 0;
            const int c0 = vx0 ? (sx >> 5) : 0, c1 = vx1 ? ((sx + 1) >> 5) : 0;
            const unsigned w00 = vis[r0 + c0], w01 = vis[r0 + c1], w10 = vis[r1 + c0], w11 = vis[r1 + c1];
            const unsigned b0 = (row_ok && vy0 && vx0) ? ((w00 >> (sx & 31)) & 1u) : 0u;
            const unsigned b1 = (row_ok && vy0 && vx1) ? ((w01 >> ((sx + 1) & 31)) & 1u) : 0u;
            const unsigned b2 = (row_ok && vy1 && vx0) ? ((w10 >> (sx & 31)) & 1u) : 0u;
            const unsigned b3 = (row_ok && vy1 && vx1) ? ((w11 >> ((sx + 1) & 31)) & 1u) : 0u;
            bits |= (b0 | (b1 << 1) | (b2 << 2) | (b3 << 3)) << (4 * k);
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            const unsigned bk = (bits >> (4 * k)) & 15u;
            float v = 0.0f;
            if (__ballot(bk != 0u) != 0ull) {
                const int sx = xq[k] >> 5, sy = yq[k] >> 5;
                const int qy0 = min(abs(sy - cq), cq) * Q, qy1 = min(abs(sy + 1 - cq), cq) * Q;
                const int qx0 = min(abs(sx - cq), cq), qx1 = min(abs(sx + 1 - cq), cq);
                const float q0 = quad[qy0 + qx0], q1 = quad[qy0 + qx1], q2 = quad[qy1 + qx0], q3 = quad[qy1 + qx1];
                const float t0 = (bk & 1u) ? q0 : 0.0f, t1 = (bk & 2u) ? q1 : 0.0f, t2 = (bk & 4u) ? q2 : 0.0f, t3 = (bk & 8u) ? q3 : 0.0f;
                // BilinearTab_f weights are products of k/32 in float (exact); accumulate in double like remapBilinear<double>
                const float fx = (float)(xq[k] & 31) * 0.03125f, fy = (float)(yq[k] & 31) * 0.03125f;
                const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
                v = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)t0, (double)w0), __dmul_rn((double)t1, (double)w1)),
                                                __dmul_rn((double)t2, (double)w2)), __dmul_rn((double)t3, (double)w3));
            }
            nw[k] = v;  // curr_map is f32 (value_map.py:316-317); 0 = the cell is not touched
        }
        VM_STAMP(9);
        {   // nothing visible in this wavefront's 64 columns x R rows (three quarters of the cone's bounding box): no map
            // access at all -- the dummy reads of an all-idle wavefront would still cost it a full memory round trip
            bool any_on = false;
#pragma unroll
            for (int k = 0; k < R; k++) any_on |= nw[k] != 0.0f;
            if (__ballot(any_on) == 0ull) continue;
        }
        if (list) {
            // one LDS atomic per wavefront and half tile reserves the slots of its R rows; row k's cells follow row k - 1's
            unsigned long long mk[R];
            int tot = 0;
#pragma unroll
            for (int k = 0; k < R; k++) {
                mk[k] = __ballot(nw[k] != 0.0f);
                tot += __popcll(mk[k]);
            }
            int base = 0;
            if (lane == 0) base = atomicAdd(list_n, tot);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base + tot <= cap) {
                int off = base;
#pragma unroll
                for (int k = 0; k < R; k++) {
                    if (nw[k] != 0.0f) {
                        const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(mk[k] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk[k], 0u));
                        list[off + (int)below] = make_uint2(((unsigned)(pose.row0 + row_base + k) << 16) | (unsigned)mc, __float_as_uint(nw[k]));
                    }
                    off += __popcll(mk[k]);
                }
                continue;
            }
            // list full: the slots reserved above (if any are inside the list) become holes, the rows are fused in place below
            for (int j = base + lane; j < min(base + tot, cap); j += 64) list[j] = make_uint2(LIST_SENTINEL, 0u);
        }
        // ---- phase B: issue the map reads of every active pixel
        float old[R];
        double oldv1[R];
        int cell[R];
        unsigned act = 0u;
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int mr = pose.row0 + row_base + k;
            bool on = nw[k] != 0.0f;
            // new_map[explored == 0] = 0 (:373); the old values of such cells are cleared by the mask step.  Branch-free:
            // inactive lanes read word 0
            if (explored) on = on && ((explored[on ? (size_t)mr * ex_stride + (mc >> 5) : 0] >> (mc & 31)) & 1u);
            cell[k] = on ? mr * S + mc : 0;
            act |= on ? (1u << k) : 0u;
            old[k] = conf[cell[k]];
            if (C_STATIC == 1) oldv1[k] = value[cell[k]];
        }
        VM_STAMP(10);
        // ---- phase C: fuse and write back
#pragma unroll
        for (int k = 0; k < R; k++) {
            const bool on = (act >> k) & 1u;
            bool stored = false;
            if (on && C_STATIC == 1) {
                float c_out;
                double v_out;
                if (fuse_cell<1>(a, nw[k], old[k], &oldv1[k], vals, c_out, &v_out)) {
                    conf[cell[k]] = c_out;
                    value[cell[k]] = v_out;
                    stored = true;
                }
            }
            if (written && C_STATIC == 1) {
                // lanes are consecutive map columns: the stores of this row form at most three 32-cell words of the plane;
                // the lowest storing lane of each word ORs the word's bits in, and only when the plane does not show them yet
                const unsigned long long m = __ballot(stored);
                if (stored) {
                    const int lo = lane - (mc & 31);
                    const unsigned wm = lo >= 0 ? (unsigned)(m >> lo) : (unsigned)(m << (-lo));
                    if ((mc & 31) == __builtin_ctz(wm)) {
                        unsigned* wp = written + (size_t)(pose.row0 + row_base + k) * ex_stride + (mc >> 5);
                        if ((*wp & wm) != wm) atomicOr(wp, wm);
                    }
                }
            }
            if (!on || C_STATIC == 1) continue;
            {
                float c_out = 0.0f;
                bool wrote = false;
                for (int c = 0; c < C; c++) {
                    const double ov = value[(size_t)cell[k] * C + c];
                    double nv;
                    wrote = fuse_cell<1>(a, nw[k], old[k], &ov, vals + c, c_out, &nv);
                    if (!wrote) break;
                    value[(size_t)cell[k] * C + c] = nv;
                }
                if (wrote) {
                    conf[cell[k]] = c_out;
                    if (written) atomicOr(written + (size_t)(pose.row0 + row_base + k) * ex_stride + (mc >> 5), 1u << (mc & 31));
                }
            }
        }
      }
        VM_STAMP(11);
    }
}

// ---- Round 5: block-sparse sweep of the window.
// The tile sweep evaluates source coordinates and four visibility taps for EVERY cell of the cone's destination bounding box --
// ~40 instructions per cell for ~16 000-25 000 cells of which ~3 000 (the part of the cone the depth profile leaves visible) end
// up with a confidence: 50 000 wavefront-instructions per observation, 25 us of pure VALU time on a CU that holds one workgroup
// (tools/vm_phase_probe.py: "fuse tiles" 7 us for ONE pass of one workgroup at 1 environment, no memory system in the way).
// Here the window is cut into 4 x 4 blocks; a block is kept only if the source footprint of its cells (bounding box of the four
// corner cells' coordinates -- exact: the fixed-point coordinate is a monotone function of the row plus a monotone function of
// the column, so its extremes over a rectangle sit in the corners -- widened by the +1 taps) contains a visible bit.  The kept
// blocks (a few hundred) are listed in LDS and swept four per wavefront, one lane per cell, with the same per-cell arithmetic as
// the tile sweep; the cells that receive a confidence go to the cell list and are fused by fuse_list().
constexpr int VB = 4;   // block side

// any visible bit in rows [y0, y1] x columns [x0, x1] of the T x T plane (clipped)?
__device__ inline bool any_visible(const unsigned* vis, int words, int T, int x0, int x1, int y0, int y1) {
    x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, T - 1); y1 = min(y1, T - 1);
    if (x0 > x1 || y0 > y1) return false;
    const int w0 = x0 >> 5, w1 = x1 >> 5;
    unsigned acc = 0u;
    for (int y = y0; y <= y1; y++) {
        for (int w = w0; w <= w1; w++) {
            unsigned m = 0xFFFFFFFFu;
            if (w == w0) m &= 0xFFFFFFFFu << (x0 & 31);
            if (w == w1) m &= 0xFFFFFFFFu >> (31 - (x1 & 31));
            acc |= vis[y * words + w] & m;
        }
    }
    return acc != 0u;
}

// new confidence of ONE destination cell (x, y) of the window: fuse_tile_lds's phase A for a single cell
__device__ inline float cell_new_confidence(const unsigned* vis, const float* quad, int T, int words, int2 rxy, int2 cxy, bool ok) {
    const int cq = T >> 1, Q = cq + 1;
    const int xq = (rxy.x + cxy.x) >> 5, yq = (rxy.y + cxy.y) >> 5;
    const int sx = xq >> 5, sy = yq >> 5;
    const bool vx0 = (unsigned)sx < (unsigned)T, vx1 = (unsigned)(sx + 1) < (unsigned)T;
    const bool vy0 = (unsigned)sy < (unsigned)T, vy1 = (unsigned)(sy + 1) < (unsigned)T;
    const int r0 = vy0 ? sy * words : 0, r1 = vy1 ? (sy + 1) * words : 0;
    const int c0 = vx0 ? (sx >> 5) : 0, c1 = vx1 ? ((sx + 1) >> 5) : 0;
    const unsigned w00 = vis[r0 + c0], w01 = vis[r0 + c1], w10 = vis[r1 + c0], w11 = vis[r1 + c1];
    const unsigned b0 = (ok && vy0 && vx0) ? ((w00 >> (sx & 31)) & 1u) : 0u;
    const unsigned b1 = (ok && vy0 && vx1) ? ((w01 >> ((sx + 1) & 31)) & 1u) : 0u;
    const unsigned b2 = (ok && vy1 && vx0) ? ((w10 >> (sx & 31)) & 1u) : 0u;
    const unsigned b3 = (ok && vy1 && vx1) ? ((w11 >> ((sx + 1) & 31)) & 1u) : 0u;
    const unsigned bk = b0 | (b1 << 1) | (b2 << 2) | (b3 << 3);
    float v = 0.0f;
    if (__ballot(bk != 0u) != 0ull) {
        const int qy0 = min(abs(sy - cq), cq) * Q, qy1 = min(abs(sy + 1 - cq), cq) * Q;
        const int qx0 = min(abs(sx - cq), cq), qx1 = min(abs(sx + 1 - cq), cq);
        const float q0 = quad[qy0 + qx0], q1 = quad[qy0 + qx1], q2 = quad[qy1 + qx0], q3 = quad[qy1 + qx1];
        const float t0 = (bk & 1u) ? q0 : 0.0f, t1 = (bk & 2u) ? q1 : 0.0f, t2 = (bk & 4u) ? q2 : 0.0f, t3 = (bk & 8u) ? q3 : 0.0f;
        // BilinearTab_f weights are products of k/32 in float (exact); accumulate in double like remapBilinear<double>
        const float fx = (float)(xq & 31) * 0.03125f, fy = (float)(yq & 31) * 0.03125f;
        const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
        v = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)t0, (double)w0), __dmul_rn((double)t1, (double)w1)),
                                        __dmul_rn((double)t2, (double)w2)), __dmul_rn((double)t3, (double)w3));
    }
    return v;  // curr_map is f32 (value_map.py:316-317); 0 = the cell is not touched
}

// The collected cells, one per lane: all map reads of a lane's (up to four) cells are in flight together.
template <int C_STATIC>
__device__ inline void fuse_list(const UpdateArgs& a, const vlfm_vm_pose& pose, const uint2* list, int n, int tid, int nth,
                                 unsigned* written) {
    const int S = a.S;
    const int C = C_STATIC > 0 ? C_STATIC : a.C;
    float* conf = a.conf + (size_t)pose.env * S * S;
    double* value = a.value + (size_t)pose.env * S * S * C;
    const int ex_stride = (S + 31) >> 5;
    const unsigned* explored = a.explored ? a.explored + (size_t)pose.env * S * ex_stride : nullptr;
    const double* vals = a.values + (size_t)pose.reserved * C;
    constexpr int U = 4;
    for (int i0 = tid; i0 < n; i0 += nth * U) {
        unsigned rc[U];
        float nw[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * nth;
            const uint2 e = i < n ? list[i] : make_uint2(LIST_SENTINEL, 0u);
            rc[u] = e.x; nw[u] = __uint_as_float(e.y);
            on[u] = e.x != LIST_SENTINEL;
        }
        if (explored) {   // new_map[explored == 0] = 0 (value_map.py:373); the old values of such cells are cleared by the mask step
            unsigned ew[U];
#pragma unroll
            for (int u = 0; u < U; u++) ew[u] = explored[on[u] ? (size_t)(rc[u] >> 16) * ex_stride + ((rc[u] & 0xFFFFu) >> 5) : 0];
#pragma unroll
            for (int u = 0; u < U; u++) on[u] = on[u] && ((ew[u] >> (rc[u] & 31u)) & 1u);
        }
        int cell[U];
        float old[U];
        double oldv1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            cell[u] = on[u] ? (int)(rc[u] >> 16) * S + (int)(rc[u] & 0xFFFFu) : 0;
            old[u] = conf[cell[u]];
            if (C_STATIC == 1) oldv1[u] = value[cell[u]];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!on[u]) continue;
            bool stored = false;
            if (C_STATIC == 1) {
                float c_out;
                double v_out;
                if (fuse_cell<1>(a, nw[u], old[u], &oldv1[u], vals, c_out, &v_out)) {
                    conf[cell[u]] = c_out;
                    value[cell[u]] = v_out;
                    stored = true;
                }
            } else {
                float c_out = 0.0f;
                for (int c = 0; c < C; c++) {
                    const double ov = value[(size_t)cell[u] * C + c];
                    double nv;
                    stored = fuse_cell<1>(a, nw[u], old[u], &ov, vals + c, c_out, &nv);
                    if (!stored) break;
                    value[(size_t)cell[u] * C + c] = nv;
                }
                if (stored) conf[cell[u]] = c_out;
            }
            if (stored && written) {
                unsigned* wp = written + (size_t)(rc[u] >> 16) * ex_stride + ((rc[u] & 0xFFFFu) >> 5);
                const unsigned bit = 1u << (rc[u] & 31u);
                if (!(*wp & bit)) atomicOr(wp, bit);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ single-launch update
// ValueMap.update_map for n observations in ONE launch: grid = (G, n), 1024 threads.  Every workgroup of an observation
//   1. turns the column-max keys into the depth-profile polygon and rasterises it into LDS (the same arithmetic as
//      visible_mask_kernel; G-fold redundant, but LDS-only work of a few microseconds -- what it buys is that no
//      visibility plane travels through HBM and no second and third launch sit on the step's critical path);
//   2. (obstacle-map-synchronised mode, value_map.py:369-375) clears its share of the cells that hold a value but are
//      no longer explored.  Instead of sweeping conf/value (17 MB per environment in the reference's formulation) it
//      intersects two bit planes: `written` (cells that ever received a confidence; maintained by step 3) and the
//      obstacle map's `explored` -- 250 KB per environment, and only the cells in (written & ~explored) are touched.
//      Cells fused in step 3 are explored, cells cleared here are not: the two steps never meet, no ordering is needed;
//   3. fuses the row tiles g, g + G, ... of the 201 x 201 window (fuse_tile).
// The last workgroup of an observation to have read the keys hands the key buffer back zeroed (depth ingest's contract).
struct FusedExtra {
    unsigned* written;   // [n_envs][S][ceil(S/32)] or null
    int* counters;       // [n] zero on entry, zero again on exit
    const float* quad;   // [(T/2+1)^2] confidence quadrant (unmasked table, rows/cols >= T/2)
    int list_cap;        // entries of the LDS cell list behind the other dynamic LDS arrays (0 = fuse every tile in place)
    int block_cap;       // entries of the LDS list of active 4 x 4 blocks behind the cell list (0 = the tile sweep)
};

// ---- polygon raster with the work FLATTENED over the workgroup.
// raster_edge (raster.h) gives one lane a whole edge: a wavefront then runs for as long as its longest edge, and the profile
// polygon mixes hundreds of 1-2 pixel edges with a few that cross the whole template (measured: 14 of the kernel's 22 us).
// Here every edge is cut into independent items -- its boundary pixels (closed form of the 8-connected line iterator:
// pixel i of the walk has minor offset (2 dy i + dx - 1) / (2 dx), checked against the iterative form for every
// dx, dy <= 260) and its scanline crossings (xs + (y - y0) * dxdy) -- an exclusive scan of the item counts maps item ->
// (edge, k), and the items are dealt round-robin to the lanes.  Same bits as raster_edge; edges that need clipping (a
// vertex outside the T x T image: never for depth in [0, 1]) keep the serial path.
__device__ inline bool edge_inside(int2 p0, int2 p1, int T) {
    return (unsigned)p0.x < (unsigned)T && (unsigned)p0.y < (unsigned)T && (unsigned)p1.x < (unsigned)T &&
           (unsigned)p1.y < (unsigned)T;
}

__device__ inline void raster_item(const LdsBitmap& bm, int2 p0, int2 p1, int k) {
    int dx = p1.x - p0.x, dy = p1.y - p0.y;
    const int major = max(abs(dx), abs(dy));
    if (k <= major) {                       // boundary pixel k of the line walked towards +x (LineIterator, leftToRight)
        int x0 = p0.x, y0 = p0.y;
        if (dx < 0) { dx = -dx; dy = -dy; x0 = p1.x; y0 = p1.y; }
        int sy = 1;
        if (dy < 0) { dy = -dy; sy = -1; }
        const bool vert = dy > dx;
        const int minor = vert ? dx : dy;
        const int m = major > 0 ? (2 * minor * k + major - 1) / (2 * major) : 0;
        bm_or(bm, vert ? y0 + sy * k : y0 + sy * m, vert ? x0 + m : x0 + k);
        return;
    }
    // scanline crossing: half-open [min y, max y), 16.16 x advanced by the truncated slope.  |num| < 2^24 and |den| < 2^8, so
    // the f64 quotient truncates to exactly the integer quotient (its error is far below the 1/|den| gap to an integer).
    const int s = k - major - 1;
    const long long num = ((long long)(p1.x - p0.x)) << XY_SHIFT;
    const long long dxdy = (long long)((double)num / (double)(p1.y - p0.y));
    const int ylo = min(p0.y, p1.y);
    const long long xs = ((long long)(p0.y < p1.y ? p0.x : p1.x)) << XY_SHIFT;
    const long long c = xs + (long long)s * dxdy;
    const long long px = c >> XY_SHIFT;
    const int y = ylo + s;
    if ((c & (XY_ONE - 1)) == 0 && px >= 0 && px < bm.cols) bm_or(bm, y, (int)px);
    long long first = px + 1;
    if (first < 0) first = 0;
    if (first < bm.cols) bm_toggle_local(bm, y, (int)first);
}

// vert[0 .. n_vert) closed polygon; pref = n_vert + 1 ints of LDS; wave_tot = 8 ints of LDS.  All threads call it.
__device__ inline void raster_polygon_flat(const LdsBitmap& bm, const int2* vert, int n_vert, int* pref, int* wave_tot,
                                           int tid, int nth) {
    const int per = (n_vert + nth - 1) / nth;
    const int e0 = min(tid * per, n_vert), e1 = min(e0 + per, n_vert);
    int local = 0;
    for (int e = e0; e < e1; e++) {
        const int2 p0 = vert[e == 0 ? n_vert - 1 : e - 1], p1 = vert[e];
        int cnt = 0;
        // a zero-length edge (consecutive image columns that fall into the same template cell: most of them) only draws its
        // own vertex, which the neighbouring non-degenerate edges draw as their end point -- no items
        if (p0.x == p1.x && p0.y == p1.y) cnt = 0;
        else if (edge_inside(p0, p1, bm.rows)) cnt = max(abs(p1.x - p0.x), abs(p1.y - p0.y)) + 1 + abs(p1.y - p0.y);
        else raster_edge(bm, (long long)p0.x << XY_SHIFT, p0.y, (long long)p1.x << XY_SHIFT, p1.y);
        pref[e] = local;   // exclusive within this lane's chunk; the chunk base is added below
        local += cnt;
    }
    // exclusive scan of the per-lane totals: wavefront shuffle scan, then the wavefront totals through LDS
    const int lane = tid & 63, wave = tid >> 6;
    int incl = local;
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = incl - local;
    for (int w = 0; w < wave; w++) base += wave_tot[w];
    for (int e = e0; e < e1; e++) pref[e] += base;
    if (tid == nth - 1) pref[n_vert] = base + local;
    __syncthreads();
    // every lane takes one contiguous run of items: ONE search for the edge of its first item, then a walk
    const int total = pref[n_vert];
    const int chunk = (total + nth - 1) / nth;
    int j = tid * chunk;
    const int j_end = min(j + chunk, total);
    if (j < j_end) {
        int lo = 0, hi = n_vert;             // largest e with pref[e] <= j
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pref[mid] <= j) lo = mid; else hi = mid;
        }
        int next = pref[lo + 1];
        int2 p0 = vert[lo == 0 ? n_vert - 1 : lo - 1], p1 = vert[lo];
        int k = j - pref[lo];
        for (; j < j_end; j++, k++) {
            while (j >= next) {              // step over exhausted (and item-less) edges
                lo++;
                k = j - next;
                next = pref[lo + 1];
                p0 = p1;
                p1 = vert[lo];
            }
            raster_item(bm, p0, p1, k);
        }
    }
}

constexpr int FUSED_THREADS = 1024;   // 16 wavefronts: the raster's items are spread thin, then 4 tiles are fused at a time

template <int C_STATIC>
__global__ __launch_bounds__(FUSED_THREADS) void value_map_update_fused_kernel(UpdateArgs a, FusedExtra fx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = a.T, W = a.W, S = a.S;
    const int words = (T + 31) >> 5;
    const int n_vert = W + 2;
    unsigned* solid = reinterpret_cast<unsigned*>(smem);
    unsigned* parity = solid + T * words;      // after the resolve: the visible plane
    int2* vert = reinterpret_cast<int2*>(parity + T * words + ((2 * T * words) & 1));
    int* pref = reinterpret_cast<int*>(vert + n_vert);      // [n_vert + 1] item prefix sums of the flattened raster
    const int Q = (T >> 1) + 1;
    float* quad = reinterpret_cast<float*>(pref + n_vert + 1 + ((n_vert + 1) & 1));  // [Q * Q]
    int2* row_xy0 = reinterpret_cast<int2*>(quad + Q * Q + ((Q * Q) & 1));          // [T] row part of the source coordinate
    uint2* cell_list = reinterpret_cast<uint2*>(row_xy0 + T);                       // [fx.list_cap]
    int2* col_xy = reinterpret_cast<int2*>(cell_list + fx.list_cap);               // [T] column part of the source coordinate
    unsigned* block_list = reinterpret_cast<unsigned*>(col_xy + T);                // [fx.block_cap] (by << 16) | bx
    __shared__ int sh_list_n, sh_block_n;
    __shared__ int sh_wave_tot[FUSED_THREADS / 64];
    __shared__ int sh_last;
    __shared__ int sh_box[4];
    __shared__ int4 sh_dbox;
    const int obs = blockIdx.y, g = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;

    // the pose is the same for every lane: keep its 16 dwords in scalar registers (the six f64 coefficients alone would
    // otherwise pin 12 vector registers through the whole fuse loop)
    vlfm_vm_pose pose;
    {
        const int* src = reinterpret_cast<const int*>(a.pose + obs);
        int* dst = reinterpret_cast<int*>(&pose);
#pragma unroll
        for (int u = 0; u < (int)(sizeof(vlfm_vm_pose) / 4); u++) dst[u] = __builtin_amdgcn_readfirstlane(src[u]);
    }
    pose.reserved = obs;  // fuse_tile reads the observation's values through it
    const int ex_stride = (S + 31) >> 5;
    // step 2's first loads are independent of everything else: issue them before the raster so that they travel under it
    unsigned* written = fx.written ? fx.written + (size_t)pose.env * S * ex_stride : nullptr;
    const int plane_words = S * ex_stride;
    const int my_word = g * nth + tid;
    unsigned wr0 = 0u;
    if (written && a.explored && my_word < plane_words) wr0 = written[my_word];

    VM_PHASE(0);
    LdsBitmap bm;
    bm.solid = solid; bm.parity = parity; bm.rows = T; bm.cols = T; bm.words = words;
    for (int i = tid; i < 2 * T * words; i += nth) solid[i] = 0u;
    if (tid == 0) { sh_box[0] = T; sh_box[1] = -1; sh_box[2] = T; sh_box[3] = -1; sh_list_n = 0; sh_block_n = 0; }
    if (fx.block_cap > 0)
        for (int x = tid; x < T; x += nth)   // the column part of cv::warpAffine's fixed-point coordinate (fuse_tile_lds: adelta, bdelta)
            col_xy[x] = make_int2(__double2int_rn(__dmul_rn(__dmul_rn(pose.inv_affine[0], (double)x), 1024.0)),
                                  __double2int_rn(__dmul_rn(__dmul_rn(pose.inv_affine[3], (double)x), 1024.0)));
    for (int y = tid; y < T; y += nth)   // cv::warpAffine: X0 = round((M01 y + M02) * 1024) + 16, Y0 likewise (AB_BITS = 10)
        row_xy0[y] = make_int2(
            __double2int_rn(__dmul_rn(__dadd_rn(__dmul_rn(pose.inv_affine[1], (double)y), pose.inv_affine[2]), 1024.0)) + 16,
            __double2int_rn(__dmul_rn(__dadd_rn(__dmul_rn(pose.inv_affine[4], (double)y), pose.inv_affine[5]), 1024.0)) + 16);
    const unsigned* cm = a.colmax + (size_t)obs * W;
    // Loads in issue order keys + tangents, THEN the quadrant: the vertex arithmetic below only has to wait for the first two
    // (memory returns in order), and the quadrant's ten loads per lane travel under it.
    constexpr int KPT = 2;   // keys per lane held in registers (W <= 2048 with 1024 lanes); wider images loop below
    unsigned key_r[KPT];
    double tan_r[KPT];
#pragma unroll
    for (int u = 0; u < KPT; u++) {
        const int i = tid + u * nth;
        key_r[u] = i < W ? cm[i] : 0u;
        tan_r[u] = i < W ? a.tan_tab[i] : 0.0;
    }
    constexpr int QPT = 10;  // quadrant floats per lane in registers ((T/2+1)^2 <= 10240: T <= 201); larger templates loop
    float quad_r[QPT];
#pragma unroll
    for (int u = 0; u < QPT; u++) {
        const int i = tid + u * nth;
        quad_r[u] = i < Q * Q ? fx.quad[i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < KPT; u++) {
        const int i = tid + u * nth;
        if (i < W) {
            const unsigned key = key_r[u];
            const float raw = __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
            const float d = __fadd_rn(__fmul_rn(raw, a.depth_scale), a.depth_offset);
            const float xr = __fadd_rn(__fmul_rn(d, a.ppm_f), a.half_t_f);
            const double yl = __dadd_rn(__dmul_rn(__dmul_rn((double)d, tan_r[u]), a.ppm_d), a.half_t_d);
            vert[i + 1] = make_int2((int)(long long)yl, (int)(long long)xr);
        }
    }
#pragma unroll
    for (int u = 0; u < QPT; u++) {
        const int i = tid + u * nth;
        if (i < Q * Q) quad[i] = quad_r[u];
    }
    for (int i = tid + QPT * nth; i < Q * Q; i += nth) quad[i] = fx.quad[i];
    for (int i = tid + KPT * nth; i < W; i += nth) {
        const unsigned key = cm[i];
        const float raw = __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
        const float d = __fadd_rn(__fmul_rn(raw, a.depth_scale), a.depth_offset);        // f32 (value_map.py:234)
        const float xr = __fadd_rn(__fmul_rn(d, a.ppm_f), a.half_t_f);                   // f32 (:248)
        const double yl = __dadd_rn(__dmul_rn(__dmul_rn((double)d, a.tan_tab[i]), a.ppm_d), a.half_t_d);  // f64 (:242,249)
        vert[i + 1] = make_int2((int)(long long)yl, (int)(long long)xr);                 // astype(int): truncation
    }
    if (tid == 0) {
        vert[0] = make_int2(0, T - 1);
        vert[W + 1] = make_int2(T - 1, T - 1);
    }
    __syncthreads();
    VM_PHASE(1);
    // the keys have been read: the last workgroup of this observation to get here zeroes them for the next depth ingest
    if (tid == 0) {
        __threadfence();
        sh_last = atomicAdd(&fx.counters[obs], 1) == G - 1;
    }
    raster_polygon_flat(bm, vert, n_vert, pref, sh_wave_tot, tid, nth);
    VM_PHASE(2);
    if (sh_last) {
        unsigned* cmw = a.colmax + (size_t)obs * W;
        for (int i = tid; i < W; i += nth) cmw[i] = 0u;
        if (tid == 0) fx.counters[obs] = 0;
    }
    __syncthreads();
    VM_PHASE(3);
    // resolve (one lane per row) fused with visible = (template > 0) & ~beyond-the-profile, and the source bounding box
    {
        int r_lo = T, r_hi = -1, c_lo = T, c_hi = -1;
        for (int y = tid; y < T; y += nth) {
            unsigned carry = 0;
            for (int w = 0; w < words; w++) {
                const int i = y * words + w;
                unsigned p = parity[i];
                p ^= p << 1; p ^= p << 2; p ^= p << 4; p ^= p << 8; p ^= p << 16;
                if (carry) p = ~p;
                carry = p >> 31;
                const unsigned v = a.tmpl_bits[i] & ~(p | solid[i]);
                parity[i] = v;
                if (v) {
                    r_lo = min(r_lo, y); r_hi = max(r_hi, y);
                    c_lo = min(c_lo, w * 32 + __builtin_ctz(v)); c_hi = max(c_hi, w * 32 + 31 - __builtin_clz(v));
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            r_lo = min(r_lo, __shfl_xor(r_lo, off, 64)); r_hi = max(r_hi, __shfl_xor(r_hi, off, 64));
            c_lo = min(c_lo, __shfl_xor(c_lo, off, 64)); c_hi = max(c_hi, __shfl_xor(c_hi, off, 64));
        }
        if (lane == 0 && r_hi >= 0) {
            atomicMin(&sh_box[0], r_lo); atomicMax(&sh_box[1], r_hi); atomicMin(&sh_box[2], c_lo); atomicMax(&sh_box[3], c_hi);
        }
    }
    __syncthreads();
    VM_PHASE(4);
    if (tid == 0) {
        // destination (rotated) bounding box of everything that can receive a non-zero tap (see visible_mask_kernel)
        int4 m;
        if (sh_box[1] < 0) {
            m = make_int4(1, 0, 1, 0);
        } else {
            const double a0 = pose.inv_affine[0], a1 = pose.inv_affine[1], a2 = pose.inv_affine[2];
            const double a3 = pose.inv_affine[3], a4 = pose.inv_affine[4], a5 = pose.inv_affine[5];
            const double det = a0 * a4 - a1 * a3;
            double xlo = 1e30, xhi = -1e30, ylo = 1e30, yhi = -1e30;
            for (int k = 0; k < 4; k++) {
                const double sx = (k & 1) ? sh_box[3] + 1.0 : sh_box[2] - 1.0, sy = (k & 2) ? sh_box[1] + 1.0 : sh_box[0] - 1.0;
                const double dx = (a4 * (sx - a2) - a1 * (sy - a5)) / det, dy = (-a3 * (sx - a2) + a0 * (sy - a5)) / det;
                xlo = fmin(xlo, dx); xhi = fmax(xhi, dx); ylo = fmin(ylo, dy); yhi = fmax(yhi, dy);
            }
            if (!(fabs(det) > 1e-9) || !(xlo == xlo) || !(ylo == ylo)) { xlo = ylo = 0; xhi = yhi = T; }
            m = make_int4(max(0, (int)floor(ylo) - 2), min(T - 1, (int)ceil(yhi) + 2), max(0, (int)floor(xlo) - 2),
                          min(T - 1, (int)ceil(xhi) + 2));
        }
        sh_dbox = m;
    }
    VM_PHASE(5);
    // ---- step 2: cells that hold a value but are not explored any more (value_map.py:369-375)
    if (written && a.explored) {
        const unsigned* explored = a.explored + (size_t)pose.env * S * ex_stride;
        float* conf = a.conf + (size_t)pose.env * S * S;
        const int C = C_STATIC > 0 ? C_STATIC : a.C;
        double* value = a.value + (size_t)pose.env * S * S * C;
        for (int i = my_word; i < plane_words; i += G * nth) {
            const unsigned wr = i == my_word ? wr0 : written[i];
            if (!wr) continue;
            unsigned dead = wr & ~explored[i];
            if (!dead) continue;
            atomicAnd(&written[i], ~dead);
            const int row = i / ex_stride, c0 = (i - row * ex_stride) * 32;
            while (dead) {
                const int b = __builtin_ctz(dead);
                dead &= dead - 1u;
                const size_t cell = (size_t)row * S + c0 + b;
                conf[cell] = 0.0f;
                // `_value_map[explored_area == 0] *= 0` (:375): a product, so a negative value leaves -0.0
                for (int c = 0; c < C; c++) value[cell * C + c] = __dmul_rn(value[cell * C + c], 0.0);
            }
        }
    }
    __syncthreads();
    VM_PHASE(6);
    // ---- step 3: rotate + place + fuse, tiles g, g + G, ...
    int4 box = sh_dbox;
    box.x = __builtin_amdgcn_readfirstlane(box.x); box.y = __builtin_amdgcn_readfirstlane(box.y);
    box.z = __builtin_amdgcn_readfirstlane(box.z); box.w = __builtin_amdgcn_readfirstlane(box.w);
    if (box.z > box.w) return;
    // four wavefronts per 8-row tile (fuse_tile's shape), FUSED_THREADS / 256 tiles in flight per workgroup; only tiles
    // inside the cone's destination rows are dealt out
    const int tiles_per_pass = nth >> 8, tg = tid >> 8, t_in = tid & 255;
    const int t_lo = max(box.x, max(0, -pose.row0)) / ROWS_PER_TILE;
    const int t_hi = min(min(box.y, T - 1), S - 1 - pose.row0) / ROWS_PER_TILE;   // inclusive
    uint2* list = fx.list_cap > 0 ? cell_list : nullptr;
    if (fx.block_cap > 0 && list) {
        // ---- 3a: which 4 x 4 blocks of the destination box can hold a cell with a visible tap?
        const int by_lo = max(box.x, max(0, -pose.row0)) / VB, by_hi = min(min(box.y, T - 1), S - 1 - pose.row0) / VB;
        const int bx_lo = max(box.z, max(0, -pose.col0)) / VB, bx_hi = min(min(box.w, T - 1), S - 1 - pose.col0) / VB;
        const int nbx = bx_hi - bx_lo + 1, nby = by_hi - by_lo + 1;
        const int nblk = nbx > 0 && nby > 0 ? nbx * nby : 0;
        const int wave = tid >> 6;
        for (int b0 = g * nth; b0 < nblk; b0 += G * nth) {       // (wavefront-uniform trip count: the ballot below needs every lane)
            const int b = b0 + tid;
            bool act = false;
            int by = 0, bx = 0;
            if (b < nblk) {
                by = by_lo + b / nbx; bx = bx_lo + b % nbx;
                const int y0 = by * VB, y1 = min(y0 + VB - 1, T - 1), x0 = bx * VB, x1 = min(x0 + VB - 1, T - 1);
                const int2 ra = row_xy0[y0], rb = row_xy0[y1], ca = col_xy[x0], cb = col_xy[x1];
                // fixed-point coordinate (>> 10 = the cell's first tap) in the four corners
                const int xa = ra.x + ca.x, xb = ra.x + cb.x, xc = rb.x + ca.x, xd = rb.x + cb.x;
                const int ya = ra.y + ca.y, yb = ra.y + cb.y, yc = rb.y + ca.y, yd = rb.y + cb.y;
                const int sx0 = min(min(xa, xb), min(xc, xd)) >> 10, sx1 = (max(max(xa, xb), max(xc, xd)) >> 10) + 1;
                const int sy0 = min(min(ya, yb), min(yc, yd)) >> 10, sy1 = (max(max(ya, yb), max(yc, yd)) >> 10) + 1;
                act = any_visible(parity, words, T, sx0, sx1, sy0, sy1);
            }
            const unsigned long long m = __ballot(act);
            if (m == 0ull) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(&sh_block_n, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (act) {
                const int k = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (k < fx.block_cap) block_list[k] = ((unsigned)by << 16) | (unsigned)bx;
            }
        }
        __syncthreads();
        // ---- 3b: the cells of the listed blocks, four blocks per wavefront and round, one lane per cell
        const int nb = min(sh_block_n, fx.block_cap);   // (block_cap covers every block of the window: nothing is ever dropped)
        for (int i0 = wave * 4; i0 < nb; i0 += (nth >> 6) * 4) {
            const int bi = i0 + (lane >> 4);
            const bool have = bi < nb;
            const unsigned blk = block_list[have ? bi : 0];
            const int y = (int)(blk >> 16) * VB + ((lane >> 2) & 3), x = (int)(blk & 0xFFFFu) * VB + (lane & 3);
            const int mr = pose.row0 + y, mc = pose.col0 + x;
            const bool ok = have && y < T && x < T && (unsigned)mr < (unsigned)S && (unsigned)mc < (unsigned)S;
            const float nwv = cell_new_confidence(parity, quad, T, words, row_xy0[y < T ? y : 0], col_xy[x < T ? x : 0], ok);
            const unsigned long long m = __ballot(nwv != 0.0f);
            if (m == 0ull) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(&sh_list_n, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (nwv != 0.0f) {
                const int k = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                const uint2 e = make_uint2(((unsigned)mr << 16) | (unsigned)mc, __float_as_uint(nwv));
                if (k < fx.list_cap) list[k] = e;
                else fuse_list<C_STATIC>(a, pose, &e, 1, 0, 1, written);    // list full (a cone far wider than a camera's): in place
            }
        }
        __syncthreads();
        fuse_list<C_STATIC>(a, pose, list, min(sh_list_n, fx.list_cap), tid, nth, written);
        VM_PHASE(7);
        return;
    }
    for (int t = t_lo + g * tiles_per_pass + tg; t <= t_hi; t += G * tiles_per_pass)
        fuse_tile_lds<C_STATIC>(a, pose, parity, quad, row_xy0, box, t * ROWS_PER_TILE, t_in, written, list, &sh_list_n, fx.list_cap);
    if (list) {
        __syncthreads();
        fuse_list<C_STATIC>(a, pose, list, min(sh_list_n, fx.list_cap), tid, nth, written);
    }
    VM_PHASE(7);
}

// ------------------------------------------------------------------------------------------------ sort_waypoints
// One workgroup (256 threads) per (waypoint, channel).  Gathers the positive cells of the disc into LDS, then finds
// the median by rank counting (n <= (2r+1)^2, r = 10 -> 441; O(n^2) compares spread over the workgroup).  The map is f64
// (see the layout note at the top); `value_is_f32` says that the reference's array is still f32 in this mode, in which case
// np.median averages the two middle elements of an even count in f32.
__global__ __launch_bounds__(256) void sort_waypoints_kernel(const double* __restrict__ value, int S, int C,
                                                             const int* __restrict__ cells, int radius,
                                                             const int* __restrict__ disc_hw, int value_is_f32,
                                                             double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // dynamic LDS only (keeps the base 16-B aligned): [0] counter, [2..3] picked middles, [4..] gathered values (doubles)
    int& n_pos = *reinterpret_cast<int*>(smem);
    double* picked = reinterpret_cast<double*>(smem) + 2;
    double* vals = reinterpret_cast<double*>(smem) + 4;
    const int wp = blockIdx.x, ch = blockIdx.y;
    const int env = cells[3 * wp], row = cells[3 * wp + 1], col = cells[3 * wp + 2];
    const int side = 2 * radius + 1;
    const int r0 = row - radius < 0 ? 0 : row - radius, c0 = col - radius < 0 ? 0 : col - radius;
    const int r1 = row + radius + 1 > S ? S : row + radius + 1, c1 = col + radius + 1 > S ? S : col + radius + 1;
    const int ch_h = r1 - r0, ch_w = c1 - c0;  // crop extents; the disc stays centred at (radius, radius) OF THE CROP
    if (threadIdx.x == 0) n_pos = 0;
    __syncthreads();
    const double* vm = value + (size_t)env * S * S * C;
    for (int i = threadIdx.x; i < side * side; i += blockDim.x) {
        const int dy = i / side, dx = i - dy * side;
        if (dy >= ch_h || dx >= ch_w) continue;
        const int hw = disc_hw[dy];
        const int off = dx - radius;
        if (off < -hw || off > hw) continue;
        const double v = vm[((size_t)(r0 + dy) * S + (c0 + dx)) * C + ch];
        if (v > 0.0) vals[atomicAdd(&n_pos, 1)] = v;
    }
    __syncthreads();
    const int n = n_pos;
    if (n == 0) {
        if (threadIdx.x == 0) out[wp * C + ch] = -1.0;
        return;
    }
    // rank of element i = #(v_j < v_i) + #(v_j == v_i, j < i): a permutation of 0..n-1 (gather order is arbitrary,
    // which is fine: equal values are interchangeable for a median)
    const int k_hi = n / 2, k_lo = (n - 1) / 2;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double vi = vals[i];
        int rank = 0;
        for (int j = 0; j < n; j++) {
            const double vj = vals[j];
            rank += (vj < vi) || (vj == vi && j < i);
        }
        if (rank == k_hi) picked[0] = vi;
        if (rank == k_lo) picked[1] = vi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // np.median: middle element, or the mean of the two middle elements for even n -- np.mean in the ARRAY's dtype:
        // f64 add + exact halving for the f64 map, f32 add for a map the reference still holds in f32
        double m = picked[0];
        if (!(n & 1))
            m = value_is_f32 ? (double)__fmul_rn(__fadd_rn((float)picked[1], (float)picked[0]), 0.5f)
                             : __dmul_rn(__dadd_rn(picked[1], picked[0]), 0.5);
        out[wp * C + ch] = m;
    }
}

}  // namespace vlfm

// ================================================================================================ C ABI
using namespace vlfm;

extern "C" int vlfm_cone_template_build(const float* d_conf, const int64_t* d_poly_xy, int n_poly, int template_size,
                                        float* d_template, uint32_t* d_template_bits, void* stream) {
    if (!d_conf || !d_poly_xy || !d_template || !d_template_bits || n_poly < 3 || template_size <= 0)
        return fail(VLFM_ERR_INVALID, "cone_template_build: bad argument");
    const int T = template_size, words = (T + 31) >> 5;
    const size_t lds = (size_t)2 * T * words * sizeof(unsigned);
    if (lds > 160 * 1024) return fail(VLFM_ERR_CAPACITY, "cone_template_build: template too large for LDS");
    VLFM_KLAUNCH(cone_template_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, d_conf,
                       reinterpret_cast<const long long*>(d_poly_xy), n_poly, T, d_template, d_template_bits);
    return check_launch("cone_template_kernel");
}

#ifdef VLFM_PHASE_TIMING
extern "C" int vlfm_debug_vm_phase_clocks(long long* h_out /* [16] */) {
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(vlfm::g_vm_phase), sizeof(long long) * 16) == hipSuccess ? VLFM_OK : VLFM_ERR_HIP;
}
extern "C" int vlfm_debug_vm_span_clocks(long long* h_out /* [2048][2] */) {
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(vlfm::g_vm_span), sizeof(long long) * 4096) == hipSuccess ? VLFM_OK : VLFM_ERR_HIP;
}
#endif

static int target_override_g() {
    static const int v = [] { const char* e = getenv("VLFM_VM_TARGET_WGS"); return e ? atoi(e) : 0; }();
    return v;
}

extern "C" int vlfm_value_map_update_fused_batched(uint32_t* d_colmax_keys, int width, const double* d_tan,
                                                   const float* d_template, const uint32_t* d_template_bits,
                                                   int template_size, const vlfm_vm_pose* d_pose,
                                                   const double* d_values, int n, float* d_conf, double* d_value,
                                                   int map_size, int channels, int pixels_per_meter, double min_depth,
                                                   double max_depth, int use_max_confidence, int fusion_type,
                                                   const uint32_t* d_explored_bits, uint32_t* d_written_bits,
                                                   int32_t* d_counters, const float* d_conf_quadrant, void* stream) {
    if (n == 0) return VLFM_OK;
    if (!d_colmax_keys || !d_tan || !d_template || !d_template_bits || !d_pose || !d_values || !d_conf || !d_value ||
        !d_counters || !d_conf_quadrant || n < 0 || width <= 0 || template_size <= 0 || map_size <= 0 || channels <= 0 || fusion_type < 0 ||
        fusion_type > 2)
        return fail(VLFM_ERR_INVALID, "value_map_update_fused_batched: bad argument");
    if ((d_explored_bits == nullptr) != (d_written_bits == nullptr))
        return fail(VLFM_ERR_INVALID, "value_map_update_fused_batched: d_explored_bits and d_written_bits go together");
    UpdateArgs a;
    a.colmax = reinterpret_cast<unsigned*>(d_colmax_keys); a.tan_tab = d_tan; a.tmpl = d_template;
    a.tmpl_bits = d_template_bits; a.pose = d_pose; a.values = d_values;
    a.conf = d_conf; a.value = d_value; a.explored = d_explored_bits;
    a.W = width; a.T = template_size; a.S = map_size; a.C = channels;
    a.depth_scale = (float)(max_depth - min_depth);
    a.depth_offset = (float)min_depth;
    a.ppm_f = (float)pixels_per_meter;
    a.half_t_f = (float)(template_size / 2.0);
    a.ppm_d = (double)pixels_per_meter;
    a.half_t_d = template_size / 2.0;
    a.use_max_conf = use_max_confidence; a.fusion = fusion_type;
    FusedExtra fx{d_written_bits, d_counters, d_conf_quadrant, 0, 0};
    const int T = template_size, words = (T + 31) >> 5;
    const int n_vert = width + 2, Q = T / 2 + 1;
    const size_t lds_base = (size_t)(2 * T * words + ((2 * T * words) & 1)) * 4 + (size_t)n_vert * sizeof(int2) +
                            (size_t)(n_vert + 1 + ((n_vert + 1) & 1)) * sizeof(int) + (size_t)(Q * Q + ((Q * Q) & 1)) * 4 +
                            (size_t)T * sizeof(int2);
    size_t lds = lds_base;
    if (lds > 150 * 1024) return fail(VLFM_ERR_CAPACITY, "value_map_update_fused_batched: template/width too large for LDS");
    // the cell list takes what is left, up to 8192 entries (a 79-degree cone at 5 m marks ~3 000 cells; VLFM_VM_LIST=0: fuse in place)
    static const int list_override = [] { const char* e = getenv("VLFM_VM_LIST"); return e ? atoi(e) : -1; }();
    {
        size_t cap = (150 * 1024 - lds) / sizeof(uint2);
        if (cap > 8192) cap = 8192;
        if (list_override >= 0 && (size_t)list_override < cap) cap = (size_t)list_override;
        if (cap < 1024) cap = 0;
        if (map_size >= 65535 || T >= 65536) cap = 0;   // a list entry packs (row << 16) | col of a MAP cell, 0xFFFFFFFF is the sentinel
        fx.list_cap = (int)cap;
        lds += cap * sizeof(uint2);
        // block-sparse sweep: the column table + one list entry for EVERY 4 x 4 block of the window (VLFM_VM_BLOCKS=0: tile sweep)
        static const int blocks_off = [] { const char* e = getenv("VLFM_VM_BLOCKS"); return e && atoi(e) == 0; }();
        const int nb_side = (T + 3) / 4;
        const size_t extra = (size_t)T * sizeof(int2) + (size_t)nb_side * nb_side * 4;
        // ... when an observation has at most two workgroups: with more (small batches: G = CUs / n) the tile sweep split over
        // G workgroups is the shorter chain (measured, tools/vm_phase_probe.py, old / new: 256 obs 56 / 38 us, 128 obs 37 / 35,
        // 64 obs 27 / 36, 16 HD obs 31 / 40, 8 obs 23 / 29)
        const int cu = target_override_g() > 0 ? target_override_g() : device_cu_count();
        const bool few_wgs = (cu + n - 1) / n <= 2;
        if (!blocks_off && few_wgs && cap > 0 && lds + extra <= 150 * 1024 && T < 65536) {
            fx.block_cap = nb_side * nb_side;
            lds += extra;
        }
    }
    if (lds > 64 * 1024) {   // beyond the default dynamic-LDS limit: opt in once per device and kernel
        static LdsOptIn opt1, opt0;
        const bool ok = channels == 1 ? opt1.ensure(reinterpret_cast<const void*>(value_map_update_fused_kernel<1>), 150 * 1024)
                                      : opt0.ensure(reinterpret_cast<const void*>(value_map_update_fused_kernel<0>), 150 * 1024);
        if (!ok) {
            // no large LDS on this device: without the cell and block lists the base layout may still fit the default 64 KB
            // (fuse every tile in place, the tile sweep): slower, same results
            if (lds_base > 64 * 1024) return fail(VLFM_ERR_HIP, "value_map_update_fused_batched: cannot opt in to large LDS");
            fx.list_cap = 0;
            fx.block_cap = 0;
            lds = lds_base;
        }
    }
    const int tiles = (T + ROWS_PER_TILE - 1) / ROWS_PER_TILE;
    // workgroups per observation: a 1024-thread workgroup fills a CU (register budget), so aim for one per CU over all
    // observations; more than ceil(tiles / 4) would leave workgroups without a tile
    // (the device's CU count, asked once; VLFM_VM_TARGET_WGS -- read once per process -- is tools/vm_phase_probe.py's sweep knob)
    const int target = target_override_g() > 0 ? target_override_g() : device_cu_count();
    int G = (target + n - 1) / n;
    const int g_max = (tiles + FUSED_THREADS / 256 - 1) / (FUSED_THREADS / 256);
    if (G > g_max) G = g_max;
    if (G < 1) G = 1;
    VLFM_TIMED("value_map_update_fused_kernel", stream);
    if (channels == 1)
        VLFM_KLAUNCH(value_map_update_fused_kernel<1>, dim3(G, n), dim3(FUSED_THREADS), lds, (hipStream_t)stream, a, fx);
    else
        VLFM_KLAUNCH(value_map_update_fused_kernel<0>, dim3(G, n), dim3(FUSED_THREADS), lds, (hipStream_t)stream, a, fx);
    return check_launch("value_map_update_fused_kernel");
}

extern "C" int vlfm_value_map_sort_waypoints_batched(const double* d_value, int map_size, int channels,
                                                     const int32_t* d_cells, int m, int radius, const int32_t* d_disc,
                                                     int value_is_f32, double* d_out, void* stream) {
    if (m == 0) return VLFM_OK;
    if (!d_value || !d_cells || !d_disc || !d_out || m < 0 || radius < 0 || channels <= 0)
        return fail(VLFM_ERR_INVALID, "sort_waypoints_batched: bad argument");
    const int side = 2 * radius + 1;
    const size_t lds = ((size_t)side * side + 4) * sizeof(double);
    if (lds > 150 * 1024) return fail(VLFM_ERR_CAPACITY, "sort_waypoints_batched: radius too large");
    VLFM_TIMED("sort_waypoints_kernel", stream);
    VLFM_KLAUNCH(sort_waypoints_kernel, dim3(m, channels), dim3(256), lds, (hipStream_t)stream, d_value,
                       map_size, channels, d_cells, radius, d_disc, value_is_f32, d_out);
    return check_launch("sort_waypoints_kernel");
}
