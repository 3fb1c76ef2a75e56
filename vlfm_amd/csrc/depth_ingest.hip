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
// Column maxima are stored as order-preserving unsigned keys so that atomicMax works for any sign (work decomposition:
// see depth_ingest_kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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

// bit j of an 8-bit value -> bit 4j of the result
__device__ inline unsigned spread8(unsigned x) {
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x;
}

struct IngestArgs {
    const float* depth;             // [n][H][W]
    const vlfm_ingest_params* prm;  // [n]
    unsigned* colmax_keys;          // [n][W] (zero-initialised keys) or null
    unsigned* obstacle;             // [n_envs][S][stride] bit-packed, or null
    int* status;                    // [n][2]: word 0 = index error (sticky); word 1 = bit 0 image has zero texels,
                                    //         bit 1 the SPECULATIVE pass hit a cell off the map (both consumed by fill_small_holes)
    unsigned* hole_bits;            // [n][H][hw] OUT: bit plane of (depth == 0), or null
    const unsigned* filled_bits;    // [n][H][hw] IN: fill_small_holes' result (texel -> 1.0), or null
    int hw;                         // words per image row in hole/filled planes = ceil(W/32)
    unsigned* journal;              // [n][journal_cap] cell ids (row * S + col) of obstacle bits NEWLY set by this pass, or null
    int* journal_count;             // [n]
    int journal_cap;
    int H, W, W4, S, stride;
    int cols_per_block;             // float4 column groups handled by one workgroup in x
    int ry;                         // rows advanced per iteration (= blockDim.x / cols_per_block)
    int rows_per_block;
    double ppm;
};

struct HeightBand {  // f32 shadow of the height test, widened by a safety margin
    float t8, t9, t10, t11, inv_fx, inv_fy, lo, hi;
    float zmin, zmax;    // range of scaled depth: d in [0, 1] -> z in [offset, scale + offset]
    float gx_lo, gx_hi;  // range over the image columns of  -t9 * (u - W/2) / fx
    float depth_scale, depth_offset, depth_max;   // f32 copies of the parameter block's scalars (obstacle_map.py:92-93)
};
__device__ inline HeightBand make_band(const vlfm_ingest_params& p, int W, int H) {
    HeightBand b;
    b.t8 = (float)p.tf[8]; b.t9 = (float)p.tf[9]; b.t10 = (float)p.tf[10]; b.t11 = (float)p.tf[11];
    b.inv_fx = (float)(1.0 / p.fx); b.inv_fy = (float)(1.0 / p.fy);
    // |x_cam| <= depth_max * (W/2)/fx, |y_cam| <= depth_max * (H/2)/fy.  A dozen f32 roundings (2^-24 each) on terms of
    // that magnitude give an absolute error below 1e-6 * mag; the margin is 1e-4 * mag (100x).
    const float mag = fabsf(b.t8) * p.depth_max + fabsf(b.t9) * p.depth_max * (float)(W / 2 + 1) * fabsf(b.inv_fx) +
                      fabsf(b.t10) * p.depth_max * (float)(H / 2 + 1) * fabsf(b.inv_fy) + fabsf(b.t11) + 1.0f;
    const float margin = 1e-4f * mag;
    b.lo = (float)p.min_height - margin;
    b.hi = (float)p.max_height + margin;
    b.zmin = fminf(p.depth_offset, p.depth_offset + p.depth_scale);
    b.zmax = fmaxf(p.depth_offset, p.depth_offset + p.depth_scale);
    const float g0 = -b.t9 * (float)(0 - W / 2) * b.inv_fx, g1 = -b.t9 * (float)(W - 1 - W / 2) * b.inv_fx;
    b.gx_lo = fminf(g0, g1); b.gx_hi = fmaxf(g0, g1);
    b.depth_scale = p.depth_scale; b.depth_offset = p.depth_offset; b.depth_max = p.depth_max;
    return b;
}

// Can ANY texel of image row v land in the (widened) height band?  Z = z * g(u, v) + t11 with
// g = t8 - t9 (u - W/2)/fx - t10 (v - H/2)/fy; over the row g spans [g_lo, g_hi] and z spans [zmin, zmax] (depth is
// normalised to [0, 1] by contract), so Z spans the hull of the four corner products.  Rows that cannot hit the band (sky,
// ceiling: about half of a level camera's image) are never loaded by the scatter-only pass.  Conservative: an extra
// 1e-3 * |terms| slack on top of the band's own margin.
__device__ inline bool row_may_hit(const HeightBand& b, int v, int H) {
    const float gv = b.t8 - b.t10 * (float)(v - H / 2) * b.inv_fy;
    const float g_lo = gv + b.gx_lo, g_hi = gv + b.gx_hi;
    const float c0 = b.zmin * g_lo, c1 = b.zmin * g_hi, c2 = b.zmax * g_lo, c3 = b.zmax * g_hi;
    const float z_lo = fminf(fminf(c0, c1), fminf(c2, c3)) + b.t11, z_hi = fmaxf(fmaxf(c0, c1), fmaxf(c2, c3)) + b.t11;
    const float slack = 1e-3f * (fabsf(z_lo) + fabsf(z_hi) + fabsf(b.t11) + 1.0f);
    return !(z_hi + slack < b.lo || z_lo - slack > b.hi);  // NaN -> true
}

// ---- placement of one texel, in two halves -------------------------------------------------------------------------
// candidate(): the cheap, branch-free f32 half every texel of a row that can reach the height band goes through.
//   z = d * (max - min) + min in f32 (obstacle_map.py:92), masked by z < max_depth (:93), then a CONSERVATIVE f32 shadow of
//   the height test: the exact f64 evaluation in place_exact() is what decides, but ~90 % of the texels (floor, ceiling) are
//   far outside [min_height, max_height].  The f32 estimate of Z is within 1e-6 * (|row 3 of tf| . |point|) of the f64 value;
//   the band is widened by 1e-4 of that magnitude, and NaNs pass (to the exact path).
// place_exact(): unproject -> transform -> height band -> rint cell -> set the bit, every operation an explicitly rounded
//   IEEE f64 operation in the reference's order.  Run DENSELY (one candidate per lane, see depth_ingest_kernel).
__device__ inline bool candidate(const HeightBand& band, int W, int H, int u, int v, float d) {
    const float z = __fadd_rn(__fmul_rn(d, band.depth_scale), band.depth_offset);  // obstacle_map.py:92 (f32)
    const float xf = (float)(u - W / 2) * z * band.inv_fx, yf = (float)(v - H / 2) * z * band.inv_fy;
    const float zf = band.t8 * z - band.t9 * xf - band.t10 * yf + band.t11;
    return (z < band.depth_max) && !(zf < band.lo || zf > band.hi);                // :93, then the widened band
}
// The same test with the per-column and per-row factors of  Z = z (t8 - t9 (u - W/2)/fx - t10 (v - H/2)/fy) + t11  taken
// out of the texel loop (gx = -t9 (u - W/2)/fx is a lane constant, gy = t8 - t10 (v - H/2)/fy a row constant): two exactly
// rounded operations for z (the `z < max_depth` mask must be the reference's), one add, one FMA, three compares.  The
// re-association moves the f32 estimate by a few ulp -- the band's margin is 1e-4 of the operands' magnitude.
__device__ inline bool candidate_fast(const HeightBand& band, float gx, float gy, float d) {
    const float z = __fadd_rn(__fmul_rn(d, band.depth_scale), band.depth_offset);  // obstacle_map.py:92 (f32)
    const float zf = __builtin_fmaf(z, gx + gy, band.t11);
    return (z < band.depth_max) && !(zf < band.lo || zf > band.hi);
}

// a / b in f64, correctly rounded, with the reciprocal y = RN(1 / b) hoisted out (b is the focal length: the same for every
// texel of the launch).  q0 = RN(a y) is a faithful quotient, r = a - b q0 is exact in one FMA, and RN(q0 + r y) is then the
// correctly rounded a / b (Markstein's division step: the final FMA of the hardware's own v_div_scale / v_rcp / v_div_fmas /
// v_div_fixup expansion, without the per-call refinement of the reciprocal and without the scaling that only matters near
// overflow / underflow -- |a| <= image width x max depth, b a focal length in pixels).  Same bits as __ddiv_rn(a, b) in 3
// operations instead of ~12 (tests/test_obstacle_prims_gpu.py checks 2^24 random quotients against the division).
struct ExactRecip { double b, y; };
__device__ inline ExactRecip exact_recip(double b) { return ExactRecip{b, __ddiv_rn(1.0, b)}; }
__device__ inline double div_exact(double a, const ExactRecip& r) {
    const double q0 = __dmul_rn(a, r.y);
    const double rem = __fma_rn(-r.b, q0, a);
    return rem == 0.0 ? q0 : __fma_rn(rem, r.y, q0);   // (exact quotient: keep q0 -- the FMA would turn a -0.0 into +0.0)
}

template <bool HOLE_PASS>
__device__ inline void place_exact(const IngestArgs& a, const vlfm_ingest_params& p, const ExactRecip& rfx,
                                   const ExactRecip& rfy, unsigned* grid, int obs, int u, int v, float d) {
    const float z = __fadd_rn(__fmul_rn(d, p.depth_scale), p.depth_offset);  // obstacle_map.py:92 (f32)
    // get_point_cloud (geometry_utils.py:230-234): int64 * f32 -> f64, then / fx
    const double zd = (double)z;
    const double xc = div_exact(__dmul_rn((double)(u - a.W / 2), zd), rfx);
    const double yc = div_exact(__dmul_rn((double)(v - a.H / 2), zd), rfy);
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
    // NumPy fancy-index semantics (obstacle_map.py:101): [-S, -1] wraps, anything else outside raises IndexError.  The test
    // runs on the (integer-valued) doubles; only values inside [-S, S) are converted (one instruction instead of the
    // multi-instruction f64 -> i64 sequence).
    const double Sd = (double)a.S;
    if (!(rowf < Sd && rowf >= -Sd && colf < Sd && colf >= -Sd)) {
        // The speculative pass (fill_small_holes not evaluated yet) cannot know whether the reference would have seen this
        // texel at all: an "island" texel inside a small hole becomes 1.0 and is dropped (img_utils.py:385-388) without
        // ever reaching the scatter.  It only NOTES the hit (status word 1, bit 1); fill_small_holes_kernel promotes the
        // note to the IndexError for frames without islands, and for island frames hole_scatter_kernel's second placement
        // of the surviving texels raises it itself.
        if (!HOLE_PASS && a.journal) atomicOr(&a.status[2 * obs + 1], 2);
        else a.status[2 * obs] = VLFM_ERR_INDEX;
        return;
    }
    int row = (int)rowf, col = (int)colf;
    if (row < 0) row += a.S;
    if (col < 0) col += a.S;
    // A plain (possibly stale) read that already shows the bit lets us skip the device-scope atomic: in steady state almost
    // every in-band point re-observes a known obstacle cell.  (Bits are only cleared by reset() and by the island undo of
    // fill_small_holes_kernel, which runs strictly after this kernel.)
    unsigned* word = &grid[row * a.stride + (col >> 5)];   // S <= 2048 (checked by the host): the index fits 32 bits
    const unsigned bit = 1u << (col & 31);
    if (!(*word & bit)) {
        const unsigned old = atomicOr(word, bit);
        // Speculative single pass: remember every bit this pass is the first to set.  If the image turns out to hold valid
        // texels INSIDE a filled hole contour ("islands"), fill_small_holes_kernel takes exactly these bits back and
        // hole_scatter_kernel re-places the texels that survive.
        if (!HOLE_PASS && a.journal && !(old & bit)) {
            const int k = atomicAdd(&a.journal_count[obs], 1);
            if (k < a.journal_cap) a.journal[(size_t)obs * a.journal_cap + k] = (unsigned)(row * a.S + col);
        }
    }
}

// Work decomposition: a workgroup owns CG float4 column groups (CG*4 image columns, CG*16 contiguous bytes per row) and
// a band of rows; lane (cx, ry) walks rows ry, ry+RL, ... of the band with UNROLL 16-byte loads in flight.  With one band per image (the large-batch
// case) every column maximum is produced by exactly one workgroup -- no atomic contention; small batches split the rows
// into bands to fill the chip and merge through atomicMax on the keys.
//
// Obstacle placement by WAVE-LOCAL COMPACTION (round 3).  Only ~8 % of the texels survive candidate(), and they sit in a
// few image rows; calling the f64 placement from the streaming loop (rounds 1-2) kept both paths live in most wavefronts
// and chained one dependent plane read per texel and lane: 142 us at 256 images, 0.28 of the HBM peak.  Now the streaming
// loop only evaluates candidate() (a dozen f32 operations, no branch); the survivors of a row group are appended to the
// WAVEFRONT's own ring in LDS (four ballots + mbcnt give every lane its slots; 8 bytes per entry: packed (u, v) and the raw
// depth), and whenever the ring holds 64 entries the wavefront places 64 of them, one per lane: the f64 arithmetic runs on
// full wavefronts and the 64 plane reads of a pass are in flight together.  Everything is wave-synchronous -- no workgroup
// barrier, no atomics on the queue, the other wavefronts keep streaming -- and nothing goes through HBM (a global
// candidate list + a second launch would add ~100 MB of list traffic per 256 images).  The remainder (< 64 entries) is
// placed once, when the wavefront has finished its rows.  First form tried: one queue per workgroup filled through an LDS
// atomic and drained by all 512 lanes between two barriers per round -- 185-213 us, SLOWER than the divergent form: the
// barriers stop the streaming of all eight wavefronts for every round, and with 128 registers only two workgroups fit a CU.
constexpr int CG = 32;   // float4 column groups per workgroup
constexpr int RL = 16;   // row lanes per workgroup  -> 512 threads, a wavefront covers 2 rows x 512 B
constexpr int INGEST_UNROLL = 4;
constexpr int WQ = 512;      // ring entries per wavefront (power of two >= 63 + 256: a row group adds at most 256)

template <bool SCATTER>
__device__ __forceinline__ void depth_ingest_body(const IngestArgs& a) {
    __shared__ float4 part[RL][CG];
    __shared__ uint2 ring[SCATTER ? CG * RL / 64 : 1][SCATTER ? WQ : 1];
    __shared__ unsigned long long row_hits[SCATTER ? 1024 : 1];   // bit v: image row v can reach the height band (H <= 65535)
    constexpr int UNROLL = INGEST_UNROLL;
    const int obs = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int cx = tid % CG, ry = tid / CG;
    const int col4 = blockIdx.x * CG + cx;
    // Row bands are INTERLEAVED: band b owns the 16-row groups b, b + bands, b + 2 bands, ...  The texels that reach the
    // expensive f64 placement sit in a few adjacent image rows; contiguous bands would hand all of them to a fraction of
    // the workgroups.
    const int bands = gridDim.y, band_id = blockIdx.y;
    const int n_groups = (a.H + RL - 1) / RL;
    const bool live = col4 < a.W4;
    const vlfm_ingest_params p = a.prm[obs];   // uniform address, before any store of this kernel: scalar loads
    const HeightBand band = make_band(p, a.W, a.H);
    const float* img = a.depth + (size_t)obs * a.H * a.W;
    unsigned* grid = nullptr;
    if (SCATTER) grid = a.obstacle + (size_t)p.env * a.S * a.stride;
    unsigned* holes = a.hole_bits ? a.hole_bits + (size_t)obs * a.H * a.hw : nullptr;
    const unsigned* filled = a.filled_bits ? a.filled_bits + (size_t)obs * a.H * a.hw : nullptr;
    const float ninf = -__builtin_huge_valf();
    float4 m = make_float4(ninf, ninf, ninf, ninf);
    bool saw_zero = false;
    const bool scatter_only = a.colmax_keys == nullptr && holes == nullptr;  // the pass after fill_small_holes
    const bool do_scatter = SCATTER && (p.scatter & 1);
    uint2* q = ring[SCATTER ? (tid >> 6) : 0];
    unsigned q_head = 0u, q_cnt = 0u;   // wave-uniform: first pending entry, number of pending entries
    ExactRecip rfx{1.0, 1.0}, rfy{1.0, 1.0};
    float gx[4] = {0.f, 0.f, 0.f, 0.f};
    if (SCATTER && do_scatter) {
        // row_may_hit once per row and workgroup (it was ~25 instructions per lane and row group inside the streaming loop)
        for (int v0 = 0; v0 < a.H; v0 += CG * RL) {
            const int v = v0 + tid;
            const unsigned long long hitmask = __ballot(v < a.H && row_may_hit(band, v, a.H));
            if (lane == 0) row_hits[(v0 + tid) >> 6] = hitmask;
        }
        rfx = exact_recip(p.fx); rfy = exact_recip(p.fy);
#pragma unroll
        for (int c = 0; c < 4; c++) gx[c] = -band.t9 * (float)(col4 * 4 + c - a.W / 2) * band.inv_fx;
        __syncthreads();
    }

    // one iteration's loads: rows (g0 + k * bands) * RL + ry, k < UNROLL.  Rows that cannot reach the height band are not
    // even loaded by a scatter-only pass; a combined pass loads them (column maximum, hole bits) but never tests them.
    float4 nxt[UNROLL];
    bool nxt_ok[UNROLL], nxt_hit[UNROLL];   // lane predicates: they live in scalar register pairs, not in packed vector masks
    auto issue = [&](int g0) {
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const int r = (g0 + k * bands) * RL + ry;
            bool want = live && r < a.H;
            const bool hit = SCATTER && do_scatter && want && ((row_hits[r >> 6] >> (r & 63)) & 1ull);
            if (SCATTER && scatter_only) want = hit;
            nxt[k] = want ? reinterpret_cast<const float4*>(img + (size_t)r * a.W)[col4] : make_float4(ninf, ninf, ninf, ninf);
            nxt_ok[k] = want;
            nxt_hit[k] = hit;
        }
    };
    // every lane of the workgroup runs the same trip count (predicated), so the cross-lane packing of hole bits and of the
    // candidate ring is always executed convergently
    for (int g0 = band_id; g0 < n_groups; g0 += UNROLL * bands) {
        issue(g0);
        float4 d[UNROLL];
        bool okb[UNROLL], hitb[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) { d[k] = nxt[k]; okb[k] = nxt_ok[k]; hitb[k] = nxt_hit[k]; }
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const int r = (g0 + k * bands) * RL + ry;
            const bool ok = okb[k];
            m.x = fmaxf(m.x, d[k].x); m.y = fmaxf(m.y, d[k].y); m.z = fmaxf(m.z, d[k].z); m.w = fmaxf(m.w, d[k].w);
            unsigned zero4 = 0u;   // this lane's (depth == 0) texels
            if (holes) {
                // (depth == 0) bit plane: 4 texels per lane, 8 neighbouring lanes (same row) per 32-bit word.  A frame behind
                // depth_camera_filtering has no zero at all: three min + one compare + one ballot say so for the whole
                // wavefront, and the plane word is simply 0.  Otherwise four wave ballots (scalar unit) + a bit spread in the
                // group's leader lane; no cross-lane data movement.
                const float mn = fminf(fminf(d[k].x, d[k].y), fminf(d[k].z, d[k].w));
                unsigned word = 0u;
                if (__ballot(ok && !(mn > 0.0f)) != 0ull) {
                    zero4 = (d[k].x == 0.0f ? 1u : 0u) | (d[k].y == 0.0f ? 2u : 0u) | (d[k].z == 0.0f ? 4u : 0u) |
                            (d[k].w == 0.0f ? 8u : 0u);
                    if (!ok) zero4 = 0u;
                    const unsigned long long b0 = __ballot(zero4 & 1u), b1 = __ballot(zero4 & 2u);
                    const unsigned long long b2 = __ballot(zero4 & 4u), b3 = __ballot(zero4 & 8u);
                    saw_zero |= (b0 | b1 | b2 | b3) != 0ull;
                    if (b0 | b1 | b2 | b3) {
                        const int sh = lane & ~7;
                        word = spread8((unsigned)(b0 >> sh) & 0xFFu) | (spread8((unsigned)(b1 >> sh) & 0xFFu) << 1) |
                               (spread8((unsigned)(b2 >> sh) & 0xFFu) << 2) | (spread8((unsigned)(b3 >> sh) & 0xFFu) << 3);
                    }
                }
                if (ok && (cx & 7) == 0) holes[(size_t)r * a.hw + (col4 >> 3)] = word;
            }
            if (!SCATTER || !do_scatter) continue;
            const bool hit = hitb[k];
            if (__ballot(hit) == 0ull) continue;   // neither of this wavefront's two rows can reach the height band
            unsigned c4 = 0u;
            if (hit) {
                // fill_small_holes (img_utils.py:361-390) turned a `filled` texel into 1.0 -> z == max_depth -> masked (:93).
                // A zero texel is a hole in the depth image.  scatter bit 1: hole_area_thresh == -1 semantics
                // (obstacle_map.py:87-89), every zero becomes 1.0 and therefore falls outside max_depth.  scatter bit 2:
                // whether this zero survives fill_small_holes is not known yet -- hole_scatter_kernel places the survivors
                // from the hole bit plane (an unfilled, large hole is used as it is: depth 0 -> z = min_depth).
                unsigned skip = 0u;
                if (filled) skip = (filled[(size_t)r * a.hw + (col4 >> 3)] >> ((col4 & 7) * 4)) & 0xFu;
                if (p.scatter & 6)
                    skip |= holes ? zero4 : ((d[k].x == 0.0f ? 1u : 0u) | (d[k].y == 0.0f ? 2u : 0u) |
                                             (d[k].z == 0.0f ? 4u : 0u) | (d[k].w == 0.0f ? 8u : 0u));
                const float gy = band.t8 - band.t10 * (float)(r - a.H / 2) * band.inv_fy;
                c4 = ((candidate_fast(band, gx[0], gy, d[k].x) ? 1u : 0u) | (candidate_fast(band, gx[1], gy, d[k].y) ? 2u : 0u) |
                      (candidate_fast(band, gx[2], gy, d[k].z) ? 4u : 0u) | (candidate_fast(band, gx[3], gy, d[k].w) ? 8u : 0u)) & ~skip;
            }
            // ---- append this row group's survivors to the wavefront's ring: texel c of lane L goes behind all texels
            // c' < c of every lane and the texels c of the lanes below L
            const unsigned long long Bany = __ballot(c4 != 0u);
            if (Bany == 0ull) continue;
            const unsigned tail = q_head + q_cnt;
            const unsigned pos = ((unsigned)(col4 * 4)) | ((unsigned)r << 16);
            if (__ballot(c4 == 0xFu) == Bany) {
                // the common shape of an in-band row: every lane that has a survivor has all four -- one ballot places them
                const unsigned first = tail + 4u * __builtin_amdgcn_mbcnt_hi((unsigned)(Bany >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Bany, 0u));
                if (c4) {
                    q[(first + 0u) & (WQ - 1)] = make_uint2(pos + 0u, __float_as_uint(d[k].x));
                    q[(first + 1u) & (WQ - 1)] = make_uint2(pos + 1u, __float_as_uint(d[k].y));
                    q[(first + 2u) & (WQ - 1)] = make_uint2(pos + 2u, __float_as_uint(d[k].z));
                    q[(first + 3u) & (WQ - 1)] = make_uint2(pos + 3u, __float_as_uint(d[k].w));
                }
                q_cnt += 4u * (unsigned)__popcll(Bany);
            } else {
                const unsigned long long B0 = __ballot(c4 & 1u), B1 = __ballot(c4 & 2u), B2 = __ballot(c4 & 4u), B3 = __ballot(c4 & 8u);
                const unsigned n0 = __popcll(B0), n1 = __popcll(B1), n2 = __popcll(B2), n3 = __popcll(B3);
                const unsigned below0 = __builtin_amdgcn_mbcnt_hi((unsigned)(B0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)B0, 0u));
                const unsigned below1 = __builtin_amdgcn_mbcnt_hi((unsigned)(B1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)B1, 0u));
                const unsigned below2 = __builtin_amdgcn_mbcnt_hi((unsigned)(B2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)B2, 0u));
                const unsigned below3 = __builtin_amdgcn_mbcnt_hi((unsigned)(B3 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)B3, 0u));
                if (c4 & 1u) q[(tail + below0) & (WQ - 1)] = make_uint2(pos + 0u, __float_as_uint(d[k].x));
                if (c4 & 2u) q[(tail + n0 + below1) & (WQ - 1)] = make_uint2(pos + 1u, __float_as_uint(d[k].y));
                if (c4 & 4u) q[(tail + n0 + n1 + below2) & (WQ - 1)] = make_uint2(pos + 2u, __float_as_uint(d[k].z));
                if (c4 & 8u) q[(tail + n0 + n1 + n2 + below3) & (WQ - 1)] = make_uint2(pos + 3u, __float_as_uint(d[k].w));
                q_cnt += n0 + n1 + n2 + n3;
            }
            // ---- full passes: 64 pending entries -> one per lane
            while (q_cnt >= 64u) {
                const uint2 e = q[(q_head + (unsigned)lane) & (WQ - 1)];
                place_exact<false>(a, p, rfx, rfy, grid, obs, (int)(e.x & 0xFFFFu), (int)(e.x >> 16), __uint_as_float(e.y));
                q_head += 64u;
                q_cnt -= 64u;
            }
        }
    }
    if (SCATTER && do_scatter && q_cnt) {   // the wavefront's remainder (< 64 entries)
        if ((unsigned)lane < q_cnt) {
            const uint2 e = q[(q_head + (unsigned)lane) & (WQ - 1)];
            place_exact<false>(a, p, rfx, rfy, grid, obs, (int)(e.x & 0xFFFFu), (int)(e.x >> 16), __uint_as_float(e.y));
        }
    }
    if (saw_zero) atomicOr(&a.status[2 * obs + 1], 1);
    if (a.colmax_keys == nullptr) return;
    part[ry][cx] = m;
    __syncthreads();
    if (ry == 0 && live) {
#pragma unroll
        for (int k = 1; k < RL; k++) {
            const float4 o = part[k][cx];
            m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w);
        }
        unsigned* out = a.colmax_keys + (size_t)obs * a.W + col4 * 4;
        atomicMax(out + 0, f32_key(m.x));
        atomicMax(out + 1, f32_key(m.y));
        atomicMax(out + 2, f32_key(m.z));
        atomicMax(out + 3, f32_key(m.w));
    }
}

// Also measured and removed (round 3, tools/ingest_probe.py, 256 x 640x480, all bit-identical planes):
//  * an f32 tier in front of the exact placement -- X, Y, Z in f32 with a rigorous bound B = 16 * 2^-24 * M on their distance
//    from the reference's f64 values; a texel whose Z is farther than B from both band edges and whose cell coordinates are
//    farther than B ppm from a rounding boundary is committed from the f32 values, the other ~0.1 % go to an exact pass at the
//    end of the workgroup.  15 f32 operations instead of 47 f64 ones per candidate, and SLOWER: 114-121 us against 95 us.  By
//    then the kernel is no longer bound by its arithmetic (SQ counters: VALU busy 54 % of the time, waves parked on
//    s_waitcnt 67 % of their cycles); the tier's 20 uniform floats cost registers (99 instead of 79: two workgroups per CU
//    instead of three) and that occupancy is what the streaming half lives on;
//  * a global candidate list + a second placement launch (the two-pass form): 341 + 129 us -- one returning atomic per 64
//    entries on ONE counter per image serialises at ~260 ns each, and 4096 workgroups reading the same few hundred plane words
//    at once are slower than the same reads spread over the streaming pass;
//  * one queue per workgroup drained by all 512 lanes between two barriers: 185-213 us (every round stops all eight wavefronts).
// Round 5 (tools/ingest_probe.py, same checksums): (a) a wavefront-level reject in front of the four candidate tests -- the hull
// of fma(z, gx + gy, t11) over a lane's four texels (exact: the estimate is monotone in depth and column) against the band, one
// ballot: 96.7 us against 95.7, no gain -- the rows row_may_hit() lets through are not where the time goes; (b) the f32 placement
// tier INSIDE the drain (one candidate per lane, rigorous 16 x 2^-24 x M bound, exact chain for the ~3 per mille it cannot decide):
// the tier's 19 uniform floats beside the exact chain's 24 overflow the scalar file -- 98 vector registers with the constants in
// scalar registers, 154-168 with the constants in LDS (volatile or not), 121 + a 336-byte stack frame as a noinline function:
// every form gives up the third workgroup per CU before it saves an instruction.  The kernel stays as it is.
// Register budget / prefetch variants measured at 256 x 640x480 (tools/ingest_probe.py, round 3): the compiler's own
// allocation without software prefetch (79 VGPRs, 3 workgroups per CU) 95 us; prefetch of the next iteration's rows held to
// 4 waves per SIMD 96 us, to 6 waves per SIMD (spills) 130 us; no prefetch held to 4 waves 107 us.  One form is kept.
template <bool SCATTER>
__global__ __launch_bounds__(CG * RL) void depth_ingest_kernel(IngestArgs a) { depth_ingest_body<SCATTER>(a); }

// The zero texels that fill_small_holes left alone (holes of area >= hole_area_thresh), placed from the bit planes: the
// depth images are not read again.  One thread per 32-texel word of (hole & ~filled); frames without a zero texel
// (counts[3] == 0, the common case) cost one early exit per workgroup.
__global__ __launch_bounds__(256) void hole_scatter_kernel(IngestArgs a, const int* counts) {
    const int obs = blockIdx.y;
    if (counts[(size_t)obs * 4 + 3] == 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.H * a.hw) return;
    const vlfm_ingest_params p = a.prm[obs];
    if (!(p.scatter & 1) || (p.scatter & 2)) return;
    const unsigned hole = a.hole_bits[(size_t)obs * a.H * a.hw + i], filled = a.filled_bits[(size_t)obs * a.H * a.hw + i];
    unsigned w = hole & ~filled;
    const int v = i / a.hw, u0 = (i % a.hw) * 32;
    const HeightBand band = make_band(p, a.W, a.H);
    // Island frame (counts[3] bit 1): the speculative pass placed valid texels that lie inside a filled hole contour;
    // its newly set bits were taken back by fill_small_holes_kernel, so every valid texel OUTSIDE the filled area is
    // placed again here, from the image (rows that cannot reach the height band are skipped).
    unsigned redo = 0u;
    if ((counts[(size_t)obs * 4 + 3] & 2) && a.depth && row_may_hit(band, v, a.H)) {
        redo = ~hole & ~filled;
        if (u0 + 32 > a.W) redo &= (a.W - u0 >= 32) ? 0xFFFFFFFFu : ((1u << (a.W - u0)) - 1u);
    }
    if (!(w | redo)) return;
    unsigned* grid = a.obstacle + (size_t)p.env * a.S * a.stride;
    const ExactRecip rfx = exact_recip(p.fx), rfy = exact_recip(p.fy);
    while (w) {
        const int b = __builtin_ctz(w);
        w &= w - 1u;
        if (u0 + b < a.W && candidate(band, a.W, a.H, u0 + b, v, 0.0f)) place_exact<true>(a, p, rfx, rfy, grid, obs, u0 + b, v, 0.0f);
    }
    const float* row = a.depth ? a.depth + ((size_t)obs * a.H + v) * a.W + u0 : nullptr;
    while (redo) {
        const int b = __builtin_ctz(redo);
        redo &= redo - 1u;
        if (candidate(band, a.W, a.H, u0 + b, v, row[b])) place_exact<true>(a, p, rfx, rfy, grid, obs, u0 + b, v, row[b]);
    }
}

// self-test of div_exact against the IEEE division (tests/test_obstacle_prims_gpu.py)
__global__ __launch_bounds__(256) void div_exact_check_kernel(const double* __restrict__ a, int n, double b, int* mismatches) {
    const ExactRecip r = exact_recip(b);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double want = __ddiv_rn(a[i], b), got = div_exact(a[i], r);
        if (__double_as_longlong(want) != __double_as_longlong(got)) atomicAdd(mismatches, 1);
    }
}

}  // namespace vlfm

using namespace vlfm;

extern "C" int vlfm_selftest_div_exact(const double* d_numerators, int n, double divisor, int32_t* d_mismatches, void* stream) {
    if (!d_numerators || !d_mismatches || n < 0) return fail(VLFM_ERR_INVALID, "selftest_div_exact: bad argument");
    if (n == 0) return VLFM_OK;
    hipLaunchKernelGGL(div_exact_check_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, d_numerators, n, divisor,
                       d_mismatches);
    return check_launch("div_exact_check_kernel");
}

extern "C" int vlfm_depth_scatter_holes_batched(const vlfm_ingest_params* d_params, int n, int height, int width,
                                                const uint32_t* d_hole_bits, const uint32_t* d_filled_bits,
                                                const int32_t* d_hole_counts, uint32_t* d_obstacle, int map_size,
                                                int pixels_per_meter, int32_t* d_status, const float* d_depth,
                                                void* stream) {
    if (n == 0) return VLFM_OK;
    if (!d_params || !d_hole_bits || !d_filled_bits || !d_hole_counts || !d_obstacle || !d_status || n < 0 ||
        height <= 0 || width <= 0)
        return fail(VLFM_ERR_INVALID, "depth_scatter_holes_batched: bad argument");
    IngestArgs a{};
    a.prm = d_params; a.obstacle = d_obstacle; a.status = d_status; a.depth = d_depth;
    a.hole_bits = const_cast<uint32_t*>(d_hole_bits); a.filled_bits = d_filled_bits; a.hw = (width + 31) / 32;
    a.H = height; a.W = width; a.W4 = width / 4; a.S = map_size; a.stride = (map_size + 31) / 32;
    a.ppm = (double)pixels_per_meter;
    VLFM_TIMED("hole_scatter_kernel", stream);
    VLFM_KLAUNCH(hole_scatter_kernel, dim3((a.H * a.hw + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, a,
                 d_hole_counts);
    return check_launch("hole_scatter_kernel");
}

extern "C" int vlfm_depth_ingest_batched(const float* d_depth, int n, int height, int width,
                                         const vlfm_ingest_params* d_params, uint32_t* d_colmax_keys, uint32_t* d_obstacle,
                                         int map_size, int pixels_per_meter, int32_t* d_status, uint32_t* d_hole_bits,
                                         const uint32_t* d_filled_bits, const vlfm_scatter_journal* journal,
                                         void* stream) {
    if (n == 0) return VLFM_OK;
    if (!d_depth || !d_params || !d_status || n < 0 || height <= 0 || width <= 0)
        return fail(VLFM_ERR_INVALID, "depth_ingest_batched: bad argument");
    if (width % 4 != 0) return fail(VLFM_ERR_INVALID, "depth_ingest_batched: width must be a multiple of 4");
    if (width > 65535 || height > 65535) return fail(VLFM_ERR_INVALID, "depth_ingest_batched: image sides must fit 16 bits");
    if (!d_colmax_keys && !d_obstacle && !d_hole_bits) return VLFM_OK;
    hipStream_t s = (hipStream_t)stream;
    IngestArgs a;
    a.depth = d_depth; a.prm = d_params; a.colmax_keys = reinterpret_cast<unsigned*>(d_colmax_keys);
    a.obstacle = d_obstacle; a.status = d_status;
    a.hole_bits = d_hole_bits; a.filled_bits = d_filled_bits; a.hw = (width + 31) / 32;
    a.journal = nullptr; a.journal_count = nullptr; a.journal_cap = 0;
    if (journal && journal->d_cells && journal->d_count && journal->capacity > 0) {
        a.journal = journal->d_cells; a.journal_count = journal->d_count; a.journal_cap = journal->capacity;
    }
    a.H = height; a.W = width; a.W4 = width / 4; a.S = map_size; a.stride = (map_size + 31) / 32;
    a.ppm = (double)pixels_per_meter;
    a.cols_per_block = CG; a.ry = RL;
    const int gx = (a.W4 + CG - 1) / CG;
    // row bands: aim for ~2048 workgroups (8 per CU) so that enough 16-byte loads are in flight to cover HBM latency,
    // but never fewer than RL rows per band
    int bands = (int)((2048 + (long)n * gx - 1) / ((long)n * gx));
    static const int bands_override = [] { const char* e = getenv("VLFM_INGEST_BANDS"); return e ? atoi(e) : 0; }();
    if (bands_override > 0) bands = bands_override;
    const int max_bands = (height + RL - 1) / RL;
    if (bands > max_bands) bands = max_bands;
    if (bands < 1) bands = 1;
    a.rows_per_block = (height + bands - 1) / bands;  // informational: rows per workgroup (interleaved in 16-row groups)
    const int gy = bands;
    if (d_obstacle) {
        // profile name: the streaming pass (column maxima [+ hole bits]) is "depth_ingest_kernel"; a pass that ALSO or
        // ONLY scatters obstacle points is reported separately
        VLFM_TIMED(d_colmax_keys ? "depth_ingest_scatter_kernel" : "depth_scatter_kernel", s);
        VLFM_KLAUNCH(depth_ingest_kernel<true>, dim3(gx, gy, n), dim3(CG * RL), 0, s, a);
    } else {
        VLFM_TIMED("depth_ingest_kernel", s);
        VLFM_KLAUNCH(depth_ingest_kernel<false>, dim3(gx, gy, n), dim3(CG * RL), 0, s, a);
    }
    return check_launch("depth_ingest_kernel");
}
