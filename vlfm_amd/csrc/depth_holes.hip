// depth_holes.hip -- fill_small_holes (vlfm/utils/img_utils.py:361-390) on the GPU (gfx950), called by
// ObstacleMap.update_map before the obstacle scatter (vlfm/mapping/obstacle_map.py:86-91).
//
//   binary = (depth == 0)                                   bit plane written by depth_ingest_kernel (one pass, no extra read)
//   contours = cv2.findContours(binary, RETR_TREE, SIMPLE)  scan_list (bitmap.h): every outer and hole border
//   for cnt: if cv2.contourArea(cnt) < thresh: cv2.drawContours(filled, [cnt], 0, 1, -1)
//                                                           shoelace in exact integer-valued f64, then the shared polygon
//                                                           rasteriser (raster.h) on the contour's bounding rows
//   depth = where(filled == 1, 1, depth)                    consumed as a bit plane by the scatter pass of depth ingest
//
// One workgroup per image.  Images without a zero texel (the normal case behind depth_camera_filtering.filter_depth) cost
// one early exit.  The border walk is inherently sequential (one lane); the scan for border starts, the area sums and
// the fills are wave/workgroup-parallel.  Integer/bit work, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vlfm_amd.h"
#include "bitmap.h"
#include "profile.h"
#include "raster.h"
#include "status.h"

namespace vlfm {

struct HoleArgs {
    const unsigned* holes;   // [n][H][hw]
    int* status;             // [n][2]; word 1: bit 0 = image has zeros, bit 1 = the speculative scatter pass hit a cell off the
                             //         map (consumed: reset to 0 here; bit 1 is promoted to word 0 = VLFM_ERR_INDEX unless the
                             //         frame has islands, whose surviving texels hole_scatter_kernel places -- and judges -- again)
    unsigned* traced;        // [n][H][hw] scratch
    unsigned* neg;           // [n][H][hw]
    unsigned* fs;            // [n][H][hw] polygon fill: solid
    unsigned* fp;            // [n][H][hw] polygon fill: parity
    unsigned* filled;        // [n][H][hw] OUT
    int2* pts;               // [n][cap_pts]
    int* starts;             // [n][cap_contours]
    int* lens;               // [n][cap_contours]
    int* counts;             // [n][4]: contours, filled contours, overflow, bit 0 image had zeros | bit 1 island frame
    int H, W, hw, cap_pts, cap_contours;
    double area_thresh;
    // undo journal of the speculative scatter pass (depth_ingest.hip), or null
    const vlfm_ingest_params* prm;  // [n] (env slot of each observation)
    unsigned* obstacle;             // [n_envs][S][stride]
    unsigned* journal;              // [n][journal_cap]
    int* journal_count;             // [n]
    int journal_cap, S, stride;
};

__global__ __launch_bounds__(256) void fill_small_holes_kernel(HoleArgs a) {
    __shared__ int sh_i[4];
    __shared__ double sh_area[4];
    __shared__ int sh_box[4][2];
    const int obs = blockIdx.x, tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const size_t poff = (size_t)obs * a.H * a.hw;
    const int plane_words = a.H * a.hw;
    unsigned* filled = a.filled + poff;
    int* counts = a.counts + (size_t)obs * 4;
    const int st = a.status[2 * obs + 1];
    if ((st & 1) == 0) {
        if (counts[3]) {  // the plane still holds a previous frame's fill
            for (int i = tid; i < plane_words; i += nth) filled[i] = 0u;
        }
        __syncthreads();
        if (tid == 0) {
            counts[0] = 0; counts[1] = 0; counts[2] = 0; counts[3] = 0;
            if (a.journal_count) a.journal_count[obs] = 0;  // nothing to take back: the speculative pass stands
            if (st & 2) {   // ... including its off-map hit: no hole, hence no island, the reference scatters that texel too
                a.status[2 * obs] = VLFM_ERR_INDEX;
                a.status[2 * obs + 1] = 0;
            }
        }
        return;
    }
    unsigned* traced = a.traced + poff;
    unsigned* neg = a.neg + poff;
    unsigned* fs = a.fs + poff;
    unsigned* fp = a.fp + poff;
    for (int i = tid; i < plane_words; i += nth) { traced[i] = 0u; neg[i] = 0u; fs[i] = 0u; fp[i] = 0u; filled[i] = 0u; }
    __threadfence();
    __syncthreads();
    int2* pts = a.pts + (size_t)obs * a.cap_pts;
    int* cstart = a.starts + (size_t)obs * a.cap_contours;
    int* clen = a.lens + (size_t)obs * a.cap_contours;
    if (wave == 0) {
        ContourSink sink;
        sink.pts = pts; sink.start = cstart; sink.len = clen; sink.cap_pts = a.cap_pts; sink.cap_contours = a.cap_contours;
        sink.n_pts = 0; sink.n_contours = 0; sink.overflow = 0;
        Bits b{a.holes + poff, a.hw, a.H, a.W};
        scan_list(b, traced, neg, 2 /* CHAIN_APPROX_SIMPLE */, sink);
        if (lane == 0) { sh_i[0] = sink.n_contours; sh_i[1] = sink.overflow; }
    }
    __threadfence();
    __syncthreads();
    const int n_contours = sh_i[0], overflow = sh_i[1];
    int n_filled = 0;
    if (!overflow) {
        LdsBitmap fb;  // global-memory bitmaps here (rare path); the rasteriser only needs atomics on them
        fb.solid = fs; fb.parity = fp; fb.rows = a.H; fb.cols = a.W; fb.words = a.hw;
        for (int c = 0; c < n_contours; c++) {
            const int n = clen[c];
            const int2* cp = pts + cstart[c];
            // cv2.contourArea: |sum(x_prev * y - y_prev * x)| / 2 over float-converted integer points.  Every product and
            // partial sum is an integer below 2^53, so the f64 sum is exact in any order.
            double part = 0.0;
            int ylo = a.H, yhi = -1;
            for (int i = tid; i < n; i += nth) {
                const int2 p0 = cp[i == 0 ? n - 1 : i - 1], p1 = cp[i];
                part += (double)p0.x * (double)p1.y - (double)p0.y * (double)p1.x;
                ylo = min(ylo, p1.y); yhi = max(yhi, p1.y);
            }
            for (int off = 32; off > 0; off >>= 1) {
                part += __shfl_down(part, off, 64);
                ylo = min(ylo, __shfl_down(ylo, off, 64));
                yhi = max(yhi, __shfl_down(yhi, off, 64));
            }
            if (lane == 0) { sh_area[wave] = part; sh_box[wave][0] = ylo; sh_box[wave][1] = yhi; }
            __syncthreads();
            const double a00 = sh_area[0] + sh_area[1] + sh_area[2] + sh_area[3];
            const int y0 = min(min(sh_box[0][0], sh_box[1][0]), min(sh_box[2][0], sh_box[3][0]));
            const int y1 = max(max(sh_box[0][1], sh_box[1][1]), max(sh_box[2][1], sh_box[3][1]));
            __syncthreads();
            if (!(fabs(a00 * 0.5) < a.area_thresh)) continue;  // img_utils.py:383
            n_filled++;
            for (int i = tid; i < n; i += nth) {
                const int2 p0 = cp[i == 0 ? n - 1 : i - 1], p1 = cp[i];
                raster_edge(fb, (long long)p0.x << XY_SHIFT, p0.y, (long long)p1.x << XY_SHIFT, p1.y);
            }
            __threadfence();
            __syncthreads();
            // resolve the touched rows, OR them into the result and clear the work planes for the next contour
            for (int y = y0 + tid; y <= y1; y += nth) {
                unsigned carry = 0;
                for (int w = 0; w < a.hw; w++) {
                    const int i = y * a.hw + w;
                    unsigned p = fp[i];
                    p ^= p << 1; p ^= p << 2; p ^= p << 4; p ^= p << 8; p ^= p << 16;
                    if (carry) p = ~p;
                    carry = p >> 31;
                    const unsigned cov = p | fs[i];
                    if (cov) filled[i] |= cov;
                    fs[i] = 0u; fp[i] = 0u;
                }
            }
            __threadfence();
            __syncthreads();
        }
        // bits beyond the image width are not pixels
        if (a.W & 31) {
            const unsigned tail = (1u << (a.W & 31)) - 1u;
            for (int y = tid; y < a.H; y += nth) filled[y * a.hw + a.hw - 1] &= tail;
        }
    }
    __threadfence();
    __syncthreads();
    // Valid texels inside a filled contour ("islands")?  The reference's drawContours(.., -1) covers them, so they become
    // 1.0 and are dropped (img_utils.py:385-388), but the speculative pass has already placed them.  Take back every
    // obstacle bit that pass was the first to set (its journal); hole_scatter_kernel then re-places the valid texels
    // outside the filled area from the image.  (Bits that were set before this step are not in the journal and stay.)
    int island = 0, joverflow = 0;
    if (a.journal && !overflow) {
        const unsigned* holes = a.holes + poff;
        int any = 0;
        for (int i = tid; i < plane_words; i += nth) any |= (filled[i] & ~holes[i]) != 0u;
        island = __syncthreads_or(any);
        if (island) {
            const int nj_raw = a.journal_count[obs];
            joverflow = nj_raw > a.journal_cap;
            const int nj = min(nj_raw, a.journal_cap);
            unsigned* grid = a.obstacle + (size_t)a.prm[obs].env * a.S * a.stride;
            const unsigned* jr = a.journal + (size_t)obs * a.journal_cap;
            for (int i = tid; i < nj; i += nth) {
                const unsigned cell = jr[i];
                const unsigned row = cell / (unsigned)a.S, col = cell % (unsigned)a.S;
                atomicAnd(&grid[(size_t)row * a.stride + (col >> 5)], ~(1u << (col & 31)));
            }
            __threadfence();
        }
    }
    __syncthreads();
    if (tid == 0) {
        counts[0] = n_contours; counts[1] = n_filled; counts[2] = overflow | (joverflow ? 2 : 0);
        counts[3] = 1 | (island ? 2 : 0);
        a.status[2 * obs + 1] = 0;  // consumed
        // an off-map hit of the speculative pass stands unless this is an island frame (then every surviving valid texel is
        // placed a second time by hole_scatter_kernel, which reports an off-map cell itself)
        if ((st & 2) && !island) a.status[2 * obs] = VLFM_ERR_INDEX;
        if (a.journal_count) a.journal_count[obs] = 0;
    }
}

struct HoleLayout { size_t plane, off_traced, off_neg, off_fs, off_fp, off_pts, off_starts, off_lens, total; };
static HoleLayout hole_layout(int n, int H, int W, int cap_pts, int cap_contours) {
    HoleLayout L;
    const size_t hw = (W + 31) / 32;
    L.plane = (size_t)n * H * hw * 4;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
    L.off_traced = take(L.plane); L.off_neg = take(L.plane); L.off_fs = take(L.plane); L.off_fp = take(L.plane);
    L.off_pts = take((size_t)n * cap_pts * sizeof(int2));
    L.off_starts = take((size_t)n * cap_contours * 4);
    L.off_lens = take((size_t)n * cap_contours * 4);
    L.total = o;
    return L;
}

}  // namespace vlfm

using namespace vlfm;

extern "C" size_t vlfm_hole_scratch_bytes(int n, int height, int width, int cap_pts, int cap_contours) {
    if (n <= 0 || height <= 0 || width <= 0 || cap_pts <= 0 || cap_contours <= 0) return 0;
    return hole_layout(n, height, width, cap_pts, cap_contours).total;
}

extern "C" int vlfm_fill_small_holes_batched(const uint32_t* d_hole_bits, const int32_t* d_status, int n, int height,
                                             int width, double area_thresh, void* d_scratch, size_t scratch_bytes,
                                             int cap_pts, int cap_contours, uint32_t* d_filled_bits, int32_t* d_counts,
                                             const vlfm_ingest_params* d_params, uint32_t* d_obstacle, int map_size,
                                             const vlfm_scatter_journal* journal, void* stream) {
    if (n == 0) return VLFM_OK;
    if (!d_hole_bits || !d_status || !d_scratch || !d_filled_bits || !d_counts || n < 0 || height <= 0 || width <= 0 ||
        width > 2048 || cap_pts <= 0 || cap_contours <= 0)
        return fail(VLFM_ERR_INVALID, "fill_small_holes_batched: bad argument (width <= 2048)");
    const HoleLayout L = hole_layout(n, height, width, cap_pts, cap_contours);
    if (scratch_bytes < L.total) return fail(VLFM_ERR_CAPACITY, "fill_small_holes_batched: scratch too small");
    unsigned char* base = (unsigned char*)d_scratch;
    HoleArgs a;
    a.holes = d_hole_bits; a.status = const_cast<int*>(d_status);
    a.traced = (unsigned*)(base + L.off_traced); a.neg = (unsigned*)(base + L.off_neg);
    a.fs = (unsigned*)(base + L.off_fs); a.fp = (unsigned*)(base + L.off_fp);
    a.filled = d_filled_bits;
    a.pts = (int2*)(base + L.off_pts); a.starts = (int*)(base + L.off_starts); a.lens = (int*)(base + L.off_lens);
    a.counts = d_counts;
    a.H = height; a.W = width; a.hw = (width + 31) / 32; a.cap_pts = cap_pts; a.cap_contours = cap_contours;
    a.area_thresh = area_thresh;
    a.prm = nullptr; a.obstacle = nullptr; a.journal = nullptr; a.journal_count = nullptr; a.journal_cap = 0;
    a.S = map_size; a.stride = (map_size + 31) / 32;
    if (journal && journal->d_cells && journal->d_count && journal->capacity > 0) {
        if (!d_params || !d_obstacle || map_size <= 0)
            return fail(VLFM_ERR_INVALID, "fill_small_holes_batched: a journal needs d_params, d_obstacle and map_size");
        a.prm = d_params; a.obstacle = d_obstacle;
        a.journal = journal->d_cells; a.journal_count = journal->d_count; a.journal_cap = journal->capacity;
    }
    VLFM_TIMED("fill_small_holes_kernel", stream);
    VLFM_KLAUNCH(fill_small_holes_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("fill_small_holes_kernel");
}
