// obstacle_map.hip -- gfx950 kernels for vlfm.mapping.ObstacleMap
// (reference: /root/reference/vlfm/mapping/obstacle_map.py; third-party steps restated per SURVEY.md App. B2/B3).
//
// All planes are bit-packed (bitmap.h): obstacles (written by depth_ingest's scatter), navigable, explored, plus
// scratch planes.  Per step and environment:
//
//   navigable_kernel     navigable = ~dilate(obstacles, k x k) ; explored &= navigable         obstacle_map.py:105-109,127
//   fog_of_war_kernel    one workgroup per env, everything in LDS on a window around the agent: cone sector raster,
//                        obstacle blobs in the cone -> shadow-casting thick lines, visible component nearest the agent,
//                        filled, dilated 3x3, OR-ed into explored                               obstacle_map.py:115-126 + B2
//   explored_select      external components of explored; if more than one keep (and fill) the agent's
//                                                                                               obstacle_map.py:128-146
//   frontier_kernel      dilate explored 5x5, small-pocket filter, border chain of the explored region, frontier runs
//                        and their arc-length midpoints                                         obstacle_map.py:155-169 + B3
//
// Synchronisation between the phases of a kernel is WORKGROUP scope everywhere (wg_sync_global, border_parallel.h): one workgroup owns
// an environment and is its own consumer.  Round 4: the explored-selection and frontier kernels still used __threadfence() -- a
// device-scope release, i.e. an L2 write-back -- at 17 phase boundaries; alone in the machine that costs a few microseconds, with 256
// workgroups doing it at once the phases around global-memory hand-offs took 10 x as long (tools/phase_probe.py: window copy 4 -> 47 us,
// offset + pick 6 -> 116 us, bad flags 14 -> 136 us at 1 -> 256 environments).
// Dense work (dilations, masks, rasterisation, packing) is data-parallel over words/pixels; the order-dependent part
// (Suzuki-Abe border chains) is followed by the WHOLE workgroup: successor tables + list ranking by pointer jumping
// (border_parallel.h), with the one-lane walk of bitmap.h kept for short borders and as the fallback.  No MFMA; bit and
// integer work.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vlfm_amd.h"
#include "bitmap.h"
#include "border_parallel.h"
#include "profile.h"
#include "raster.h"
#include "status.h"

namespace vlfm {

// ------------------------------------------------------------------------------------------------ pack / unpack
__global__ void pack_u8_kernel(const unsigned char* __restrict__ src, unsigned* __restrict__ dst, int rows, int cols,
                               int stride) {
    const int y = blockIdx.y, wi = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = blockIdx.z;
    if (wi >= stride) return;
    const unsigned char* row = src + (plane * rows + y) * (size_t)cols;
    unsigned w = 0;
    for (int b = 0; b < 32; b++) {
        const int x = wi * 32 + b;
        if (x < cols && row[x]) w |= 1u << b;
    }
    dst[(plane * rows + y) * (size_t)stride + wi] = w;
}

__global__ void unpack_u8_kernel(const unsigned* __restrict__ src, unsigned char* __restrict__ dst, int rows, int cols,
                                 int stride) {
    const int y = blockIdx.y, x = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = blockIdx.z;
    if (x >= cols) return;
    dst[(plane * rows + y) * (size_t)cols + x] = (src[(plane * rows + y) * (size_t)stride + (x >> 5)] >> (x & 31)) & 1u;
}

// ------------------------------------------------------------------------------------------------ dilation on words
// horizontal dilation of one row word by radius r (r <= 31) given its neighbours
__device__ inline unsigned hdilate(unsigned prev, unsigned cur, unsigned next, int r) {
    unsigned out = cur;
    for (int d = 1; d <= r; d++) out |= (cur << d) | (prev >> (32 - d)) | (cur >> d) | (next << (32 - d));
    return out;
}

__device__ inline unsigned row_word(const unsigned* plane, int stride, int rows, int y, int wi) {
    if ((unsigned)y >= (unsigned)rows || (unsigned)wi >= (unsigned)stride) return 0u;
    return plane[(size_t)y * stride + wi];
}

__device__ inline unsigned tail_mask(int cols, int wi) {
    const int rem = cols - wi * 32;
    return rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
}

// dst = dilate(src, (2rx+1) x (2ry+1)) over the rows [y_lo, y_hi] of each listed plane.
// mode 0: plain;  mode 1: dst = ~dilated (navigable) and and_plane &= dst (explored &= navigable)
__global__ __launch_bounds__(256) void dilate_bits_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst,
                                                          unsigned* __restrict__ and_plane, const int* __restrict__ env,
                                                          int rows, int cols, int stride, int rx, int ry, int mode) {
    const int e = env ? env[blockIdx.z] : (int)blockIdx.z;
    const size_t off = (size_t)e * rows * stride;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * stride) return;
    const int y = idx / stride, wi = idx - y * stride;
    const unsigned* s = src + off;
    unsigned acc = 0;
    for (int dy = -ry; dy <= ry; dy++) {
        const unsigned c = row_word(s, stride, rows, y + dy, wi);
        const unsigned p = row_word(s, stride, rows, y + dy, wi - 1), n = row_word(s, stride, rows, y + dy, wi + 1);
        acc |= hdilate(p, c, n, rx);
    }
    const unsigned m = tail_mask(cols, wi);
    if (mode == 0) {
        dst[off + idx] = acc & m;
    } else {
        const unsigned nav = ~acc & m;
        dst[off + idx] = nav;
        if (and_plane) and_plane[off + idx] &= nav;
    }
}

// ------------------------------------------------------------------------------------------------ contour test entry
// One wavefront per plane: RETR_EXTERNAL contours of a bit-packed image (used by the parity tests and by hosts that
// want raw contours).
__global__ __launch_bounds__(64) void find_contours_kernel(const unsigned* __restrict__ img, unsigned* __restrict__ traced,
                                                           unsigned* __restrict__ neg, int rows, int cols, int stride,
                                                           int method, int2* __restrict__ pts, int cap_pts,
                                                           int* __restrict__ starts, int* __restrict__ lens,
                                                           int cap_contours, int* __restrict__ counts /* [planes][3] */) {
    const size_t plane = blockIdx.x;
    Bits b{img + plane * rows * stride, stride, rows, cols};
    unsigned* t = traced + plane * rows * stride;
    unsigned* ng = neg + plane * rows * stride;
    for (int i = threadIdx.x; i < rows * stride; i += 64) { t[i] = 0u; ng[i] = 0u; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    ContourSink sink;
    sink.pts = pts + plane * cap_pts; sink.start = starts + plane * cap_contours; sink.len = lens + plane * cap_contours;
    sink.cap_pts = cap_pts; sink.cap_contours = cap_contours; sink.n_pts = 0; sink.n_contours = 0; sink.overflow = 0;
    scan_external(b, t, ng, 0, rows - 1, method, sink);
    if (threadIdx.x == 0) {
        counts[plane * 3 + 0] = sink.n_contours;
        counts[plane * 3 + 1] = sink.n_pts;
        counts[plane * 3 + 2] = sink.overflow;
    }
}


// The same scan by a whole workgroup (border_parallel.h) on a padded LDS copy of the plane: what fog_of_war / explored_select /
// frontier do on their windows, exposed for images that fit (3 planes of (rows + 2) x (stride + 2) words in 144 KB) so that the
// parallel follower can be checked against cv2.findContours on arbitrary bitmaps (tests/test_obstacle_prims_gpu.py).
struct WgContourWork { unsigned* planes6; int* pixbase; int plane_words, cap_states, use_lds; };
__global__ __launch_bounds__(1024) void find_contours_wg_kernel(const unsigned* __restrict__ img, int rows, int cols, int stride,
                                                                int method, WgContourWork wk, unsigned lds_total,
                                                                int2* __restrict__ pts, int cap_pts,
                                                                int* __restrict__ starts, int* __restrict__ lens,
                                                                int cap_contours, int* __restrict__ counts /* [planes][3] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_win[];
    __shared__ int sh_wg[WG_SH_INTS];
    const size_t plane = blockIdx.x;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int pw = (stride + 2) | 1, wn = (rows + 2) * pw;   // (odd row stride: border_parallel.h, row_next_start_lane)
    unsigned* L_img = lds_win;
    unsigned* L_tr = lds_win + wn;
    unsigned* L_ng = lds_win + 2 * wn;
    const unsigned* src = img + plane * rows * stride;
    for (int i = tid; i < wn; i += nth) {
        const int ly = i / pw - 1, lw = i % pw - 1;
        const bool real = (unsigned)ly < (unsigned)rows && (unsigned)lw < (unsigned)stride;
        unsigned w = real ? src[(size_t)ly * stride + lw] : 0u;
        if (real && lw == stride - 1 && (cols & 31)) w &= (1u << (cols & 31)) - 1u;
        L_img[i] = w; L_tr[i] = 0u; L_ng[i] = 0u;
    }
    __syncthreads();
    ContourSink sink;
    sink.pts = pts + plane * cap_pts; sink.start = starts + plane * cap_contours; sink.len = lens + plane * cap_contours;
    sink.cap_pts = cap_pts; sink.cap_contours = cap_contours; sink.n_pts = 0; sink.n_contours = 0; sink.overflow = 0;
    Bits b{L_img + pw + 1, pw, rows, stride * 32, 1};
    WalkTables T;
    unsigned* base = wk.planes6 + plane * 6 * (size_t)wk.plane_words;
    T.bmask = base; T.wprefix = reinterpret_cast<int*>(base + wk.plane_words);
    T.next = reinterpret_cast<int*>(base + 2 * (size_t)wk.plane_words); T.sinfo = base + 3 * (size_t)wk.plane_words;
    T.jd0 = base + 4 * (size_t)wk.plane_words; T.jd1 = base + 5 * (size_t)wk.plane_words;
    T.pixbase = wk.pixbase + plane * cap_pts;
    T.cap_bp = cap_pts; T.cap_states = wk.cap_states;
    T.wrows = rows; T.wwords = stride;
    // tables of the follower in what is left of the LDS (border_parallel.h, round 4); wk.use_lds = 0 forces the global-table form
    const unsigned used = 3u * (unsigned)wn * 4u;
    wg_scan_external_lds(b, L_tr + pw + 1, L_ng + pw + 1, method, sink, T, wk.use_lds ? lds_win + 3 * wn : nullptr,
                         lds_total > used ? lds_total - used : 0u, sh_wg);
    if (tid == 0) {
        counts[plane * 3 + 0] = sink.n_contours;
        counts[plane * 3 + 1] = sink.n_pts;
        counts[plane * 3 + 2] = sink.overflow;
    }
}

// Optional phase timing (compile with -DVLFM_PHASE_TIMING; tools/phase_probe.py): workgroup 0's thread 0 stamps the
// constant-rate 100 MHz counter at phase boundaries.  Zero cost when the macro is not defined.
#ifdef VLFM_PHASE_TIMING
__device__ long long g_phase_clock[3][16];
__device__ int g_phase_block;              // the workgroup that stamps (vlfm_debug_phase_block)
__device__ long long g_wg_first[3][1024];  // every workgroup's first and last stamp: which environment is the slow one?
__device__ long long g_wg_last[3][1024];
#define VLFM_PHASE(kernel_id, k)                                                                  \
    do {                                                                                          \
        __syncthreads();                                                                          \
        if (threadIdx.x == 0) {                                                                   \
            const long long t_ = wall_clock64();                                                  \
            if ((int)blockIdx.x == g_phase_block) g_phase_clock[kernel_id][k] = t_;               \
            if (blockIdx.x < 1024) { if ((k) == 0) g_wg_first[kernel_id][blockIdx.x] = t_; g_wg_last[kernel_id][blockIdx.x] = t_; } \
        }                                                                                         \
    } while (0)
// the same without the barrier, for a stamp inside single-wavefront code
#define VLFM_STAMP(kernel_id, k)                                                                  \
    do {                                                                                          \
        if ((int)blockIdx.x == g_phase_block && (threadIdx.x & 63) == 0) g_phase_clock[kernel_id][k] = wall_clock64(); \
    } while (0)
#else
#define VLFM_PHASE(kernel_id, k) do {} while (0)
#define VLFM_STAMP(kernel_id, k) do {} while (0)
#endif

// ================================================================================================ drawing helpers
// Window view of a bit plane living in LDS: wn x wn cells, top-left = image cell (ox, oy); image is S x S.
struct Win {
    int ox, oy, wn, words, S;
};

__device__ inline void win_clear_span(unsigned* plane, const Win& w, int y, int x1, int x2) {  // image coords, inclusive
    const int ly = y - w.oy;
    if ((unsigned)ly >= (unsigned)w.wn) return;
    int a = x1 - w.ox, b = x2 - w.ox;
    if (a < 0) a = 0;
    if (b > w.wn - 1) b = w.wn - 1;
    if (a > b) return;
    for (int wi = a >> 5; wi <= (b >> 5); wi++) {
        const int lo = wi == (a >> 5) ? (a & 31) : 0, hi = wi == (b >> 5) ? (b & 31) : 31;
        const unsigned m = (hi == 31 ? 0xFFFFFFFFu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
        atomicAnd(&plane[ly * w.words + wi], ~m);
    }
}
__device__ inline void win_clear_px(unsigned* plane, const Win& w, long long x, long long y) {
    if (x < 0 || y < 0 || x >= w.S || y >= w.S) return;
    win_clear_span(plane, w, (int)y, (int)x, (int)x);
}

// cv Line2: DDA between 16.16 endpoints, clipped to the image.  seg / nseg: only that share of the DDA's steps (the steps are
// x1 + k, y1 + k * y_step -- exact integer arithmetic, so a share can start anywhere), so that one line can be spread over lanes
__device__ inline void line2_clear(unsigned* plane, const Win& w, long long x1, long long y1, long long x2, long long y2,
                                   int seg = 0, int nseg = 1) {
    if (!clip_line((long long)w.S << XY_SHIFT, (long long)w.S << XY_SHIFT, x1, y1, x2, y2)) return;
    long long dx = x2 - x1, dy = y2 - y1;
    const long long j = dx < 0 ? -1 : 0, i = dy < 0 ? -1 : 0;
    const long long ax = (dx ^ j) - j, ay = (dy ^ i) - i;
    long long x_step, y_step;
    int ecount;
    if (ax > ay) {
        dy = (dy ^ j) - j;
        if (j) { long long t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
        x_step = XY_ONE;
        y_step = (dy << XY_SHIFT) / (ax | 1);
        ecount = (int)((x2 - x1) >> XY_SHIFT);
    } else {
        dx = (dx ^ i) - i;
        if (i) { long long t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
        x_step = (dx << XY_SHIFT) / (ay | 1);
        y_step = XY_ONE;
        ecount = (int)((y2 - y1) >> XY_SHIFT);
    }
    x1 += XY_ONE >> 1;
    y1 += XY_ONE >> 1;
    if (seg == 0) win_clear_px(plane, w, (x2 + (XY_ONE >> 1)) >> XY_SHIFT, (y2 + (XY_ONE >> 1)) >> XY_SHIFT);
    const int n = ecount + 1;                     // steps k = 0 .. ecount
    const int k0 = n > 0 ? (int)((long long)n * seg / nseg) : 0;
    int left = n > 0 ? (int)((long long)n * (seg + 1) / nseg) - k0 : 0;
    if (ax > ay) {
        x1 = (x1 >> XY_SHIFT) + k0;
        y1 += (long long)k0 * y_step;
        while (left-- > 0) { win_clear_px(plane, w, x1, y1 >> XY_SHIFT); x1++; y1 += y_step; }
    } else {
        y1 = (y1 >> XY_SHIFT) + k0;
        x1 += (long long)k0 * x_step;
        while (left-- > 0) { win_clear_px(plane, w, x1 >> XY_SHIFT, y1); x1 += x_step; y1++; }
    }
}

// cv FillConvexPoly for a 16.16 quadrilateral (LINE_8), painting zeros
// part < 0: everything; part 0..3: only outline edge `part`; part 4: only the interior scanlines (all parts clear bits, so
// they may run concurrently on different lanes).  seg / nseg: that share of the edge's DDA steps, or of the interior's scanlines
// inside the window (rows in front of the share are stepped over edge event by edge event: the edges are linear in between)
__device__ inline void fill_convex_quad_clear(unsigned* plane, const Win& w, const long long* vx, const long long* vy,
                                              int part = -1, int seg = 0, int nseg = 1) {
    const int npts = 4, shift = XY_SHIFT;
    const int delta = 1 << shift >> 1;
    struct { int idx, di; long long x, dx; int ye; } edge[2];
    int imin = 0, edges = npts;
    long long xmin = vx[0], xmax = vx[0], ymin = vy[0], ymax = vy[0];
    long long p0x = vx[npts - 1], p0y = vy[npts - 1];
    for (int i = 0; i < npts; i++) {
        if (vy[i] < ymin) { ymin = vy[i]; imin = i; }
        if (vy[i] > ymax) ymax = vy[i];
        if (vx[i] > xmax) xmax = vx[i];
        if (vx[i] < xmin) xmin = vx[i];
        if (part < 0) line2_clear(plane, w, p0x, p0y, vx[i], vy[i]);
        else if (part == i) line2_clear(plane, w, p0x, p0y, vx[i], vy[i], seg, nseg);
        p0x = vx[i]; p0y = vy[i];
    }
    if (part >= 0 && part < 4) return;
    xmin = (xmin + delta) >> shift; xmax = (xmax + delta) >> shift;
    ymin = (ymin + delta) >> shift; ymax = (ymax + delta) >> shift;
    if ((int)xmax < 0 || (int)ymax < 0 || (int)xmin >= w.S || (int)ymin >= w.S) return;
    if (ymax > w.S - 1) ymax = w.S - 1;
    // rows this call paints: the rows of the quadrilateral that lie in the image AND in the window (rows outside the window
    // paint nothing), then the caller's share of them
    int ya = (int)ymin > 0 ? (int)ymin : 0, yb = (int)ymax;
    if (ya < w.oy) ya = w.oy;
    if (yb > w.oy + w.wn - 1) yb = w.oy + w.wn - 1;
    if (ya > yb) return;
    if (nseg > 1) {
        const int n = yb - ya + 1, a0 = ya;
        ya = a0 + (int)((long long)n * seg / nseg);
        yb = a0 + (int)((long long)n * (seg + 1) / nseg) - 1;
        if (ya > yb) return;
    }
    edge[0].idx = edge[1].idx = imin;
    int y = (int)ymin;
    edge[0].ye = edge[1].ye = y;
    edge[0].di = 1; edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -XY_ONE;
    edge[0].dx = edge[1].dx = 0;
    do {
        for (int i = 0; i < 2; i++) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx, di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                int ty = 0;
                for (; edges-- > 0;) {
                    ty = (int)((vy[idx] + delta) >> shift);
                    if (ty > y) {
                        const long long xs = vx[idx0], xe = vx[idx];
                        edge[i].ye = ty;
                        edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                        edge[i].x = xs;
                        edge[i].idx = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += di;
                    if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        if (y < ya) {   // nothing to paint before row ya: on to the next edge event or to ya, whichever comes first
            int ny = ya;
            if (edge[0].ye < ny) ny = edge[0].ye;
            if (edge[1].ye < ny) ny = edge[1].ye;
            if (ny <= y) ny = y + 1;
            const long long k = ny - y;
            edge[0].x += edge[0].dx * k;
            edge[1].x += edge[1].dx * k;
            y = ny - 1;
            continue;
        }
        if (y >= 0) {
            int left = 0, right = 1;
            if (edge[0].x > edge[1].x) { left = 1; right = 0; }
            int xx1 = (int)((edge[left].x + (XY_ONE >> 1)) >> XY_SHIFT);
            int xx2 = (int)((edge[right].x + (XY_ONE >> 1)) >> XY_SHIFT);
            if (xx2 >= 0 && xx1 < w.S) {
                if (xx1 < 0) xx1 = 0;
                if (xx2 >= w.S) xx2 = w.S - 1;
                win_clear_span(plane, w, y, xx1, xx2);
            }
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
        if (y >= yb) break;
    } while (++y <= (int)ymax);
}

// cv Circle(center, radius, fill) painting zeros
__device__ inline void circle_clear(unsigned* plane, const Win& w, int cx, int cy, int radius) {
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    auto span = [&](int y, int x1, int x2) {
        if ((unsigned)y >= (unsigned)w.S) return;
        if (x1 < 0) x1 = 0;
        if (x2 > w.S - 1) x2 = w.S - 1;
        if (x1 <= x2) win_clear_span(plane, w, y, x1, x2);
    };
    while (dx >= dy) {
        span(cy - dy, cx - dx, cx + dx); span(cy + dy, cx - dx, cx + dx);
        span(cy - dx, cx - dy, cx + dy); span(cy + dx, cx - dy, cx + dy);
        dy++;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= 2 & mask;
    }
}

// cv ThickLine(thickness = 2) from integer p0 to p1, painting zeros (frontier_exploration's shadow cuts)
// part < 0: the whole line; parts 0..6 = the four outline edges, the interior, the two end discs (independent: every part
// only clears bits), so that one line can be spread over seven lanes
constexpr int THICK_LINE_PARTS = 7;
constexpr int THICK_LINE_SEGS = 4;   // shares per outline edge and of the interior (a 105-px edge on one lane took 50 us)
__device__ inline void thick_line2_clear(unsigned* plane, const Win& w, int p0x, int p0y, int p1x, int p1y, int part = -1,
                                         int seg = 0, int nseg = 1) {
    const long long a0x = (long long)p0x << XY_SHIFT, a0y = (long long)p0y << XY_SHIFT;
    const long long a1x = (long long)p1x << XY_SHIFT, a1y = (long long)p1y << XY_SHIFT;
    const double INV = 1. / XY_ONE;
    const double dx = (double)(a0x - a1x) * INV, dy = (double)(a1y - a0y) * INV;
    double r = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
    const long long thickness = 2LL << (XY_SHIFT - 1);
    if (fabs(r) > 2.220446049250313e-16) {
        r = __ddiv_rn((double)thickness, sqrt(r));
        const long long dpx = __double2ll_rn(__dmul_rn(dy, r)), dpy = __double2ll_rn(__dmul_rn(dx, r));
        const long long vx[4] = {a0x + dpx, a0x - dpx, a1x - dpx, a1x + dpx};
        const long long vy[4] = {a0y + dpy, a0y - dpy, a1y - dpy, a1y + dpy};
        if (part < 5) fill_convex_quad_clear(plane, w, vx, vy, part, seg, nseg);
    }
    const int rad = (int)((thickness + (XY_ONE >> 1)) >> XY_SHIFT);
    if (part < 0 || part == 5) circle_clear(plane, w, p0x, p0y, rad);
    if (part < 0 || part == 6) circle_clear(plane, w, p1x, p1y, rad);
}

// |cv::pointPolygonTest(contour, pt, true)| and the inside flag for an integer contour; evaluated by one wavefront.
// Returns squared-distance quotient (num/denom as a double) in *d2 and the crossing parity in *inside.
__device__ inline void wave_point_polygon(const int2* c, int n, int ptx, int pty, double* d2_out, int* inside_out) {
    const int lane = threadIdx.x & 63;
    double best = 3.4028234663852886e38;  // FLT_MAX / 1
    int counter = 0;
    const float fx = (float)ptx, fy = (float)pty;
    for (int i = lane; i < n; i += 64) {
        const int2 a = c[i == 0 ? n - 1 : i - 1], b = c[i];
        const float v0x = (float)a.x, v0y = (float)a.y, vx = (float)b.x, vy = (float)b.y;
        const double dx = vx - v0x, dy = vy - v0y, dx1 = fx - v0x, dy1 = fy - v0y, dx2 = fx - vx, dy2 = fy - vy;
        double num, den = 1;
        if (dx1 * dx + dy1 * dy <= 0) num = dx1 * dx1 + dy1 * dy1;
        else if (dx2 * dx + dy2 * dy >= 0) num = dx2 * dx2 + dy2 * dy2;
        else { num = dy1 * dx - dx1 * dy; num *= num; den = dx * dx + dy * dy; }
        const double q = num / den;
        if (q < best) best = q;
        if (!((v0y <= fy && vy <= fy) || (v0y > fy && vy > fy) || (v0x < fx && vx < fx))) {
            double t = dy1 * dx - dx1 * dy;
            if (dy < 0) t = -t;
            counter += t > 0;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(best, off, 64);
        if (o < best) best = o;
        counter += __shfl_down(counter, off, 64);
    }
    *d2_out = __shfl(best, 0, 64);
    *inside_out = __shfl(counter, 0, 64) & 1;
}

// ================================================================================================ navigable
// navigable = ~dilate(obstacles, k x k) (obstacle_map.py:105-109); explored &= navigable (obstacle_map.py:127, applied
// before this step's reveal, which only ever adds navigable cells)
// -> dilate_bits_kernel mode 1.

// ================================================================================================ fog of war
constexpr int FOG_MAX_POLY = VLFM_FOG_MAX_POLY;
using FogParams = vlfm_fog_params;  // one per environment and step, host-computed (vlfm_fog_params_host)

struct MapPlanes {
    const unsigned* obstacle;  // [n_envs][S][stride]
    unsigned* navigable;    // [n_envs][S][stride]
    unsigned* explored;     // [n_envs][S][stride]
    int S, stride;
};

struct FogScratch {         // per environment slices of global scratch
    int2* pts;              // [n_envs][cap_pts]
    int* starts; int* lens; // [n_envs][cap_contours]
    int4* lines;            // [n_envs][cap_pts]
    int* status;            // [n_envs][4]: overflow flag, n obstacle contours, n lines, selected dist*1000
    int* bbox;              // [n_envs][4]: explored rows/cols touched so far (ymin, ymax, xmin, xmax)
    int cap_pts, cap_contours;
    unsigned* walk[6];      // parallel border follower (border_parallel.h): six [n_envs][walk_words] planes the later kernels own
    int* walk_pixbase;      // [n_envs][cap_pts]
    int walk_words;
    int lds_states;         // states whose list ranking fits the kernel's LDS (2 words each)
};

__device__ inline WalkTables fog_walk_tables(const FogScratch& sc, int env, int wrows, int wwords) {
    WalkTables T;
    const size_t o = (size_t)env * sc.walk_words;
    T.bmask = sc.walk[0] + o; T.wprefix = reinterpret_cast<int*>(sc.walk[1] + o);
    T.next = reinterpret_cast<int*>(sc.walk[2] + o); T.sinfo = sc.walk[3] + o;
    T.jd0 = sc.walk[4] + o; T.jd1 = sc.walk[5] + o;
    T.pixbase = sc.walk_pixbase + (size_t)env * sc.cap_pts;
    T.cap_bp = sc.cap_pts; T.cap_states = sc.walk_words < 65535 ? sc.walk_words : 65535;
    T.wrows = wrows; T.wwords = wwords;
    T.lds_states = sc.lds_states;
    return T;
}

// navigable = ~dilate(obstacles, k x k) (obstacle_map.py:105-109); explored &= navigable (obstacle_map.py:127 -- applied
// ahead of this step's reveal, which only ever adds navigable cells, so the order is immaterial).
//
// DIRTY WINDOWS (round 3).  The reference recomputes both over the whole S x S map every step; but obstacle bits only change
// within the camera's reach of the frames ingested since the last recompute, so `navigable` only changes inside that
// window grown by the dilation radius; and `explored &= navigable` only has to visit the bounding box of everything ever
// revealed (no explored bit exists outside it; the box is persistent device state maintained by fog_of_war_kernel).  The
// box, not a window of `navigable` changes: explored_select's filled contour (obstacle_map.py:145) may set explored bits
// on non-navigable cells anywhere inside the explored area, which the NEXT step's masking takes away again.
// The caller accumulates the windows per environment (ObstacleMapBatch: full map after reset(), after a frame whose reach
// leaves the map -- negative indices wrap, obstacle_map.py:101 -- or with a non-rigid camera transform) and hands three of
// them over per observation, inclusive (y0, y1, x0, x1), empty when y1 < y0:
//   win[0..3]   where `navigable` has to be recomputed
//   win[4..7]   where `explored` has to be masked: the caller's mirror of the revealed area's bounding box BEFORE this step
//   win[8..11]  where the frontier stage's derived planes have to be refreshed (frontier_prepare_kernel): the windows of
//               `navigable` changes since the last explore step, united with the bounding box AFTER this step grown by 3
// and sizes the launch for the largest window of the batch: the kernels walk (row, word) items of their window in a
// grid-stride loop, so that a step costs the window's ~3 000 words per environment instead of the plane's 32 000 (round 2:
// 101 us at 256 environments; early-exit workgroups over the full plane: 53 us; window-sized grid: see profiles/).
// win == null: the full-plane pass.
struct DirtyWin { int y0, y1, x0, x1; };
__device__ inline DirtyWin load_win(const int* __restrict__ w, int S) {
    if (!w) return DirtyWin{0, S - 1, 0, S - 1};
    return DirtyWin{w[0], w[1], w[2], w[3]};
}
__device__ inline bool win_empty(const DirtyWin& d) { return d.y1 < d.y0 || d.x1 < d.x0; }
__device__ inline DirtyWin win_union(const DirtyWin& a, const DirtyWin& b) {
    if (win_empty(a)) return b;
    if (win_empty(b)) return a;
    return DirtyWin{min(a.y0, b.y0), max(a.y1, b.y1), min(a.x0, b.x0), max(a.x1, b.x1)};
}
__device__ inline bool word_hit(const DirtyWin& d, int y, int wi) {
    return !win_empty(d) && y >= d.y0 && y <= d.y1 && wi >= (d.x0 >> 5) && wi <= (d.x1 >> 5);
}

__global__ __launch_bounds__(256) void navigable_kernel(const FogParams* __restrict__ prm, MapPlanes mp, int radius,
                                                        int update_obstacles, int explore, const int* __restrict__ windows) {
    const int e = prm[blockIdx.z].env;
    const int S = mp.S, stride = mp.stride;
    const size_t off = (size_t)e * S * stride;
    const int* w = windows ? windows + 12 * blockIdx.z : nullptr;
    const bool do_mask = explore && prm[blockIdx.z].n_poly > 0;   // only the explore branch masks (:127)
    DirtyWin w_nav = load_win(w, S), w_mask = load_win(w ? w + 4 : nullptr, S);
    if (!update_obstacles) w_nav = DirtyWin{0, -1, 0, -1};
    if (!do_mask) w_mask = DirtyWin{0, -1, 0, -1};
    const DirtyWin U = win_union(w_nav, w_mask);
    if (win_empty(U)) return;
    const int wx0 = U.x0 >> 5, ww = (U.x1 >> 5) - wx0 + 1, n_items = (U.y1 - U.y0 + 1) * ww;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x) {
        const int y = U.y0 + i / ww, wi = wx0 + i % ww;
        const bool recompute = word_hit(w_nav, y, wi), mask = word_hit(w_mask, y, wi);
        if (!recompute && !mask) continue;
        const int idx = y * stride + wi;
        unsigned nav;
        if (recompute) {
            const unsigned* src = mp.obstacle + off;
            unsigned acc = 0;
            for (int dy = -radius; dy <= radius; dy++)
                acc |= hdilate(row_word(src, stride, S, y + dy, wi - 1), row_word(src, stride, S, y + dy, wi),
                               row_word(src, stride, S, y + dy, wi + 1), radius);
            nav = ~acc & tail_mask(S, wi);
            mp.navigable[off + idx] = nav;
        } else {
            nav = mp.navigable[off + idx];
        }
        if (mask) {
            const unsigned ex = mp.explored[off + idx];
            if (ex & ~nav) mp.explored[off + idx] = ex & nav;
        }
    }
}

__device__ inline unsigned load_window_word(const unsigned* plane, int S, int stride, int ox, int oy, int ly, int lw) {
    const int y = oy + ly;
    if ((unsigned)y >= (unsigned)S) return 0u;
    const int x0 = ox + lw * 32;  // image x of bit 0 of this window word (may be negative)
    unsigned out = 0;
    const unsigned* row = plane + (size_t)y * stride;
    // assemble from up to two image words
    const int wi = x0 >= 0 ? (x0 >> 5) : -((-x0 + 31) >> 5);
    const int sh = x0 - wi * 32;  // 0..31
    const unsigned lo = (wi >= 0 && wi < stride) ? row[wi] : 0u;
    const unsigned hi = (wi + 1 >= 0 && wi + 1 < stride) ? row[wi + 1] : 0u;
    out = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
    return out;
}

__global__ __launch_bounds__(1024) void fog_of_war_kernel(const FogParams* __restrict__ prm, MapPlanes mp, FogScratch sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FogParams& P = prm[blockIdx.x];
    if (P.n_poly <= 0) return;
    const int S = mp.S;
    const int R = P.radius;
    const int wn = 2 * R + 5, words = (wn + 31) >> 5, plane_words = wn * words;
    const int ox = P.ax - R - 2, oy = P.ay - R - 2;
    unsigned* cone = reinterpret_cast<unsigned*>(smem);
    unsigned* par = cone + plane_words;     // parity scratch (cone), later fill parity
    unsigned* nav = par + plane_words;
    unsigned* obst = nav + plane_words;
    unsigned* vis = obst + plane_words;
    unsigned* fill = vis + plane_words;
    // padded copies for the two border scans (Bits::padded): image + both label planes, (wn + 2) x (words + 2) words each
    const int pw = (words + 2) | 1, pad_words = (wn + 2) * pw;
    unsigned* p_img = fill + plane_words;
    unsigned* p_tr = p_img + pad_words;
    unsigned* p_ng = p_tr + pad_words;
    int* sh_i = reinterpret_cast<int*>(p_ng + pad_words);  // small shared ints (16)
    unsigned* l_jd = reinterpret_cast<unsigned*>(sh_i + 16);   // 2 x sc.lds_states words: list-ranking buffers of the border follower
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const size_t eoff = (size_t)P.env * S * mp.stride;
    const unsigned last_mask = (wn & 31) ? ((1u << (wn & 31)) - 1u) : 0xFFFFFFFFu;
    Win W{ox, oy, wn, words, S};
    int2* pts = sc.pts + (size_t)P.env * sc.cap_pts;
    int* cstart = sc.starts + (size_t)P.env * sc.cap_contours;
    int* clen = sc.lens + (size_t)P.env * sc.cap_contours;
    int4* lines = sc.lines + (size_t)P.env * sc.cap_pts;
    int* status = sc.status + (size_t)P.env * 4;

    for (int i = tid; i < 6 * plane_words; i += nth) cone[i] = 0u;
    if (tid < 16) sh_i[tid] = 0;
    __syncthreads();

    VLFM_PHASE(0, 0);
    // ---- 1. cone sector (cv2.ellipse filled) into cone/par
    LdsBitmap bm;
    bm.solid = cone; bm.parity = par; bm.rows = S; bm.cols = S; bm.words = words;
    bm.ox = ox; bm.oy = oy; bm.wrows = wn; bm.wcols = wn;
    // (eight lanes per edge: raster.h, raster_edge_shared -- the cone's two straight sides are 100 pixels long)
    for (int t = tid; t < P.n_poly * 8; t += nth) {
        const int i = t >> 3, j = i == 0 ? P.n_poly - 1 : i - 1;
        const long long ax = P.poly[2 * j], ay = (P.poly[2 * j + 1] + (XY_ONE >> 1)) >> XY_SHIFT;
        const long long bx = P.poly[2 * i], by = (P.poly[2 * i + 1] + (XY_ONE >> 1)) >> XY_SHIFT;
        raster_edge_shared(bm, ax, (int)ay, bx, (int)by, t & 7, 8);
    }
    __syncthreads();
    resolve_rows(bm, tid, nth);
    __syncthreads();
    VLFM_PHASE(0, 1);
    // ---- 2. navigable window; obstacles_in_cone = cone & ~nav; visible = cone & nav
    for (int i = tid; i < plane_words; i += nth) {
        const int ly = i / words, lw = i - ly * words;
        unsigned c = cone[i];
        if (lw == words - 1) c &= last_mask;
        cone[i] = c;
        const unsigned n = load_window_word(mp.navigable + eoff, S, mp.stride, ox, oy, ly, lw);
        nav[i] = n;
        obst[i] = c & ~n;
        vis[i] = c & n;
        par[i] = 0u;
    }
    __syncthreads();
    VLFM_PHASE(0, 2);
    // ---- 3. external contours (SIMPLE) of the obstacle blobs: wave 0 walks a padded copy
    for (int i = tid; i < pad_words; i += nth) {
        const int ly = i / pw - 1, lw = i % pw - 1;
        p_img[i] = ((unsigned)ly < (unsigned)wn && (unsigned)lw < (unsigned)words) ? obst[ly * words + lw] : 0u;
        p_tr[i] = 0u; p_ng[i] = 0u;
    }
    __syncthreads();
    __shared__ int sh_wg[WG_SH_INTS];
    {
        ContourSink sink;
        sink.pts = pts; sink.start = cstart; sink.len = clen; sink.cap_pts = sc.cap_pts; sink.cap_contours = sc.cap_contours;
        sink.n_pts = 0; sink.n_contours = 0; sink.overflow = 0;
        Bits b{p_img + pw + 1, pw, wn, wn, 1};
        WalkTables T = fog_walk_tables(sc, P.env, wn, words);
        T.ljd0 = l_jd; T.ljd1 = l_jd + sc.lds_states;
        wg_scan_external_lds(b, p_tr + pw + 1, p_ng + pw + 1, 2, sink, T, l_jd, 8u * (unsigned)sc.lds_states, sh_wg);   // the whole workgroup (border_parallel.h)
        if (tid == 0) { sh_i[0] = sink.n_contours; sh_i[1] = sink.n_pts; sh_i[2] = sink.overflow; }
    }
    __threadfence_block();
    __syncthreads();
    const int n_obst = sh_i[0];
    if (sh_i[2]) { if (tid == 0) status[0] = 1; return; }
    if (tid == 0) { status[1] = n_obst; }
    if (n_obst == 0) return;  // "no obstacles in the cone": fog returned unchanged -> nothing revealed this step
    VLFM_PHASE(0, 3);
    // ---- 4. shadow-casting points: convex blobs contribute their two angular extremes, the others every vertex.
    // One WAVEFRONT per blob (round 3; before: the whole workgroup per blob with three barriers each, and ONE lane evaluating
    // the f64 atan2 of every vertex of a convex blob in a serial loop).  The cuts of step 5 only clear bits, so the order in
    // which blobs append their points is immaterial; np.argmin / np.argmax keep the FIRST extreme, i.e. the smallest index
    // among equal angles -- the wave reduction orders (angle, index) pairs accordingly.
    const int acx = P.ax - ox, acy = P.ay - oy;  // agent in window coordinates (contour points are window-local)
    for (int c = wave; c < n_obst; c += nth >> 6) {
        const int n = clen[c];
        const int2* cp = pts + cstart[c];
        // cv::isContourConvex on the SIMPLE vertices
        int flags = 0;
        for (int i = lane; i < n; i += 64) {
            const int2 p2 = cp[(i - 2 + 2 * n) % n], p1 = cp[(i - 1 + n) % n], p = cp[i];
            const int dx0 = p1.x - p2.x, dy0 = p1.y - p2.y, dx = p.x - p1.x, dy = p.y - p1.y;
            const int dxdy0 = dx * dy0, dydx0 = dy * dx0;
            flags |= dydx0 > dxdy0 ? 1 : (dydx0 < dxdy0 ? 2 : 3);
        }
        for (int off = 32; off > 0; off >>= 1) flags |= __shfl_xor(flags, off, 64);
        const bool convex = n > 0 && flags != 3;
        if (convex) {
            // get_two_farthest_points: rotate (pt - source) by the upstream matrix, atan2, first argmin / argmax
            double amin = 0, amax = 0;
            int imin = 0x7FFFFFFF, imax = 0x7FFFFFFF;
            for (int i = lane; i < n; i += 64) {
                const double px = (double)(cp[i].x - acx), py = (double)(cp[i].y - acy);
                // np.matmul accumulates from +0.0: a sum of two NEGATIVE zeros (the agent's own pixel, px = py = 0, under a rotation
                // with negative entries) comes out as +0.0 there, and atan2(+0, +0) = 0 where atan2(+0, -0) would be pi -- a
                // different angular extreme, different shadow lines (found by tests/test_obstacle_map_gpu.py's random headings:
                // the agent standing inside a convex obstacle blob)
                const double rx = __dadd_rn(__dadd_rn(0.0, __dmul_rn(px, P.rot_c)), __dmul_rn(py, P.rot_s));
                const double ry = __dadd_rn(__dadd_rn(0.0, __dmul_rn(px, -P.rot_s)), __dmul_rn(py, P.rot_c));
                const double a = atan2(ry, rx);
                if (imin == 0x7FFFFFFF || a < amin) { amin = a; imin = i; }
                if (imax == 0x7FFFFFFF || a > amax) { amax = a; imax = i; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const double oa = __shfl_xor(amin, off, 64), ob = __shfl_xor(amax, off, 64);
                const int oi = __shfl_xor(imin, off, 64), oj = __shfl_xor(imax, off, 64);
                if (oi != 0x7FFFFFFF && (imin == 0x7FFFFFFF || oa < amin || (oa == amin && oi < imin))) { amin = oa; imin = oi; }
                if (oj != 0x7FFFFFFF && (imax == 0x7FFFFFFF || ob > amax || (ob == amax && oj < imax))) { amax = ob; imax = oj; }
            }
            if (lane == 0) {
                const int base = atomicAdd(&sh_i[3], 2);
                lines[base] = make_int4(cp[imin].x, cp[imin].y, 0, 0);
                lines[base + 1] = make_int4(cp[imax].x, cp[imax].y, 0, 0);
            }
        } else {
            int base = 0;
            if (lane == 0) base = atomicAdd(&sh_i[3], n);
            base = __shfl(base, 0, 64);
            for (int i = lane; i < n; i += 64) lines[base + i] = make_int4(cp[i].x, cp[i].y, 0, 0);
        }
    }
    __threadfence_block();
    __syncthreads();
    const int n_lines = sh_i[3];
    if (tid == 0) status[2] = n_lines;
    VLFM_PHASE(0, 4);
    // ---- 5. cut the visible mask with 2-px lines from every point away from the agent (cv2.polylines, color 0).  End
    // points first (one f64 atan2 / cos / sin per LINE, not per task), then one (part, line) task per lane, PART-MAJOR so that
    // a wavefront runs one code path (edge / interior / end disc): the critical path becomes one edge + one interior + one
    // disc instead of a whole line
    for (int i = tid; i < n_lines; i += nth) {
        const int px = lines[i].x + ox, py = lines[i].y + oy;  // image coordinates
        const double ang = atan2((double)(py - P.ay), (double)(px - P.ax));
        const double ex = __dadd_rn((double)px, __dmul_rn(P.line_len, cos(ang)));
        const double ey = __dadd_rn((double)py, __dmul_rn(P.line_len, sin(ang)));
        lines[i].z = (int)ex; lines[i].w = (int)ey;            // .astype(np.int32): truncation
    }
    __threadfence_block();
    __syncthreads();
    VLFM_PHASE(0, 5);
    // tasks, part-major: [edge 0 .. 3, interior] x THICK_LINE_SEGS shares, then the two end discs
    constexpr int kCutTasks = 5 * THICK_LINE_SEGS + 2;
    for (int t = tid; t < n_lines * kCutTasks; t += nth) {
        const int p = t / n_lines, i = t - p * n_lines;
        const int part = p < 5 * THICK_LINE_SEGS ? p / THICK_LINE_SEGS : p - 5 * THICK_LINE_SEGS + 5;
        const int seg = p < 5 * THICK_LINE_SEGS ? p % THICK_LINE_SEGS : 0;
#ifdef VLFM_CUT_SKIP   // diagnostic builds only (tools/phase_probe.py): leave parts out to see what each costs
        if ((VLFM_CUT_SKIP >> part) & 1) continue;
#endif
        const int4 ln = lines[i];
        thick_line2_clear(vis, W, ln.x + ox, ln.y + oy, ln.z, ln.w, part, seg, part < 5 ? THICK_LINE_SEGS : 1);
    }
    __syncthreads();
    VLFM_PHASE(0, 6);
    // ---- 6. external contours of what is left; keep the one nearest the agent (|pointPolygonTest|, <= 3 px)
    for (int i = tid; i < pad_words; i += nth) {
        const int ly = i / pw - 1, lw = i % pw - 1;
        p_img[i] = ((unsigned)ly < (unsigned)wn && (unsigned)lw < (unsigned)words) ? vis[ly * words + lw] : 0u;
        p_tr[i] = 0u; p_ng[i] = 0u;
    }
    __syncthreads();
    ContourSink sink;
    sink.pts = pts; sink.start = cstart; sink.len = clen; sink.cap_pts = sc.cap_pts; sink.cap_contours = sc.cap_contours;
    sink.n_pts = 0; sink.n_contours = 0; sink.overflow = 0;
    {
        Bits b{p_img + pw + 1, pw, wn, wn, 1};
        WalkTables T = fog_walk_tables(sc, P.env, wn, words);
        T.ljd0 = l_jd; T.ljd1 = l_jd + sc.lds_states;
        wg_scan_external_lds(b, p_tr + pw + 1, p_ng + pw + 1, 2, sink, T, l_jd, 8u * (unsigned)sc.lds_states, sh_wg);
    }
    // distance of the agent to every contour: one wavefront per contour, 16 at a time (the line list is free: it holds the
    // distances); then wavefront 0 picks.  OpenCV lists contours in reverse discovery order and the reference's loop keeps the
    // FIRST strict minimum: in discovery order that is the LAST contour with the minimal distance.
    double* d2s = reinterpret_cast<double*>(lines);
    const int n_vis = sink.overflow ? 0 : (sink.n_contours < sc.cap_pts ? sink.n_contours : sc.cap_pts);
    for (int c = wave; c < n_vis; c += nth >> 6) {
        double d2; int inside;
        wave_point_polygon(pts + cstart[c], clen[c], acx, acy, &d2, &inside);
        if (lane == 0) d2s[c] = d2;
    }
    __threadfence_block();
    __syncthreads();
    if (wave == 0) {
        int best = -1;
        double best_d2 = 0;
        for (int c = lane; c < n_vis; c += 64) {
            const double d2 = d2s[c];
            if (best < 0 || d2 <= best_d2) { best = c; best_d2 = d2; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const int ob = __shfl_xor(best, off, 64);
            const double od = __shfl_xor(best_d2, off, 64);
            if (ob >= 0 && (best < 0 || od < best_d2 || (od == best_d2 && ob > best))) { best = ob; best_d2 = od; }
        }
        if (lane == 0) { sh_i[5] = best; sh_i[6] = sink.overflow; sh_i[7] = (best >= 0 && sqrt(best_d2) > 3.0) ? 1 : 0; }
    }
    __threadfence_block();
    __syncthreads();
    if (sh_i[6]) { if (tid == 0) status[0] = 1; return; }
    const int best = sh_i[5];
    if (best < 0 || sh_i[7]) return;  // nothing visible / closest contour too far away
    VLFM_PHASE(0, 7);
    // ---- 7. drawContours(fog, [visible_area], 0, 1, -1): fill the chosen outline (component + enclosed holes)
    {
        LdsBitmap fb;
        fb.solid = fill; fb.parity = par; fb.rows = wn; fb.cols = wn; fb.words = words;  // window-local polygon
        const int n = clen[best];
        const int2* cp = pts + cstart[best];
        const int G = n * 8 <= nth ? 8 : (n * 4 <= nth ? 4 : (n * 2 <= nth ? 2 : 1));   // lanes per edge (raster_edge_shared)
        for (int t = tid; t < n * G; t += nth) {
            const int i = t / G;
            const int2 a = cp[i == 0 ? n - 1 : i - 1], b = cp[i];
            raster_edge_shared(fb, (long long)a.x << XY_SHIFT, a.y, (long long)b.x << XY_SHIFT, b.y, t - i * G, G);
        }
        __syncthreads();
        resolve_rows(fb, tid, nth);
        __syncthreads();
    }
    VLFM_PHASE(0, 8);
    // ---- 8. dilate 3x3 (obstacle_map.py:125), keep navigable cells (:127), OR into the explored plane (:126)
    unsigned* expl = mp.explored + eoff;
    for (int i = tid; i < plane_words; i += nth) {
        const int ly = i / words, lw = i - ly * words;
        unsigned acc = 0;
        for (int dy = -1; dy <= 1; dy++) {
            const int yy = ly + dy;
            if ((unsigned)yy >= (unsigned)wn) continue;
            unsigned c = fill[yy * words + lw], p = lw > 0 ? fill[yy * words + lw - 1] : 0u,
                     n = lw + 1 < words ? fill[yy * words + lw + 1] : 0u;
            if (lw == words - 1) c &= last_mask;
            if (lw + 1 == words - 1) n &= last_mask;
            acc |= hdilate(p, c, n, 1);
        }
        if (lw == words - 1) acc &= last_mask;
        acc &= nav[i];
        if (!acc) continue;
        // scatter the window word into (up to) two image words
        const int y = oy + ly;
        if ((unsigned)y >= (unsigned)S) continue;
        const int x0 = ox + lw * 32;
        const int wi = x0 >= 0 ? (x0 >> 5) : -((-x0 + 31) >> 5);
        const int sh = x0 - wi * 32;
        const unsigned lo = acc << sh, hi = sh ? (acc >> (32 - sh)) : 0u;
        if (lo && wi >= 0 && wi < mp.stride) atomicOr(&expl[(size_t)y * mp.stride + wi], lo & tail_mask(S, wi));
        if (hi && wi + 1 >= 0 && wi + 1 < mp.stride) atomicOr(&expl[(size_t)y * mp.stride + wi + 1], hi & tail_mask(S, wi + 1));
    }
    VLFM_PHASE(0, 9);
    if (tid == 0) {
        int* bb = sc.bbox + (size_t)P.env * 4;
        atomicMin(&bb[0], max(oy, 0)); atomicMax(&bb[1], min(oy + wn - 1, S - 1));
        atomicMin(&bb[2], max(ox, 0)); atomicMax(&bb[3], min(ox + wn - 1, S - 1));
    }
}

// ================================================================================================ explored selection
// obstacle_map.py:128-146: if the explored area has split into several external components keep the one that contains
// the agent (else the nearest), redrawn filled.  One workgroup per environment; wave 0 scans.
struct SelectScratch {
    unsigned* traced; unsigned* neg;       // [n_envs][S][stride] each
    unsigned* fill_solid; unsigned* fill_par;
    int2* pts; int* starts; int* lens; int* status;  // status [n_envs][4]: overflow, n contours, chosen, refilled
    int cap_pts, cap_contours;
    unsigned lds_bytes;                    // dynamic LDS available for the window copy
    unsigned* walk_jd; int* walk_pixbase;  // parallel border follower (border_parallel.h): [n_envs][2 * walk_states], [n_envs][cap_pts]
    int walk_states;
};

__global__ __launch_bounds__(1024) void explored_select_kernel(const FogParams* __restrict__ prm, MapPlanes mp,
                                                              SelectScratch sc, const int* __restrict__ bbox) {
    const FogParams& P = prm[blockIdx.x];
    if (P.n_poly <= 0) return;
    __shared__ int sh_i[8];
    const int S = mp.S, stride = mp.stride;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const size_t eoff = (size_t)P.env * S * stride;
    unsigned* expl = mp.explored + eoff;
    unsigned* traced = sc.traced + eoff;
    unsigned* neg = sc.neg + eoff;
    const int* bb = bbox + (size_t)P.env * 4;
    const int y_lo = max(bb[0] - 1, 0), y_hi = min(bb[1] + 1, S - 1);
    if (y_lo > y_hi) return;  // nothing explored yet
    VLFM_PHASE(1, 0);
    // The border walk is a single-lane pointer chase: run it on an LDS copy of the explored window (everything ever revealed
    // lies inside the persistent bounding box) whenever image + two label planes fit; the global planes are the fallback.
    extern __shared__ __attribute__((aligned(16))) unsigned lds_win[];
    const int w_lo = max(bb[2] - 1, 0) >> 5, w_hi = min(bb[3] + 1, S - 1) >> 5;
    const int wrows = y_hi - y_lo + 1, wwords = w_hi - w_lo + 1;
    const int pw = (wwords + 2) | 1, wn = (wrows + 2) * pw;  // padded: one zero row / word all around (Bits::padded), odd stride
    const bool in_lds = (size_t)3 * wn * sizeof(unsigned) <= sc.lds_bytes;
    unsigned* L_img = lds_win;
    unsigned* L_tr = lds_win + wn;
    unsigned* L_ng = lds_win + 2 * wn;
    if (in_lds) {
        for (int i = tid; i < wn; i += nth) {
            const int ly = i / pw - 1, lw = i % pw - 1;
            const bool real = (unsigned)ly < (unsigned)wrows && (unsigned)lw < (unsigned)wwords;
            L_img[i] = real ? expl[(size_t)(y_lo + ly) * stride + w_lo + lw] : 0u;
            L_tr[i] = 0u; L_ng[i] = 0u;
        }
    } else {
        for (int i = tid + y_lo * stride; i < (y_hi + 1) * stride; i += nth) { traced[i] = 0u; neg[i] = 0u; }
    }
    wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
    VLFM_PHASE(1, 1);
    int2* pts = sc.pts + (size_t)P.env * sc.cap_pts;
    int* cstart = sc.starts + (size_t)P.env * sc.cap_contours;
    int* clen = sc.lens + (size_t)P.env * sc.cap_contours;
    int* status = sc.status + (size_t)P.env * 4;
    __shared__ int sh_wg[WG_SH_INTS];
    ContourSink sink;
    sink.pts = pts; sink.start = cstart; sink.len = clen; sink.cap_pts = sc.cap_pts; sink.cap_contours = sc.cap_contours;
    sink.n_pts = 0; sink.n_contours = 0; sink.overflow = 0;
    if (in_lds) {   // the whole workgroup follows the borders (border_parallel.h); tables in the global planes this path leaves free
        Bits b{L_img + pw + 1, pw, wrows, wwords * 32, 1};
        WalkTables T;
        T.bmask = traced; T.wprefix = reinterpret_cast<int*>(neg);
        T.next = reinterpret_cast<int*>(sc.fill_solid + eoff); T.sinfo = sc.fill_par + eoff;
        T.jd0 = sc.walk_jd + (size_t)P.env * 2 * sc.walk_states; T.jd1 = T.jd0 + sc.walk_states;
        T.pixbase = sc.walk_pixbase + (size_t)P.env * sc.cap_pts;
        T.cap_bp = sc.cap_pts; T.cap_states = min(sc.walk_states, S * stride);
        T.wrows = wrows; T.wwords = wwords;
        wg_scan_external_lds(b, L_tr + pw + 1, L_ng + pw + 1, 2, sink, T, lds_win + 3 * wn, sc.lds_bytes - 12u * (unsigned)wn, sh_wg);
        VLFM_STAMP(1, 2);
        const int npt = sink.n_pts < sc.cap_pts ? sink.n_pts : sc.cap_pts;
        for (int i = tid; i < npt; i += nth) { int2 q = pts[i]; q.x += w_lo * 32; q.y += y_lo; pts[i] = q; }  // -> image coords
        wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
    }
    if (wave == 0) {
        if (!in_lds) {
            Bits b{expl, stride, S, S};
            scan_external(b, traced, neg, y_lo, y_hi, 2, sink);
        }
        int chosen = -1;
        if (!sink.overflow && sink.n_contours > 1) {
            // OpenCV order = reverse discovery.  First contour with dist >= 0 wins outright; otherwise the first strict
            // minimum of |dist| (best_idx starts at 0 = the last discovered contour).
            double min_d2 = 1e300;
            chosen = sink.n_contours - 1;
            for (int c = sink.n_contours - 1; c >= 0; c--) {
                double d2; int inside;
                wave_point_polygon(pts + cstart[c], clen[c], P.ax, P.ay, &d2, &inside);
                if (inside || d2 == 0.0) { chosen = c; break; }   // dist >= 0: inside or on the outline
                if (d2 < min_d2) { min_d2 = d2; chosen = c; }
            }
        }
        if (lane == 0) { sh_i[0] = sink.n_contours; sh_i[1] = sink.overflow; sh_i[2] = chosen; }
        VLFM_STAMP(1, 3);
    }
    wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
    if (tid == 0) { status[0] = sh_i[1]; status[1] = sh_i[0]; status[2] = sh_i[2]; status[3] = 0; }
    VLFM_PHASE(1, 4);
    if (sh_i[1] || sh_i[0] <= 1) return;
    // redraw the chosen outline filled on an empty plane (global-memory bitmaps; rare path)
    const int chosen = sh_i[2];
    unsigned* fs = sc.fill_solid + eoff;
    unsigned* fp = sc.fill_par + eoff;
    for (int i = tid; i < S * stride; i += nth) { fs[i] = 0u; fp[i] = 0u; }
    wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
    LdsBitmap fb;
    fb.solid = fs; fb.parity = fp; fb.rows = S; fb.cols = S; fb.words = stride;
    const int n = clen[chosen];
    const int2* cp = pts + cstart[chosen];
    for (int i = tid; i < n; i += nth) {
        const int2 a = cp[i == 0 ? n - 1 : i - 1], b = cp[i];
        raster_edge(fb, (long long)a.x << XY_SHIFT, a.y, (long long)b.x << XY_SHIFT, b.y);
    }
    wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
    resolve_rows(fb, tid, nth);
    wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
    for (int i = tid; i < S * stride; i += nth) expl[i] = fs[i] & tail_mask(S, i % stride);
    if (tid == 0) status[3] = 1;
}

// ================================================================================================ frontiers
// obstacle_map.py:155-169 + frontier_exploration.detect_frontier_waypoints.
struct FrontierScratch {
    unsigned* explored_d;                  // [n_envs][S][stride]  dilate5(explored) & navigable, then "filtered"
    unsigned* unexplored;                  // [n_envs][S][stride]
    unsigned* traced; unsigned* neg;       // [n_envs][S][stride]
    unsigned* fill_solid; unsigned* fill_par;
    int2* pts; int* starts; int* lens;     // chain points
    unsigned char* bad;                    // [n_envs][cap_pts]
    int* pieces;                           // [n_envs][cap_contours * 6]: (chain base, chain n, s1, l1, s2, l2)
    double* out_xy;                        // [n_envs][cap_frontiers][2]
    int* out_n;                            // [n_envs][4]: n frontiers, overflow, n contours, n chain points
    int cap_pts, cap_contours, cap_frontiers;
    double area_thresh;
    unsigned lds_bytes;  // dynamic LDS available for the window copy of the border walk
    int* fog_status;        // [n_envs][4] of fog_of_war_kernel: word 0 = contour scratch overflow (sticky; consumed here)
    const int* sel_status;  // [n_envs][4] of explored_select_kernel: word 0 = contour scratch overflow
    unsigned* walk_jd; int* walk_pixbase;  // parallel border follower, as in SelectScratch
    int walk_states;
    int* derived_dirty;     // [n_envs]: explored_d holds this step's pocket fills -> frontier_prepare_kernel rebuilds the plane
};

__device__ inline unsigned ring_all_set(const unsigned* plane, int S, int stride, int tid, int nth) {
    // every pixel of the image border ring set?  returns 0/1 partial (thread-local AND)
    unsigned ok = 1;
    for (int i = tid; i < S; i += nth) {
        ok &= bit_get(plane, stride, S, S, i, 0) & bit_get(plane, stride, S, S, i, S - 1) &
              bit_get(plane, stride, S, S, 0, i) & bit_get(plane, stride, S, S, S - 1, i);
    }
    return ok;
}

__device__ inline int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i < 0 ? 0 : i;
}

// obstacle_map.py:159-163 + the first lines of detect_frontier_waypoints: explored_d = dilate(explored, 5x5) & navigable,
// unexplored = navigable & ~explored_d, for every listed environment, one thread per 32-cell word.
// Both planes are pure functions of (explored, navigable) with a 5 x 5 footprint, and they persist per environment: they
// only have to be refreshed where an input may have changed since the last explore step -- win[8..11] (see navigable_kernel).
// Exception: last step's filter_out_small_unexplored marked a pocket explored IN `explored_d` (frontier_kernel, the rare
// non-shortcut path; flagged in `derived_dirty`): then the whole plane is rebuilt (by the same window-sized grid, striding).
__global__ __launch_bounds__(256) void frontier_prepare_kernel(const FogParams* __restrict__ prm, MapPlanes mp,
                                                               unsigned* __restrict__ explored_d,
                                                               unsigned* __restrict__ unexplored,
                                                               const int* __restrict__ windows,
                                                               const int* __restrict__ derived_dirty) {
    const FogParams& P = prm[blockIdx.z];
    if (P.n_poly <= 0) return;
    const int S = mp.S, stride = mp.stride;
    const bool full = !windows || derived_dirty[P.env] != 0;
    const DirtyWin U = load_win(full ? nullptr : windows + 12 * blockIdx.z + 8, S);
    if (win_empty(U)) return;
    const int wx0 = U.x0 >> 5, ww = (U.x1 >> 5) - wx0 + 1, n_items = (U.y1 - U.y0 + 1) * ww;
    const size_t eoff = (size_t)P.env * S * stride;
    const unsigned* expl = mp.explored + eoff;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x) {
        const int y = U.y0 + i / ww, wi = wx0 + i % ww;
        const int idx = y * stride + wi;
        unsigned acc = 0;
        for (int dy = -2; dy <= 2; dy++)
            acc |= hdilate(row_word(expl, stride, S, y + dy, wi - 1), row_word(expl, stride, S, y + dy, wi),
                           row_word(expl, stride, S, y + dy, wi + 1), 2);
        const unsigned nv = mp.navigable[eoff + idx];
        acc &= nv & tail_mask(S, wi);
        explored_d[eoff + idx] = acc;
        unexplored[eoff + idx] = nv & ~acc;
    }
}

__global__ __launch_bounds__(1024) void frontier_kernel(const FogParams* __restrict__ prm, MapPlanes mp, FrontierScratch sc,
                                                       const int* __restrict__ bbox) {
    const FogParams& P = prm[blockIdx.x];
    if (P.n_poly <= 0) return;
    __shared__ int sh_i[16];
    const int S = mp.S, stride = mp.stride;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const size_t eoff = (size_t)P.env * S * stride;
    const unsigned* expl = mp.explored + eoff;
    const unsigned* navp = mp.navigable + eoff;
    unsigned* ed = sc.explored_d + eoff;
    unsigned* un = sc.unexplored + eoff;
    unsigned* traced = sc.traced + eoff;
    unsigned* neg = sc.neg + eoff;
    int2* pts = sc.pts + (size_t)P.env * sc.cap_pts;
    int* cstart = sc.starts + (size_t)P.env * sc.cap_contours;
    int* clen = sc.lens + (size_t)P.env * sc.cap_contours;
    unsigned char* bad = sc.bad + (size_t)P.env * sc.cap_pts;
    int* pieces = sc.pieces + (size_t)P.env * sc.cap_contours * 6;
    double* out_xy = sc.out_xy + (size_t)P.env * sc.cap_frontiers * 2;
    int* out_n = sc.out_n + (size_t)P.env * 4;
    if (tid < 16) sh_i[tid] = 0;
    if (tid == 0) sh_i[0] = 1;
    __syncthreads();
    VLFM_PHASE(2, 0);
    // ---- a. explored_d = dilate(explored, 5x5) & navigable ; unexplored = navigable & ~explored_d: full-plane, embarrassingly
    // parallel -> done for all environments at once by frontier_prepare_kernel (a single workgroup took ~125 us for it)
    unsigned ring_ok = 1;
    VLFM_PHASE(2, 1);
    // ---- b. filter_out_small_unexplored.  Exact shortcut: when the border ring of `unexplored` is fully set, that one
    // component encloses every other one, RETR_EXTERNAL returns it alone and its contour area is (S-1)^2.
    ring_ok = ring_all_set(un, S, stride, tid, nth);
    if (!ring_ok) atomicAnd(&sh_i[0], 0);
    __syncthreads();
    const bool shortcut = sh_i[0] && ((double)(S - 1) * (double)(S - 1) >= sc.area_thresh);
    if (tid == 0) sc.derived_dirty[P.env] = shortcut ? 0 : 1;   // the other branch may write pocket fills into explored_d
    if (!shortcut) {
        for (int i = tid; i < S * stride; i += nth) { traced[i] = 0u; neg[i] = 0u; }
        wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
        if (wave == 0) {
            ContourSink sink;
            sink.pts = pts; sink.start = cstart; sink.len = clen; sink.cap_pts = sc.cap_pts; sink.cap_contours = sc.cap_contours;
            sink.n_pts = 0; sink.n_contours = 0; sink.overflow = 0;
            Bits b{un, stride, S, S};
            scan_external(b, traced, neg, 0, S - 1, 2, sink);
            if (lane == 0) { sh_i[1] = sink.n_contours; sh_i[2] = sink.overflow; }
        }
        wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
        if (sh_i[2]) { if (tid == 0) { out_n[0] = 0; out_n[1] = 1; } return; }
        const int nc = sh_i[1];
        unsigned* fs = sc.fill_solid + eoff;
        unsigned* fp = sc.fill_par + eoff;
        for (int c = 0; c < nc; c++) {
            const int n = clen[c];
            const int2* cp = pts + cstart[c];
            // cv::contourArea (exact: integer cross products accumulate without rounding in a double)
            if (tid == 0) { sh_i[3] = 0; sh_i[4] = 0; }
            __syncthreads();
            long long part = 0;
            for (int i = tid; i < n; i += nth) {
                const int2 a = cp[i == 0 ? n - 1 : i - 1], b = cp[i];
                part += (long long)a.x * b.y - (long long)a.y * b.x;
            }
            // 64-bit sum through two 32-bit shared atomics is awkward; use a global-memory-free tree: wave shuffle + LDS
            for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
            __shared__ long long wsum[16];
            if (lane == 0) wsum[wave] = part;
            __syncthreads();
            long long twice = 0;
            for (int w = 0; w < (nth + 63) / 64; w++) twice += wsum[w];
            const double area = fabs((double)twice * 0.5);
            if (!(area < sc.area_thresh)) { __syncthreads(); continue; }
            // mask = filled outline; keep it only if every covered cell is unexplored-navigable
            for (int i = tid; i < S * stride; i += nth) { fs[i] = 0u; fp[i] = 0u; }
            wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
            LdsBitmap fb;
            fb.solid = fs; fb.parity = fp; fb.rows = S; fb.cols = S; fb.words = stride;
            for (int i = tid; i < n; i += nth) {
                const int2 a = cp[i == 0 ? n - 1 : i - 1], b = cp[i];
                raster_edge(fb, (long long)a.x << XY_SHIFT, a.y, (long long)b.x << XY_SHIFT, b.y);
            }
            wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
            resolve_rows(fb, tid, nth);
            wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
            for (int i = tid; i < S * stride; i += nth) {
                const unsigned m = fs[i] & tail_mask(S, i % stride);
                if (m & ~un[i]) atomicOr(&sh_i[3], 1);  // covers something that is not unexplored
                if (m) atomicOr(&sh_i[4], 1);
            }
            __syncthreads();
            if (sh_i[4] && !sh_i[3])
                for (int i = tid; i < S * stride; i += nth) ed[i] |= fs[i] & tail_mask(S, i % stride);
            wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
        }
        for (int i = tid; i < S * stride; i += nth) { traced[i] = 0u; neg[i] = 0u; }
        wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
    }
    VLFM_PHASE(2, 2);
    // ---- c. border chain (CHAIN_APPROX_NONE) of the filtered explored mask
    const int* bb = bbox + (size_t)P.env * 4;
    {
        // single-lane pointer chase again: walk an LDS copy of the window when it fits (see explored_select_kernel).  In the
        // shortcut case the filtered mask is dilate(explored, 5x5) & navigable, i.e. inside the bounding box grown by 2.
        extern __shared__ __attribute__((aligned(16))) unsigned lds_win[];
        const int y_lo = shortcut ? max(bb[0] - 3, 0) : 0, y_hi = shortcut ? min(bb[1] + 3, S - 1) : S - 1;
        const int w_lo = shortcut ? (max(bb[2] - 3, 0) >> 5) : 0, w_hi = shortcut ? (min(bb[3] + 3, S - 1) >> 5) : stride - 1;
        const int wrows = y_hi - y_lo + 1, wwords = w_hi - w_lo + 1;
        const int pw = (wwords + 2) | 1, wn = (wrows + 2) * pw;  // padded (Bits::padded), odd stride
        const bool in_lds = wrows > 0 && (size_t)3 * wn * sizeof(unsigned) <= sc.lds_bytes;
        unsigned* L_img = lds_win;
        unsigned* L_tr = lds_win + wn;
        unsigned* L_ng = lds_win + 2 * wn;
        if (in_lds) {
            for (int i = tid; i < wn; i += nth) {
                const int ly = i / pw - 1, lw = i % pw - 1;
                const bool real = (unsigned)ly < (unsigned)wrows && (unsigned)lw < (unsigned)wwords;
                L_img[i] = real ? ed[(size_t)(y_lo + ly) * stride + w_lo + lw] : 0u;
                L_tr[i] = 0u; L_ng[i] = 0u;
            }
        } else if (shortcut) {  // (the other branch zeroed the whole label planes after its own scan)
            for (int i = tid + y_lo * stride; i < (y_hi + 1) * stride; i += nth) { traced[i] = 0u; neg[i] = 0u; }
        }
        wg_sync_global();
        VLFM_PHASE(2, 3);
        ContourSink sink;
        sink.pts = pts; sink.start = cstart; sink.len = clen; sink.cap_pts = sc.cap_pts; sink.cap_contours = sc.cap_contours;
        sink.n_pts = 0; sink.n_contours = 0; sink.overflow = 0;
        if (in_lds) {   // the whole workgroup follows the border (border_parallel.h)
            __shared__ int sh_wg[WG_SH_INTS];
            Bits b{L_img + pw + 1, pw, wrows, wwords * 32, 1};
            WalkTables T;
            T.bmask = traced; T.wprefix = reinterpret_cast<int*>(neg);
            T.next = reinterpret_cast<int*>(sc.fill_solid + eoff); T.sinfo = sc.fill_par + eoff;
            T.jd0 = sc.walk_jd + (size_t)P.env * 2 * sc.walk_states; T.jd1 = T.jd0 + sc.walk_states;
            T.pixbase = sc.walk_pixbase + (size_t)P.env * sc.cap_pts;
            T.cap_bp = sc.cap_pts; T.cap_states = min(sc.walk_states, S * stride);
            T.wrows = wrows; T.wwords = wwords;
            wg_scan_external_lds(b, L_tr + pw + 1, L_ng + pw + 1, 1, sink, T, lds_win + 3 * wn, sc.lds_bytes - 12u * (unsigned)wn,
                                 sh_wg);
            VLFM_STAMP(2, 4);
            const int npt = sink.n_pts < sc.cap_pts ? sink.n_pts : sc.cap_pts;
            for (int i = tid; i < npt; i += nth) { int2 q = pts[i]; q.x += w_lo * 32; q.y += y_lo; pts[i] = q; }
            if (tid == 0) { sh_i[5] = sink.n_contours; sh_i[6] = sink.n_pts; sh_i[7] = sink.overflow; }
        } else if (wave == 0) {
            Bits b{ed, stride, S, S};
            scan_external(b, traced, neg, y_lo, y_hi, 1, sink);
            if (lane == 0) { sh_i[5] = sink.n_contours; sh_i[6] = sink.n_pts; sh_i[7] = sink.overflow; }
        }
    }
    wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
    const int nc = sh_i[5], npts_all = sh_i[6];
    if (sh_i[7]) { if (tid == 0) { out_n[0] = 0; out_n[1] = 1; } return; }
    VLFM_PHASE(2, 5);
    // ---- d. a chain point is "bad" when no unexplored-navigable cell lies in its 3x3 neighbourhood
    //         (cv2.blur 3x3, BORDER_REFLECT_101, of 255*(navigable & ~filtered) is zero there)
    // the flags are consumed by a single lane below: keep them in LDS (the walk window is free again) when they fit
    {
        extern __shared__ __attribute__((aligned(16))) unsigned lds_win[];
        if ((unsigned)npts_all <= sc.lds_bytes) bad = reinterpret_cast<unsigned char*>(lds_win);
    }
    for (int i = tid; i < npts_all; i += nth) {
        const int2 p = pts[i];
        unsigned any = 0;
        if (p.x >= 1 && p.x + 1 < S && p.y >= 1 && p.y + 1 < S) {
            // interior point (all but the map's rim): the 3 x 3 cells are bits x-1 .. x+1 of three rows -- six 64-bit windows
            // requested together (bit_get per cell was 18 dependent global loads)
            const int px = p.x - 1, wi = px >> 5, sh = px & 31;
            unsigned long long nv[3], ev[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const size_t o = (size_t)(p.y - 1 + r) * stride + wi;
                const bool two = wi + 1 < stride;
                nv[r] = (unsigned long long)navp[o] | (two ? (unsigned long long)navp[o + 1] << 32 : 0ull);
                ev[r] = (unsigned long long)ed[o] | (two ? (unsigned long long)ed[o + 1] << 32 : 0ull);
            }
#pragma unroll
            for (int r = 0; r < 3; r++) any |= (unsigned)((nv[r] & ~ev[r]) >> sh) & 7u;
        } else {
            for (int dy = -1; dy <= 1; dy++) {
                const int yy = reflect101(p.y + dy, S);
                for (int dx = -1; dx <= 1; dx++) {
                    const int xx = reflect101(p.x + dx, S);
                    any |= bit_get(navp, stride, S, S, xx, yy) & (1u ^ bit_get(ed, stride, S, S, xx, yy));
                }
            }
        }
        bad[i] = any ? 0 : 1;
    }
    wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
    VLFM_PHASE(2, 6);
    // ---- e. frontier runs + arc-length midpoints.  Contours in OpenCV order (reverse discovery); the chain handed to
    // contour_to_frontiers is the contour rotated by one (interpolate_contour emits end points only).
    // Wavefront 0 lists the kept pieces, 64 chain positions at a time (a serial pass over a 2 300-point chain took 370 us);
    // then the midpoints are computed piece by piece.
    if (wave == 0) {
        // np.split(chain, bad) -> pieces [0,b0) [b0,b1) ... [b_last,n); a piece is kept when it has more than two
        // points (its leading bad point is then dropped) -- or is the head piece of a chain whose ends both lie on a
        // frontier (front_last_split), in which case the LAST kept piece is finally glued in front of it.
        // Boundary k (the k-th bad position, and finally j == n) closes piece k = [previous boundary or 0, boundary): everything
        // a piece needs is the previous boundary and its ordinal, i.e. a ballot and two popcounts per 64 positions.
        int np = 0, overflow = 0;   // uniform
        const unsigned long long below = (1ull << lane) - 1ull;
        for (int c = nc - 1; c >= 0; c--) {
            const int n = clen[c], base = cstart[c];
            if (n < 2) continue;  // a single-pixel contour interpolates to nothing
            auto is_bad = [&](int j) { return bad[base + (j + 1 == n ? 0 : j + 1)] != 0; };  // chain rotated by one
            int nbad = 0, first_bad = -1, last_bad = -1;
            for (int j0 = 0; j0 < n; j0 += 64) {
                const int j = j0 + lane;
                const unsigned long long m = __ballot(j < n && is_bad(j));
                if (m) {
                    if (first_bad < 0) first_bad = j0 + __builtin_ctzll(m);
                    last_bad = j0 + 63 - __builtin_clzll(m);
                    nbad += __builtin_popcountll(m);
                }
            }
            const bool fls = nbad > 0 && first_bad != 0 && last_bad < n - 2;
            const int np_base = np;
            int prev_boundary = 0, ordinal = 0;   // carried across the 64-position chunks (uniform)
            for (int j0 = 0; j0 <= n; j0 += 64) {
                const int j = j0 + lane;
                const bool boundary = j == n || (j < n && is_bad(j));
                const unsigned long long m = __ballot(boundary);
                if (!m) continue;
                const unsigned long long lower = m & below;
                const int start = lower ? j0 + 63 - __builtin_clzll(lower) : prev_boundary;
                const int idx = ordinal + __builtin_popcountll(lower);
                const int len = j - start;
                const bool keep = boundary && (len > 2 || (idx == 0 && fls));
                const unsigned long long km = __ballot(keep);
                const int slot = np + __builtin_popcountll(km & below);
                if (keep) {
                    if (slot < sc.cap_contours) {
                        int* pc = pieces + 6 * slot;
                        pc[0] = base; pc[1] = n;
                        pc[2] = idx == 0 ? start : start + 1; pc[3] = idx == 0 ? len : len - 1;
                        pc[4] = 0; pc[5] = 0;
                    }
                }
                np += __builtin_popcountll(km);
                if (np > sc.cap_contours) { np = sc.cap_contours; overflow = 1; }
                prev_boundary = j0 + 63 - __builtin_clzll(m);
                ordinal += __builtin_popcountll(m);
            }
            const int kept = np - np_base;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (kept > 1 && fls) {  // kept[0] = concat(kept.pop(), kept[0])
                if (lane == 0) {
                    int* head = pieces + 6 * np_base;
                    int* tail = pieces + 6 * (np - 1);
                    const int hs = head[2], hl = head[3];
                    head[2] = tail[2]; head[3] = tail[3]; head[4] = hs; head[5] = hl;
                }
                np--;
            }
        }
        if (lane == 0) { sh_i[8] = np; sh_i[9] = overflow; }
    }
    wg_sync_global();   // (workgroup scope: the consumers are this workgroup's own wavefronts)
    VLFM_PHASE(2, 7);
    const int np = sh_i[8];
    // get_frontier_midpoint per piece.  The arc-length cumsum is sequential by definition (np.cumsum's rounding order), but
    // the segment lengths are not: all lanes compute them (f64 sqrt, chain indexing) into LDS, then one lane only adds.
    {
        extern __shared__ __attribute__((aligned(16))) unsigned lds_win[];
        const size_t seg_off = ((size_t)npts_all + 7) & ~(size_t)7;   // after the bad flags (when those live in LDS)
        const bool bad_in_lds = (unsigned)npts_all <= sc.lds_bytes;
        double* seg = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(lds_win) + (bad_in_lds ? seg_off : 0));
        const size_t seg_cap = (sc.lds_bytes - (bad_in_lds ? seg_off : 0)) / sizeof(double);
        // chain point j of piece f / length of its segment k (f64, the reference's expression)
        auto q = [&](const int* pc, int j) {
            const int base = pc[0], n = pc[1], s1 = pc[2], l1 = pc[3], s2 = pc[4];
            const int k = j < l1 ? s1 + j : s2 + (j - l1);
            int idx = k + 1;
            if (idx >= n) idx -= n;
            if (idx >= n) idx %= n;
            return pts[base + idx];
        };
        auto seglen = [&](const int* pc, int k) {
            const int2 a = q(pc, k), b = q(pc, k + 1);
            const double ddx = (double)(a.x - b.x), ddy = (double)(a.y - b.y);
            return sqrt(__dadd_rn(__dmul_rn(ddx, ddx), __dmul_rn(ddy, ddy)));
        };
        // all segment lengths of all pieces at once (piece f owns seg[off[f] .. off[f] + m_f - 1)), then one lane PER PIECE adds
        // them up -- 16 pieces at a time instead of one (17 pieces took 70 us, two barriers each)
        __shared__ int sh_off[257];
        __shared__ int sh_scan[32];
        const int npc = np < 256 ? np : 256;
        {   // segment offsets of the pieces: one piece per lane + a workgroup scan (one lane reading every piece's lengths from
            // global memory in turn was most of this phase: 20 us for 17 pieces)
            int mine = 0;
            if (tid < npc) { const int m = pieces[6 * tid + 3] + pieces[6 * tid + 5]; mine = m >= 2 ? m - 1 : 0; }
            int ex, ex2, tot, tot2;
            wg_scan2(mine, 0, sh_scan, ex, ex2, tot, tot2);
            if (tid < npc) sh_off[tid] = ex;
            if (tid == 0) sh_off[npc] = tot;
        }
        __syncthreads();
        const int total = sh_off[npc];
        const bool staged = (size_t)total <= seg_cap && np <= 256;
        if (staged) {
            for (int e = tid; e < total; e += nth) {
                int lo = 0, hi = npc - 1;            // last piece whose offset is <= e and that owns segments
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sh_off[mid] <= e) lo = mid; else hi = mid - 1; }
                while (sh_off[lo + 1] == sh_off[lo]) lo++;   // (pieces without segments share their successor's offset)
                seg[e] = seglen(pieces + 6 * lo, e - sh_off[lo]);
            }
        }
        __syncthreads();
        // one lane per piece (pieces dealt round-robin to the wavefronts, so that up to 16 x 64 of them run at once); the sums are
        // sequential (np.cumsum's rounding order) but the LDS reads are not: eight segment lengths are requested at a time
        const int nwaves = (nth + 63) >> 6;
        for (int f0 = 0; f0 < np; f0 += staged ? nth : 1) {
            const int f = staged ? f0 + (lane * nwaves + wave) : f0;
            if (staged ? f >= np : tid != 0) continue;
            const int* pc = pieces + 6 * f;
            const int m = pc[3] + pc[5];
            const double* sg = seg + (staged ? sh_off[f] : 0);
            double ox_ = 0, oy_ = 0;
            if (m < 2) {
                if (m == 1) { const int2 p2 = q(pc, 0); ox_ = p2.x; oy_ = p2.y; }
            } else {
                double totlen = 0;
                if (staged) {
                    for (int k0 = 0; k0 + 1 < m; k0 += 8) {
                        double v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) v[u] = k0 + u + 1 < m ? sg[k0 + u] : 0.0;
#pragma unroll
                        for (int u = 0; u < 8; u++) if (k0 + u + 1 < m) totlen = __dadd_rn(totlen, v[u]);
                    }
                } else {
                    for (int k = 0; k + 1 < m; k++) totlen = __dadd_rn(totlen, seglen(pc, k));
                }
                const double half = totlen / 2;
                double cum = 0, upto = 0, sl = 0;
                int idx = 0;
                bool found = false;
                if (staged) {
                    for (int k0 = 0; k0 + 1 < m && !found; k0 += 8) {
                        double v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) v[u] = k0 + u + 1 < m ? sg[k0 + u] : 0.0;
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            if (!found && k0 + u + 1 < m) {
                                const double c2 = __dadd_rn(cum, v[u]);
                                if (c2 > half) { idx = k0 + u; upto = idx > 0 ? cum : 0; sl = v[u]; found = true; }
                                else cum = c2;
                            }
                        }
                    }
                } else {
                    for (int k = 0; k + 1 < m; k++) {
                        const double l = seglen(pc, k);
                        const double c2 = __dadd_rn(cum, l);
                        if (c2 > half) { idx = k; upto = k > 0 ? cum : 0; sl = l; found = true; break; }
                        cum = c2;
                    }
                }
                if (!found) { idx = 0; upto = 0; sl = staged ? sg[0] : seglen(pc, 0); }
                const double prop = __ddiv_rn(__dsub_rn(half, upto), sl);
                const int2 a = q(pc, idx), b = q(pc, idx + 1);
                ox_ = __dadd_rn((double)a.x, __dmul_rn(prop, (double)(b.x - a.x)));
                oy_ = __dadd_rn((double)a.y, __dmul_rn(prop, (double)(b.y - a.y)));
            }
            if (f < sc.cap_frontiers) { out_xy[2 * f] = ox_; out_xy[2 * f + 1] = oy_; }
        }
        __syncthreads();
    }
    VLFM_PHASE(2, 8);
    if (tid == 0) {
        out_n[0] = np < sc.cap_frontiers ? np : sc.cap_frontiers;
        // one flag for the whole explore pipeline of this environment: a capacity overflow in the fog-of-war or the
        // component-selection kernel leaves the explored area short of the reference's, so it must surface on the same
        // read-back path as this kernel's own overflow (ObstacleMapBatch._read_frontiers raises)
        int* fog_st = sc.fog_status + (size_t)P.env * 4;
        const int upstream = fog_st[0] | sc.sel_status[(size_t)P.env * 4];
        fog_st[0] = 0;
        out_n[1] = sh_i[9] || np > sc.cap_frontiers || upstream != 0;
        out_n[2] = nc; out_n[3] = npts_all;
    }
}

}  // namespace vlfm

using namespace vlfm;

extern "C" int vlfm_bits_pack(const uint8_t* d_src, uint32_t* d_dst, int planes, int rows, int cols, void* stream) {
    if (!d_src || !d_dst || planes <= 0 || rows <= 0 || cols <= 0) return fail(VLFM_ERR_INVALID, "bits_pack: bad argument");
    const int stride = (cols + 31) / 32;
    VLFM_KLAUNCH(pack_u8_kernel, dim3((stride + 63) / 64, rows, planes), dim3(64), 0, (hipStream_t)stream, d_src,
                       d_dst, rows, cols, stride);
    return check_launch("pack_u8_kernel");
}

extern "C" int vlfm_bits_unpack(const uint32_t* d_src, uint8_t* d_dst, int planes, int rows, int cols, void* stream) {
    if (!d_src || !d_dst || planes <= 0 || rows <= 0 || cols <= 0) return fail(VLFM_ERR_INVALID, "bits_unpack: bad argument");
    const int stride = (cols + 31) / 32;
    VLFM_KLAUNCH(unpack_u8_kernel, dim3((cols + 255) / 256, rows, planes), dim3(256), 0, (hipStream_t)stream,
                       d_src, d_dst, rows, cols, stride);
    return check_launch("unpack_u8_kernel");
}

extern "C" int vlfm_bits_dilate(const uint32_t* d_src, uint32_t* d_dst, int planes, int rows, int cols, int kernel_w,
                                int kernel_h, void* stream) {
    if (!d_src || !d_dst || planes <= 0 || rows <= 0 || cols <= 0 || kernel_w < 1 || kernel_h < 1 || !(kernel_w & 1) ||
        !(kernel_h & 1) || kernel_w > 63)
        return fail(VLFM_ERR_INVALID, "bits_dilate: bad argument (odd kernel sizes, width <= 63)");
    const int stride = (cols + 31) / 32;
    VLFM_TIMED("dilate_bits_kernel", stream);
    VLFM_KLAUNCH(dilate_bits_kernel, dim3((rows * stride + 255) / 256, 1, planes), dim3(256), 0,
                       (hipStream_t)stream, d_src, d_dst, (unsigned*)nullptr, (const int*)nullptr, rows, cols, stride,
                       kernel_w / 2, kernel_h / 2, 0);
    return check_launch("dilate_bits_kernel");
}

extern "C" int vlfm_find_contours_external(const uint32_t* d_img, int planes, int rows, int cols, int method,
                                           uint32_t* d_scratch /* [planes][2][rows][stride] */, int32_t* d_pts,
                                           int cap_pts, int32_t* d_starts, int32_t* d_lens, int cap_contours,
                                           int32_t* d_counts, void* stream) {
    if (!d_img || !d_scratch || !d_pts || !d_starts || !d_lens || !d_counts || planes <= 0 || rows <= 0 || cols <= 0 ||
        (method != 1 && method != 2) || cols > 2048)
        return fail(VLFM_ERR_INVALID, "find_contours_external: bad argument (method 1|2, cols <= 2048)");
    const int stride = (cols + 31) / 32;
    unsigned* traced = d_scratch;
    unsigned* neg = d_scratch + (size_t)planes * rows * stride;
    VLFM_TIMED("find_contours_kernel", stream);
    VLFM_KLAUNCH(find_contours_kernel, dim3(planes), dim3(64), 0, (hipStream_t)stream, d_img, traced, neg, rows,
                       cols, stride, method, reinterpret_cast<int2*>(d_pts), cap_pts, d_starts, d_lens, cap_contours,
                       d_counts);
    return check_launch("find_contours_kernel");
}

// dynamic LDS of the border walks of explored_select / frontier: image + two label planes of a window up to ~640 x 640 cells, and
// behind them the tables of the workgroup-parallel follower (border_parallel.h).  The chip has 160 KB per CU and these kernels run
// one workgroup per environment, so occupancy is irrelevant; 2 KB stay free for the kernels' static shared variables.
constexpr unsigned kWalkLdsBytes = 158 * 1024;

extern "C" size_t vlfm_find_contours_wg_scratch_bytes(int planes, int rows, int cols, int cap_pts) {
    if (planes <= 0 || rows <= 0 || cols <= 0 || cap_pts <= 0) return 0;
    const size_t stride = (cols + 31) / 32;
    size_t plane_words = (size_t)rows * stride;
    if (plane_words < (size_t)2 * cap_pts) plane_words = (size_t)2 * cap_pts;   // room for the states (~5 per border pixel at most 8)
    if (plane_words > 65535) plane_words = 65535;
    if (plane_words < (size_t)rows * stride) return 0;                           // image too large for 16-bit state ids
    return (size_t)planes * (6 * plane_words + (size_t)cap_pts) * 4;
}

extern "C" int vlfm_find_contours_external_wg(const uint32_t* d_img, int planes, int rows, int cols, int method,
                                              void* d_scratch, size_t scratch_bytes, int32_t* d_pts, int cap_pts,
                                              int32_t* d_starts, int32_t* d_lens, int cap_contours, int32_t* d_counts,
                                              void* stream) {
    const int global_tables = (method & 0x100) != 0;   // test switch: the follower's tables in global memory (the fallback form)
    method &= 0xFF;
    if (!d_img || !d_scratch || !d_pts || !d_starts || !d_lens || !d_counts || planes <= 0 || rows <= 0 || cols <= 0 ||
        (method != 1 && method != 2) || cols > 2048)
        return fail(VLFM_ERR_INVALID, "find_contours_external_wg: bad argument (method 1|2, cols <= 2048)");
    const int stride = (cols + 31) / 32;
    const size_t lds = (size_t)3 * (rows + 2) * ((stride + 2) | 1) * 4;
    const size_t need = vlfm_find_contours_wg_scratch_bytes(planes, rows, cols, cap_pts);
    if (lds > 144 * 1024 || need == 0) return fail(VLFM_ERR_CAPACITY, "find_contours_external_wg: the image does not fit the LDS window");
    const size_t lds_launch = global_tables ? lds : (size_t)kWalkLdsBytes;
    if (scratch_bytes < need) return fail(VLFM_ERR_CAPACITY, "find_contours_external_wg: scratch too small");
    static LdsOptIn opt;
    if (!opt.ensure(reinterpret_cast<const void*>(find_contours_wg_kernel), kWalkLdsBytes))
        return fail(VLFM_ERR_HIP, "find_contours_external_wg: cannot opt in to 158 KB of LDS");
    size_t plane_words = (size_t)rows * stride;
    if (plane_words < (size_t)2 * cap_pts) plane_words = (size_t)2 * cap_pts;
    if (plane_words > 65535) plane_words = 65535;
    WgContourWork wk{(unsigned*)d_scratch, (int*)((unsigned*)d_scratch + (size_t)planes * 6 * plane_words), (int)plane_words,
                     (int)plane_words, !global_tables};
    VLFM_TIMED("find_contours_wg_kernel", stream);
    VLFM_KLAUNCH(find_contours_wg_kernel, dim3(planes), dim3(1024), lds_launch, (hipStream_t)stream, d_img, rows, cols, stride, method,
                 wk, (unsigned)lds_launch, reinterpret_cast<int2*>(d_pts), cap_pts, d_starts, d_lens, cap_contours, d_counts);
    return check_launch("find_contours_wg_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
namespace {

struct ScratchLayout {
    size_t plane_words, total;
    size_t off_planes[8], off_pts, off_lines, off_starts, off_lens, off_bad, off_pieces, off_status, off_pixbase;
};
ScratchLayout layout(int n_envs, int S, int cap_pts, int cap_contours) {
    ScratchLayout L;
    const size_t stride = (S + 31) / 32;
    L.plane_words = (size_t)n_envs * S * stride;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
    // planes 0-5: scratch shared by the stages of one step (label planes, fill planes, the fog kernel's follower tables);
    // planes 6-7: the frontier stage's derived planes explored_d / unexplored, which PERSIST between steps (they are only
    // refreshed inside the dirty windows, frontier_prepare_kernel) and therefore belong to nobody else
    for (int k = 0; k < 8; k++) L.off_planes[k] = take(L.plane_words * 4);
    L.off_pts = take((size_t)n_envs * cap_pts * sizeof(int2));
    L.off_lines = take((size_t)n_envs * cap_pts * sizeof(int4));
    L.off_starts = take((size_t)n_envs * cap_contours * 4);
    L.off_lens = take((size_t)n_envs * cap_contours * 4);
    L.off_bad = take((size_t)n_envs * cap_pts);
    L.off_pieces = take((size_t)n_envs * cap_contours * 6 * 4);
    L.off_status = take((size_t)n_envs * 16 * 4);
    L.off_pixbase = take((size_t)n_envs * cap_pts * 4);
    L.total = o;
    return L;
}
}  // namespace

extern "C" size_t vlfm_obstacle_scratch_bytes(int n_envs, int map_size, int cap_pts, int cap_contours) {
    if (n_envs <= 0 || map_size <= 0 || cap_pts <= 0 || cap_contours <= 0) return 0;
    return layout(n_envs, map_size, cap_pts, cap_contours).total;
}

extern "C" int vlfm_obstacle_map_update_batched(const vlfm_fog_params* d_prm, int n, const uint32_t* d_obstacle,
                                                uint32_t* d_navigable, uint32_t* d_explored, int32_t* d_bbox,
                                                int n_envs, int map_size, int kernel_size, int fog_radius,
                                                double area_thresh_px, void* d_scratch, size_t scratch_bytes,
                                                int cap_pts, int cap_contours, double* d_frontiers, int cap_frontiers,
                                                int32_t* d_counts, int update_obstacles, int explore,
                                                const int32_t* d_windows, int window_blocks_navigable,
                                                int window_blocks_prepare, void* stream) {
    if (n == 0) return VLFM_OK;
    if (!d_prm || !d_obstacle || !d_navigable || !d_explored || !d_bbox || !d_scratch || !d_frontiers || !d_counts ||
        n < 0 || map_size <= 0 || map_size > 2048 || kernel_size < 1 || !(kernel_size & 1) || kernel_size > 63 ||
        fog_radius <= 0)
        return fail(VLFM_ERR_INVALID, "obstacle_map_update_batched: bad argument");
    const ScratchLayout L = layout(n_envs, map_size, cap_pts, cap_contours);
    if (scratch_bytes < L.total) return fail(VLFM_ERR_CAPACITY, "obstacle_map_update_batched: scratch too small");
    hipStream_t s = (hipStream_t)stream;
    const int stride = (map_size + 31) / 32;
    unsigned char* base = (unsigned char*)d_scratch;
    MapPlanes mp{d_obstacle, d_navigable, d_explored, map_size, stride};
    unsigned* planes[8];
    for (int k = 0; k < 8; k++) planes[k] = (unsigned*)(base + L.off_planes[k]);
    int2* pts = (int2*)(base + L.off_pts);
    int4* lines = (int4*)(base + L.off_lines);
    int* starts = (int*)(base + L.off_starts);
    int* lens = (int*)(base + L.off_lens);
    int* status = (int*)(base + L.off_status);
    if (update_obstacles || explore) {
        VLFM_TIMED("navigable_kernel", s);
        const int full_blocks = (map_size * stride + 255) / 256;
        const int nb = (d_windows && window_blocks_navigable > 0 && window_blocks_navigable < full_blocks)
                           ? window_blocks_navigable : full_blocks;
        VLFM_KLAUNCH(navigable_kernel, dim3(nb, 1, n), dim3(256), 0, s, d_prm, mp, kernel_size / 2, update_obstacles, explore,
                     d_windows);
    }
    int rc = check_launch("navigable_kernel");
    if (rc != VLFM_OK || !explore) return rc;
    {
        FogScratch fs{pts, starts, lens, lines, status, d_bbox, cap_pts, cap_contours,
                      {planes[0], planes[1], planes[2], planes[3], planes[4], planes[5]}, (int*)(base + L.off_pixbase),
                      map_size * stride, 0};
        const int wn = 2 * fog_radius + 5, words = (wn + 31) / 32;
        size_t lds = (size_t)6 * wn * words * 4 + (size_t)3 * (wn + 2) * ((words + 2) | 1) * 4 + 64;
        if (lds > 160 * 1024) return fail(VLFM_ERR_CAPACITY, "obstacle_map_update_batched: fog window too large for LDS");
        // what is left of 152 KB holds the border follower's list-ranking buffers (two words per state; a window of 205 x 205
        // cells has a few thousand states): a ranking round is then an LDS round trip instead of an L2 one
        // (0 when the planes alone pass 152 KB: the follower then ranks in its global buffers, border_parallel.h)
        const size_t budget = (size_t)152 * 1024;
        fs.lds_states = lds < budget ? (int)((budget - lds) / 8 < 16384 ? (budget - lds) / 8 : 16384) : 0;
        lds += (size_t)fs.lds_states * 8;
        static LdsOptIn opt_fog;   // beyond the default dynamic-LDS limit of 64 KB: opt in (once per device and size)
        if (lds > 64 * 1024 && !opt_fog.ensure(reinterpret_cast<const void*>(fog_of_war_kernel), lds))
            return fail(VLFM_ERR_HIP, "obstacle_map_update_batched: cannot opt in to the fog-of-war kernel's LDS window");
        VLFM_TIMED("fog_of_war_kernel", s);
        VLFM_KLAUNCH(fog_of_war_kernel, dim3(n), dim3(1024), lds, s, d_prm, mp, fs);
    }
    rc = check_launch("fog_of_war_kernel");
    if (rc != VLFM_OK) return rc;
    {
        static LdsOptIn opt_select, opt_frontier;
        if (!opt_select.ensure(reinterpret_cast<const void*>(explored_select_kernel), kWalkLdsBytes) ||
            !opt_frontier.ensure(reinterpret_cast<const void*>(frontier_kernel), kWalkLdsBytes))
            return fail(VLFM_ERR_HIP, "obstacle_map_update_batched: cannot opt in to 158 KB of LDS");
        // the fog kernel's line list (cap_pts x int4 per environment) is free from here on: two pointer-jumping buffers of
        // 2 * cap_pts words for the parallel border follower
        SelectScratch ss{planes[0], planes[1], planes[2], planes[3], pts, starts, lens, status + (size_t)n_envs * 4,
                         cap_pts, cap_contours, kWalkLdsBytes, (unsigned*)lines, (int*)(base + L.off_pixbase),
                         2 * cap_pts > 65535 ? 65535 : 2 * cap_pts};
        VLFM_TIMED("explored_select_kernel", s);
        VLFM_KLAUNCH(explored_select_kernel, dim3(n), dim3(1024), kWalkLdsBytes, s, d_prm, mp, ss, (const int*)d_bbox);
    }
    rc = check_launch("explored_select_kernel");
    if (rc != VLFM_OK) return rc;
    {
        FrontierScratch fr{planes[6], planes[7], planes[0], planes[1], planes[2], planes[3], pts, starts, lens,
                           (unsigned char*)(base + L.off_bad), (int*)(base + L.off_pieces), d_frontiers, d_counts,
                           cap_pts, cap_contours, cap_frontiers, area_thresh_px, kWalkLdsBytes, status,
                           status + (size_t)n_envs * 4, (unsigned*)lines, (int*)(base + L.off_pixbase),
                           2 * cap_pts > 65535 ? 65535 : 2 * cap_pts, status + (size_t)n_envs * 8};
        {
            VLFM_TIMED("frontier_prepare_kernel", s);
            const int full_blocks = (map_size * stride + 255) / 256;
            const int nb = (d_windows && window_blocks_prepare > 0 && window_blocks_prepare < full_blocks)
                               ? window_blocks_prepare : full_blocks;
            VLFM_KLAUNCH(frontier_prepare_kernel, dim3(nb, 1, n), dim3(256), 0, s, d_prm, mp, planes[6], planes[7], d_windows,
                         (const int*)(status + (size_t)n_envs * 8));
        }
        VLFM_TIMED("frontier_kernel", s);
        VLFM_KLAUNCH(frontier_kernel, dim3(n), dim3(1024), kWalkLdsBytes, s, d_prm, mp, fr, (const int*)d_bbox);
    }
    return check_launch("frontier_kernel");
}

#ifdef VLFM_PHASE_TIMING
extern "C" int vlfm_debug_phase_clocks(long long* h_out /* [3][16] */) {
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(vlfm::g_phase_clock), sizeof(long long) * 48) == hipSuccess ? VLFM_OK : VLFM_ERR_HIP;
}
extern "C" int vlfm_debug_phase_block(int block) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vlfm::g_walk_block), &block, sizeof(int));
    return hipMemcpyToSymbol(HIP_SYMBOL(vlfm::g_phase_block), &block, sizeof(int)) == hipSuccess ? VLFM_OK : VLFM_ERR_HIP;
}
extern "C" int vlfm_debug_wg_spans(long long* h_first /* [3][1024] */, long long* h_last) {
    if (hipMemcpyFromSymbol(h_first, HIP_SYMBOL(vlfm::g_wg_first), sizeof(long long) * 3 * 1024) != hipSuccess) return VLFM_ERR_HIP;
    return hipMemcpyFromSymbol(h_last, HIP_SYMBOL(vlfm::g_wg_last), sizeof(long long) * 3 * 1024) == hipSuccess ? VLFM_OK : VLFM_ERR_HIP;
}
extern "C" int vlfm_debug_parallel_walk_clocks(long long* h_out16) {
    return hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(vlfm::g_walk_clk), sizeof(long long) * 16) == hipSuccess ? VLFM_OK : VLFM_ERR_HIP;
}
extern "C" int vlfm_debug_walk_stats(long long* h_out /* ticks, points, calls; reset afterwards */) {
    long long z = 0;
    if (hipMemcpyFromSymbol(&h_out[0], HIP_SYMBOL(vlfm::g_walk_ticks), 8) != hipSuccess) return VLFM_ERR_HIP;
    (void)hipMemcpyFromSymbol(&h_out[1], HIP_SYMBOL(vlfm::g_walk_points), 8);
    (void)hipMemcpyFromSymbol(&h_out[2], HIP_SYMBOL(vlfm::g_walk_calls), 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vlfm::g_walk_ticks), &z, 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vlfm::g_walk_points), &z, 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vlfm::g_walk_calls), &z, 8);
    return VLFM_OK;
}
#endif

extern "C" int vlfm_walk_path_counters(long long* h_out4, int reset) {
    // borders traced from the LDS tables / (unused) / by one lane after the tables declined; images whose tables were in LDS
    if (!h_out4) return fail(VLFM_ERR_INVALID, "walk_path_counters: bad argument");
    if (hipMemcpyFromSymbol(h_out4, HIP_SYMBOL(vlfm::g_walk_paths), 32) != hipSuccess) return check_launch("walk_path_counters");
    if (reset) {
        const long long z[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(vlfm::g_walk_paths), z, 32) != hipSuccess) return check_launch("walk_path_counters");
    }
    return VLFM_OK;
}

extern "C" int vlfm_obstacle_status(const void* d_scratch, int n_envs, int map_size, int cap_pts, int cap_contours,
                                    int32_t* h_out /* [n_envs][8] */) {
    if (!d_scratch || !h_out) return fail(VLFM_ERR_INVALID, "obstacle_status: bad argument");
    const ScratchLayout L = layout(n_envs, map_size, cap_pts, cap_contours);
    if (hipMemcpy(h_out, (const unsigned char*)d_scratch + L.off_status, (size_t)n_envs * 8 * 4, hipMemcpyDeviceToHost) !=
        hipSuccess)
        return check_launch("obstacle_status");
    return VLFM_OK;
}
