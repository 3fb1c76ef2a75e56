// raster.h -- device-side polygon rasteriser shared by the value-map and obstacle-map kernels (gfx950).
//
// Reproduces the pixel set that OpenCV 4.5.5's scanline polygon fill produces (boundary drawn with the
// 8-connected integer line iterator, interior by the even-odd rule on 16.16 fixed-point edge crossings, spans
// [ceil(x_left), floor(x_right)], edges half-open in y), but formulated for a wavefront machine instead of a
// sorted active-edge list:
//
//   * every polygon edge is owned by one lane; the lane walks the edge's scanlines and, per scanline, XORs one
//     bit into an LDS "parity" bitmap at the first pixel strictly right of the crossing, and ORs one bit into an
//     LDS "solid" bitmap when the crossing lies exactly on a pixel centre;
//   * the same lane draws the edge's boundary pixels into the "solid" bitmap;
//   * after a barrier one lane per bitmap row turns toggles into coverage with a 5-step shift/XOR prefix scan per
//     32-bit word (carry = bit 31 of the previous word), and ORs in the solid bits.
//
// Why this equals the sorted-active-list fill: with crossings c_0<=c_1<=... on a scanline, pixel X (=x<<16) lies in
// some closed pair [c_2k, c_2k+1] iff some crossing equals X, or the number of crossings strictly below X is odd.
// No MFMA anywhere: this is integer/bit work on LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vlfm {

constexpr int XY_SHIFT = 16;
constexpr long long XY_ONE = 1LL << XY_SHIFT;

// rows/cols are the extents of the IMAGE the polygon is drawn on (all clipping follows them); the bitmaps may cover
// only a window of it: wrows x wcols cells whose top-left cell is image cell (ox, oy).  ox = oy = 0 and
// wrows/wcols = rows/cols is the plain whole-image case.
struct LdsBitmap {
    unsigned* solid;   // [wrows][words]  OR-accumulated
    unsigned* parity;  // [wrows][words]  XOR-accumulated toggles
    int rows, cols, words;
    int ox = 0, oy = 0, wrows = -1, wcols = -1;
    __device__ int win_rows() const { return wrows < 0 ? rows : wrows; }
    __device__ int win_cols() const { return wcols < 0 ? cols : wcols; }
};

__device__ inline void bm_or(const LdsBitmap& bm, int y, int x) {  // image coordinates
    const int lx = x - bm.ox, ly = y - bm.oy;
    if ((unsigned)lx >= (unsigned)bm.win_cols() || (unsigned)ly >= (unsigned)bm.win_rows()) return;
    atomicOr(&bm.solid[ly * bm.words + (lx >> 5)], 1u << (lx & 31));
}
__device__ inline void bm_toggle_local(const LdsBitmap& bm, int ly, int lx) {
    atomicXor(&bm.parity[ly * bm.words + (lx >> 5)], 1u << (lx & 31));
}

// cv::clipLine on 64-bit points against [0,width) x [0,height)
__device__ inline bool clip_line(long long width, long long height, long long& x1, long long& y1, long long& x2,
                                 long long& y2) {
    const long long right = width - 1, bottom = height - 1;
    if (width <= 0 || height <= 0) return false;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// 8-connected integer line, always walked towards +x (LineIterator with leftToRight=true), clipped first.
template <typename Put>
__device__ inline void line8(int rows, int cols, long long x0, long long y0, long long x1, long long y1, Put put) {
    if ((unsigned long long)x0 >= (unsigned long long)cols || (unsigned long long)x1 >= (unsigned long long)cols ||
        (unsigned long long)y0 >= (unsigned long long)rows || (unsigned long long)y1 >= (unsigned long long)rows) {
        if (!clip_line(cols, rows, x0, y0, x1, y1)) return;
    }
    int dx = (int)(x1 - x0), dy = (int)(y1 - y0);
    int sx = 1, sy = 1;
    long long px = x0, py = y0;
    if (dx < 0) { dx = -dx; dy = -dy; px = x1; py = y1; }
    if (dy < 0) { dy = -dy; sy = -1; }
    const bool vert = dy > dx;
    if (vert) { int t = dx; dx = dy; dy = t; t = sx; sx = sy; sy = t; }
    int err = dx - (dy + dy);
    const int plus = dx + dx, minus = -(dy + dy);
    const int count = dx + 1;
    for (int i = 0; i < count; i++) {
        put((int)px, (int)py);
        const bool step_minor = err < 0;
        err += minus + (step_minor ? plus : 0);
        if (!vert) { px += sx; if (step_minor) py += sy; }
        else       { py += sx; if (step_minor) px += sy; }
    }
}

// One polygon edge between vertices a and b.  x is 16.16 fixed point, y is an integer scanline.
// Draws the boundary (solid) and the per-scanline parity toggles.
__device__ inline void raster_edge(const LdsBitmap& bm, long long ax, int ay, long long bx, int by) {
    const long long tx0 = (ax + (XY_ONE >> 1)) >> XY_SHIFT, tx1 = (bx + (XY_ONE >> 1)) >> XY_SHIFT;
    line8(bm.rows, bm.cols, tx0, ay, tx1, by, [&](int x, int y) { bm_or(bm, y, x); });
    if (ay == by) return;
    const long long dxdy = (bx - ax) / (long long)(by - ay);  // truncating, same value for either orientation
    int y0, y1;
    long long xs;
    if (ay < by) { y0 = ay; y1 = by; xs = ax; } else { y0 = by; y1 = ay; xs = bx; }
    int ys = y0 < 0 ? 0 : y0;
    int ye = y1 > bm.rows ? bm.rows : y1;  // half-open [y0, y1), clipped to the image
    if (ys < bm.oy) ys = bm.oy;            // ... and to the window rows
    if (ye > bm.oy + bm.win_rows()) ye = bm.oy + bm.win_rows();
    for (int y = ys; y < ye; y++) {
        const long long c = xs + (long long)(y - y0) * dxdy;
        const long long px = c >> XY_SHIFT;  // floor
        if ((c & (XY_ONE - 1)) == 0 && px >= 0 && px < bm.cols) bm_or(bm, y, (int)px);
        long long first = px + 1;  // first pixel strictly right of the crossing
        if (first < 0) first = 0;
        if (first < bm.cols) {
            long long lf = first - bm.ox;  // a toggle left of the window flips the whole window row
            if (lf < 0) lf = 0;
            if (lf < bm.win_cols()) bm_toggle_local(bm, y - bm.oy, (int)lf);
        }
    }
}

// raster_edge by a GROUP of lanes: lane g of G takes every G-th step of the boundary line and every G-th scanline of the edge.
// The line iterator has a closed form -- before major step i it has made m_i = max(0, ceil((2 dy i - dx) / (2 dx))) minor steps
// (err_i = dx - 2 dy (i + 1) + 2 dx m_i, and a minor step is taken iff err_i < 0; checked against the iterative form for all
// dx, dy < 80) -- and the crossings are linear in y already, so nothing in an edge is sequential: the two 100-pixel sides of the
// fog-of-war cone were 22 us on one lane each.
__device__ inline void raster_edge_shared(const LdsBitmap& bm, long long ax, int ay, long long bx, int by, int g, int G) {
    const long long tx0 = (ax + (XY_ONE >> 1)) >> XY_SHIFT, tx1 = (bx + (XY_ONE >> 1)) >> XY_SHIFT;
    {
        long long x0 = tx0, y0 = ay, x1 = tx1, y1 = by;
        bool ok = true;
        if ((unsigned long long)x0 >= (unsigned long long)bm.cols || (unsigned long long)x1 >= (unsigned long long)bm.cols ||
            (unsigned long long)y0 >= (unsigned long long)bm.rows || (unsigned long long)y1 >= (unsigned long long)bm.rows)
            ok = clip_line(bm.cols, bm.rows, x0, y0, x1, y1);
        if (ok) {
            int dx = (int)(x1 - x0), dy = (int)(y1 - y0);
            int sx = 1, sy = 1;
            long long px = x0, py = y0;
            if (dx < 0) { dx = -dx; dy = -dy; px = x1; py = y1; }
            if (dy < 0) { dy = -dy; sy = -1; }
            const bool vert = dy > dx;
            if (vert) { int t = dx; dx = dy; dy = t; t = sx; sx = sy; sy = t; }
            const int count = dx + 1;
            for (int i = g; i < count; i += G) {
                const int num = 2 * dy * i - dx;
                const int m = num <= 0 ? 0 : (int)(((unsigned)num + 2u * (unsigned)dx - 1u) / (2u * (unsigned)dx));
                if (!vert) bm_or(bm, (int)py + sy * m, (int)px + sx * i);
                else bm_or(bm, (int)py + sx * i, (int)px + sy * m);
            }
        }
    }
    if (ay == by) return;
    const long long dxdy = (bx - ax) / (long long)(by - ay);  // truncating, same value for either orientation
    int y0, y1;
    long long xs;
    if (ay < by) { y0 = ay; y1 = by; xs = ax; } else { y0 = by; y1 = ay; xs = bx; }
    int ys = y0 < 0 ? 0 : y0;
    int ye = y1 > bm.rows ? bm.rows : y1;  // half-open [y0, y1), clipped to the image
    if (ys < bm.oy) ys = bm.oy;            // ... and to the window rows
    if (ye > bm.oy + bm.win_rows()) ye = bm.oy + bm.win_rows();
    for (int y = ys + g; y < ye; y += G) {
        const long long c = xs + (long long)(y - y0) * dxdy;
        const long long px = c >> XY_SHIFT;  // floor
        if ((c & (XY_ONE - 1)) == 0 && px >= 0 && px < bm.cols) bm_or(bm, y, (int)px);
        long long first = px + 1;  // first pixel strictly right of the crossing
        if (first < 0) first = 0;
        if (first < bm.cols) {
            long long lf = first - bm.ox;  // a toggle left of the window flips the whole window row
            if (lf < 0) lf = 0;
            if (lf < bm.win_cols()) bm_toggle_local(bm, y - bm.oy, (int)lf);
        }
    }
}

// After all edges: coverage[row] = prefix_xor(parity[row]) | solid[row], written back into `solid`.
__device__ inline void resolve_rows(const LdsBitmap& bm, int tid, int nthreads) {
    for (int y = tid; y < bm.win_rows(); y += nthreads) {
        unsigned carry = 0;
        for (int w = 0; w < bm.words; w++) {
            unsigned p = bm.parity[y * bm.words + w];
            p ^= p << 1; p ^= p << 2; p ^= p << 4; p ^= p << 8; p ^= p << 16;
            if (carry) p = ~p;
            carry = p >> 31;
            bm.solid[y * bm.words + w] |= p;
        }
    }
}

__device__ inline bool bm_test(const unsigned* bits, int words, int y, int x) {
    return (bits[y * words + (x >> 5)] >> (x & 31)) & 1u;
}

}  // namespace vlfm
