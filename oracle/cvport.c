/*
 * oracle/cvport.c -- TEST INFRASTRUCTURE ONLY (parity oracle, CPU baseline).
 *
 * CPU restatement, in plain C, of the OpenCV 4.5.5 primitives that the VLFM
 * perception+mapping hot path calls (the reference pins opencv-python==4.5.5.64,
 * /root/reference/pyproject.toml:28).  OpenCV is NOT vendored in /root/reference and
 * is not installable here, so every routine below restates the published 4.5.x
 * algorithm (modules/imgproc/src/{drawing,imgwarp,contours,shapedescr,morph,
 * box_filter,geometry}.cpp) from its documented behaviour.  PARITY UNPINNED: the
 * reference has no golden vectors for this path; the pins we
 * can make are the hand-derived known answers in tests/test_oracle_cv.py.
 *
 * Reference call sites each primitive serves:
 *   cvp_ellipse_fill      cv2.ellipse(...,-1)        vlfm/mapping/value_map.py:325-334
 *                                                    + frontier_exploration reveal_fog_of_war [ext]
 *   cvp_fill_poly         cv2.drawContours(..,-1)    value_map.py:260, obstacle_map.py:145,
 *                                                    vlfm/utils/img_utils.py:385
 *   cvp_warp_affine_f64   cv2.warpAffine             img_utils.py:25-26
 *   cvp_circle_fill       cv2.circle(..,-1)          img_utils.py:247-253
 *   cvp_dilate_rect       cv2.dilate                 obstacle_map.py:105-109,125,159-163
 *   cvp_find_contours     cv2.findContours           obstacle_map.py:128-132, img_utils.py:377
 *   cvp_contour_area      cv2.contourArea            img_utils.py:383
 *   cvp_point_polygon_test cv2.pointPolygonTest      obstacle_map.py:137
 *   cvp_is_contour_convex, cvp_polylines_thick, cvp_blur3x3   frontier_exploration [ext]
 *
 * Nothing under vlfm_amd/ may link or call this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <limits.h>

#define XY_SHIFT 16
#define XY_ONE (1 << XY_SHIFT)

typedef struct { int64_t x, y; } P2l;

static inline int cv_round(double v) { return (int)lrint(v); } /* round-half-even */

/* ------------------------------------------------------------------ clipLine */
static int clip_line(int64_t width, int64_t height, P2l *p1, P2l *p2) {
    int c1, c2;
    int64_t right = width - 1, bottom = height - 1;
    if (width <= 0 || height <= 0) return 0;
    int64_t *x1 = &p1->x, *y1 = &p1->y, *x2 = &p2->x, *y2 = &p2->y;
    c1 = (*x1 < 0) + (*x1 > right) * 2 + (*y1 < 0) * 4 + (*y1 > bottom) * 8;
    c2 = (*x2 < 0) + (*x2 > right) * 2 + (*y2 < 0) * 4 + (*y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int64_t a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            *x1 += (int64_t)((double)(a - *y1) * (*x2 - *x1) / (*y2 - *y1));
            *y1 = a;
            c1 = (*x1 < 0) + (*x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            *x2 += (int64_t)((double)(a - *y2) * (*x2 - *x1) / (*y2 - *y1));
            *y2 = a;
            c2 = (*x2 < 0) + (*x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                *y1 += (int64_t)((double)(a - *x1) * (*y2 - *y1) / (*x2 - *x1));
                *x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                *y2 += (int64_t)((double)(a - *x2) * (*y2 - *y1) / (*x2 - *x1));
                *x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

/* ------------------------------------------------------------------ Line (8-connected Bresenham, LineIterator leftToRight=true) */
static void line8(uint8_t *img, int rows, int cols, int64_t x0, int64_t y0, int64_t x1, int64_t y1,
                  uint8_t color) {
    P2l pt1 = {x0, y0}, pt2 = {x1, y1};
    if ((uint64_t)pt1.x >= (uint64_t)cols || (uint64_t)pt2.x >= (uint64_t)cols ||
        (uint64_t)pt1.y >= (uint64_t)rows || (uint64_t)pt2.y >= (uint64_t)rows) {
        if (!clip_line(cols, rows, &pt1, &pt2)) return;
    }
    int delta_x = 1, delta_y = 1;
    int dx = (int)(pt2.x - pt1.x), dy = (int)(pt2.y - pt1.y);
    if (dx < 0) { /* leftToRight: swap so that we always walk +x */
        dx = -dx;
        dy = -dy;
        pt1 = pt2;
    }
    if (dy < 0) {
        dy = -dy;
        delta_y = -1;
    }
    int vert = dy > dx;
    if (vert) {
        int t = dx; dx = dy; dy = t;
        t = delta_x; delta_x = delta_y; delta_y = t;
    }
    int err = dx - (dy + dy);
    int plusDelta = dx + dx, minusDelta = -(dy + dy);
    /* in "major" axis we always step delta_x (after swap); in minor axis delta_y when err<0 */
    int minusShiftX, minusStepY, plusShiftX, plusStepY;
    if (!vert) { minusShiftX = delta_x; minusStepY = 0; plusShiftX = 0; plusStepY = delta_y; }
    else       { minusShiftX = 0; minusStepY = delta_x; plusShiftX = delta_y; plusStepY = 0; }
    int count = dx + 1;
    int64_t px = pt1.x, py = pt1.y;
    for (int i = 0; i < count; i++) {
        img[py * cols + px] = color;
        int mask = err < 0 ? -1 : 0;
        err += minusDelta + (plusDelta & mask);
        px += minusShiftX + (plusShiftX & mask);
        py += minusStepY + (plusStepY & mask);
    }
}

void cvp_line8(uint8_t *img, int rows, int cols, long x0, long y0, long x1, long y1, int color) {
    line8(img, rows, cols, x0, y0, x1, y1, (uint8_t)color);
}

static inline void hline(uint8_t *row, int x1, int x2, uint8_t color) {
    for (int x = x1; x <= x2; x++) row[x] = color;
}

/* ------------------------------------------------------------------ Line2 (fixed-point 16.16 endpoints) */
static void line2(uint8_t *img, int rows, int cols, P2l pt1, P2l pt2, uint8_t color) {
    int64_t dx, dy, ax, ay, i, j, x_step, y_step;
    int ecount;
    if (!clip_line((int64_t)cols << XY_SHIFT, (int64_t)rows << XY_SHIFT, &pt1, &pt2)) return;
    dx = pt2.x - pt1.x;
    dy = pt2.y - pt1.y;
    j = dx < 0 ? -1 : 0;
    ax = (dx ^ j) - j;
    i = dy < 0 ? -1 : 0;
    ay = (dy ^ i) - i;
    if (ax > ay) {
        dy = (dy ^ j) - j;
        pt1.x ^= pt2.x & j; pt2.x ^= pt1.x & j; pt1.x ^= pt2.x & j;
        pt1.y ^= pt2.y & j; pt2.y ^= pt1.y & j; pt1.y ^= pt2.y & j;
        x_step = XY_ONE;
        y_step = (dy << XY_SHIFT) / (ax | 1);
        ecount = (int)((pt2.x - pt1.x) >> XY_SHIFT);
    } else {
        dx = (dx ^ i) - i;
        pt1.x ^= pt2.x & i; pt2.x ^= pt1.x & i; pt1.x ^= pt2.x & i;
        pt1.y ^= pt2.y & i; pt2.y ^= pt1.y & i; pt1.y ^= pt2.y & i;
        x_step = (dx << XY_SHIFT) / (ay | 1);
        y_step = XY_ONE;
        ecount = (int)((pt2.y - pt1.y) >> XY_SHIFT);
    }
    pt1.x += (XY_ONE >> 1);
    pt1.y += (XY_ONE >> 1);
#define PUT_POINT(_x, _y) do { int64_t xx = (_x), yy = (_y); \
        if (0 <= xx && xx < cols && 0 <= yy && yy < rows) img[yy * cols + xx] = color; } while (0)
    PUT_POINT((pt2.x + (XY_ONE >> 1)) >> XY_SHIFT, (pt2.y + (XY_ONE >> 1)) >> XY_SHIFT);
    if (ax > ay) {
        pt1.x >>= XY_SHIFT;
        while (ecount >= 0) {
            PUT_POINT(pt1.x, pt1.y >> XY_SHIFT);
            pt1.x++;
            pt1.y += y_step;
            ecount--;
        }
    } else {
        pt1.y >>= XY_SHIFT;
        while (ecount >= 0) {
            PUT_POINT(pt1.x >> XY_SHIFT, pt1.y);
            pt1.x += x_step;
            pt1.y++;
            ecount--;
        }
    }
#undef PUT_POINT
}

/* ------------------------------------------------------------------ CollectPolyEdges + FillEdgeCollection */
typedef struct PolyEdge {
    int y0, y1;
    int64_t x, dx;
    struct PolyEdge *next;
} PolyEdge;

typedef struct { PolyEdge *e; int n, cap; } EdgeVec;

static void ev_push(EdgeVec *v, PolyEdge e) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 64;
        v->e = (PolyEdge *)realloc(v->e, sizeof(PolyEdge) * (size_t)v->cap);
    }
    v->e[v->n++] = e;
}

/* v in fixed point with `shift` fractional bits (shift==0: integer pixels, shift==16: 16.16) */
static void collect_poly_edges(uint8_t *img, int rows, int cols, const P2l *v, int count, EdgeVec *edges,
                               uint8_t color, int shift) {
    int i, delta = (1 << shift) >> 1;
    P2l pt0 = v[count - 1], pt1;
    pt0.x = pt0.x << (XY_SHIFT - shift);
    pt0.y = (pt0.y + delta) >> shift;
    for (i = 0; i < count; i++, pt0 = pt1) {
        PolyEdge edge;
        pt1 = v[i];
        pt1.x = pt1.x << (XY_SHIFT - shift);
        pt1.y = (pt1.y + delta) >> shift;
        {
            int64_t t0x = (pt0.x + (XY_ONE >> 1)) >> XY_SHIFT, t1x = (pt1.x + (XY_ONE >> 1)) >> XY_SHIFT;
            line8(img, rows, cols, t0x, pt0.y, t1x, pt1.y, color);
        }
        if (pt0.y == pt1.y) continue;
        if (pt0.y < pt1.y) {
            edge.y0 = (int)pt0.y; edge.y1 = (int)pt1.y; edge.x = pt0.x;
        } else {
            edge.y0 = (int)pt1.y; edge.y1 = (int)pt0.y; edge.x = pt1.x;
        }
        edge.dx = (pt1.x - pt0.x) / (pt1.y - pt0.y);
        edge.next = 0;
        ev_push(edges, edge);
    }
}

static int cmp_edges(const void *a, const void *b) {
    const PolyEdge *e1 = (const PolyEdge *)a, *e2 = (const PolyEdge *)b;
    if (e1->y0 != e2->y0) return e1->y0 < e2->y0 ? -1 : 1;
    if (e1->x != e2->x) return e1->x < e2->x ? -1 : 1;
    if (e1->dx != e2->dx) return e1->dx < e2->dx ? -1 : 1;
    return 0;
}

static void fill_edge_collection(uint8_t *img, int rows, int cols, EdgeVec *ev, uint8_t color) {
    PolyEdge tmp;
    int i, y, total = ev->n;
    PolyEdge *e;
    int y_max = INT_MIN, y_min = INT_MAX;
    int64_t x_max = -1, x_min = 0x7FFFFFFFFFFFFFFFLL;
    if (total < 2) return;
    for (i = 0; i < total; i++) {
        PolyEdge *e1 = &ev->e[i];
        int64_t x1 = e1->x + (e1->y1 - e1->y0) * e1->dx;
        if (e1->y0 < y_min) y_min = e1->y0;
        if (e1->y1 > y_max) y_max = e1->y1;
        if (e1->x < x_min) x_min = e1->x;
        if (e1->x > x_max) x_max = e1->x;
        if (x1 < x_min) x_min = x1;
        if (x1 > x_max) x_max = x1;
    }
    if (y_max < 0 || y_min >= rows || x_max < 0 || x_min >= ((int64_t)cols << XY_SHIFT)) return;
    qsort(ev->e, (size_t)total, sizeof(PolyEdge), cmp_edges);
    memset(&tmp, 0, sizeof(tmp));
    tmp.y0 = INT_MAX;
    ev_push(ev, tmp); /* sentinel; no more pushes -> pointers are stable */
    i = 0;
    tmp.next = 0;
    e = &ev->e[i];
    if (y_max > rows) y_max = rows;
    for (y = e->y0; y < y_max; y++) {
        PolyEdge *last, *prelast, *keep_prelast;
        int sort_flag = 0, draw = 0, clipline = y < 0;
        prelast = &tmp;
        last = tmp.next;
        while (last || e->y0 == y) {
            if (last && last->y1 == y) {
                prelast->next = last->next;
                last = last->next;
                continue;
            }
            keep_prelast = prelast;
            if (last && (e->y0 > y || last->x < e->x)) {
                prelast = last;
                last = last->next;
            } else if (i < total) {
                prelast->next = e;
                e->next = last;
                prelast = e;
                e = &ev->e[++i];
            } else
                break;
            if (draw) {
                if (!clipline) {
                    uint8_t *timg = img + (size_t)y * cols;
                    int x1, x2;
                    if (keep_prelast->x > prelast->x) {
                        x1 = (int)((prelast->x + XY_ONE - 1) >> XY_SHIFT);
                        x2 = (int)(keep_prelast->x >> XY_SHIFT);
                    } else {
                        x1 = (int)((keep_prelast->x + XY_ONE - 1) >> XY_SHIFT);
                        x2 = (int)(prelast->x >> XY_SHIFT);
                    }
                    if (x1 < cols && x2 >= 0) {
                        if (x1 < 0) x1 = 0;
                        if (x2 >= cols) x2 = cols - 1;
                        hline(timg, x1, x2, color);
                    }
                }
                keep_prelast->x += keep_prelast->dx;
                prelast->x += prelast->dx;
            }
            draw ^= 1;
        }
        /* bubble-sort the active list by x */
        keep_prelast = 0;
        do {
            prelast = &tmp;
            last = tmp.next;
            sort_flag = 0;
            while (last != keep_prelast && last && last->next != 0) {
                PolyEdge *te = last->next;
                if (last->x > te->x) {
                    prelast->next = te;
                    last->next = te->next;
                    te->next = last;
                    prelast = te;
                    sort_flag = 1;
                } else {
                    prelast = last;
                    last = te;
                }
            }
            keep_prelast = prelast;
        } while (sort_flag && keep_prelast != tmp.next && keep_prelast != &tmp);
    }
}

/* fillPoly / drawContours(thickness=-1): all contours' edges go into ONE even-odd edge collection.
 * pts: concatenated (x,y) int32 pairs; npts[c] = vertex count of contour c. */
void cvp_fill_poly(uint8_t *img, int rows, int cols, const int32_t *pts, const int32_t *npts, int ncontours,
                   int color) {
    EdgeVec ev = {0, 0, 0};
    const int32_t *p = pts;
    for (int c = 0; c < ncontours; c++) {
        int n = npts[c];
        if (n > 0) {
            P2l *v = (P2l *)malloc(sizeof(P2l) * (size_t)n);
            for (int k = 0; k < n; k++) { v[k].x = p[2 * k]; v[k].y = p[2 * k + 1]; }
            collect_poly_edges(img, rows, cols, v, n, &ev, (uint8_t)color, 0);
            free(v);
        }
        p += 2 * n;
    }
    fill_edge_collection(img, rows, cols, &ev, (uint8_t)color);
    free(ev.e);
}

/* ------------------------------------------------------------------ FillConvexPoly (used by thick lines) */
static void fill_convex_poly(uint8_t *img, int rows, int cols, const P2l *v, int npts, uint8_t color, int shift) {
    struct { int idx, di; int64_t x, dx; int ye; } edge[2];
    int delta = 1 << shift >> 1;
    int i, y, imin = 0;
    int edges = npts;
    int64_t xmin, xmax, ymin, ymax;
    P2l p0;
    int delta1 = XY_ONE >> 1, delta2 = XY_ONE >> 1;
    p0 = v[npts - 1];
    p0.x <<= XY_SHIFT - shift;
    p0.y <<= XY_SHIFT - shift;
    xmin = xmax = v[0].x;
    ymin = ymax = v[0].y;
    for (i = 0; i < npts; i++) {
        P2l p = v[i];
        if (p.y < ymin) { ymin = p.y; imin = i; }
        if (p.y > ymax) ymax = p.y;
        if (p.x > xmax) xmax = p.x;
        if (p.x < xmin) xmin = p.x;
        p.x <<= XY_SHIFT - shift;
        p.y <<= XY_SHIFT - shift;
        if (shift == 0)
            line8(img, rows, cols, p0.x >> XY_SHIFT, p0.y >> XY_SHIFT, p.x >> XY_SHIFT, p.y >> XY_SHIFT, color);
        else
            line2(img, rows, cols, p0, p, color);
        p0 = p;
    }
    xmin = (xmin + delta) >> shift;
    xmax = (xmax + delta) >> shift;
    ymin = (ymin + delta) >> shift;
    ymax = (ymax + delta) >> shift;
    if (npts < 3 || (int)xmax < 0 || (int)ymax < 0 || (int)xmin >= cols || (int)ymin >= rows) return;
    if (ymax > rows - 1) ymax = rows - 1;
    edge[0].idx = edge[1].idx = imin;
    edge[0].ye = edge[1].ye = y = (int)ymin;
    edge[0].di = 1;
    edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -XY_ONE;
    edge[0].dx = edge[1].dx = 0;
    do {
        for (i = 0; i < 2; i++) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx, di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                int ty = 0;
                for (; edges-- > 0;) {
                    ty = (int)((v[idx].y + delta) >> shift);
                    if (ty > y) {
                        int64_t xs = v[idx0].x, xe = v[idx].x;
                        if (shift != XY_SHIFT) { xs <<= XY_SHIFT - shift; xe <<= XY_SHIFT - shift; }
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
        if (y >= 0) {
            int left = 0, right = 1;
            if (edge[0].x > edge[1].x) { left = 1; right = 0; }
            int xx1 = (int)((edge[left].x + delta1) >> XY_SHIFT);
            int xx2 = (int)((edge[right].x + delta2) >> XY_SHIFT);
            if (xx2 >= 0 && xx1 < cols) {
                if (xx1 < 0) xx1 = 0;
                if (xx2 >= cols) xx2 = cols - 1;
                hline(img + (size_t)y * cols, xx1, xx2, color);
            }
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
    } while (++y <= (int)ymax);
}

/* ------------------------------------------------------------------ Circle (midpoint, filled) */
static void circle_fill(uint8_t *img, int rows, int cols, int cx, int cy, int radius, uint8_t color) {
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    int inside = cx >= radius && cx < cols - radius && cy >= radius && cy < rows - radius;
    while (dx >= dy) {
        int mask;
        int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
        int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
        if (inside) {
            hline(img + (size_t)y11 * cols, x11, x12, color);
            hline(img + (size_t)y12 * cols, x11, x12, color);
            hline(img + (size_t)y21 * cols, x21, x22, color);
            hline(img + (size_t)y22 * cols, x21, x22, color);
        } else if (x11 < cols && x12 >= 0 && y21 < rows && y22 >= 0) {
            if (x11 < 0) x11 = 0;
            if (x12 > cols - 1) x12 = cols - 1;
            if ((unsigned)y11 < (unsigned)rows) hline(img + (size_t)y11 * cols, x11, x12, color);
            if ((unsigned)y12 < (unsigned)rows) hline(img + (size_t)y12 * cols, x11, x12, color);
            if (x21 < cols && x22 >= 0) {
                if (x21 < 0) x21 = 0;
                if (x22 > cols - 1) x22 = cols - 1;
                if ((unsigned)y21 < (unsigned)rows) hline(img + (size_t)y21 * cols, x21, x22, color);
                if ((unsigned)y22 < (unsigned)rows) hline(img + (size_t)y22 * cols, x21, x22, color);
            }
        }
        dy++;
        err += plus;
        plus += 2;
        mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= 2 & mask;
    }
}

void cvp_circle_fill(uint8_t *img, int rows, int cols, int cx, int cy, int radius, int color) {
    circle_fill(img, rows, cols, cx, cy, radius, (uint8_t)color);
}

/* ------------------------------------------------------------------ ThickLine / polylines(thickness>1) */
static void thick_line(uint8_t *img, int rows, int cols, P2l p0, P2l p1, uint8_t color, int thickness, int flags) {
    static const double INV_XY_ONE = 1. / XY_ONE;
    p0.x <<= XY_SHIFT; p0.y <<= XY_SHIFT; p1.x <<= XY_SHIFT; p1.y <<= XY_SHIFT;
    if (thickness <= 1) {
        line8(img, rows, cols, (p0.x + (XY_ONE >> 1)) >> XY_SHIFT, (p0.y + (XY_ONE >> 1)) >> XY_SHIFT,
              (p1.x + (XY_ONE >> 1)) >> XY_SHIFT, (p1.y + (XY_ONE >> 1)) >> XY_SHIFT, color);
        return;
    }
    P2l pt[4], dp = {0, 0};
    double dx = (p0.x - p1.x) * INV_XY_ONE, dy = (p1.y - p0.y) * INV_XY_ONE;
    double r = dx * dx + dy * dy;
    int i, oddThickness = thickness & 1;
    thickness <<= XY_SHIFT - 1;
    if (fabs(r) > DBL_EPSILON) {
        r = (thickness + oddThickness * XY_ONE * 0.5) / sqrt(r);
        dp.x = cv_round(dy * r);
        dp.y = cv_round(dx * r);
        pt[0].x = p0.x + dp.x; pt[0].y = p0.y + dp.y;
        pt[1].x = p0.x - dp.x; pt[1].y = p0.y - dp.y;
        pt[2].x = p1.x - dp.x; pt[2].y = p1.y - dp.y;
        pt[3].x = p1.x + dp.x; pt[3].y = p1.y + dp.y;
        fill_convex_poly(img, rows, cols, pt, 4, color, XY_SHIFT);
    }
    for (i = 0; i < 2; i++) {
        if (flags & (i + 1)) {
            int cx = (int)((p0.x + (XY_ONE >> 1)) >> XY_SHIFT);
            int cy = (int)((p0.y + (XY_ONE >> 1)) >> XY_SHIFT);
            circle_fill(img, rows, cols, cx, cy, (thickness + (XY_ONE >> 1)) >> XY_SHIFT, color);
        }
        p0 = p1;
    }
}

/* cv2.polylines(img, lines, isClosed=False, color, thickness): lines = nlines open 2-point polylines
 * (frontier_exploration passes (N,2,2) int32).  PolyLine: flags = 2 + !is_closed, then flags=2 per segment. */
void cvp_polylines2_thick(uint8_t *img, int rows, int cols, const int32_t *segs, int nsegs, int color,
                          int thickness) {
    for (int s = 0; s < nsegs; s++) {
        P2l p0 = {segs[4 * s + 0], segs[4 * s + 1]}, p1 = {segs[4 * s + 2], segs[4 * s + 3]};
        /* PolyLine with count=2, is_closed=0: i starts at 1, flags = 2 + 1 = 3 for the only segment */
        thick_line(img, rows, cols, p0, p1, (uint8_t)color, thickness, 3);
    }
}

/* ------------------------------------------------------------------ ellipse2Poly + EllipseEx (filled sector) */
static float g_sin_table[451];
static int g_sin_init = 0;
static void init_sin_table(void) {
    if (g_sin_init) return;
    /* OpenCV's SinTable[] holds sin(i deg), i=0..450, written as 7-decimal float literals. */
    for (int i = 0; i <= 450; i++) {
        double s = sin(i * 3.14159265358979323846 / 180.0);
        double r = round(s * 1e7) / 1e7;
        g_sin_table[i] = (float)r;
    }
    g_sin_init = 1;
}

typedef struct { double x, y; } P2d;

static int ellipse2poly(P2d center, double aw, double ah, int angle, int arc_start, int arc_end, int delta,
                        P2d *pts /* cap >= 400 */) {
    float alpha, beta;
    int i, n = 0;
    init_sin_table();
    while (angle < 0) angle += 360;
    while (angle > 360) angle -= 360;
    if (arc_start > arc_end) { i = arc_start; arc_start = arc_end; arc_end = i; }
    while (arc_start < 0) { arc_start += 360; arc_end += 360; }
    while (arc_end > 360) { arc_end -= 360; arc_start -= 360; }
    if (arc_end - arc_start > 360) { arc_start = 0; arc_end = 360; }
    beta = g_sin_table[angle];
    alpha = g_sin_table[450 - angle];
    for (i = arc_start; i < arc_end + delta; i += delta) {
        double x, y;
        int a = i;
        if (a > arc_end) a = arc_end;
        if (a < 0) a += 360;
        x = aw * g_sin_table[450 - a];
        y = ah * g_sin_table[a];
        pts[n].x = center.x + x * alpha - y * beta;
        pts[n].y = center.y + x * beta + y * alpha;
        n++;
    }
    if (n == 1) { pts[0] = center; pts[1] = center; n = 2; }
    return n;
}

/* cv2.ellipse(img, (cx,cy), (ax,ay), angle, start, end, color, thickness=-1), LINE_8, shift=0.
 * Also exports the 16.16 polygon (for tests / the HIP host side pins). Returns vertex count. */
int cvp_ellipse_poly(int cx, int cy, int ax, int ay, double angle, double start_angle, double end_angle,
                     int64_t *out_xy /* cap 2*402 */) {
    int _angle = cv_round(angle), _start = cv_round(start_angle), _end = cv_round(end_angle);
    P2l center = {(int64_t)cx << XY_SHIFT, (int64_t)cy << XY_SHIFT};
    int64_t aw = llabs((int64_t)ax << XY_SHIFT), ah = llabs((int64_t)ay << XY_SHIFT);
    int64_t mx = aw > ah ? aw : ah;
    int delta = (int)((mx + (XY_ONE >> 1)) >> XY_SHIFT);
    delta = delta < 3 ? 90 : delta < 10 ? 30 : delta < 15 ? 18 : 5;
    P2d _v[402];
    P2d c = {(double)center.x, (double)center.y};
    int n = ellipse2poly(c, (double)aw, (double)ah, _angle, _start, _end, delta, _v);
    int m = 0;
    P2l prev = {(int64_t)-1, (int64_t)-1};
    for (int i = 0; i < n; i++) {
        P2l pt;
        pt.x = (int64_t)cv_round(_v[i].x / XY_ONE) << XY_SHIFT;
        pt.y = (int64_t)cv_round(_v[i].y / XY_ONE) << XY_SHIFT;
        pt.x += cv_round(_v[i].x - pt.x);
        pt.y += cv_round(_v[i].y - pt.y);
        if (pt.x != prev.x || pt.y != prev.y) {
            out_xy[2 * m] = pt.x; out_xy[2 * m + 1] = pt.y; m++;
            prev = pt;
        }
    }
    if (m == 1) { out_xy[2] = center.x; out_xy[3] = center.y; out_xy[0] = center.x; out_xy[1] = center.y; m = 2; }
    if (_end - _start < 360) { /* EllipseEx: sectors get the centre appended, full ellipses do not */
        out_xy[2 * m] = center.x; out_xy[2 * m + 1] = center.y; m++;
    }
    return m;
}

void cvp_ellipse_fill(uint8_t *img, int rows, int cols, int cx, int cy, int ax, int ay, double angle,
                      double start_angle, double end_angle, int color) {
    int64_t xy[2 * 404];
    int _start = cv_round(start_angle), _end = cv_round(end_angle);
    int span = _end - _start;
    int m = cvp_ellipse_poly(cx, cy, ax, ay, angle, start_angle, end_angle, xy);
    P2l v[404];
    for (int i = 0; i < m; i++) { v[i].x = xy[2 * i]; v[i].y = xy[2 * i + 1]; }
    if (span >= 360) {
        fill_convex_poly(img, rows, cols, v, m, (uint8_t)color, XY_SHIFT);
    } else {
        EdgeVec ev = {0, 0, 0};
        collect_poly_edges(img, rows, cols, v, m, &ev, (uint8_t)color, XY_SHIFT);
        fill_edge_collection(img, rows, cols, &ev, (uint8_t)color);
        free(ev.e);
    }
}

/* ------------------------------------------------------------------ warpAffine (CV_64F, INTER_LINEAR, BORDER_CONSTANT) */
#define INTER_BITS 5
#define INTER_TAB_SIZE (1 << INTER_BITS)
#define AB_BITS 10
#define AB_SCALE (1 << AB_BITS)

static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

/* M: forward 2x3 matrix as returned by getRotationMatrix2D (no WARP_INVERSE_MAP) */
void cvp_warp_affine_f64(const double *src, double *dst, int rows, int cols, const double *Min, double border) {
    double M[6];
    memcpy(M, Min, sizeof(M));
    {
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D;
        M[3] *= -D; M[4] = A22;
        double b1 = -M[0] * M[2] - M[1] * M[5];
        double b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
    }
    int *adelta = (int *)malloc(sizeof(int) * (size_t)cols * 2), *bdelta = adelta + cols;
    for (int x = 0; x < cols; x++) {
        adelta[x] = cv_round(M[0] * x * AB_SCALE);
        bdelta[x] = cv_round(M[3] * x * AB_SCALE);
    }
    const int round_delta = AB_SCALE / INTER_TAB_SIZE / 2;
    for (int y = 0; y < rows; y++) {
        int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        for (int x = 0; x < cols; x++) {
            int X = (X0 + adelta[x]) >> (AB_BITS - INTER_BITS);
            int Y = (Y0 + bdelta[x]) >> (AB_BITS - INTER_BITS);
            int sx = sat_short(X >> INTER_BITS), sy = sat_short(Y >> INTER_BITS);
            int fxi = X & (INTER_TAB_SIZE - 1), fyi = Y & (INTER_TAB_SIZE - 1);
            float fx = (float)fxi * (1.f / INTER_TAB_SIZE), fy = (float)fyi * (1.f / INTER_TAB_SIZE);
            /* BilinearTab_f: w[k1*2+k2] = ((k1? fy : 1-fy) * (k2 ? fx : 1-fx)) as float products */
            float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
            double out;
            if (sx >= cols || sx + 1 < 0 || sy >= rows || sy + 1 < 0) {
                out = border;
            } else {
                int sx1 = sx + 1, sy1 = sy + 1;
                double v0 = (sx >= 0 && sy >= 0 && sx < cols && sy < rows) ? src[(size_t)sy * cols + sx] : border;
                double v1 = (sx1 >= 0 && sy >= 0 && sx1 < cols && sy < rows) ? src[(size_t)sy * cols + sx1] : border;
                double v2 = (sx >= 0 && sy1 >= 0 && sx < cols && sy1 < rows) ? src[(size_t)sy1 * cols + sx] : border;
                double v3 = (sx1 >= 0 && sy1 >= 0 && sx1 < cols && sy1 < rows) ? src[(size_t)sy1 * cols + sx1] : border;
                out = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
            }
            dst[(size_t)y * cols + x] = out;
        }
    }
    free(adelta);
}

/* ------------------------------------------------------------------ dilate with a kw x kh all-ones kernel, centre anchor */
void cvp_dilate_rect(const uint8_t *src, uint8_t *dst, int rows, int cols, int kw, int kh) {
    /* rectangular structuring element, anchor at the centre, border = "no contribution" (cv2's default border value for
     * dilation never wins a maximum): separable running maxima written as whole-row shifted max passes so that the
     * compiler vectorises them (full 1000 x 1000 maps, several times per step) */
    int ax = kw / 2, ay = kh / 2;
    uint8_t *tmp = (uint8_t *)malloc((size_t)rows * cols);
    for (int y = 0; y < rows; y++) {
        const uint8_t *s = src + (size_t)y * cols;
        uint8_t *t = tmp + (size_t)y * cols;
        memcpy(t, s, (size_t)cols);
        for (int k = 1; k <= ax; k++) {                 /* window [x - ax, x - ax + kw - 1] */
            for (int x = k; x < cols; x++) { uint8_t v = s[x - k]; if (v > t[x]) t[x] = v; }
        }
        for (int k = 1; k <= kw - 1 - ax; k++) {
            for (int x = 0; x + k < cols; x++) { uint8_t v = s[x + k]; if (v > t[x]) t[x] = v; }
        }
    }
    for (int y = 0; y < rows; y++) {
        int lo = y - ay, hi = y - ay + kh - 1;
        if (lo < 0) lo = 0;
        if (hi > rows - 1) hi = rows - 1;
        uint8_t *d = dst + (size_t)y * cols;
        memcpy(d, tmp + (size_t)lo * cols, (size_t)cols);
        for (int k = lo + 1; k <= hi; k++) {
            const uint8_t *t = tmp + (size_t)k * cols;
            for (int x = 0; x < cols; x++) { uint8_t v = t[x]; if (v > d[x]) d[x] = v; }
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------ blur 3x3, u8, BORDER_REFLECT_101, normalized */
static inline int reflect101_(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i < 0 ? 0 : i;
}
void cvp_blur3x3(const uint8_t *src, uint8_t *dst, int rows, int cols) {
    /* sums are integers <= 9 * 255: the rounding of s * (1/9) is tabulated once; horizontal 3-sums per row, then three
     * rows added -- same values as the direct 9-tap evaluation */
    static uint8_t lut[9 * 255 + 1];
    static int have = 0;
    if (!have) {
        for (int s = 0; s <= 9 * 255; s++) { int v = cv_round(s * (1. / 9)); lut[s] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
        have = 1;
    }
    uint16_t *h = (uint16_t *)malloc((size_t)rows * cols * sizeof(uint16_t));
    for (int y = 0; y < rows; y++) {
        const uint8_t *s = src + (size_t)y * cols;
        uint16_t *t = h + (size_t)y * cols;
        for (int x = 1; x + 1 < cols; x++) t[x] = (uint16_t)(s[x - 1] + s[x] + s[x + 1]);
        t[0] = (uint16_t)(s[reflect101_(-1, cols)] + s[0] + s[reflect101_(1, cols)]);
        if (cols > 1) t[cols - 1] = (uint16_t)(s[reflect101_(cols - 2, cols)] + s[cols - 1] + s[reflect101_(cols, cols)]);
    }
    for (int y = 0; y < rows; y++) {
        const uint16_t *a = h + (size_t)reflect101_(y - 1, rows) * cols, *b = h + (size_t)y * cols,
                       *c = h + (size_t)reflect101_(y + 1, rows) * cols;
        uint8_t *d = dst + (size_t)y * cols;
        for (int x = 0; x < cols; x++) d[x] = lut[a[x] + b[x] + c[x]];
    }
    free(h);
}

/* ------------------------------------------------------------------ findContours (Suzuki-Abe, legacy C implementation semantics)
 * mode: 0 = RETR_EXTERNAL, 1 = RETR_LIST (RETR_CCOMP/RETR_TREE trace exactly the same borders; only the
 *       hierarchy differs, which no hot-path caller reads -- img_utils.py:377 discards it).
 * method: 1 = CHAIN_APPROX_NONE, 2 = CHAIN_APPROX_SIMPLE.
 * Output: points appended to out_pts (x,y int32), per-contour lengths in out_len, per-contour is_hole flag.
 * Contours are returned in OpenCV's order (reverse discovery order: cvInsertNodeIntoTree pushes at head).
 * Returns number of contours, or -(needed) if a capacity was exceeded. */
static const int code_dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int code_dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

typedef struct { int32_t *pts; long n, cap; } PtVec;
static void pv_push(PtVec *v, int x, int y) {
    if (v->n + 2 > v->cap) {
        v->cap = v->cap ? v->cap * 2 : 4096;
        v->pts = (int32_t *)realloc(v->pts, sizeof(int32_t) * (size_t)v->cap);
    }
    v->pts[v->n++] = x;
    v->pts[v->n++] = y;
}

static long fetch_contour(int8_t *ptr, int step, int px, int py, int is_hole, int method, PtVec *out) {
    const int8_t nbd = 2;
    int deltas[16];
    int8_t *i0 = ptr, *i1, *i3, *i4 = 0;
    int prev_s = -1, s, s_end;
    long start_n = out->n;
    method -= 1; /* 0 = NONE, 1 = SIMPLE */
    deltas[0] = 1; deltas[1] = -step + 1; deltas[2] = -step; deltas[3] = -step - 1;
    deltas[4] = -1; deltas[5] = step - 1; deltas[6] = step; deltas[7] = step + 1;
    memcpy(deltas + 8, deltas, 8 * sizeof(int));
    s_end = s = is_hole ? 0 : 4;
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (*i1 == 0 && s != s_end);
    if (s == s_end) { /* single pixel domain */
        *i0 = (int8_t)(nbd | -128);
        pv_push(out, px, py);
    } else {
        i3 = i0;
        prev_s = s ^ 4;
        for (;;) {
            s_end = s;
            for (;;) {
                i4 = i3 + deltas[++s];
                if (*i4 != 0) break;
            }
            s &= 7;
            if ((unsigned)(s - 1) < (unsigned)s_end) {
                *i3 = (int8_t)(nbd | -128);
            } else if (*i3 == 1) {
                *i3 = nbd;
            }
            if (s != prev_s || method == 0) {
                pv_push(out, px, py);
                prev_s = s;
            }
            px += code_dx[s];
            py += code_dy[s];
            if (i4 == i0 && i3 == i1) break;
            i3 = i4;
            s = (s + 4) & 7;
        }
    }
    return (out->n - start_n) / 2;
}

int cvp_find_contours(const uint8_t *image, int rows, int cols, int mode, int method, int32_t *out_pts,
                      long pts_cap /* in points */, int32_t *out_len, int32_t *out_hole, int max_contours) {
    /* working image: nonzero->1, 1-pixel zero frame (cvStartFindContours zeroes the border rows/cols of the
     * image itself; cv::findContours copies into a +1 padded buffer first so no input pixel is lost). */
    int W = cols + 2, H = rows + 2, step = W;
    int8_t *img = (int8_t *)calloc((size_t)W * H, 1);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) img[(size_t)(y + 1) * step + x + 1] = image[(size_t)y * cols + x] != 0;
    PtVec pv = {0, 0, 0};
    int ncont = 0, cap_c = 256;
    long *starts = (long *)malloc(sizeof(long) * (size_t)cap_c);
    int *lens = (int *)malloc(sizeof(int) * (size_t)cap_c);
    int *holes = (int *)malloc(sizeof(int) * (size_t)cap_c);
    for (int y = 1; y < H - 1; y++) {
        int8_t *row = img + (size_t)y * step;
        int prev = 0;
        int lnbd_x = 0; /* lnbd.y == y, lnbd.x resets to 0 (a zero pixel) at each row */
        for (int x = 1; x < W - 1; x++) {
            int p = row[x];
            if (p != prev) {
                int is_hole = 0;
                int origin_x = x;
                int skip = 0;
                if (!(prev == 0 && p == 1)) {
                    /* candidate hole border: previous pixel nonzero (>=1), this one zero */
                    if (p != 0 || prev < 1) skip = 1;
                    else {
                        if (prev & -2) lnbd_x = x - 1;
                        is_hole = 1;
                    }
                }
                if (!skip && mode == 0 && (is_hole || row[lnbd_x] > 0)) skip = 1;
                if (!skip) {
                    if (is_hole) origin_x = x - 1;
                    if (ncont == cap_c) {
                        cap_c *= 2;
                        starts = (long *)realloc(starts, sizeof(long) * (size_t)cap_c);
                        lens = (int *)realloc(lens, sizeof(int) * (size_t)cap_c);
                        holes = (int *)realloc(holes, sizeof(int) * (size_t)cap_c);
                    }
                    starts[ncont] = pv.n / 2;
                    /* contour coordinates are reported relative to the un-padded image (offset -1,-1) */
                    lens[ncont] = (int)fetch_contour(row + origin_x, step, origin_x - 1, y - 1, is_hole, method, &pv);
                    holes[ncont] = is_hole;
                    ncont++;
                    p = row[x];
                    /* after tracing, lnbd.x = x - is_hole (pixel that now carries a border label) */
                    lnbd_x = x - is_hole;
                }
                /* resume_scan */
                prev = p;
                if (prev & -2) lnbd_x = x;
            }
        }
    }
    int ret = ncont;
    if (ncont > max_contours || pv.n / 2 > pts_cap) {
        ret = -1;
    } else {
        long w = 0;
        for (int c = ncont - 1, k = 0; c >= 0; c--, k++) { /* reverse discovery order */
            memcpy(out_pts + 2 * w, pv.pts + 2 * starts[c], sizeof(int32_t) * 2 * (size_t)lens[c]);
            out_len[k] = lens[c];
            out_hole[k] = holes[c];
            w += lens[c];
        }
    }
    free(img); free(pv.pts); free(starts); free(lens); free(holes);
    return ret;
}

/* ------------------------------------------------------------------ contourArea / pointPolygonTest / isContourConvex */
double cvp_contour_area(const int32_t *pts, int n) {
    if (n == 0) return 0.;
    double a00 = 0;
    float px = (float)pts[2 * (n - 1)], py = (float)pts[2 * (n - 1) + 1];
    for (int i = 0; i < n; i++) {
        float x = (float)pts[2 * i], y = (float)pts[2 * i + 1];
        a00 += (double)px * y - (double)py * x;
        px = x; py = y;
    }
    return fabs(a00 * 0.5);
}

double cvp_point_polygon_test(const int32_t *pts, int n, double ptx, double pty, int measure_dist) {
    double result = 0;
    int counter = 0;
    if (n == 0) return measure_dist ? -DBL_MAX : -1;
    float fx = (float)ptx, fy = (float)pty; /* Point2f pt */
    float vx = (float)pts[2 * (n - 1)], vy = (float)pts[2 * (n - 1) + 1], v0x, v0y;
    if (!measure_dist) {
        for (int i = 0; i < n; i++) {
            double dist;
            v0x = vx; v0y = vy;
            vx = (float)pts[2 * i]; vy = (float)pts[2 * i + 1];
            if ((v0y <= fy && vy <= fy) || (v0y > fy && vy > fy) || (v0x < fx && vx < fx)) {
                if (fy == vy && (fx == vx || (fy == v0y && ((v0x <= fx && fx <= vx) || (vx <= fx && fx <= v0x)))))
                    return 0;
                continue;
            }
            dist = (double)(fy - v0y) * (vx - v0x) - (double)(fx - v0x) * (vy - v0y);
            if (dist == 0) return 0;
            if (vy < v0y) dist = -dist;
            counter += dist > 0;
        }
        return counter % 2 == 0 ? -1 : 1;
    }
    double min_dist_num = FLT_MAX, min_dist_denom = 1;
    for (int i = 0; i < n; i++) {
        double dx, dy, dx1, dy1, dx2, dy2, dist_num, dist_denom = 1;
        v0x = vx; v0y = vy;
        vx = (float)pts[2 * i]; vy = (float)pts[2 * i + 1];
        dx = vx - v0x; dy = vy - v0y;
        dx1 = fx - v0x; dy1 = fy - v0y;
        dx2 = fx - vx; dy2 = fy - vy;
        if (dx1 * dx + dy1 * dy <= 0)
            dist_num = dx1 * dx1 + dy1 * dy1;
        else if (dx2 * dx + dy2 * dy >= 0)
            dist_num = dx2 * dx2 + dy2 * dy2;
        else {
            dist_num = (dy1 * dx - dx1 * dy);
            dist_num *= dist_num;
            dist_denom = dx * dx + dy * dy;
        }
        if (dist_num * min_dist_denom < min_dist_num * dist_denom) {
            min_dist_num = dist_num;
            min_dist_denom = dist_denom;
            if (min_dist_num == 0) break;
        }
        if ((v0y <= fy && vy <= fy) || (v0y > fy && vy > fy) || (v0x < fx && vx < fx)) continue;
        dist_num = dy1 * dx - dx1 * dy;
        if (dy < 0) dist_num = -dist_num;
        counter += dist_num > 0;
    }
    result = sqrt(min_dist_num / min_dist_denom);
    if (counter % 2 == 0) result = -result;
    return result;
}

int cvp_is_contour_convex(const int32_t *p, int n) {
    if (n == 0) return 0;
    int pi = (n - 2 + n) % n;
    int prev_x = p[2 * pi], prev_y = p[2 * pi + 1];
    int cur_x = p[2 * (n - 1)], cur_y = p[2 * (n - 1) + 1];
    int dx0 = cur_x - prev_x, dy0 = cur_y - prev_y;
    int orientation = 0;
    for (int i = 0; i < n; i++) {
        int dxdy0, dydx0, dx, dy;
        prev_x = cur_x; prev_y = cur_y;
        cur_x = p[2 * i]; cur_y = p[2 * i + 1];
        dx = cur_x - prev_x; dy = cur_y - prev_y;
        dxdy0 = dx * dy0;
        dydx0 = dy * dx0;
        orientation |= (dydx0 > dxdy0) ? 1 : ((dydx0 < dxdy0) ? 2 : 3);
        if (orientation == 3) return 0;
        dx0 = dx; dy0 = dy;
    }
    return 1;
}

/* ------------------------------------------------------------------ frontier_exploration helpers that upstream numba-jits [ext]
 * contour_to_frontiers + get_frontier_midpoint fused: contour = (n,2) int32 (already "interpolated"),
 * unexplored = blurred mask.  Writes midpoints (x,y) float64; returns count. */
int cvp_contour_frontier_midpoints(const int32_t *contour, int n, const uint8_t *unexplored, int rows, int cols,
                                   double *out_xy, int cap) {
    (void)rows;
    if (n <= 0) return 0;
    int *bad = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    int nbad = 0;
    for (int i = 0; i < n; i++) {
        int x = contour[2 * i], y = contour[2 * i + 1];
        if (unexplored[(size_t)y * cols + x] == 0) bad[nbad++] = i;
    }
    /* np.split(contour, bad) -> nbad+1 pieces: [0,bad0), [bad0,bad1), ..., [bad_last, n) */
    int npieces = nbad + 1;
    int front_last_split = (nbad > 0) && (bad[0] != 0) && (bad[nbad - 1] < n - 2);
    /* kept pieces as (start,len) after dropping the leading bad element for idx>0 */
    int *ks = (int *)malloc(sizeof(int) * (size_t)npieces * 2);
    int nk = 0;
    for (int idx = 0; idx < npieces; idx++) {
        int s = idx == 0 ? 0 : bad[idx - 1];
        int e = idx == nbad ? n : bad[idx];
        int len = e - s;
        if (len > 2 || (idx == 0 && front_last_split)) {
            if (idx == 0) { ks[2 * nk] = s; ks[2 * nk + 1] = len; }
            else { ks[2 * nk] = s + 1; ks[2 * nk + 1] = len - 1; }
            nk++;
        }
    }
    int nout = 0;
    /* merged first frontier = last piece ++ first piece when front_last_split and nk>1 */
    int merge = (nk > 1 && front_last_split);
    int nfront = merge ? nk - 1 : nk;
    double *seg = (double *)malloc(sizeof(double) * (size_t)(n + 2) * 2);
    for (int f = 0; f < nfront && nout < cap; f++) {
        int m = 0;
        if (f == 0 && merge) {
            int s = ks[2 * (nk - 1)], l = ks[2 * (nk - 1) + 1];
            for (int k = 0; k < l; k++) { seg[2 * m] = contour[2 * (s + k)]; seg[2 * m + 1] = contour[2 * (s + k) + 1]; m++; }
        }
        {
            int s = ks[2 * f], l = ks[2 * f + 1];
            for (int k = 0; k < l; k++) { seg[2 * m] = contour[2 * (s + k)]; seg[2 * m + 1] = contour[2 * (s + k) + 1]; m++; }
        }
        /* get_frontier_midpoint: arc-length midpoint */
        if (m < 2) { /* upstream would index an empty cumsum; a 1-point frontier cannot survive the len>2 filter
                        except idx==0 with front_last_split: emit the point itself (documented divergence guard) */
            if (m == 1) { out_xy[2 * nout] = seg[0]; out_xy[2 * nout + 1] = seg[1]; nout++; }
            continue;
        }
        double total = 0;
        for (int k = 0; k + 1 < m; k++) {
            double ddx = seg[2 * k] - seg[2 * (k + 1)], ddy = seg[2 * k + 1] - seg[2 * (k + 1) + 1];
            total += sqrt(ddx * ddx + ddy * ddy);
        }
        double half = total / 2, cum = 0, upto = 0;
        int idx = 0;
        double seglen = 0;
        {
            double c = 0;
            int found = 0;
            for (int k = 0; k + 1 < m; k++) {
                double ddx = seg[2 * k] - seg[2 * (k + 1)], ddy = seg[2 * k + 1] - seg[2 * (k + 1) + 1];
                double l = sqrt(ddx * ddx + ddy * ddy);
                double c2 = c + l;
                if (c2 > half) { idx = k; upto = k > 0 ? c : 0; seglen = l; found = 1; break; }
                c = c2;
            }
            if (!found) { /* np.argmax of all-False -> 0 */
                idx = 0; upto = 0;
                double ddx = seg[0] - seg[2], ddy = seg[1] - seg[3];
                seglen = sqrt(ddx * ddx + ddy * ddy);
            }
            (void)cum;
        }
        double prop = (half - upto) / seglen;
        out_xy[2 * nout] = seg[2 * idx] + prop * (seg[2 * (idx + 1)] - seg[2 * idx]);
        out_xy[2 * nout + 1] = seg[2 * idx + 1] + prop * (seg[2 * (idx + 1) + 1] - seg[2 * idx + 1]);
        nout++;
    }
    free(bad); free(ks); free(seg);
    return nout;
}
