// bitmap.h -- bit-packed binary maps (1 bit per cell, row = `stride` 32-bit words, bit x of a row lives in word x>>5 at
// position x&31) and the wave-cooperative border follower used by the obstacle-map kernels (gfx950).
//
// Why bit-packed: ObstacleMap's planes (obstacles, navigable, explored) are boolean 1000x1000 grids.  Packed they are
// 125 KB each (L2/MALL-resident for dozens of environments), a k x k dilation is k word-shifts + ORs, and a 205x205
// fog-of-war window is 5.7 KB of LDS.
//
// Border following restates the Suzuki-Abe scan of OpenCV's legacy findContours (RETR_EXTERNAL; CHAIN_APPROX_NONE /
// _SIMPLE) for a 64-wide wavefront: the raster scan for border starts is done 64 words at a time with ballot/ctz, the
// "last labelled pixel to the left" test (lnbd) is a masked clz over the row's label words, and only the inherently
// sequential chain walk is executed by a single lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vlfm {

#ifdef VLFM_PHASE_TIMING
__device__ long long g_walk_ticks, g_walk_points, g_walk_calls;  // diagnostics: time inside follow_border (all scans)
#endif

struct Bits {            // read view
    const unsigned* w;
    int stride;          // words per row
    int rows, cols;
    int padded = 0;      // 1: the plane (and its label planes) has one zero row above/below and one zero word left/right of
                         // every row, i.e. w points at word [1][1] of a (rows + 2) x stride array whose real words are
                         // 1 .. stride - 2 of each row: the border walk then needs no bounds checks at all
};

__device__ inline unsigned bit_get(const unsigned* w, int stride, int rows, int cols, int x, int y) {
    if ((unsigned)x >= (unsigned)cols || (unsigned)y >= (unsigned)rows) return 0u;
    return (w[(size_t)y * stride + (x >> 5)] >> (x & 31)) & 1u;
}
__device__ inline unsigned bit_get(const Bits& b, int x, int y) { return bit_get(b.w, b.stride, b.rows, b.cols, x, y); }

// three consecutive bits x-1, x, x+1 of row y (zero outside) -> bits 0..2
__device__ inline unsigned bits3(const Bits& b, int x, int y) {
    if ((unsigned)y >= (unsigned)b.rows) return 0u;
    const unsigned* row = b.w + (size_t)y * b.stride;
    const int sh = x & 31, wi = x >> 5;
    // fast path (15/16 of all positions): the three bits live in one word and none is beyond the last column
    if (sh != 0 && sh != 31 && x > 0 && x + 1 < b.cols) return (row[wi] >> (sh - 1)) & 7u;
    const int xm = x - 1;
    unsigned long long win;
    if (xm < 0) {
        win = ((unsigned long long)row[0]) << 1;  // bit 0 = x=-1 (zero), bit1 = x=0 ...
        unsigned r = (unsigned)(win & 7u);
        if (b.cols < 2) r &= 3u;
        return r;
    }
    const int wm = xm >> 5, shm = xm & 31;
    win = row[wm];
    if (wm + 1 < b.stride) win |= ((unsigned long long)row[wm + 1]) << 32;
    unsigned r = (unsigned)((win >> shm) & 7u);
    // mask columns beyond cols
    if (x + 1 >= b.cols) r &= (x >= b.cols) ? 1u : 3u;
    return r;
}

// 8-neighbourhood of (x,y) as a mask indexed by chain code: 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE (y grows downwards)
__device__ inline unsigned nbr8(const Bits& b, int x, int y) {
    const unsigned up = bits3(b, x, y - 1), mid = bits3(b, x, y), dn = bits3(b, x, y + 1);
    return ((mid >> 2) & 1u)            /* E  */
         | (((up >> 2) & 1u) << 1)      /* NE */
         | (((up >> 1) & 1u) << 2)      /* N  */
         | ((up & 1u) << 3)             /* NW */
         | ((mid & 1u) << 4)            /* W  */
         | ((dn & 1u) << 5)             /* SW */
         | (((dn >> 1) & 1u) << 6)      /* S  */
         | (((dn >> 2) & 1u) << 7);     /* SE */
}

// chain-code steps dx = {1,1,0,-1,-1,-1,0,1}, dy = {0,-1,-1,-1,0,1,1,1} as 2-bit fields of a register constant: a table in
// memory would put a dependent memory load into every step of the single-lane border walk
__device__ inline int code_dx(int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; }
__device__ inline int code_dy(int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; }

struct ContourSink {         // per-environment output of a scan
    int2* pts;               // [cap_pts]
    int* start;              // [cap_contours]
    int* len;                // [cap_contours]
    int cap_pts, cap_contours;
    int n_pts, n_contours;   // running counts (uniform across the wave)
    int overflow;
    unsigned char* hole = nullptr;  // [cap_contours] 1 = hole border (RETR_LIST scans only), optional
};

// Follow one outer border starting at (x0,y0) (the scan guarantees: pixel set, left neighbour clear, not yet traced).
// Executed by ONE lane.  Labels: `traced` = pixel carries a border label, `neg` = the label is the negative one
// (the pixel's east neighbour was examined and found empty).  method: 1 = every chain point, 2 = direction changes only.
// is_hole: the border is a hole border (starts at the pixel WEST of the first hole pixel; the initial search direction
// is east instead of west -- cvFindNextContour / icvFetchContour).
__device__ inline int follow_border_padded(const Bits& img, unsigned* traced, unsigned* neg, int x0, int y0, int method,
                                           int2* out, int cap, int is_hole);

__device__ inline int follow_border(const Bits& img, unsigned* traced, unsigned* neg, int x0, int y0, int method,
                                    int2* out, int cap, int is_hole = 0) {
    if (img.padded) return follow_border_padded(img, traced, neg, x0, y0, method, out, cap, is_hole);
    int n = 0;
    // labels are only ever set, by this single lane: plain OR stores (no read-back needed for the negative label)
    auto mark_neg = [&](int x, int y) {
        const size_t wi = (size_t)y * img.stride + (x >> 5);
        const unsigned m = 1u << (x & 31);
        traced[wi] |= m;
        neg[wi] |= m;
    };
    auto emit = [&](int x, int y) {
        if (n < cap) out[n] = make_int2(x, y);
        n++;
    };
    // first set direction at or after `from` (mod 8), searching towards increasing chain codes: rotate + ctz
    auto next_ccw = [](unsigned nb, int from) {
        const unsigned rot = ((nb | (nb << 8)) >> (from & 7)) & 0xFFu;
        return (from & 7) + __builtin_ctz(rot);  // caller guarantees nb != 0
    };
    unsigned nb = nbr8(img, x0, y0);
    int s_end = is_hole ? 0 : 4, s = s_end;
    // the reference's start loop tries s_end - 1, s_end - 2, ... and gives up when it is back at s_end WITHOUT testing that
    // direction: mask it out (the scan guarantees it is clear anyway)
    const unsigned nb0 = nb & ~(1u << s_end);
    if (nb0 == 0u) {  // isolated pixel
        mark_neg(x0, y0);
        emit(x0, y0);
        return n;
    }
    {   // clockwise search for the first neighbour: s = s_end - 1, s_end - 2, ... (mod 8)
        const unsigned mir = __brev(nb0) >> 24;                   // bit k of mir = bit (7 - k) of nb0
        const int from = (8 - s_end) & 7;                         // direction s_end - 1  <->  mirrored index 8 - s_end
        const unsigned rot = ((mir | (mir << 8)) >> from) & 0xFFu;
        const int k = __builtin_ctz(rot);
        s = (s_end - 1 - k) & 7;
    }
    const int x1 = x0 + code_dx(s), y1 = y0 + code_dy(s);
    int x3 = x0, y3 = y0;
    int prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        nb = nbr8(img, x3, y3);
        if (nb == 0u) break;       // cannot happen (we came from a neighbour); guards the ctz below
        s = next_ccw(nb, s + 1) & 7;   // counter-clockwise search for the next border pixel, then "s &= 7"
        if ((unsigned)(s - 1) < (unsigned)s_end) {
            mark_neg(x3, y3);
        } else {
            const size_t wi = (size_t)y3 * img.stride + (x3 >> 5);
            traced[wi] |= 1u << (x3 & 31);   // positive label unless already labelled: OR-ing `traced` alone is exactly that
        }
        if (s != prev_s || method == 1) {
            emit(x3, y3);
            prev_s = s;
        }
        const int x4 = x3 + code_dx(s), y4 = y3 + code_dy(s);
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
        x3 = x4; y3 = y4;
        s = (s + 4) & 7;
    }
    return n;
}

// follow_border for a PADDED plane (Bits::padded): same labels, same emitted points, ~4x fewer instructions per step --
// the walk is one lane executing a dependent chain, so its speed is its instruction count.  Three 64-bit windows (two
// adjacent words, always in bounds thanks to the padding) give the 3x3 neighbourhood without a branch; marks are
// fire-and-forget ORs; the only data-dependent branch left is the emit.
__device__ inline int follow_border_padded(const Bits& img, unsigned* traced, unsigned* neg, int x0, int y0, int method,
                                           int2* out, int cap, int is_hole) {
    const int pw = img.stride;
    const unsigned* base = img.w;  // word [1][1]: pixel (0,0) is bit 0 of base[0]; base[-1] and base[-pw] are padding
    auto nbr = [&](int x, int y) -> unsigned {
        const int px = x + 31;                       // bit index of pixel x-1 counted from the left padding word
        const int wi = y * pw + (px >> 5) - 1, sh = px & 31;
        const unsigned long long up = ((unsigned long long)base[wi - pw + 1] << 32) | base[wi - pw];
        const unsigned long long mid = ((unsigned long long)base[wi + 1] << 32) | base[wi];
        const unsigned long long dn = ((unsigned long long)base[wi + pw + 1] << 32) | base[wi + pw];
        const unsigned u = (unsigned)(up >> sh) & 7u, m = (unsigned)(mid >> sh) & 7u, d = (unsigned)(dn >> sh) & 7u;
        return (m >> 2) | ((u >> 2) << 1) | (((u >> 1) & 1u) << 2) | ((u & 1u) << 3) | ((m & 1u) << 4) | ((d & 1u) << 5) |
               (((d >> 1) & 1u) << 6) | ((d >> 2) << 7);
    };
    int n = 0;
    unsigned nb = nbr(x0, y0);
    int s_end = is_hole ? 0 : 4, s;
    const unsigned nb0 = nb & ~(1u << s_end);
    if (nb0 == 0u) {  // isolated pixel
        const int wi = y0 * pw + (x0 >> 5);
        const unsigned m = 1u << (x0 & 31);
        traced[wi] |= m; neg[wi] |= m;
        if (n < cap) out[n] = make_int2(x0, y0);
        return 1;
    }
    {
        const unsigned mir = __brev(nb0) >> 24;
        const unsigned rot = ((mir | (mir << 8)) >> ((8 - s_end) & 7)) & 0xFFu;
        s = (s_end - 1 - __builtin_ctz(rot)) & 7;
    }
    const int x1 = x0 + code_dx(s), y1 = y0 + code_dy(s);
    int x3 = x0, y3 = y0, prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        nb = nbr(x3, y3);
        if (nb == 0u) break;
        const int from = (s + 1) & 7;
        s = (from + __builtin_ctz(((nb | (nb << 8)) >> from) & 0xFFu)) & 7;
        const int wi = y3 * pw + (x3 >> 5);
        const unsigned m = 1u << (x3 & 31);
        // result-less atomics = fire-and-forget ds_or: no read-modify-write round trip in the dependent chain
        atomicOr(&traced[wi], m);
        atomicOr(&neg[wi], (unsigned)(s - 1) < (unsigned)s_end ? m : 0u);
        if (s != prev_s || method == 1) {
            if (n < cap) out[n] = make_int2(x3, y3);
            n++;
            prev_s = s;
        }
        const int x4 = x3 + code_dx(s), y4 = y3 + code_dy(s);
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
        x3 = x4; y3 = y4;
        s = (s + 4) & 7;
    }
    return n;
}

// RETR_EXTERNAL scan of rows [y_lo, y_hi] by one wavefront (all 64 lanes must call).  `traced`/`neg` must be zero in
// that row range on entry.  stride <= 64 words (maps up to 2048 columns).
//
// A border start is a set pixel with a clear west neighbour that is not labelled yet and whose nearest labelled pixel to
// the left in the row (lnbd) does not carry a positive label.  A mid-episode explored area has a handful of outer borders
// but thousands of run starts (every ragged row, every pillar), so the candidates are NOT visited one by one: each lane
// decides all the run starts of its word at once -- "the nearest label below this bit is positive" is a flood of the
// positive labels upwards through the unlabelled bits (5 shift/AND/OR steps), with the state entering the word taken from
// the nearest lower word that holds a label (two ballots) -- and the wavefront only stops at rows where a start survives.
// Rows without a surviving start cost three LDS reads and a ballot; after a walk the row is re-evaluated with the new
// labels, for the candidates to the right of the one just traced (the sequential scan never revisits earlier ones).
__device__ inline unsigned fill_up_through(unsigned seed, unsigned open) {
    // bits reachable from `seed` moving towards higher bit positions through consecutive `open` bits (seed bits included)
    unsigned f = seed, m = open;
    f |= m & (f << 1); m &= m << 1;
    f |= m & (f << 2); m &= m << 2;
    f |= m & (f << 4); m &= m << 4;
    f |= m & (f << 8); m &= m << 8;
    f |= m & (f << 16);
    return f;
}

__device__ inline void scan_external(const Bits& img, unsigned* traced, unsigned* neg, int y_lo, int y_hi, int method,
                                     ContourSink& sink) {
    const int lane = threadIdx.x & 63;
    if (y_lo < 0) y_lo = 0;
    if (y_hi > img.rows - 1) y_hi = img.rows - 1;
    for (int y = y_lo; y <= y_hi; y++) {
        const unsigned* row = img.w + (size_t)y * img.stride;
        const size_t roww = (size_t)y * img.stride;
        const unsigned w = lane < img.stride ? row[lane] : 0u;
        const unsigned left = __shfl_up(w, 1, 64);
        const unsigned carry = lane > 0 ? (left >> 31) : 0u;
        const unsigned starts = w & ~((w << 1) | carry);
        if (__ballot(starts != 0u) == 0ull) continue;
        int x_done = -1;   // candidates at or left of this column have been decided
        for (;;) {
            const unsigned tw = lane < img.stride ? traced[roww + lane] : 0u;
            const unsigned ng = lane < img.stride ? neg[roww + lane] : 0u;
            const unsigned pos = tw & ~ng;
            // state entering this word: label kind of the highest labelled bit of the nearest lower word that has one
            const unsigned long long have = __ballot(tw != 0u);
            const unsigned long long top_pos = __ballot(tw != 0u && ((pos >> (31 - __builtin_clz(tw | 1u))) & 1u));
            const unsigned long long lower = have & ((1ull << lane) - 1ull);
            unsigned enter = 0u;
            if (lower) enter = (unsigned)(top_pos >> (63 - __builtin_clzll(lower))) & 1u;
            // inside[x] = the nearest labelled bit below x (in this word, else the entering state) is positive
            unsigned seed = pos << 1;                       // a positive label at bit b covers bits b+1 ...
            if (enter) seed |= 1u;                          // ... and the entering state covers bit 0 ...
            const unsigned inside = fill_up_through(seed & ~tw, ~tw);   // ... upwards through unlabelled bits only
            unsigned need = starts & ~tw & ~inside;
            if (x_done >= 0) {                              // only candidates to the right of the last traced start
                const int wd = x_done >> 5;
                if (lane < wd) need = 0u;
                else if (lane == wd) need &= ~((2u << (x_done & 31)) - 1u);
            }
            const unsigned long long any = __ballot(need != 0u);
            if (!any) break;
            const int L = __builtin_ctzll(any);
            const int x = L * 32 + __builtin_ctz(__shfl(need, L, 64));
            int n = 0;
            if (lane == 0) {
                const int room = sink.cap_pts - sink.n_pts;
#ifdef VLFM_PHASE_TIMING
                const long long t0_ = wall_clock64();
#endif
                n = follow_border(img, traced, neg, x, y, method, sink.pts + sink.n_pts, room > 0 ? room : 0);
#ifdef VLFM_PHASE_TIMING
                g_walk_ticks += wall_clock64() - t0_; g_walk_points += n; g_walk_calls += 1;
#endif
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            n = __shfl(n, 0, 64);
            if (sink.n_contours < sink.cap_contours && sink.n_pts + n <= sink.cap_pts) {
                if (lane == 0) { sink.start[sink.n_contours] = sink.n_pts; sink.len[sink.n_contours] = n; }
            } else {
                sink.overflow = 1;
            }
            sink.n_contours++;
            sink.n_pts += n;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            x_done = x;
        }
    }
}

// RETR_LIST / RETR_TREE scan (every outer AND hole border, discovery order) of a whole image by one wavefront; the
// hierarchy is not produced (no caller on the path reads it: img_utils.py:377 discards it).  Border starts, per row and in
// x order (cvFindNextContour): an unlabelled set pixel whose west neighbour is clear starts an OUTER border; a clear
// pixel whose west neighbour is set and does not carry a negative label starts a HOLE border at that west neighbour.
// `traced`/`neg` must be zero on entry.  stride <= 64 words.
__device__ inline void scan_list(const Bits& img, unsigned* traced, unsigned* neg, int method, ContourSink& sink) {
    const int lane = threadIdx.x & 63;
    for (int y = 0; y < img.rows; y++) {
        const unsigned* row = img.w + (size_t)y * img.stride;
        const unsigned w = lane < img.stride ? row[lane] : 0u;
        const unsigned left = __shfl_up(w, 1, 64);
        const unsigned carry = lane > 0 ? (left >> 31) : 0u;
        const unsigned west = (w << 1) | carry;       // bit x = pixel x-1
        unsigned valid = 0u;                          // columns < cols in this word
        if (lane * 32 < img.cols) valid = (img.cols - lane * 32 >= 32) ? 0xFFFFFFFFu : ((1u << (img.cols - lane * 32)) - 1u);
        unsigned cand = ((w & ~west) | (~w & west)) & valid;
        unsigned long long any = __ballot(cand != 0u);
        while (any) {
            const int L = __builtin_ctzll(any);
            const unsigned cL = __shfl(cand, L, 64);
            const unsigned wL = __shfl(w, L, 64);
            const int bit = __builtin_ctz(cL);
            const int x = L * 32 + bit;
            if (lane == L) cand &= cand - 1;          // consume the candidate
            const int is_hole = !((wL >> bit) & 1u);
            const int xo = x - is_hole;               // origin pixel of the border
            const size_t wi = (size_t)y * img.stride + (xo >> 5);
            const unsigned tb = (traced[wi] >> (xo & 31)) & 1u, nb = (neg[wi] >> (xo & 31)) & 1u;
            const bool start = is_hole ? !(tb && nb) : !tb;
            if (start) {
                int n = 0;
                if (lane == 0) {
                    const int room = sink.cap_pts - sink.n_pts;
                    n = follow_border(img, traced, neg, xo, y, method, sink.pts + sink.n_pts, room > 0 ? room : 0, is_hole);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                n = __shfl(n, 0, 64);
                if (sink.n_contours < sink.cap_contours && sink.n_pts + n <= sink.cap_pts) {
                    if (lane == 0) {
                        sink.start[sink.n_contours] = sink.n_pts;
                        sink.len[sink.n_contours] = n;
                        if (sink.hole) sink.hole[sink.n_contours] = (unsigned char)is_hole;
                    }
                } else {
                    sink.overflow = 1;
                }
                sink.n_contours++;
                sink.n_pts += n;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            any = __ballot(cand != 0u);
        }
    }
}

}  // namespace vlfm
