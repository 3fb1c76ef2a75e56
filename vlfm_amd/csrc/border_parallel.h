// border_parallel.h -- Suzuki-Abe border following by a whole workgroup (gfx950), for the padded LDS windows of bitmap.h.
//
// follow_border() is one lane executing a dependent chain: ~600 cycles per border pixel, i.e. 0.7 ms for the 2 300-pixel outline
// of a mid-episode explored area, in explored_select_kernel and again in frontier_kernel (tools/phase_probe.py) -- while 1023 of
// the 1024 lanes wait.  The walk is a deterministic map on STATES (pixel p, direction s_back to the pixel it was entered from):
//     f(p, s_back) = (p', s_back'),  p' = first set neighbour of p counter-clockwise after s_back,  s_back' = direction p' -> p.
// f is injective (the predecessor of (p', s_back') is recovered by the clockwise search from p), so every valid state lies on a
// cycle, and the border OpenCV traces from a start pixel i0 is exactly the cycle through state0 = (i0, direction to i1): the loop
// of cvFindContours ends when it is about to step from i1 to i0, i.e. when it would re-enter state0.  Hence:
//   1. once per image: enumerate the border pixels (set, with a clear 8-neighbour) and their valid states in raster order
//      (two workgroup prefix sums) and store every state's successor -- all local 3x3 work, fully parallel;
//   2. per traced border: LIST RANKING by pointer jumping from state0 (log2(#states) rounds, every state in parallel) gives each
//      state on the cycle its position in the chain; states of other borders never reach state0 and are ignored;
//   3. points, the CHAIN_APPROX_SIMPLE filter (a state emits iff its outgoing direction differs from the one it was entered
//      with, which is s_back ^ 4: local) and the +/- labels are written by all lanes at once.
// Identical output to follow_border (same points, same order, same labels): the golden fixtures and property tests of
// tests/ do not distinguish the two.  Borders that close within a few dozen steps are still walked by one lane (cheaper than
// a ranking), and any situation the tables cannot express (capacity, a successor that is not a border pixel) falls back to it.
#pragma once
#include "bitmap.h"

namespace vlfm {

// 3x3 neighbourhood of (x, y) in a padded plane as a chain-code mask (bit s = neighbour in direction s)
__device__ inline unsigned nbr8_padded(const unsigned* base, int pw, int x, int y) {
    const int px = x + 31;
    const int wi = y * pw + (px >> 5) - 1, sh = px & 31;
    const unsigned long long up = ((unsigned long long)base[wi - pw + 1] << 32) | base[wi - pw];
    const unsigned long long mid = ((unsigned long long)base[wi + 1] << 32) | base[wi];
    const unsigned long long dn = ((unsigned long long)base[wi + pw + 1] << 32) | base[wi + pw];
    const unsigned u = (unsigned)(up >> sh) & 7u, m = (unsigned)(mid >> sh) & 7u, d = (unsigned)(dn >> sh) & 7u;
    return (m >> 2) | ((u >> 2) << 1) | (((u >> 1) & 1u) << 2) | ((u & 1u) << 3) | ((m & 1u) << 4) | ((d & 1u) << 5) |
           (((d >> 1) & 1u) << 6) | ((d >> 2) << 7);
}

struct WalkTables {          // per-environment global scratch (L2-resident: a few hundred KB)
    unsigned* bmask;         // [wrows * wwords]  border pixels of the window
    int* wprefix;            // [wrows * wwords]  border pixels in front of this word, raster order
    int* pixbase;            // [cap_bp]          first state of border pixel #rank
    int* next;               // [cap_states]      successor state
    unsigned* sinfo;         // [cap_states]      x | y << 11 | s_back << 22 | s_out << 25
    unsigned* jd0;           // [cap_states]      pointer-jumping buffers: dist << 16 | jump
    unsigned* jd1;
    int cap_bp, cap_states;  // cap_states <= 65535
    int wrows, wwords;
    int n_states = 0, ok = 0;   // results of wg_build_walk_tables (uniform)
};

constexpr int WG_SH_INTS = 48;   // shared scratch the functions below need (ints)

// exclusive prefix sums of two ints over the workgroup (<= 16 wavefronts); sh: 32 ints
__device__ inline void wg_scan2(int a, int b, int* sh, int& a_ex, int& b_ex, int& a_tot, int& b_tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int ia = a, ib = b;
    for (int off = 1; off < 64; off <<= 1) {
        const int ta = __shfl_up(ia, off, 64), tb = __shfl_up(ib, off, 64);
        if (lane >= off) { ia += ta; ib += tb; }
    }
    if (lane == 63) { sh[wave] = ia; sh[16 + wave] = ib; }
    __syncthreads();
    int wa = 0, wb = 0, ta = 0, tb = 0;
    for (int w = 0; w < nw; w++) {
        const int va = sh[w], vb = sh[16 + w];
        if (w < wave) { wa += va; wb += vb; }
        ta += va; tb += vb;
    }
    a_ex = wa + ia - a; b_ex = wb + ib - b; a_tot = ta; b_tot = tb;
    __syncthreads();
}

// Step 1: border pixels, their states and every state's successor.  All threads of the workgroup; img is a PADDED view.
__device__ inline void wg_build_walk_tables(const Bits& img, WalkTables& T, int* sh) {
    const int tid = threadIdx.x, nth = blockDim.x;
    const unsigned* base = img.w;
    const int pw = img.stride, wwords = T.wwords, W = T.wrows * T.wwords;
    const int per = (W + nth - 1) / nth;
    const int k0 = min(tid * per, W), k1 = min(k0 + per, W);
    auto border_word = [&](int ly, int lw) -> unsigned {   // set pixels with at least one clear 8-neighbour
        const unsigned* r = base + ly * pw + lw;
        auto h = [](const unsigned* q) { const unsigned c = q[0]; return c & ((c << 1) | (q[-1] >> 31)) & ((c >> 1) | (q[1] << 31)); };
        return r[0] & ~(h(r - pw) & h(r) & h(r + pw));
    };
    if (tid == 0) sh[32] = 0;
    int nb = 0, ns = 0;
    for (int k = k0; k < k1; k++) {
        const int ly = k / wwords, lw = k - ly * wwords;
        unsigned bm = border_word(ly, lw);
        nb += __builtin_popcount(bm);
        while (bm) {
            const int bit = __builtin_ctz(bm);
            bm &= bm - 1;
            ns += __builtin_popcount(nbr8_padded(base, pw, lw * 32 + bit, ly));
        }
    }
    int nb_ex, ns_ex, nb_tot, ns_tot;
    wg_scan2(nb, ns, sh, nb_ex, ns_ex, nb_tot, ns_tot);
    T.n_states = ns_tot;
    T.ok = nb_tot <= T.cap_bp && ns_tot <= T.cap_states && ns_tot > 0;
    if (!T.ok) return;
    {
        int rank = nb_ex, sb = ns_ex;
        for (int k = k0; k < k1; k++) {
            const int ly = k / wwords, lw = k - ly * wwords;
            unsigned bm = border_word(ly, lw);
            T.bmask[k] = bm;
            T.wprefix[k] = rank;
            while (bm) {
                const int bit = __builtin_ctz(bm);
                bm &= bm - 1;
                T.pixbase[rank++] = sb;
                sb += __builtin_popcount(nbr8_padded(base, pw, lw * 32 + bit, ly));
            }
        }
    }
    __threadfence();
    __syncthreads();
    {
        int sb = ns_ex, bad = 0;
        for (int k = k0; k < k1; k++) {
            const int ly = k / wwords, lw = k - ly * wwords;
            unsigned bm = T.bmask[k];
            while (bm) {
                const int bit = __builtin_ctz(bm);
                bm &= bm - 1;
                const int x = lw * 32 + bit;
                const unsigned nbm = nbr8_padded(base, pw, x, ly);
                unsigned dirs = nbm;
                int t = 0;
                while (dirs) {
                    const int d = __builtin_ctz(dirs);   // this state was entered from direction d
                    dirs &= dirs - 1;
                    const int from = (d + 1) & 7;
                    const int s = (from + __builtin_ctz(((nbm | (nbm << 8)) >> from) & 0xFFu)) & 7;
                    const int x2 = x + code_dx(s), y2 = ly + code_dy(s), sb2 = (s + 4) & 7;
                    const int k2 = y2 * wwords + (x2 >> 5);
                    const unsigned bm2 = T.bmask[k2], bit2 = 1u << (x2 & 31);
                    int id2 = sb + t;
                    if (bm2 & bit2) {
                        const int rank2 = T.wprefix[k2] + __builtin_popcount(bm2 & (bit2 - 1u));
                        const unsigned nb2 = nbr8_padded(base, pw, x2, y2);
                        id2 = T.pixbase[rank2] + __builtin_popcount(nb2 & ((1u << sb2) - 1u));
                    } else {
                        bad = 1;   // the walk would enter an interior pixel: not expressible here
                    }
                    T.next[sb + t] = id2;
                    T.sinfo[sb + t] = (unsigned)x | ((unsigned)ly << 11) | ((unsigned)d << 22) | ((unsigned)s << 25);
                    t++;
                }
                sb += t;
            }
        }
        if (bad) atomicOr(&sh[32], 1);
    }
    __threadfence();
    __syncthreads();
    if (sh[32]) T.ok = 0;
    __syncthreads();
}

// follow_border_padded with a step budget: -1 when the border did not close within max_steps (labels written so far are a
// subset of the final ones; the caller redoes the border)
__device__ inline int follow_border_short(const Bits& img, unsigned* traced, unsigned* neg, int x0, int y0, int method, int2* out,
                                          int cap, int max_steps) {
    const int pw = img.stride;
    const unsigned* base = img.w;
    int n = 0;
    unsigned nb = nbr8_padded(base, pw, x0, y0);
    int s_end = 4, s;
    const unsigned nb0 = nb & ~(1u << s_end);
    if (nb0 == 0u) {
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
    for (int step = 0; step < max_steps; step++) {
        s_end = s;
        nb = nbr8_padded(base, pw, x3, y3);
        if (nb == 0u) return n;
        const int from = (s + 1) & 7;
        s = (from + __builtin_ctz(((nb | (nb << 8)) >> from) & 0xFFu)) & 7;
        const int wi = y3 * pw + (x3 >> 5);
        const unsigned m = 1u << (x3 & 31);
        atomicOr(&traced[wi], m);
        atomicOr(&neg[wi], (unsigned)(s - 1) < (unsigned)s_end ? m : 0u);
        if (s != prev_s || method == 1) {
            if (n < cap) out[n] = make_int2(x3, y3);
            n++;
            prev_s = s;
        }
        const int x4 = x3 + code_dx(s), y4 = y3 + code_dy(s);
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) return n;
        x3 = x4; y3 = y4;
        s = (s + 4) & 7;
    }
    return -1;
}

// Steps 2 + 3 for the outer border starting at (x0, y0) (scan guarantees: set, west neighbour clear, unlabelled).  All threads;
// returns the number of emitted points (uniform).  Window coordinates, like follow_border on the same view.
__device__ inline int wg_follow_border(const Bits& img, const WalkTables& T, unsigned* traced, unsigned* neg, int x0, int y0,
                                       int method, int2* out, int cap, int* sh) {
    const int tid = threadIdx.x, nth = blockDim.x;
    const unsigned* base = img.w;
    const int pw = img.stride;
    if (tid == 0) {
        int n = follow_border_short(img, traced, neg, x0, y0, method, out, cap, 48);
        sh[33] = n;
        if (n < 0) {
            const unsigned nb = nbr8_padded(base, pw, x0, y0);
            const unsigned mir = __brev(nb & ~(1u << 4)) >> 24;
            const unsigned rot = ((mir | (mir << 8)) >> 4) & 0xFFu;
            const int s = (3 - __builtin_ctz(rot)) & 7;                 // direction of i1 = s_back of state0
            const int k = y0 * T.wwords + (x0 >> 5);
            const int rank = T.wprefix[k] + __builtin_popcount(T.bmask[k] & ((1u << (x0 & 31)) - 1u));
            const int id0 = T.pixbase[rank] + __builtin_popcount(nb & ((1u << s) - 1u));
            sh[34] = id0;
            sh[35] = T.next[id0];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    const int n_short = sh[33];
    if (n_short >= 0) { __syncthreads(); return n_short; }
    const int id0 = sh[34], succ0 = sh[35], N = T.n_states;
    __syncthreads();
    // ---- list ranking: after the last round jump == id0 exactly for the states on state0's cycle, dist = steps to reach it
    unsigned* A = T.jd0;
    unsigned* B = T.jd1;
    for (int i = tid; i < N; i += nth) A[i] = i == id0 ? (unsigned)id0 : ((1u << 16) | (unsigned)T.next[i]);
    __threadfence();
    __syncthreads();
    const int rounds = 32 - __builtin_clz((unsigned)N);
    for (int r = 0; r < rounds; r++) {
        for (int i = tid; i < N; i += nth) {
            unsigned a = A[i];
            const unsigned j = a & 0xFFFFu;
            if ((int)j != id0) {
                const unsigned b = A[j];
                a = ((a & 0xFFFF0000u) + (b & 0xFFFF0000u)) | (b & 0xFFFFu);   // distances add (overflow only off the cycle)
            }
            B[i] = a;
        }
        __threadfence();
        __syncthreads();
        unsigned* t = A; A = B; B = t;
    }
    const int L = (succ0 == id0) ? 1 : (int)(A[succ0] >> 16) + 1;
    // ---- labels and points
    if (method == 1) {
        for (int i = tid; i < N; i += nth) {
            const unsigned a = A[i];
            if ((int)(a & 0xFFFFu) != id0) continue;
            const int pos = i == id0 ? 0 : L - (int)(a >> 16);
            const unsigned info = T.sinfo[i];
            const int x = info & 2047u, y = (info >> 11) & 2047u, sb = (info >> 22) & 7u, so = (info >> 25) & 7u;
            const int wi = y * pw + (x >> 5);
            const unsigned m = 1u << (x & 31);
            atomicOr(&traced[wi], m);
            if ((unsigned)(so - 1) < (unsigned)sb) atomicOr(&neg[wi], m);
            if (pos < cap) out[pos] = make_int2(x, y);
        }
        __threadfence();
        __syncthreads();
        return L;
    }
    // CHAIN_APPROX_SIMPLE: positions into B (free now), then an ordered compaction of the flagged ones
    for (int i = tid; i < N; i += nth) {
        const unsigned a = A[i];
        if ((int)(a & 0xFFFFu) != id0) continue;
        const int pos = i == id0 ? 0 : L - (int)(a >> 16);
        const unsigned info = T.sinfo[i];
        const int x = info & 2047u, y = (info >> 11) & 2047u, sb = (info >> 22) & 7u, so = (info >> 25) & 7u;
        const int wi = y * pw + (x >> 5);
        const unsigned m = 1u << (x & 31);
        atomicOr(&traced[wi], m);
        if ((unsigned)(so - 1) < (unsigned)sb) atomicOr(&neg[wi], m);
        B[pos] = (info & 0x3FFFFFu) | (so != (sb ^ 4) ? 0x80000000u : 0u);
    }
    __threadfence();
    __syncthreads();
    const int per = (L + nth - 1) / nth;
    const int p0 = min(tid * per, L), p1 = min(p0 + per, L);
    int cnt = 0;
    for (int p = p0; p < p1; p++) cnt += B[p] >> 31;
    int ex, dummy_ex, tot, dummy_tot;
    wg_scan2(cnt, 0, sh, ex, dummy_ex, tot, dummy_tot);
    for (int p = p0; p < p1; p++) {
        const unsigned v = B[p];
        if (v >> 31) {
            if (ex < cap) out[ex] = make_int2((int)(v & 2047u), (int)((v >> 11) & 2047u));
            ex++;
        }
    }
    __threadfence();
    __syncthreads();
    return tot;
}

// One step of scan_external's raster scan, resumable: the next border start at or after row y (to the right of x_done in
// that row).  One wavefront; y / x_done are its scan position and are advanced.
__device__ inline bool scan_next_start(const Bits& img, const unsigned* traced, const unsigned* neg, int y_hi, int& y, int& x_done,
                                       int& x_out) {
    const int lane = threadIdx.x & 63;
    for (; y <= y_hi; y++, x_done = -1) {
        const unsigned* row = img.w + (size_t)y * img.stride;
        const size_t roww = (size_t)y * img.stride;
        const unsigned w = lane < img.stride ? row[lane] : 0u;
        const unsigned left = __shfl_up(w, 1, 64);
        const unsigned carry = lane > 0 ? (left >> 31) : 0u;
        const unsigned starts = w & ~((w << 1) | carry);
        if (__ballot(starts != 0u) == 0ull) continue;
        const unsigned tw = lane < img.stride ? traced[roww + lane] : 0u;
        const unsigned ng = lane < img.stride ? neg[roww + lane] : 0u;
        const unsigned pos = tw & ~ng;
        const unsigned long long have = __ballot(tw != 0u);
        const unsigned long long top_pos = __ballot(tw != 0u && ((pos >> (31 - __builtin_clz(tw | 1u))) & 1u));
        const unsigned long long lower = have & ((1ull << lane) - 1ull);
        unsigned enter = 0u;
        if (lower) enter = (unsigned)(top_pos >> (63 - __builtin_clzll(lower))) & 1u;
        unsigned seed = pos << 1;
        if (enter) seed |= 1u;
        const unsigned inside = fill_up_through(seed & ~tw, ~tw);
        unsigned need = starts & ~tw & ~inside;
        if (x_done >= 0) {
            const int wd = x_done >> 5;
            if (lane < wd) need = 0u;
            else if (lane == wd) need &= ~((2u << (x_done & 31)) - 1u);
        }
        const unsigned long long any = __ballot(need != 0u);
        if (!any) continue;
        const int L = __builtin_ctzll(any);
        x_out = L * 32 + __builtin_ctz(__shfl(need, L, 64));
        x_done = x_out;
        return true;
    }
    return false;
}

// RETR_EXTERNAL scan of a padded LDS window by the whole workgroup (all threads must call; labels zero on entry).
// sink's counters end up identical in every thread.  sh: WG_SH_INTS ints of shared memory.
__device__ inline void wg_scan_external(const Bits& img, unsigned* traced, unsigned* neg, int method, ContourSink& sink,
                                        WalkTables& T, int* sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    wg_build_walk_tables(img, T, sh);
    int y = 0, x_done = -1;
    for (;;) {
        if (wave == 0) {
            int x = 0;
            const bool f = scan_next_start(img, traced, neg, img.rows - 1, y, x_done, x);
            if (lane == 0) { sh[40] = f; sh[41] = x; sh[42] = y; }
        }
        __syncthreads();
        if (!sh[40]) break;
        const int x = sh[41], yy = sh[42];
        const int room = sink.cap_pts - sink.n_pts > 0 ? sink.cap_pts - sink.n_pts : 0;
        int n;
        if (T.ok) {
            n = wg_follow_border(img, T, traced, neg, x, yy, method, sink.pts + sink.n_pts, room, sh);
        } else {
            if (tid == 0) sh[43] = follow_border(img, traced, neg, x, yy, method, sink.pts + sink.n_pts, room);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            n = sh[43];
        }
        if (sink.n_contours < sink.cap_contours && sink.n_pts + n <= sink.cap_pts) {
            if (tid == 0) { sink.start[sink.n_contours] = sink.n_pts; sink.len[sink.n_contours] = n; }
        } else {
            sink.overflow = 1;
        }
        sink.n_contours++;
        sink.n_pts += n;
        __threadfence();
        __syncthreads();
    }
}

}  // namespace vlfm
