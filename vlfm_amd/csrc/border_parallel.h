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
// tests/ do not distinguish the two.  Borders that close within WG_SERIAL_STEPS steps are still walked by one lane (cheaper than
// a ranking), and any situation the tables cannot express (capacity, a successor that is not a border pixel) falls back to it.
#pragma once
#include "bitmap.h"

namespace vlfm {

#ifdef VLFM_PHASE_TIMING
__device__ long long g_walk_clk[16];   // one workgroup's stamps inside the parallel follower (the last call wins)
__device__ int g_walk_block;
#define WALK_STAMP(k)                                                                              \
    do {                                                                                           \
        if ((int)blockIdx.x == g_walk_block && threadIdx.x == 0) g_walk_clk[k] = wall_clock64();     \
    } while (0)
#else
#define WALK_STAMP(k) do {} while (0)
#endif

// Barrier that also orders GLOBAL-memory traffic between the wavefronts of this workgroup.  They share one CU and its L1, so
// workgroup scope is enough; __threadfence() is a device-scope release (L2 write-back on gfx950) and cost ~8 us per
// pointer-jumping round here.
__device__ inline void wg_sync_global() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

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
    unsigned* ljd0 = nullptr;   // optional LDS copies of the two buffers for images with at most lds_states states (a round then
    unsigned* ljd1 = nullptr;   // costs an LDS round trip instead of an L2 one: the fog-of-war windows)
    int lds_states = 0;
    int cap_bp, cap_states;  // cap_states <= 65535
    int wrows, wwords;
    int n_states = 0, ok = 0;   // results of wg_build_walk_tables (uniform)
};

constexpr int WG_SH_INTS = 48;   // shared scratch the functions below need (ints)
// Borders that close within this many steps are walked by one lane (specks, single pixels); measured at 256 envs: 16 beats 64
// (fog of war 0.62 vs 0.71 ms) although a ranking costs ~60 us whatever the border's length.
constexpr int WG_SERIAL_STEPS = 16;

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

// Step 1: border pixels, their states and every state's successor.  All threads of the workgroup; img is a PADDED view whose
// two label planes (lt, ln: same geometry, zero on entry) serve as LDS scratch for the border mask and the per-word rank
// while the tables are built, and are zero again on return.
__device__ inline void wg_build_walk_tables(const Bits& img, unsigned* lt, unsigned* ln, WalkTables& T, int* sh) {
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
    WALK_STAMP(0);
    // a. border mask and (border pixels | states << 8) of every word; words are dealt round-robin so that a ragged row is
    //    everybody's problem (with contiguous chunks the thread that owned it took 30 us)
    int* wcount = T.wprefix;                            // packed counts now, ranks after the scan
    int* wsbase = reinterpret_cast<int*>(T.sinfo);      // first state of a word's first border pixel (free until step d)
    for (int k = tid; k < W; k += nth) {
        const int ly = k / wwords, lw = k - ly * wwords;
        const unsigned bm0 = border_word(ly, lw);
        unsigned bm = bm0;
        int ns = 0;
        while (bm) {
            const int bit = __builtin_ctz(bm);
            bm &= bm - 1;
            ns += __builtin_popcount(nbr8_padded(base, pw, lw * 32 + bit, ly));
        }
        lt[ly * pw + lw] = bm0;
        T.bmask[k] = bm0;
        wcount[k] = __builtin_popcount(bm0) | (ns << 8);
    }
    wg_sync_global();
    // b. raster-order prefix sums over the words (contiguous chunks: plain integer adds)
    int nb = 0, ns = 0;
    for (int k = k0; k < k1; k++) { const int v = wcount[k]; nb += v & 255; ns += v >> 8; }
    int nb_ex, ns_ex, nb_tot, ns_tot;
    wg_scan2(nb, ns, sh, nb_ex, ns_ex, nb_tot, ns_tot);
    WALK_STAMP(1);
    T.n_states = ns_tot;
    T.ok = nb_tot <= T.cap_bp && nb_tot <= T.cap_states && ns_tot <= T.cap_states && ns_tot > 0 && W <= T.cap_states;
    if (!T.ok) {
        for (int k = tid; k < W; k += nth) { const int ly = k / wwords, lw = k - ly * wwords; lt[ly * pw + lw] = 0u; }
        __syncthreads();
        return;
    }
    for (int k = k0; k < k1; k++) {
        const int v = wcount[k];
        const int ly = k / wwords, lw = k - ly * wwords;
        T.wprefix[k] = nb_ex;
        ln[ly * pw + lw] = (unsigned)nb_ex;
        wsbase[k] = ns_ex;
        nb_ex += v & 255; ns_ex += v >> 8;
    }
    wg_sync_global();
    // c. first state and coordinates of every border pixel
    int* pixxy = reinterpret_cast<int*>(T.jd0);   // rank -> x | y << 11 (the jumping buffers are free until a border is ranked)
    for (int k = tid; k < W; k += nth) {
        const int ly = k / wwords, lw = k - ly * wwords;
        unsigned bm = lt[ly * pw + lw];
        int rank = (int)ln[ly * pw + lw], sb = wsbase[k];
        while (bm) {
            const int bit = __builtin_ctz(bm);
            bm &= bm - 1;
            const int x = lw * 32 + bit;
            T.pixbase[rank] = sb;
            pixxy[rank] = x | (ly << 11);
            rank++;
            sb += __builtin_popcount(nbr8_padded(base, pw, x, ly));
        }
    }
    wg_sync_global();
    WALK_STAMP(2);
    // d. successors: one border pixel per thread and step; the mask and rank of the successor's word come from LDS, only its
    //    first-state index is a global read
    for (int r = tid; r < nb_tot; r += nth) {
        const int xy = pixxy[r], x = xy & 2047, ly = xy >> 11;
        const int sb = T.pixbase[r];
        const unsigned nbm = nbr8_padded(base, pw, x, ly);
        unsigned dirs = nbm;
        int t = 0;
        while (dirs) {
            const int d = __builtin_ctz(dirs);   // this state was entered from direction d
            dirs &= dirs - 1;
            const int from = (d + 1) & 7;
            const int s = (from + __builtin_ctz(((nbm | (nbm << 8)) >> from) & 0xFFu)) & 7;
            const int x2 = x + code_dx(s), y2 = ly + code_dy(s), sb2 = (s + 4) & 7;
            const int w2 = y2 * pw + (x2 >> 5);
            const unsigned bm2 = lt[w2], bit2 = 1u << (x2 & 31);
            int id2 = sb + t;   // a successor that is not a border pixel: self-loop (such a state is on no traced border; if
                                // one ever were, the ranking would not close and the border is walked serially)
            if (bm2 & bit2) {
                const int rank2 = (int)ln[w2] + __builtin_popcount(bm2 & (bit2 - 1u));
                const unsigned nb2 = nbr8_padded(base, pw, x2, y2);
                id2 = T.pixbase[rank2] + __builtin_popcount(nb2 & ((1u << sb2) - 1u));
            }
            T.next[sb + t] = id2;
            T.sinfo[sb + t] = (unsigned)x | ((unsigned)ly << 11) | ((unsigned)d << 22) | ((unsigned)s << 25);
            t++;
        }
    }
    wg_sync_global();
    for (int k = tid; k < W; k += nth) {   // the label planes are labels again
        const int ly = k / wwords, lw = k - ly * wwords;
        lt[ly * pw + lw] = 0u;
        ln[ly * pw + lw] = 0u;
    }
    __syncthreads();
    WALK_STAMP(3);
#ifdef VLFM_PHASE_TIMING
    if ((int)blockIdx.x == g_walk_block && threadIdx.x == 0) { g_walk_clk[12] = T.ok; g_walk_clk[13] = T.n_states; g_walk_clk[14] = nb_tot; }
#endif
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
        int n = follow_border_short(img, traced, neg, x0, y0, method, out, cap, WG_SERIAL_STEPS);
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
    WALK_STAMP(4);
    // ---- list ranking: after the last round jump == id0 exactly for the states on state0's cycle, dist = steps to reach it
    unsigned* A = N <= T.lds_states ? T.ljd0 : T.jd0;
    unsigned* B = N <= T.lds_states ? T.ljd1 : T.jd1;
    for (int i = tid; i < N; i += nth) A[i] = i == id0 ? (unsigned)id0 : ((1u << 16) | (unsigned)T.next[i]);
    wg_sync_global();
    WALK_STAMP(5);
    // a round: dist[i] += dist[jump[i]], jump[i] = jump[jump[i]] for every state.  The random reads of a thread's states are
    // issued eight at a time (the buffers may not alias, but the compiler cannot know: a plain loop serialises on them); the
    // ranking is complete as soon as state0's successor -- the farthest state of the cycle -- has reached state0.
    const int rounds = 32 - __builtin_clz((unsigned)N);
    for (int r = 0; r < rounds; r++) {
        const unsigned* __restrict__ src = A;
        unsigned* __restrict__ dst = B;
        for (int i0 = tid; i0 < N; i0 += 8 * nth) {
            unsigned a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = i0 + u * nth; a[u] = i < N ? src[i] : (unsigned)id0; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const unsigned j = a[u] & 0xFFFFu; b[u] = (int)j != id0 ? src[j] : 0u; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + u * nth;
                const unsigned j = a[u] & 0xFFFFu;
                if (i < N) dst[i] = (int)j != id0 ? (((a[u] & 0xFFFF0000u) + (b[u] & 0xFFFF0000u)) | (b[u] & 0xFFFFu)) : a[u];
            }
        }
        wg_sync_global();
        unsigned* t = A; A = B; B = t;
        if (succ0 == id0 || (int)(A[succ0] & 0xFFFFu) == id0) break;
    }
    WALK_STAMP(6);
    if (succ0 != id0 && (int)(A[succ0] & 0xFFFFu) != id0) {   // the cycle did not close inside the tables: one lane walks it
        if (tid == 0) sh[33] = follow_border(img, traced, neg, x0, y0, method, out, cap);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        const int n_serial = sh[33];
        __syncthreads();
        return n_serial;
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
        wg_sync_global();
        WALK_STAMP(7);
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
    wg_sync_global();
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
    wg_sync_global();
    WALK_STAMP(7);
    return tot;
}

// scan_external's decision for ONE row: the first border start to the right of x_done (-1: whole row) given the labels so far.
// One wavefront.
__device__ inline bool row_next_start(const Bits& img, const unsigned* traced, const unsigned* neg, int y, int x_done, int& x_out) {
    const int lane = threadIdx.x & 63;
    const unsigned* row = img.w + (size_t)y * img.stride;
    const size_t roww = (size_t)y * img.stride;
    const unsigned w = lane < img.stride ? row[lane] : 0u;
    const unsigned left = __shfl_up(w, 1, 64);
    const unsigned carry = lane > 0 ? (left >> 31) : 0u;
    const unsigned starts = w & ~((w << 1) | carry);
    if (__ballot(starts != 0u) == 0ull) return false;
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
    if (!any) return false;
    const int L = __builtin_ctzll(any);
    x_out = L * 32 + __builtin_ctz(__shfl(need, L, 64));
    return true;
}

// scan_external's decision for ONE row by ONE LANE (same rule as row_next_start, the words taken left to right with the
// entering state carried along): the first border start to the right of x_done (-1: whole row), or -1.  A row per wavefront cost
// ~60 instructions per row and wavefront; a row per lane costs them per 64 rows.  (The callers keep the padded row stride odd,
// so that 64 lanes reading the same word of 64 consecutive rows hit 32 different banks.)
__device__ __forceinline__ int row_next_start_lane(const Bits& img, const unsigned* traced, const unsigned* neg, int y, int x_done) {
    const unsigned* row = img.w + (size_t)y * img.stride;
    const unsigned* tr = traced + (size_t)y * img.stride;
    const unsigned* ngp = neg + (size_t)y * img.stride;
    const int words = (img.cols + 31) >> 5;
    unsigned prev = 0u, enter = 0u;
    for (int wi = 0; wi < words; wi++) {
        const unsigned w = row[wi], tw = tr[wi];
        unsigned need = w & ~((w << 1) | (prev >> 31)) & ~tw;     // run starts without a label
        prev = w;
        if ((need | tw) == 0u) continue;
        const unsigned pos = tw ? (tw & ~ngp[wi]) : 0u;
        if (need) {
            // inside[x] = the nearest labelled bit below x (in this word, else the entering state) is positive
            unsigned seed = pos << 1;
            if (enter) seed |= 1u;
            need &= ~fill_up_through(seed & ~tw, ~tw);
            if (x_done >= 0) {
                const int wd = x_done >> 5;
                if (wi < wd) need = 0u;
                else if (wi == wd) need &= ~((2u << (x_done & 31)) - 1u);
            }
            if (need) return wi * 32 + __builtin_ctz(need);
        }
        if (tw) enter = (pos >> (31 - __builtin_clz(tw))) & 1u;
    }
    return -1;
}

// The next border start in raster order at or after row y (right of x_done in row y itself), by the whole workgroup: one row per
// lane (labels only change between searches, so the rows can be judged independently), the earliest hit wins.  One barrier
// per search -- the block-wise, row-per-wavefront search this replaces spent 18 us on the LAST search of a 450-row window
// (nothing left to find: 7 blocks x 3 barriers), in each of the four scans of a step.  `slot` alternates between calls
// (sh[44], sh[45]).
__device__ __forceinline__ bool wg_next_start(const Bits& img, const unsigned* traced, const unsigned* neg, int y, int x_done, int* sh,
                                              int slot, int& fx, int& fy) {
    const int tid = threadIdx.x, lane = tid & 63, nth = blockDim.x;
    int* best = &sh[44 + (slot & 1)];
    if (slot == 0) {                      // first search of a scan: both slots
        if (tid == 0) { sh[44] = 0x7FFFFFFF; sh[45] = 0x7FFFFFFF; }
        __syncthreads();
    }
    for (int r0 = y; r0 < img.rows; r0 += nth) {
        const int r = r0 + tid;
        const int x = r < img.rows ? row_next_start_lane(img, traced, neg, r, r == y ? x_done : -1) : -1;
        const unsigned long long hit = __ballot(x >= 0);
        if (hit) {
            if (lane == __builtin_ctzll(hit)) atomicMin(best, (r << 11) | x);
            break;
        }
    }
    __syncthreads();
    const int v = *best;
    // the slot of the NEXT search is reset now: every path from here to that search's atomicMin passes a barrier (tracing a
    // border always does), and nobody reads that slot before
    if (tid == 0) sh[44 + ((slot + 1) & 1)] = 0x7FFFFFFF;
    if (v == 0x7FFFFFFF) return false;
    fx = v & 2047; fy = v >> 11;
    return true;
}

// RETR_EXTERNAL scan of a padded LDS window by the whole workgroup (all threads must call; labels zero on entry).
// sink's counters end up identical in every thread.  sh: WG_SH_INTS ints of shared memory.
__device__ inline void wg_scan_external(const Bits& img, unsigned* traced, unsigned* neg, int method, ContourSink& sink,
                                        WalkTables& T, int* sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    wg_build_walk_tables(img, traced, neg, T, sh);
    int y = 0, x_done = -1;              // scan position (uniform)
    for (int call = 0;; call++) {
        int fx = 0, fy = 0;
        const bool found = wg_next_start(img, traced, neg, y, x_done, sh, call, fx, fy);
        if (!found) break;
        y = fy; x_done = fx;
        const int x = fx, yy = fy;
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
        wg_sync_global();
    }
}


// =====================================================================================================================
// Round 4: the follower with ALL its tables in LDS, every border of the image ranked at once.
//
// What the form above costs (tools/phase_probe.py, 256 environments): ~45 us to build the tables and ~47 us to rank ONE border of a
// mid-episode explored area -- a chain of ~40 barrier-separated passes, each a round trip through L2 because the tables live in
// global memory, over ~5 states per border pixel of which the traced border uses one.  This form:
//   * STATES are (border pixel p, direction d of a BORDER-pixel neighbour): a traced border only ever visits border pixels (set
//     pixels with a clear 8-neighbour; checked on 400 000 chain points of outer and hole borders), so states entered from an
//     interior pixel are on no border -- ~3 states per border pixel instead of ~5.3.  A state whose successor pixel is not a
//     border pixel (walking a border the wrong way round dives into the interior: about a third of them) is not given an id at
//     all; a state whose successor is such a state becomes a fixed point marked DEAD, and DEAD spreads backwards through the chains that end there; f stays injective everywhere else, so a state
//     is either dead or on a genuine cycle -- a start state that is dead (it never is) sends its border to the one-lane walk.
//   * every cycle is ranked ONCE per image, in place: word A = (jump, dist) doubles as usual, word B = (m, dm) carries the
//     LARGEST state id in the window [i, jump) and the distance to its first occurrence; when the window has wrapped the
//     cycle, m is the cycle's HEAD h (its largest state) and dm the distance to it (largest, so that DEAD = 0xFFFF spreads by
//     the same rule).  A border traced from state0 is then
//     { i : m[i] == h } in the order pos(i) = (dm[state0] - dm[i]) mod L, L = the cycle's length (kept in the head's own dm
//     slot) -- a border costs one pass over the states, no ranking of its own (fog of war traces ~3-10 blobs per window).
//     In-place rounds converge at least as fast as synchronous ones; a reader takes A[j] before B[j], a writer publishes B[i]
//     before A[i] (workgroup-scope fences between), so an observed B window is never shorter than the observed A distance and
//     the union [i, jump_i) + [jump_i, ..) stays gap-free.  The rounds end after one in which every state found its m of the
//     round's START equal to the m it read from its jump target (possibly newer, hence larger): then m_i >= m_jump(i) held for
//     the whole start-of-round snapshot, along the chain i -> jump -> ... the values cannot decrease, the chain ends in a
//     periodic part whose windows tile the cycle and therefore hold its maximum, so every m was the maximum already.
//     When the LDS has room for it, A and B are ONE 64-bit word per state (a round is then two LDS round trips and no fence).
//   * tables: B (4 B / state), the s_back nibble (1/2), position flags of the CHAIN_APPROX_SIMPLE compaction (1/4) and
//     6 B per border pixel sit behind the window planes in LDS; A (4 B / state) borrows the two label planes, which are all
//     zero while the borders are ranked (no border has been traced yet) and are zeroed again afterwards -- what does not fit
//     there (bit planes grow with the window's area, states with the outlines' length) follows B.
// Whatever does not fit (LDS, 16-bit ids) goes through the form above, and a start state that is not in the tables or is dead through
// the one-lane walk: same output in every case (tests/test_obstacle_prims_gpu.py).
__device__ unsigned long long g_walk_paths[4];   // contours traced by: this form, the global-table form, one lane; images
constexpr unsigned WALK_DEAD = 0xFFFFu;   // m of a state whose chain ends in a fixed point (ids stay below it)
struct WalkLds {
    unsigned* A;               // [n_lab] label planes while ranking ...
    unsigned* Aext;            // [N - n_lab] ... and behind B what they cannot hold
    int n_lab;
    unsigned* B;               // [N]   m | dm << 16; head: h | L << 16
    int* pixxy;                // [nb]  x | y << 11 | (directions with a state) << 22, raster order
    unsigned short* pixbase;   // [nb + 1] first state of border pixel #rank; [nb] = N
    unsigned* posflag;         // [ceil(N / 32) + 1]
    int* posprefix;            // [ceil(N / 32) + 1]
    unsigned long long* AB;    // [N]   packed form: A | B << 32 while ranking (then B is extracted to the front of it)
    unsigned short* rowrank;   // [rows + 1] rank of the first border pixel of every row
    int N, nb, ok, packed;
};

__device__ __forceinline__ size_t walk_pix_bytes(int nb) { return ((size_t)4 * nb + (size_t)2 * (nb + 1) + 7) & ~(size_t)7; }
// directions d of a border pixel's border-pixel neighbours (bnb) from which the walk goes on to a border pixel: the states that get ids
__device__ __forceinline__ unsigned v2_alloc_mask(unsigned nbm, unsigned bnb) {
    unsigned am = 0u, dirs = bnb;
    const unsigned nn = nbm | (nbm << 8);
    while (dirs) {
        const int d = __builtin_ctz(dirs);
        dirs &= dirs - 1;
        const int from = (d + 1) & 7;
        const int s = (from + __builtin_ctz((nn >> from) & 0xFFu)) & 7;
        if ((bnb >> s) & 1u) am |= 1u << d;
    }
    return am;
}

// One in-place round over all states (see the header of this section); returns whether one of this lane's states is still open.
// MODE 0: A in the label planes; 1: A split between the label planes and the arena; 2: A and B packed in 64-bit words.
template <int U, int MODE>
__device__ __forceinline__ bool v2_round(const WalkLds& T, int tid, int nth) {
    const int N = T.N;
    bool open = false;
    if (MODE == 2) {
        unsigned long long* const AB = T.AB;
        for (int i0 = tid; i0 < N; i0 += U * nth) {
            unsigned long long own[U], tgt[U];
#pragma unroll
            for (int u = 0; u < U; u++) { const int i = i0 + u * nth; own[u] = i < N ? AB[i] : 0ull; }
#pragma unroll
            for (int u = 0; u < U; u++)
                tgt[u] = __hip_atomic_load(&AB[(unsigned)own[u] & 0xFFFFu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned a = (unsigned)own[u], bi = (unsigned)(own[u] >> 32), aj = (unsigned)tgt[u], bj = (unsigned)(tgt[u] >> 32);
                const unsigned mj = bj & 0xFFFFu, mi = bi & 0xFFFFu;
                const unsigned nbw = mj > mi ? (mj | ((a & 0xFFFF0000u) + (bj & 0xFFFF0000u))) : bi;
                const unsigned na = (aj & 0xFFFFu) | ((a & 0xFFFF0000u) + (aj & 0xFFFF0000u));
                const int i = i0 + u * nth;
                if (i < N) {
                    if (mj != mi) open = true;
                    __hip_atomic_store(&AB[i], (unsigned long long)na | ((unsigned long long)nbw << 32), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        return open;
    }
    unsigned* const B = T.B;
    unsigned* const A_lab = T.A;
    unsigned* const A_ext = T.Aext - T.n_lab;
    const int n_lab = T.n_lab;
    auto A = [&](int i) -> unsigned* { return (MODE == 0 || i < n_lab) ? A_lab + i : A_ext + i; };
    for (int i0 = tid; i0 < N; i0 += U * nth) {
        unsigned a[U], aj[U], bi[U], bj[U], na[U], nbw[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * nth; a[u] = i < N ? *A(i) : 0u; bi[u] = i < N ? B[i] : 0u; }
#pragma unroll
        for (int u = 0; u < U; u++) aj[u] = __hip_atomic_load(A((int)(a[u] & 0xFFFFu)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int u = 0; u < U; u++) bj[u] = __hip_atomic_load(&B[a[u] & 0xFFFFu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned mj = bj[u] & 0xFFFFu, mi = bi[u] & 0xFFFFu;
            nbw[u] = mj > mi ? (mj | ((a[u] & 0xFFFF0000u) + (bj[u] & 0xFFFF0000u))) : bi[u];
            na[u] = (aj[u] & 0xFFFFu) | ((a[u] & 0xFFFF0000u) + (aj[u] & 0xFFFF0000u));
            // my m of the round's start against the (possibly newer) m of the target
            if (i0 + u * nth < N && mj != mi) open = true;
        }
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * nth; if (i < N) __hip_atomic_store(&B[i], nbw[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * nth; if (i < N) __hip_atomic_store(A(i), na[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
    return open;
}

template <int MODE>
__device__ __forceinline__ bool v2_round_cnt(const WalkLds& T, int cnt, int tid, int nth) {
    // all of a lane's states in ONE batch when there are at most six of them: a batch is a chain of dependent LDS round trips,
    // and a second, nearly empty batch doubled the round
    if (cnt <= 1) return v2_round<1, MODE>(T, tid, nth);
    if (cnt <= 2) return v2_round<2, MODE>(T, tid, nth);
    if (cnt <= 3) return v2_round<3, MODE>(T, tid, nth);
    if (cnt <= 4 || cnt == 7 || cnt == 8) return v2_round<4, MODE>(T, tid, nth);
    if (cnt <= 5) return v2_round<5, MODE>(T, tid, nth);
    return v2_round<6, MODE>(T, tid, nth);
}

// Barrier behind writes to the per-pixel tables: they are in global memory when GPIX (a window so large that its three planes leave
// no room for them in LDS: late in an episode the explored area spans the whole 20 m world), else in LDS.
template <bool GPIX>
__device__ __forceinline__ void sync_pix() {
    if (GPIX) wg_sync_global();
    else __syncthreads();
}

// Tables + ranking of every cycle.  All threads; img padded, lt / ln = its two label planes (contiguous: ln == lt + plane words,
// zero on entry, zero again on return).  T.ok = 0: nothing was changed, use another form.
template <bool GPIX>
__device__ __forceinline__ void wg_build_rank_lds(const Bits& img, unsigned* lt, unsigned* ln, unsigned* arena, unsigned arena_bytes,
                                         int* gpixxy, unsigned short* gpixbase, int gpix_cap, int wrows, int wwords, WalkLds& T,
                                         int* sh) {
    const int tid = threadIdx.x, nth = blockDim.x;
    const unsigned* base = img.w;
    const int pw = img.stride, W = wrows * wwords;
    const int plane_words = (wrows + 2) * pw;
    T.ok = 0; T.N = 0; T.nb = 0; T.n_lab = 0; T.packed = 0;
    if (ln != lt + plane_words || arena == nullptr) return;
    if (((unsigned)(size_t)arena & 7u) && arena_bytes >= 4u) { arena += 1; arena_bytes -= 4u; }   // the packed words are 8 bytes
    const size_t rr_bytes = ((size_t)2 * (wrows + 1) + 7) & ~(size_t)7;
    if (rr_bytes > (size_t)arena_bytes) return;
    unsigned* lab0 = lt - (pw + 1);            // the padded origin of the label planes: 2 * plane_words words
    auto border_word = [&](int ly, int lw) -> unsigned {
        const unsigned* r = base + ly * pw + lw;
        auto h = [](const unsigned* q) { const unsigned c = q[0]; return c & ((c << 1) | (q[-1] >> 31)) & ((c >> 1) | (q[1] << 31)); };
        return r[0] & ~(h(r - pw) & h(r) & h(r + pw));
    };
    WALK_STAMP(0);
    // a. border mask; raster-order rank of every word's first border pixel (the per-word loops only count and copy: a word of a
    //    horizontal outline holds 32 border pixels, and anything heavier per pixel made its lane the critical path -- 36 us per pass)
    for (int k = tid; k < W; k += nth) { const int ly = k / wwords, lw = k - ly * wwords; lt[ly * pw + lw] = border_word(ly, lw); }
    __syncthreads();
    const int per = (W + nth - 1) / nth;
    const int k0 = min(tid * per, W), k1 = min(k0 + per, W);
    int nb = 0;
    for (int k = k0; k < k1; k++) { const int ly = k / wwords, lw = k - ly * wwords; nb += __builtin_popcount(lt[ly * pw + lw]); }
    int nb_ex, d_ex, nb_tot, d_tot;
    wg_scan2(nb, 0, sh, nb_ex, d_ex, nb_tot, d_tot);
    const size_t pix_bytes = GPIX ? 0 : walk_pix_bytes(nb_tot);
    // (the last clause: with fewer than ~1.25 states per border pixel to come the LDS would still be too small -- do not spend the
    // next passes on finding that out; only the choice of the form depends on it)
    if (nb_tot > 65535 || rr_bytes + pix_bytes > (size_t)arena_bytes || (GPIX && nb_tot + 1 > gpix_cap) ||
        (!GPIX && rr_bytes + pix_bytes + (size_t)5 * nb_tot > (size_t)arena_bytes)) {
        for (int k = tid; k < W; k += nth) { const int ly = k / wwords, lw = k - ly * wwords; lt[ly * pw + lw] = 0u; }
        __syncthreads();
        return;
    }
    T.nb = nb_tot;
    T.rowrank = reinterpret_cast<unsigned short*>(arena);
    if (GPIX) {
        T.pixxy = gpixxy;
        T.pixbase = gpixbase;
    } else {
        T.pixxy = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(arena) + rr_bytes);
        T.pixbase = reinterpret_cast<unsigned short*>(T.pixxy + nb_tot);
    }
    for (int k = k0; k < k1; k++) {
        const int ly = k / wwords, lw = k - ly * wwords;
        ln[ly * pw + lw] = (unsigned)nb_ex;
        nb_ex += __builtin_popcount(lt[ly * pw + lw]);
    }
    __syncthreads();
    for (int k = tid; k < W; k += nth) {
        const int ly = k / wwords, lw = k - ly * wwords;
        unsigned bm = lt[ly * pw + lw];
        int rank = (int)ln[ly * pw + lw];
        if (lw == 0) T.rowrank[ly] = (unsigned short)rank;
        while (bm) { const int bit = __builtin_ctz(bm); bm &= bm - 1; T.pixxy[rank++] = (lw * 32 + bit) | (ly << 11); }
    }
    if (tid == 0) T.rowrank[wrows] = (unsigned short)nb_tot;
    sync_pix<GPIX>();
    WALK_STAMP(1);
    // b. ONE BORDER PIXEL PER LANE from here on: the directions that get a state (bits 22..29 of the pixel's entry), their count
    for (int r = tid; r < nb_tot; r += nth) {
        const int xy = T.pixxy[r], x = xy & 2047, ly = xy >> 11;
        T.pixxy[r] = xy | (int)(v2_alloc_mask(nbr8_padded(base, pw, x, ly), nbr8_padded(lt, pw, x, ly)) << 22);
    }
    sync_pix<GPIX>();
    const int pper = (nb_tot + nth - 1) / nth;
    const int r0 = min(tid * pper, nb_tot), r1 = min(r0 + pper, nb_tot);
    int ns = 0;
    for (int rb = r0; rb < r1; rb += 8) {      // (eight entries requested before the first is used: they may be in global memory)
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = rb + u < r1 ? T.pixxy[rb + u] : 0;
#pragma unroll
        for (int u = 0; u < 8; u++) ns += __builtin_popcount(((unsigned)v[u] >> 22) & 0xFFu);
    }
    int ns_ex, ns_tot;
    wg_scan2(ns, 0, sh, ns_ex, d_ex, ns_tot, d_tot);
    const size_t left = (size_t)arena_bytes - rr_bytes - pix_bytes;
    const size_t w32b = (size_t)8 * ((size_t)(ns_tot + 31) / 32 + 1);
    const bool packed = (size_t)8 * ns_tot + w32b <= left;
    const size_t n_lab_max = (size_t)2 * plane_words;
    const bool fits = ns_tot < (int)WALK_DEAD &&
                      (packed || (size_t)4 * ns_tot + (size_t)4 * ((size_t)ns_tot > n_lab_max ? ns_tot - n_lab_max : 0) + w32b <= left);
    if (!fits || ns_tot == 0) {
        for (int k = tid; k < W; k += nth) { const int ly = k / wwords, lw = k - ly * wwords; lt[ly * pw + lw] = 0u; ln[ly * pw + lw] = 0u; }
        __syncthreads();
        T.ok = fits;     // no states at all: every border is a single pixel (wg_emit_border_lds needs no table for those)
        return;
    }
    const int N = ns_tot;
    T.N = N;
    T.packed = packed;
    const int w32 = (N + 31) / 32 + 1;
    T.B = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(arena) + rr_bytes + pix_bytes);
    T.AB = reinterpret_cast<unsigned long long*>(T.B);
    T.n_lab = N < 2 * plane_words ? N : 2 * plane_words;
    T.Aext = T.B + N;
    T.posflag = packed ? T.B + 2 * N : T.Aext + (N - T.n_lab);
    T.posprefix = reinterpret_cast<int*>(T.posflag + w32);
    T.A = lab0;
    for (int rb = r0; rb < r1; rb += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = rb + u < r1 ? T.pixxy[rb + u] : 0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (rb + u < r1) T.pixbase[rb + u] = (unsigned short)ns_ex;
            ns_ex += __builtin_popcount(((unsigned)v[u] >> 22) & 0xFFu);
        }
    }
    for (int i = tid; i < w32; i += nth) T.posflag[i] = 0u;
    if (tid == 0) T.pixbase[nb_tot] = (unsigned short)N;
    sync_pix<GPIX>();
    WALK_STAMP(2);
    // c. successors: A = (next, 1), B = (self, 0) -- (DEAD, 0) for the fixed points.  The split form parks `next` in B first (the
    //    label planes are still border mask and ranks) and moves it over in a second pass.
    for (int r = tid; r < nb_tot; r += nth) {
        const int xy = T.pixxy[r], x = xy & 2047, ly = (xy >> 11) & 2047;
        const int sb0 = T.pixbase[r];
        const unsigned nbm = nbr8_padded(base, pw, x, ly);
        const unsigned dirs0 = ((unsigned)xy >> 22) & 0xFFu;
        // every successor's table entries are requested before the first is used (global memory when GPIX)
        int rank2[8], xy2[8];
        unsigned short pb2[8];
        {
            unsigned dirs = dirs0;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                rank2[t] = 0;
                if (dirs) {
                    const int d = __builtin_ctz(dirs);
                    dirs &= dirs - 1;
                    const int from = (d + 1) & 7;
                    const int s = (from + __builtin_ctz(((nbm | (nbm << 8)) >> from) & 0xFFu)) & 7;   // a border pixel: the state has an id
                    const int x2 = x + code_dx(s), y2 = ly + code_dy(s);
                    const int w2 = y2 * pw + (x2 >> 5);
                    rank2[t] = (int)ln[w2] + __builtin_popcount(lt[w2] & ((1u << (x2 & 31)) - 1u));
                }
            }
#pragma unroll
            for (int t = 0; t < 8; t++) { xy2[t] = T.pixxy[rank2[t]]; pb2[t] = T.pixbase[rank2[t]]; }
        }
        unsigned dirs = dirs0;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            if (!dirs) break;
            const int d = __builtin_ctz(dirs);
            dirs &= dirs - 1;
            const int from = (d + 1) & 7;
            const int s = (from + __builtin_ctz(((nbm | (nbm << 8)) >> from) & 0xFFu)) & 7;
            const int id = sb0 + t, sb2 = (s + 4) & 7;
            const unsigned am2 = ((unsigned)xy2[t] >> 22) & 0xFFu;
            // successor state without an id: fixed point
            const int id2 = (am2 >> sb2) & 1u ? (int)pb2[t] + __builtin_popcount(am2 & ((1u << sb2) - 1u)) : id;
            if (packed) T.AB[id] = (unsigned long long)((unsigned)id2 | (1u << 16)) | ((unsigned long long)(id2 == id ? WALK_DEAD : (unsigned)id) << 32);
            else T.B[id] = (unsigned)id2;
        }
    }
    __syncthreads();
    if (!packed) {
        for (int i = tid; i < N; i += nth) {
            const unsigned nx = T.B[i];
            *(i < T.n_lab ? T.A + i : T.Aext + (i - T.n_lab)) = nx | (1u << 16);
            T.B[i] = (int)nx == i ? WALK_DEAD : (unsigned)i;
        }
    }
    if (tid < 3) sh[36 + tid] = 0;
    __syncthreads();
    WALK_STAMP(3);
    // e. rounds, in place
    const int cap_rounds = 33 - __builtin_clz((unsigned)N);
    const int cnt = (N + nth - 1) / nth;
    const int mode = packed ? 2 : (N > T.n_lab ? 1 : 0);
    for (int r = 0; r < cap_rounds; r++) {
        const bool open = mode == 2 ? v2_round_cnt<2>(T, cnt, tid, nth) : (mode == 1 ? v2_round_cnt<1>(T, cnt, tid, nth) : v2_round_cnt<0>(T, cnt, tid, nth));
        if (open) sh[36 + r % 3] = 1;
        __syncthreads();
        const int any = sh[36 + r % 3];
        if (tid == 0) sh[36 + (r + 2) % 3] = 0;
        if (!any) break;
    }
    __syncthreads();
    if (packed) {   // B to the front of the packed words: blocks of 8 states per lane in ascending order, read -- barrier -- write
                    // (B[i] lands on AB[i / 2], a word of this block or an earlier one)
        for (int i0 = 0; i0 < N; i0 += 8 * nth) {
            unsigned hi[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = i0 + tid + u * nth; hi[u] = i < N ? (unsigned)(T.AB[i] >> 32) : 0u; }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = i0 + tid + u * nth; if (i < N) T.B[i] = hi[u]; }
            __syncthreads();
        }
    }
    WALK_STAMP(4);
    // f. cycle lengths into the heads' dm slots; the label planes are labels again
    unsigned* B = T.B;
    for (int i = tid; i < N; i += nth) {
        const unsigned w = B[i], m = w & 0xFFFFu;
        if ((int)m != i && m != WALK_DEAD) atomicMax(&B[m], m | (((w >> 16) + 1u) << 16));
    }
    __syncthreads();
    for (int i = tid; i < 2 * plane_words; i += nth) lab0[i] = 0u;
    __syncthreads();
    T.ok = 1;
    WALK_STAMP(5);
#ifdef VLFM_PHASE_TIMING
    if ((int)blockIdx.x == g_walk_block && threadIdx.x == 0) { g_walk_clk[12] = GPIX ? 3 : 2; g_walk_clk[13] = N; g_walk_clk[14] = nb_tot; }
#endif
}

// One outer border from (x0, y0) out of the ranked tables.  All threads; returns the number of emitted points (uniform), or -1
// when the border is not in the tables (nothing was written: the caller lets one lane walk it).
template <bool GPIX>
__device__ __forceinline__ int wg_emit_border_lds(const Bits& img, const WalkLds& T, unsigned* traced, unsigned* neg, int x0, int y0,
                                         int method, int2* out, int cap, int* sh) {
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
    const unsigned* base = img.w;
    const int pw = img.stride;
    const unsigned nb = nbr8_padded(base, pw, x0, y0);
    const unsigned nb0 = nb & ~(1u << 4);
    if (nb0 == 0u) {                                     // isolated pixel
        if (tid == 0) {
            const int wi = y0 * pw + (x0 >> 5);
            const unsigned m = 1u << (x0 & 31);
            traced[wi] |= m; neg[wi] |= m;
            if (cap > 0) out[0] = make_int2(x0, y0);
        }
        __syncthreads();
        return 1;
    }
    if (T.N == 0) return -1;
    const unsigned mir = __brev(nb0) >> 24;
    const unsigned rot = ((mir | (mir << 8)) >> 4) & 0xFFu;
    const int s = (3 - __builtin_ctz(rot)) & 7;          // direction of i1 = s_back of state0
    // the start pixel's table entry: the border pixels of row y0 are ranks rowrank[y0] .. rowrank[y0 + 1] - 1, 64 of them per probe
    // (every wavefront does the same probe: no exchange)
    const int key0 = x0 | (y0 << 11);
    int r0 = -1, xy0 = 0;
    for (int q = T.rowrank[y0], q1 = T.rowrank[y0 + 1]; q < q1 && r0 < 0; q += 64) {
        const int v = q + lane < q1 ? T.pixxy[q + lane] : -1;
        const unsigned long long hit = __ballot(v >= 0 && (v & 0x3FFFFF) == key0);
        if (hit) { const int l = __builtin_ctzll(hit); r0 = q + l; xy0 = __shfl(v, l, 64); }
    }
    if (r0 < 0) return -1;
    const unsigned am0 = ((unsigned)xy0 >> 22) & 0xFFu;
    if (!((am0 >> s) & 1u)) return -1;                   // i1 is not a border pixel: cannot happen on a traced border
    const int id0 = (int)T.pixbase[r0] + __builtin_popcount(am0 & ((1u << s) - 1u));
    const unsigned w0 = T.B[id0];
    if ((w0 & 0xFFFFu) == WALK_DEAD) return -1;          // the start state's chain ends in the interior: cannot happen either
    const int h = (int)(w0 & 0xFFFFu);
    const int d0 = id0 == h ? 0 : (int)(w0 >> 16);
    const int L = (int)(T.B[h] >> 16);
    if (L < 2 || d0 >= L) return -1;
    // ---- labels and points: one BORDER PIXEL per lane and step -- its states are consecutive ids, its coordinates and the
    // directions of its states one table entry (no search from a state back to its pixel); four entries requested at a time
    const int pwords = (L + 31) >> 5;
    for (int rb = tid; rb < T.nb; rb += 4 * nth) {
        int xy4[4];
        unsigned short pb4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int r = rb + u * nth; xy4[u] = r < T.nb ? T.pixxy[r] : 0; pb4[u] = r < T.nb ? T.pixbase[r] : (unsigned short)0; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int xy = xy4[u];
            unsigned am = ((unsigned)xy >> 22) & 0xFFu;
            if (!am) continue;
            const int x = xy & 2047, y = (xy >> 11) & 2047;
            const unsigned nbm = nbr8_padded(base, pw, x, y);
            for (int i = pb4[u]; am; am &= am - 1, i++) {
                const unsigned w = T.B[i];
                if ((int)(w & 0xFFFFu) != h) continue;
                const int dm = i == h ? 0 : (int)(w >> 16);
                int pos = d0 - dm;
                if (pos < 0) pos += L;
                const int sb = __builtin_ctz(am);
                const int from = (sb + 1) & 7;
                const int so = (from + __builtin_ctz(((nbm | (nbm << 8)) >> from) & 0xFFu)) & 7;
                const int wi = y * pw + (x >> 5);
                const unsigned m = 1u << (x & 31);
                atomicOr(&traced[wi], m);
                if ((unsigned)(so - 1) < (unsigned)sb) atomicOr(&neg[wi], m);
                if (method == 1) { if (pos < cap) out[pos] = make_int2(x, y); }
                else if (so != (sb ^ 4)) atomicOr(&T.posflag[pos >> 5], 1u << (pos & 31));
            }
        }
    }
    __syncthreads();
    if (method == 1) return L;
    // CHAIN_APPROX_SIMPLE: ordered compaction through the position flags.  Borders of up to 256 states (the fragments a fog-of-war
    // window is cut into: a dozen per window, each of them used to pay for a workgroup scan) add the few flag words up themselves.
    const bool short_border = pwords <= 8;
    int total = 0;
    if (short_border) {
        for (int wi = 0; wi < pwords; wi++) total += __builtin_popcount(T.posflag[wi]);
    } else {
        for (int w0i = 0; w0i < pwords; w0i += nth) {        // (one pass unless the border has more than 32 768 states)
            const int wi = w0i + tid;
            const int c = wi < pwords ? __builtin_popcount(T.posflag[wi]) : 0;
            int ex, dummy_ex, tot, dummy_tot;
            wg_scan2(c, 0, sh, ex, dummy_ex, tot, dummy_tot);
            if (wi < pwords) T.posprefix[wi] = total + ex;
            total += tot;
        }
        __syncthreads();
    }
    for (int rb = tid; rb < T.nb; rb += 4 * nth) {
        int xy4[4];
        unsigned short pb4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int r = rb + u * nth; xy4[u] = r < T.nb ? T.pixxy[r] : 0; pb4[u] = r < T.nb ? T.pixbase[r] : (unsigned short)0; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int xy = xy4[u];
            unsigned am = ((unsigned)xy >> 22) & 0xFFu;
            for (int i = pb4[u]; am; am &= am - 1, i++) {
                const unsigned w = T.B[i];
                if ((int)(w & 0xFFFFu) != h) continue;
                const int dm = i == h ? 0 : (int)(w >> 16);
                int pos = d0 - dm;
                if (pos < 0) pos += L;
                const unsigned fw = T.posflag[pos >> 5];
                if (!((fw >> (pos & 31)) & 1u)) continue;
                int idx = __builtin_popcount(fw & ((1u << (pos & 31)) - 1u));
                if (short_border) { for (int wi = 0; wi < (pos >> 5); wi++) idx += __builtin_popcount(T.posflag[wi]); }
                else idx += T.posprefix[pos >> 5];
                if (idx < cap) out[idx] = make_int2(xy & 2047, (xy >> 11) & 2047);
            }
        }
    }
    __syncthreads();
    for (int wi = tid; wi < pwords; wi += nth) T.posflag[wi] = 0u;
    __syncthreads();
    return total;
}

// The RETR_EXTERNAL scan over ranked tables (wg_scan_external's loop)
template <bool GPIX>
__device__ __forceinline__ void wg_scan_ranked(const Bits& img, const WalkLds& T, unsigned* traced, unsigned* neg, int method,
                                      ContourSink& sink, int* sh) {
    const int tid = threadIdx.x;
    int y = 0, x_done = -1;
    int n_fast = 0, n_serial = 0;
    for (int call = 0;; call++) {
        int fx = 0, fy = 0;
        const bool found = wg_next_start(img, traced, neg, y, x_done, sh, call, fx, fy);
        if (!found) break;
        y = fy; x_done = fx;
        const int room = sink.cap_pts - sink.n_pts > 0 ? sink.cap_pts - sink.n_pts : 0;
        if (sink.n_contours == 0) WALK_STAMP(6);
        int n = wg_emit_border_lds<GPIX>(img, T, traced, neg, fx, fy, method, sink.pts + sink.n_pts, room, sh);
        if (sink.n_contours == 0) WALK_STAMP(7);
        if (n < 0) {
            if (tid == 0) sh[46] = follow_border(img, traced, neg, fx, fy, method, sink.pts + sink.n_pts, room);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            n = sh[46];
            n_serial++;
        } else {
            n_fast++;
        }
        if (sink.n_contours < sink.cap_contours && sink.n_pts + n <= sink.cap_pts) {
            if (tid == 0) { sink.start[sink.n_contours] = sink.n_pts; sink.len[sink.n_contours] = n; }
        } else {
            sink.overflow = 1;
        }
        sink.n_contours++;
        sink.n_pts += n;
    }
    wg_sync_global();     // the points and the contour list (global memory) for whoever reads them next in this workgroup
    WALK_STAMP(8);
    if (tid == 0) {
        if (n_fast) atomicAdd(&g_walk_paths[0], (unsigned long long)n_fast);
        if (n_serial) atomicAdd(&g_walk_paths[2], (unsigned long long)n_serial);
        atomicAdd(&g_walk_paths[GPIX ? 1 : 3], 1ull);
    }
}

// wg_scan_external with the ranked tables: everything in LDS when it fits behind the window planes (arena), the per-pixel tables
// in the global scratch of the fallback form when that is what it takes (large windows), the fallback form itself otherwise.
__device__ __forceinline__ void wg_scan_external_lds(const Bits& img, unsigned* traced, unsigned* neg, int method, ContourSink& sink,
                                            WalkTables& Tglobal, unsigned* arena, unsigned arena_bytes, int* sh) {
    WalkLds T;
    wg_build_rank_lds<false>(img, traced, neg, arena, arena_bytes, nullptr, nullptr, 0, Tglobal.wrows, Tglobal.wwords, T, sh);
    if (T.ok) { wg_scan_ranked<false>(img, T, traced, neg, method, sink, sh); return; }
    const int gcap = min(Tglobal.cap_states, 2 * Tglobal.cap_bp);
    wg_build_rank_lds<true>(img, traced, neg, arena, arena_bytes, reinterpret_cast<int*>(Tglobal.jd0),
                            reinterpret_cast<unsigned short*>(Tglobal.pixbase), gcap, Tglobal.wrows, Tglobal.wwords, T, sh);
    if (T.ok) { wg_scan_ranked<true>(img, T, traced, neg, method, sink, sh); return; }
    wg_scan_external(img, traced, neg, method, sink, Tglobal, sh);
}

}  // namespace vlfm
