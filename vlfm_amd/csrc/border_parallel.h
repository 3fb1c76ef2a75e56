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

// RETR_EXTERNAL scan of a padded LDS window by the whole workgroup (all threads must call; labels zero on entry).
// sink's counters end up identical in every thread.  sh: WG_SH_INTS ints of shared memory.
__device__ inline void wg_scan_external(const Bits& img, unsigned* traced, unsigned* neg, int method, ContourSink& sink,
                                        WalkTables& T, int* sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    wg_build_walk_tables(img, traced, neg, T, sh);
    const int nwaves = (blockDim.x + 63) >> 6;
    constexpr int RPW = 4;               // rows per wavefront and search step
    int y = 0, x_done = -1;              // scan position (uniform)
    for (;;) {
        // the next start in raster order: every wavefront examines RPW rows of a block of nwaves * RPW; labels only change
        // between searches, so the rows can be judged independently (a row costs ~0.15 us: one wavefront alone spent
        // 115 us on a 700-row window)
        bool found = false;
        int fx = 0, fy = 0;
        for (int yb = y; yb < img.rows && !found; yb += nwaves * RPW) {
            if (tid == 0) sh[40] = 0x7FFFFFFF;
            __syncthreads();
            for (int q = 0; q < RPW; q++) {
                const int r = yb + wave * RPW + q;
                int x = 0;
                if (r < img.rows && row_next_start(img, traced, neg, r, r == y ? x_done : -1, x)) {
                    if (lane == 0) atomicMin(&sh[40], (r << 11) | x);
                    break;
                }
            }
            __syncthreads();
            const int v = sh[40];
            if (v != 0x7FFFFFFF) { found = true; fx = v & 2047; fy = v >> 11; }
            __syncthreads();
        }
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

}  // namespace vlfm
