// gemm_f16.hip -- C[M][N] = epilogue(X[M][K] . W[N][K]^T + bias[N]) in f16 with f32 accumulation on the gfx950 matrix
// cores, written for the ViT-g MLP of the BLIP-2 forward (reference: LAVIS eva_vit's Mlp behind vlfm/vlm/blip2itm.py:52):
// fc1 is [65 792 x 1408] . [6144 x 1408]^T followed by an exact (erf) GELU.  hipBLASLt offers GELU only in its tanh form, so
// the library path pays a separate 5.4 TB/s elementwise pass over the 808 MB activation; here the GELU runs on the f32
// accumulators in the epilogue and the activation is written once.
//
// Structure (one workgroup = 256 x 256 output tile, 8 wavefronts as 2 (n) x 4 (m), K-tile 64):
//   * both operands are K-contiguous ("NT" GEMM): a K-tile of W and of X is 256 rows x 128 B each.  They are staged by
//     global_load_lds (16 B per lane, no registers, no ds_write): a wavefront's instruction fills 8 rows x 128 B of LDS
//     linearly; the 16-byte slot a lane FETCHES is permuted within its row (slot ^ ((row >> 1) & 7)), so that the MFMA
//     fragment reads -- the four 16-lane groups of ds_read_b128, 16 B from 16 different rows each -- hit 16 different bank
//     groups: conflict-free, while the 8 lanes of a row still read one whole 128-byte line of global memory.
//   * two LDS buffers (128 KB, one workgroup per CU, two wavefronts per SIMD).
//   * operand roles are swapped (A-operand = W rows, B-operand = X rows): a lane then holds 4 CONSECUTIVE n of one m, i.e.
//     8 contiguous bytes of the output row after conversion; the epilogue transposes through LDS (row stride 272 B:
//     conflict-free 8-byte writes, 16-byte reads) and stores 256-byte row segments with 16 B per lane.
//   * the schedule is a PING-PONG between the two wavefronts of a SIMD in the 8-phase form (Gemm8p: four phases of 16 MFMAs per K-tile,
//     half-tile staging, counted vmcnt waits) as a PERSISTENT kernel (gemm_f16_8pp_kernel: one workgroup per CU walks the tile list,
//     the next tile's first K-tile is requested before the epilogue, the stores drain under the next tile) with the GELU in packed
//     f32 (gelu_f16.h).  Round 6: the earlier schedules (round 2's 4-phase ping-pong, the lock-step kernel, the one-tile-per-workgroup
//     8-phase kernel, the 4-wavefront kernel) were A/B baselines only -- none is faster on any ViT shape -- and are gone from the
//     tree (git history: round 5; their numbers: DESIGN.md section 6c / 6g).
// Measured at 256 images (tools/gemm_f16_probe.py, tools/mlp_probe.py; DESIGN.md section 6c has the table): fc1 + GELU 1.11-1.16 ms
// against 1.41-1.45 ms for hipBLASLt + the GELU pass on random data, 1.19 against 1.24 ms on the network's own activations; as a
// plain GEMM 1.0-1.1 PFLOP/s, i.e. 5-10 % BELOW hipBLASLt -- so only the fused fc1 uses it (vlfm_amd/vlm/ops.py:linear_gelu).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/vlfm_amd.h"
#include "gelu_f16.h"
#include "profile.h"
#include "status.h"

namespace vlfm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// s_waitcnt through the builtin (the compiler's own scoreboard sees it; beside inline-asm waits it re-waits for everything at
// the loop head): simm16 = vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]; 0xC07F = lgkmcnt(0), 0x0F70 =
// vmcnt(0), 0x0F78 = vmcnt(8)

constexpr int GB = 256;            // tile rows (both m and n)
constexpr int GK = 64;             // K per tile
constexpr int ROWB = GK * 2;       // bytes per staged row
constexpr int OPER = GB * ROWB;    // bytes per operand tile (32 KB)
constexpr int BUF = 2 * OPER;      // W tile + X tile
constexpr int EPI_ROW = 272;       // epilogue staging row stride (128 n x 2 B + 16)
constexpr int EPI_WAVE = 64 * EPI_ROW;
constexpr int GEMM_LDS = 8 * EPI_WAVE > 2 * BUF ? 8 * EPI_WAVE : 2 * BUF;
enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_ACCUM = 2 };   // 2: C += X . W^T (+ bias): the residual-stream GEMMs (projection, fc2)

struct GemmArgs {
    const _Float16* x;     // [M][K]
    const _Float16* w;     // [N][K]
    const _Float16* bias;  // [N] or null
    _Float16* c;           // [M][N]
    int M, N, K;
    int tiles_m, tiles_n;
    int group_m;          // tile order of the ping-pong kernel: m-tiles per group (1 = plain n-fastest order)
    int split_ragged;     // persistent kernel: split the tiles of a ragged last round into n-halves (0: diagnostic VLFM_GEMM_NO_SPLIT)
};

using lds_ptr = __attribute__((address_space(3))) unsigned char*;
using gbl_ptr = const __attribute__((address_space(1))) unsigned char*;

// ------------------------------------------------------------------------------------------------ 8-phase schedule
// The two wavefronts of a SIMD (w and w + 4) never do the same thing at the same time (wavefronts 4-7 run one barrier behind 0-3).
// A K-tile is FOUR phases of 16 MFMAs (one quadrant of the wavefront's 128 x 64 accumulator block each) instead of
// two of 32, an operand tile is staged as two HALF-tiles, one half-tile (2 global_load_lds per wavefront) per phase, and the loads
// are never drained inside the loop: ONE counted s_waitcnt vmcnt(4) per K-tile leaves the two half-tiles issued last in flight across
// every barrier (the guide's 256^2 8-phase template; a schedule that waits vmcnt(0) once per K-tile needs every load of the next tile to land
// inside ~3 phases of the current one, with the wait on the critical path of all 8 wavefronts).
//
// LDS: two K-tile buffers of four 16 KB slots, in the order the phases need them:
//     slot 0 = Q_H0   X rows {wm * 64 + [ 0, 32)}  for the four wm      read in phase 1 (4 ds_read_b128 per wavefront)
//     slot 1 = P_H0   W rows {wn * 128 + [ 0, 64)} for the two wn       read in phase 1 (8)
//     slot 2 = Q_H1   X rows {wm * 64 + [32, 64)}                       read in phase 2 (4)
//     slot 3 = P_H1   W rows {wn * 128 + [64,128)}                      read in phase 3 (8)
//   (a "half" is the same half of EVERY wavefront's rows, so one slot serves all eight wavefronts in one phase.)
// Quadrants: phase 1 (P_H0, Q_H0), phase 2 (P_H0, Q_H1), phase 3 (P_H1, Q_H1), phase 4 (P_H1, Q_H0): one new fragment set per phase,
// none in phase 4; P fragments 32 VGPRs, Q_H0 and Q_H1 16 each.
// Staging while tile t (buffer b) is computed:    phase 1: slot 2 of tile t + 1 (buffer b ^ 1)   phase 2: slot 3 of tile t + 1
//                                                 phase 3: slot 0 of tile t + 2 (buffer b)       phase 4: slot 1 of tile t + 2
//   WAR: a slot is re-staged >= 2 phases after the phase that read it last (slot 0 / 1: read in 1, staged in 3 / 4; slot 2: read in
//        2 of tile t - 1, staged in 1 of tile t; slot 3: read in 3, staged in 2), and every wavefront retires its reads of phase p
//        (lgkmcnt(0)) right behind the first barrier of phase p -- the late group one barrier later, still before phase p + 2.
//   RAW: the wait of phase 4 -- before that phase's FIRST barrier -- leaves only the 4 loads of phases 3 and 4 (tile t + 2) in flight:
//        tile t + 1 is complete as far as this wavefront's shares go; the early group reads it two barriers later, the late group
//        three, and by then every wavefront has executed its own wait (the late group's sits one barrier behind the early group's).
// (Round 5 instrumented this schedule with per-workgroup and per-barrier clock stamps -- tools/gemm_stamp_probe.py of that tree,
// profiles/r05_gemm_stamps.txt; the instrumentation left with the one-tile kernel it was written for.)
template <int EPI, int BAL, int SYNC = 2>
struct Gemm8p {
    static constexpr int SLOT = 128 * ROWB;      // 16 KB
    static constexpr int KBUF = 4 * SLOT;        // one K-tile: 64 KB

    const GemmArgs& a;
    lds_ptr lds;
    const unsigned char* smem;
    int wave, lane;
    bool half;                  // half tile: at most 128 valid W rows (the last n-tile of N = 1408 / 4224).  The two wavefront groups
                                // then take 64 rows each (P_H0 only) and phases 3 and 4 carry no work -- with the full-tile mapping
                                // one group would own all 128 rows, the other none, and the ping-pong would degenerate to "read, then
                                // compute" at three quarters of a full tile's time for half the work
    bool active;                // this wavefront's W rows are not all beyond N
    uint32_t voff[4][2];        // per slot and chunk: byte offset of this lane's 16 B inside the operand, without the K-tile term
    uint32_t rdP[2], rdQ[2];    // fragment read offsets inside a slot, by kk (the swizzle turns kk into an XOR of 64 B)
    half8 fp[4][2], fq0[BAL ? 2 : 1][2][2], fq1[2][2];   // BAL: Q_H0 of the NEXT tile is read in phase 4 into the other set
    floatx4 acc[8][4];

    __device__ Gemm8p(const GemmArgs& a_, unsigned char* smem_, int m0, int n0) : a(a_), lds((lds_ptr)smem_), smem(smem_) {
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wn = wave >> 2, wm = wave & 3;
        half = false;   // (half-tile mode: measured, no gain -- see tile())
        active = n0 + wn * (half ? 64 : 128) < a.N;
        offsets(m0, n0, voff);
        const int r16 = lane & 15, swz = (r16 >> 1) & 7, s0 = lane >> 4;
        const uint32_t bp = (uint32_t)(wn * 64 + r16) * ROWB + (uint32_t)((s0 ^ swz) << 4);
        const uint32_t bq = (uint32_t)(wm * 32 + r16) * ROWB + (uint32_t)((s0 ^ swz) << 4);
        rdP[0] = bp; rdP[1] = bp ^ 64u;
        rdQ[0] = bq; rdQ[1] = bq ^ 64u;
        zero_acc();
    }
    __device__ inline void zero_acc() {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    // (persistent kernel: the fragment registers are read and used under `active`; without a definition on every path they
    // would stay live -- 80 registers -- through the epilogue of every tile)
    __device__ inline void kill_fragments() {
        const half8 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
        for (int f = 0; f < 4; f++) { fp[f][0] = z; fp[f][1] = z; }
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                fq1[g][k] = z;
#pragma unroll
                for (int q = 0; q < (BAL ? 2 : 1); q++) fq0[q][g][k] = z;
            }
    }
    // staging offsets of this lane for the tile at (m0, n0)
    __device__ inline void offsets(int m0, int n0, uint32_t (&vo)[4][2]) const {
        const int sub = lane >> 3, p = lane & 7;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int rho = (wave * 2 + j) * 8 + sub;            // row inside the slot
            const int s = p ^ ((rho >> 1) & 7);                  // the 16-byte slot of the row this lane FETCHES (lands at p)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int rq = min(m0 + (rho >> 5) * 64 + h * 32 + (rho & 31), a.M - 1);
                const int rp = min(n0 + (half ? rho + h * 128 : (rho >> 6) * 128 + h * 64 + (rho & 63)), a.N - 1);
                vo[2 * h][j] = (uint32_t)rq * (uint32_t)a.K * 2u + (uint32_t)s * 16u;
                vo[2 * h + 1][j] = (uint32_t)rp * (uint32_t)a.K * 2u + (uint32_t)s * 16u;
            }
        }
    }

    // half-tile `slot` of K-tile t -> buffer b
    template <int SLOT_ID>
    __device__ inline void stage_at(int b, int t, const uint32_t (&vo)[4][2]) {
        const unsigned char* base = reinterpret_cast<const unsigned char*>((SLOT_ID & 1) ? a.w : a.x) + (size_t)t * (GK * 2);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int dst = __builtin_amdgcn_readfirstlane(b * KBUF + SLOT_ID * SLOT + (wave * 2 + j) * 1024);
            __builtin_amdgcn_global_load_lds((gbl_ptr)(base + vo[SLOT_ID][j]), lds + dst, 16, 0, 0);
        }
    }
    template <int SLOT_ID>
    __device__ inline void stage(int b, int t) { stage_at<SLOT_ID>(b, t, voff); }
    template <int B, int SLOT_ID>
    __device__ inline void read_p() {
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int f = 0; f < 4; f++)
                fp[f][kk] = *reinterpret_cast<const half8*>(smem + B * KBUF + SLOT_ID * SLOT + f * 16 * ROWB + rdP[kk]);
    }
    template <int B, int SLOT_ID>
    __device__ inline void read_q(half8 (&fq)[2][2]) {
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int g = 0; g < 2; g++)
                fq[g][kk] = *reinterpret_cast<const half8*>(smem + B * KBUF + SLOT_ID * SLOT + g * 16 * ROWB + rdQ[kk]);
    }
    template <int PI, int QJ>
    __device__ inline void mma(const half8 (&fq)[2][2]) {
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int f = 0; f < 4; f++)
#pragma unroll
                for (int g = 0; g < 2; g++)
                    acc[PI * 4 + f][QJ * 2 + g] =
                        __builtin_amdgcn_mfma_f32_16x16x32_f16(fp[f][kk], fq[g][kk], acc[PI * 4 + f][QJ * 2 + g], 0, 0, 0);
    }
    static __device__ inline void bar() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // [fragments + staging issued by the caller] | barrier | fragments have arrived | 16 MFMAs | barrier
    template <int PI, int QJ>
    __device__ inline void compute(const half8 (&fq)[2][2], bool work = true) {
        bar();
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this phase's reads are in registers -- and out of the slot
        __builtin_amdgcn_sched_barrier(0);
        if (active && work) {
            __builtin_amdgcn_s_setprio(1);
            mma<PI, QJ>(fq);
            __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // SYNC 2: a second barrier closes the phase, and the late group runs one barrier behind: strict alternation of the two
        // wavefronts of a SIMD between "read" and "MFMA".  SYNC 1: ONE barrier per phase and no stagger -- the two wavefronts of a
        // SIMD leave the barrier together, one gets the matrix pipe first, the other follows, and each reads its next fragments
        // when it is done: the offset re-creates itself every phase.  Same hazards: a slot is re-staged two barriers after the
        // barrier that followed its last read (whose lgkmcnt(0) every wavefront executes before it can arrive at the next one),
        // and the counted vmcnt wait precedes the barrier in front of the first read.
        if (SYNC == 2) bar();
    }

    // K-tile t in buffer B (compile-time: the buffer term of every ds_read offset is an immediate).
    // BAL = 1 ("balanced"): the Q_H0 fragments of tile t + 1 are read in phase 4 of tile t (which has no reads of its own) instead
    // of in phase 1 of tile t + 1 (which has the 8 P reads): 8 / 4 / 8 / 4 ds_read_b128 per phase instead of 12 / 4 / 8 / 0.  Slot 0 of
    // tile t + 1 therefore has to be complete one phase earlier: a second counted wait, vmcnt(8), in phase 3.
    // Measured and dropped (round 5, profiles/r05_gemm_*): a counted wait in EVERY phase (vmcnt(8): each half-tile gets four
    // phases to land instead of two to five) -- no gain on the N = 4224 / 6144 shapes, 7 % slower on fc2: the K-tile time is not a
    // memory-latency effect (see the barrier stamps in DESIGN.md); a half-tile mode that splits the 128 valid W rows of the last
    // n-tile over both wavefront groups and leaves phases 3 and 4 empty -- no gain either, the barriers cost what they cost.
    template <int B>
    __device__ inline void tile(int t, int NT) {
        constexpr int QS = BAL ? B : 0;
        // ---- phase 1
        if (active) {
            if (!BAL) read_q<B, 0>(fq0[QS]);
            read_p<B, 1>();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < NT) stage<2>(B ^ 1, t + 1);
        compute<0, 0>(fq0[QS]);
        // ---- phase 2
        if (active) read_q<B, 2>(fq1);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < NT) stage<3>(B ^ 1, t + 1);
        compute<0, 1>(fq1);
        // ---- phase 3
        if (active) read_p<B, 3>();
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < NT) {
            stage<0>(B, t + 2);
            if (BAL) __builtin_amdgcn_s_waitcnt(0x0F78);     // vmcnt(8): slot 0 of tile t + 1 has landed (slots 1-3 of t + 1 and
        } else if (BAL) {                                     //           slot 0 of t + 2 may still be in flight)
            __builtin_amdgcn_s_waitcnt(0x0F76);              // vmcnt(6): no slot 0 of t + 2 behind it
        }
        compute<1, 1>(fq1);
        // ---- phase 4: tile t + 1 must be complete when this phase's first barrier is passed
        if (BAL && active && t + 1 < NT) read_q<B ^ 1, 0>(fq0[BAL ? (B ^ 1) : 0]);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < NT) {
            stage<1>(B, t + 2);
            __builtin_amdgcn_s_waitcnt(0x0F74);      // vmcnt(4): only the loads of phases 3 and 4 (tile t + 2) stay in flight
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): nothing newer was issued
        }
        compute<1, 0>(fq0[QS]);
    }
};

// Tile of a workgroup.  Workgroup ids go round-robin over the 8 XCDs; every XCD gets a contiguous range of the tile list (tiles
// sharing operand panels meet in one L2), walked in groups of GM m-tiles, m fastest.  When the last column of n-tiles is at most
// half full (N = 1408: 5.5 tiles, N = 4224: 16.5) its tiles run in about half the time of a full one (the wavefronts of the empty
// half skip their reads and MFMAs): they are listed LAST within each XCD, so that they fill the ragged end of the last wave of
// workgroups instead of being scattered through it -- 257 x 6 tiles on 256 CUs were 7 rounds of full-tile time, the work is 5.6.
__host__ __device__ inline void tile_of(const GemmArgs& a, int bid, int& tm, int& tn) {
    const int nwg = a.tiles_m * a.tiles_n;
    const int xcd = bid & 7, idx = bid >> 3;
    const int tail_cols = a.N - (a.tiles_n - 1) * GB;
    const bool ragged = a.tiles_n > 1 && tail_cols <= GB / 2;
    const int ncols = ragged ? a.tiles_n - 1 : a.tiles_n;
    const int nfull = a.tiles_m * ncols, nhalf = nwg - nfull;
    const int qf = nfull >> 3, rf = nfull & 7, qh = nhalf >> 3, rh = nhalf & 7;
    const int myfull = qf + (xcd < rf ? 1 : 0);
    if (idx < myfull) {
        const int f = xcd * qf + min(xcd, rf) + idx;
        const int GM = a.group_m;
        const int per_group = GM * ncols, grp = f / per_group, in = f - grp * per_group;
        const int rows = min(GM, a.tiles_m - grp * GM);
        tn = in / rows;
        tm = grp * GM + (in - tn * rows);
    } else {
        // the XCDs {rf, rf + 1, ...} (mod 8) take the nhalf % 8 extra half tiles: together with the nfull % 8 extra full ones of
        // XCDs [0, rf) that is exactly the round-robin share of every XCD
        int before = 0;
        for (int x = 0; x < xcd; x++) before += (((x - rf) & 7) < rh) ? 1 : 0;
        tm = xcd * qh + before + (idx - myfull);
        tn = a.tiles_n - 1;
    }
}

// ------------------------------------------------------------------------------------------------ 8-phase, persistent
// One workgroup per CU walks the tile list with stride gridDim.x (a multiple of 8: a workgroup stays on the XCD whose share of the
// list it walks, and the tiles of one round are the neighbours the one-tile-per-workgroup launch would run together).  What a tile
// costs outside its K loop in that launch (round 5's stamps: ~1.5 us from entry to the first MFMA -- a cold fetch of
// K-tile 0 -- and ~3.6 us from the last MFMA to "stores have left", of 49-55 us) is taken off the critical path:
//   * K-tile 0 of the NEXT tile is requested (8 global_load_lds per wavefront, buffer 0) right behind the last operand read of this
//     one and lands while the epilogue does its arithmetic;
//   * the epilogue stages through [64 KB, 136 KB) only -- buffer 1 and the tail -- in two passes of 64 n-columns (64 rows x 144 B per
//     wavefront and pass), both passes' rows held in registers, then ONE s_waitcnt vmcnt(0) (the residual-stream loads of
//     EPI_ACCUM and the prefetch: everything this wavefront asked for) and the 16 stores back to back;
//   * the stores drain under the next tile's first phases: on gfx9 they share vmcnt with the LDS-DMA loads and retire IN ISSUE ORDER
//     with them (the assumption every counted wait of this file rests on; tests/test_gemm_f16_gpu.py's 30-repeat bitwise screen is
//     what covers it), so the first counted wait behind them -- vmcnt(8) in phase 3 of K-tile 0, BAL form -- also waits for all 16
//     stores of the previous tile; nothing waits for them earlier.
// After the epilogue one barrier (every wavefront's staging reads are over and its share of K-tile 0 is in LDS), then slots 0 / 1
// of K-tile 1 are requested and the phases start; the wait of phase 4 of K-tile 0 covers them as it does in the steady state.
constexpr int EPI2_ROW = 144;                 // 64 n x 2 B + 16
constexpr int EPI2_WAVE = 64 * EPI2_ROW;      // 9216 B; 8 wavefronts: 72 KB behind buffer 0
constexpr int PBIAS_OFF = GEMM_LDS;           // 256 bias halves behind everything else
constexpr int GEMM_LDS_P = GEMM_LDS + 512;

// (plain vector values, not HIP's uint4 struct: its copies are memcpy calls between address spaces that pin the arrays in scratch)
__device__ __forceinline__ half8 add_half8(half8 a, const half8 b) {       // 8 halves + 8 halves, summed in f32
#pragma unroll
    for (int e = 0; e < 8; e++) a[e] = (_Float16)((float)a[e] + (float)b[e]);
    return a;
}
// one pass of the persistent kernel's epilogue: n-columns [64 H, 64 H + 64) of the wavefront's block -> 8 row segments per lane
template <int EPI, int H>
__device__ __forceinline__ void epi2_pass(const floatx4 (&acc)[8][4], const half4 (&bias4)[8], unsigned char* stg, int g4, int c16, int rsub,
                                 int chunk, half8 (&v)[8]) {
#pragma unroll
    for (int ii = 0; ii < 4; ii++) {
        constexpr int I0 = H * 4;
        const float b0 = (float)bias4[I0 + ii][0], b1 = (float)bias4[I0 + ii][1], b2 = (float)bias4[I0 + ii][2], b3 = (float)bias4[I0 + ii][3];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v0 = acc[I0 + ii][j][0] + b0, v1 = acc[I0 + ii][j][1] + b1, v2 = acc[I0 + ii][j][2] + b2, v3 = acc[I0 + ii][j][3] + b3;
            if (EPI == EPI_BIAS_GELU) gelu_erf4(v0, v1, v2, v3);
            const half4 hv = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
            *reinterpret_cast<half4*>(stg + (j * 16 + c16) * EPI2_ROW + (ii * 16 + g4) * 2) = hv;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // wavefront-private rows: no workgroup barrier
#pragma unroll
    for (int it = 0; it < 8; it++) v[it] = *reinterpret_cast<const half8*>(stg + (it * 8 + rsub) * EPI2_ROW + chunk * 16);
}
__device__ __forceinline__ void stage_bias(const GemmArgs& a, lds_ptr lds, int wave, int lane, int n_first) {
    if (a.bias != nullptr && wave == 0 && lane < 32) {
        const _Float16* src = a.bias + min(n_first + lane * 8, a.N - 8);
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, lds + PBIAS_OFF, 16, 0, 0);
    }
}

// Work items of the persistent walk.  With more tiles than workgroups the last round is ragged: 6 168 tiles on 256 workgroups are
// 24 rounds + 24 tiles, and those 24 cost a 25th round (792 tiles at 32 images: 3 rounds + 24, i.e. 4).  When the leftover is at most
// half a round, each leftover tile becomes TWO items -- its n-columns [0, 128) and [128, 256), i.e. one wavefront group each (the
// other group's wavefronts skip their reads and MFMAs, as they do in the half-empty last n-tile of N = 1408) -- which take ~0.6 of a
// tile's time on twice as many workgroups.  item -> (list position of the tile, half or -1)
struct WorkList {
    int nwg, first_split, nitems;
    __host__ __device__ WorkList(int nwg_, int grid, bool allow = true) : nwg(nwg_) {
        const int left = nwg > grid ? nwg % grid : 0;
        const bool split = allow && left > 0 && 2 * left <= grid;
        first_split = split ? nwg - left : nwg;
        nitems = nwg + (split ? left : 0);
    }
    __host__ __device__ void item(int w, int& pos, int& half) const {
        if (w < first_split) { pos = w; half = -1; }
        else { const int j = w - first_split; pos = first_split + (j >> 1); half = j & 1; }
    }
};

template <int EPI>
__global__ __launch_bounds__(512) void gemm_f16_8pp_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = Gemm8p<EPI, 1, 2>;
    const int stride = gridDim.x;
    const WorkList work(a.tiles_m * a.tiles_n, stride, a.split_ragged != 0);
    int bid = blockIdx.x;        // work item
    int tm, tn, pos, half;
    work.item(bid, pos, half);
    tile_of(a, pos, tm, tn);
    int m0 = tm * GB, n0 = tn * GB;
    G g(a, smem, m0, n0);
    g.active = g.active && (half < 0 || half == (g.wave >> 2));
    const int NT = a.K / GK;
    const int wave = g.wave, lane = g.lane, wn = wave >> 2, wm = wave & 3, late = wn;
    const int g4 = (lane >> 4) * 4, c16 = lane & 15;      // accumulator layout: 4 consecutive n at g4, m = c16
    const int rsub = lane >> 3, chunk = lane & 7;         // store layout: 8 rows x 8 chunks of 16 B per instruction
    unsigned char* stg = smem + G::KBUF + wave * EPI2_WAVE;
    const bool has_bias = a.bias != nullptr;
    // bias[n0 .. n0 + 256) -> LDS behind the staging rows: one 16-byte LDS-DMA load by lanes 0-31 of wavefront 0.  Issued BEFORE the
    // tile's first operand loads, so every counted wait that covers those covers it
    stage_bias(a, g.lds, wave, lane, n0);
    g.template stage<0>(0, 0); g.template stage<1>(0, 0); g.template stage<2>(0, 0); g.template stage<3>(0, 0);
    if (NT > 1) {
        g.template stage<0>(1, 1); g.template stage<1>(1, 1);
        __builtin_amdgcn_s_waitcnt(0x0F74);
    } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    g.bar();
    for (;;) {
        g.kill_fragments();
        if (late) g.bar();
        if (g.active) g.template read_q<0, 0>(g.fq0[0]);
        for (int t = 0; t < NT; t += 2) {
            g.template tile<0>(t, NT);
            if (t + 1 < NT) g.template tile<1>(t + 1, NT);
        }
        if (!late) g.bar();       // behind it nobody reads the operand buffers any more

        // ---- the tile's 256 bias values come from LDS (stage_bias(): requested with the tile's first loads).  Nothing in the
        // epilogue's arithmetic depends on a VMEM load -- a bias quad loaded here made hipcc drain vmcnt(0), prefetch included,
        // before the first addition
        half4 bias4[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const half4 ld = *reinterpret_cast<const half4*>(smem + PBIAS_OFF + (wn * 128 + i * 16 + g4) * 2);
            bias4[i] = has_bias ? ld : half4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- K-tile 0 of the next tile -> buffer 0.  Unconditional (a branch here makes hipcc drain vmcnt at the join): the last
        // tile of a workgroup requests its own K-tile 0 again, which nobody reads
        const int nbid = bid + stride;
        const bool more = nbid < work.nitems;
        const bool store = g.active;          // (a wavefront outside the item's columns holds zeros, not results)
        work.item(more ? nbid : bid, pos, half);
        tile_of(a, pos, tm, tn);
        const int m1 = tm * GB, n1 = tn * GB;
        uint32_t von[4][2];
        g.offsets(m1, n1, von);
        g.template stage_at<0>(0, 0, von); g.template stage_at<1>(0, 0, von);
        g.template stage_at<2>(0, 0, von); g.template stage_at<3>(0, 0, von);
        __builtin_amdgcn_sched_barrier(0);
        half8 old[EPI == EPI_ACCUM ? 16 : 1];       // the residual stream's row segments: used behind the vmcnt(0) below
        if (EPI == EPI_ACCUM) {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int m = min(m0 + wm * 64 + it * 8 + rsub, a.M - 1);
                    const int n = max(min(n0 + wn * 128 + h * 64 + chunk * 8, a.N - 8), 0);
                    old[h * 8 + it] = *reinterpret_cast<const half8*>(a.c + (size_t)m * a.N + n);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- bias (+ GELU), f16, transpose through the wavefront's staging rows: two passes of 64 n
        half8 v0[8], v1[8];
        epi2_pass<EPI, 0>(g.acc, bias4, stg, g4, c16, rsub, chunk, v0);
        epi2_pass<EPI, 1>(g.acc, bias4, stg, g4, c16, rsub, chunk, v1);
        if (EPI == EPI_ACCUM) {
#pragma unroll
            for (int it = 0; it < 8; it++) {
                v0[it] = add_half8(v0[it], old[it]);
                v1[it] = add_half8(v1[it], old[8 + it]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);       // this wavefront's share of the next K-tile 0 is in LDS (no store is in flight yet)
        __builtin_amdgcn_sched_barrier(0);
        {
            const int n = n0 + wn * 128 + chunk * 8;
            if (store && n + 8 <= a.N) {
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int m = m0 + wm * 64 + it * 8 + rsub;
                    if (m < a.M) *reinterpret_cast<half8*>(a.c + (size_t)m * a.N + n) = v0[it];
                }
            }
            if (store && n + 64 + 8 <= a.N) {
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int m = m0 + wm * 64 + it * 8 + rsub;
                    if (m < a.M) *reinterpret_cast<half8*>(a.c + (size_t)m * a.N + n + 64) = v1[it];
                }
            }
        }
        if (!more) break;
        bid = nbid; m0 = m1; n0 = n1;
#pragma unroll
        for (int s = 0; s < 4; s++) { g.voff[s][0] = von[s][0]; g.voff[s][1] = von[s][1]; }
        g.active = n0 + wn * 128 < a.N && (half < 0 || half == wn);
        g.zero_acc();
        g.bar();      // staging rows are free again (they lie in buffer 1), K-tile 0 is complete for everybody
        stage_bias(a, g.lds, wave, lane, n0);
        if (NT > 1) { g.template stage<0>(1, 1); g.template stage<1>(1, 1); }
    }
}


}  // namespace vlfm

using namespace vlfm;

// C = epilogue(X . W^T + bias), f16 in / f16 out / f32 accumulate.  epilogue: 0 = bias only, 1 = bias + exact (erf) GELU,
// 2 = accumulate into C (C += X . W^T + bias: the residual-stream GEMMs).  K must be a multiple of 64, N a multiple of 8; M and N
// tails are handled.  d_bias may be NULL.  One kernel: the persistent 8-phase schedule, one workgroup per CU (a multiple of 8 of them).
template <int EPI>
static int launch_gemm(const GemmArgs& a, hipStream_t stream) {
    const void* fn = reinterpret_cast<const void*>(gemm_f16_8pp_kernel<EPI>);
    static LdsOptIn optp;
    if (!optp.ensure(fn, GEMM_LDS_P)) return fail(VLFM_ERR_HIP, "gemm_f16_nt: cannot opt in to the LDS size");
    const int n_cu = device_cu_count() & ~7;       // per device (status.h): the grid and the XCD walk follow the CURRENT device
    const int nwg = a.tiles_m * a.tiles_n;
    const dim3 grid(nwg < n_cu ? nwg : n_cu), block(512);
    // (profile name by epilogue: in the ViT block 0 = qkv, 1 = fc1 + GELU, 2 = projection and fc2, one launch each)
    VLFM_TIMED(EPI == EPI_BIAS ? "gemm_f16_8pp_kernel<0>" : EPI == EPI_BIAS_GELU ? "gemm_f16_8pp_kernel<1>" : "gemm_f16_8pp_kernel<2>", stream);
    VLFM_KLAUNCH((gemm_f16_8pp_kernel<EPI>), grid, block, GEMM_LDS_P, stream, a);
    return check_launch("gemm_f16_8pp_kernel");
}

extern "C" int vlfm_gemm_f16_nt(const void* d_x, const void* d_w, const void* d_bias, void* d_c, int m, int n, int k,
                                int epilogue, void* stream) {
    if (m == 0 || n == 0) return VLFM_OK;
    if (!d_x || !d_w || !d_c || m < 0 || n < 0 || k <= 0 || (k % GK) != 0 || (n % 8) != 0 || epilogue < 0 || epilogue > 2)
        return fail(VLFM_ERR_INVALID, "gemm_f16_nt: K must be a multiple of 64, N of 8, epilogue 0, 1 or 2");
    if ((size_t)m * (size_t)k * 2 >= (1ull << 32) || (size_t)n * (size_t)k * 2 >= (1ull << 32))
        return fail(VLFM_ERR_INVALID, "gemm_f16_nt: an operand of 4 GB or more (32-bit byte offsets inside an operand)");
    GemmArgs a;
    a.x = (const _Float16*)d_x; a.w = (const _Float16*)d_w; a.bias = (const _Float16*)d_bias; a.c = (_Float16*)d_c;
    a.M = m; a.N = n; a.K = k;
    a.tiles_m = (m + GB - 1) / GB; a.tiles_n = (n + GB - 1) / GB;
    // tile order of the persistent walk: groups of 4 m-tiles on every ViT shape (probe: 2-8 within 1 %, 1 and 16+ behind); the two
    // diagnostic switches are read once per process
    static const int env_group_m = [] { const char* e = getenv("VLFM_GEMM_GROUP_M"); return e ? atoi(e) : 0; }();
    static const bool env_no_split = getenv("VLFM_GEMM_NO_SPLIT") != nullptr;
    a.group_m = env_group_m >= 1 ? env_group_m : 4;
    a.split_ragged = env_no_split ? 0 : 1;
    if (epilogue == 0) return launch_gemm<EPI_BIAS>(a, (hipStream_t)stream);
    if (epilogue == 1) return launch_gemm<EPI_BIAS_GELU>(a, (hipStream_t)stream);
    return launch_gemm<EPI_ACCUM>(a, (hipStream_t)stream);
}

// The tile order of the 8-phase kernels, evaluated on the HOST (tests/test_gemm_tile_order_cpu.py: a bijection for every shape, the
// half n-tiles last within each XCD, the persistent walk of gridDim workgroups covering every tile once): out[2 i], out[2 i + 1] =
// (m-tile, n-tile) of list position i.  group_m <= 0: the default vlfm_gemm_f16_nt chooses for the persistent kernel.
extern "C" int vlfm_gemm_f16_tile_order(int m, int n, int group_m, int* out, int capacity_pairs) {
    if (m <= 0 || n <= 0 || !out) return fail(VLFM_ERR_INVALID, "gemm_f16_tile_order: m, n > 0 and an output array");
    GemmArgs a{};
    a.M = m; a.N = n; a.K = GK;
    a.tiles_m = (m + GB - 1) / GB; a.tiles_n = (n + GB - 1) / GB;
    a.group_m = group_m > 0 ? group_m : 4;
    const int nwg = a.tiles_m * a.tiles_n;
    if (capacity_pairs < nwg) return fail(VLFM_ERR_INVALID, "gemm_f16_tile_order: output array too small");
    for (int b = 0; b < nwg; b++) tile_of(a, b, out[2 * b], out[2 * b + 1]);
    return nwg;
}

// The persistent kernel's work items for a grid of `grid` workgroups, on the host: out[2 i], out[2 i + 1] = (list position of the tile,
// n-half 0 / 1 or -1 for the whole tile) of item i; workgroup w takes items w, w + grid, ...
extern "C" int vlfm_gemm_f16_work_items(int m, int n, int grid, int* out, int capacity_pairs) {
    if (m <= 0 || n <= 0 || grid <= 0 || !out) return fail(VLFM_ERR_INVALID, "gemm_f16_work_items: m, n, grid > 0 and an output array");
    const int nwg = ((m + GB - 1) / GB) * ((n + GB - 1) / GB);
    const WorkList work(nwg, grid < nwg ? grid : nwg);
    if (capacity_pairs < work.nitems) return fail(VLFM_ERR_INVALID, "gemm_f16_work_items: output array too small");
    for (int w = 0; w < work.nitems; w++) work.item(w, out[2 * w], out[2 * w + 1]);
    return work.nitems;
}
