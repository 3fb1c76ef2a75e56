// gemm_f16.hip -- C[M][N] = epilogue(X[M][K] . W[N][K]^T + bias[N]) in f16 with f32 accumulation on the gfx950 matrix
// cores, written for the ViT-g MLP of the BLIP-2 forward (reference: LAVIS eva_vit's Mlp behind vlfm/vlm/blip2itm.py:52):
// fc1 is [65 792 x 1408] . [6144 x 1408]^T followed by an exact (erf) GELU.  hipBLASLt offers GELU only in its tanh form, so
// the library path pays a separate 5.4 TB/s elementwise pass over the 808 MB activation; here the GELU runs on the f32
// accumulators in the epilogue and the activation is written once.
//
// Structure (one workgroup = 256 x 256 output tile, 8 wavefronts as 2 (n) x 4 (m), K-step 64):
//   * both operands are K-contiguous ("NT" GEMM): a K-tile of W and of X is 256 rows x 128 B each.  They are staged by
//     global_load_lds (16 B per lane, no registers, no ds_write): a wavefront's instruction fills 8 rows x 128 B of LDS
//     linearly; the 16-byte slot a lane FETCHES is permuted within its row (slot ^ ((row >> 1) & 7)), so that the MFMA
//     fragment reads -- 16 lanes x 16 B from 16 consecutive rows -- hit 16 different bank groups (rows r and r + 1 differ
//     in the upper / lower half of the 256-byte bank row, pairs differ in the slot): conflict-free ds_read_b128, while the
//     8 lanes of a row still read one whole 128-byte line of global memory.
//   * two LDS buffers (128 KB); tile t + 2 is requested as soon as every wavefront has finished reading tile t, one
//     s_barrier per K-tile; the wait in front of it is for exactly the loads of tile t + 1 (nothing newer is in flight).
//   * the MFMA operand fragments are software-pipelined through two register stages of half a K-tile each: the ds_reads of
//     the next half are issued in front of the 32 MFMAs of the current half.
//   * operand roles are swapped (A-operand = W rows, B-operand = X rows): a lane then holds 4 CONSECUTIVE n of one m, i.e.
//     8 contiguous bytes of the output row after conversion; the epilogue transposes through LDS (row stride 272 B:
//     conflict-free 8-byte writes, 16-byte reads) and stores 256-byte row segments with 16 B per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/vlfm_amd.h"
#include "profile.h"
#include "status.h"

namespace vlfm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
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

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1 };

struct GemmArgs {
    const _Float16* x;     // [M][K]
    const _Float16* w;     // [N][K]
    const _Float16* bias;  // [N] or null
    _Float16* c;           // [M][N]
    int M, N, K;
    int tiles_m, tiles_n;
};

// Exact-form GELU, 0.5 v (1 + erf(v / sqrt 2)), without the library erff (two polynomial branches + exp, ~36 VALU instructions
// per value once both sides of the branch run in a wavefront).  erfc(u / sqrt 2) = exp2(-u q(u)) with q a degree-7 polynomial
// (weighted minimax fit on u in [0, 5.55], monotone beyond: the tail underflows to 0 like erfc), so
//   gelu(v) = max(v, 0) - 0.5 |v| exp2(-|v| q(|v|)):   9 FMA/mul + v_exp_f32 + max, no branch, no cancellation in the negative tail.
// |gelu - f64| <= 2.8e-7 over [-12, 12] (torch's f32 erf form: 1.2e-6); tools/gemm_f16_probe.py checks it against torch.
__device__ inline float gelu_erf(float v) {
    const float u = fabsf(v);
    float q = 2.834908400e-06f;
    q = fmaf(q, u, -3.937762449e-05f);
    q = fmaf(q, u, 1.861798810e-04f);
    q = fmaf(q, u, 1.369373058e-04f);
    q = fmaf(q, u, -7.063421421e-03f);
    q = fmaf(q, u, 5.249617994e-02f);
    q = fmaf(q, u, 4.592081904e-01f);
    q = fmaf(q, u, 1.151105165e+00f);
    const float e = __builtin_amdgcn_exp2f(-(q * u));
    return fmaf(-0.5f * u, e, fmaxf(v, 0.0f));
}

using lds_ptr = __attribute__((address_space(3))) unsigned char*;
using gbl_ptr = const __attribute__((address_space(1))) unsigned char*;

// one K-tile of both operands -> LDS buffer `buf` (byte offset): 4 + 4 global_load_lds per wavefront
__device__ inline void stage_tile(const GemmArgs& a, lds_ptr lds, int buf, int n0, int m0, int k0, int wave, int lane) {
    const int sub = lane >> 3, p = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int chunk = wave * 4 + j;                 // 8 rows each
        const int r = chunk * 8 + sub;
        const int s = p ^ ((r >> 1) & 7);               // the slot this lane fetches lands at physical slot p
        const int rn = min(n0 + r, a.N - 1), rm = min(m0 + r, a.M - 1);
        const _Float16* gw = a.w + (size_t)rn * a.K + k0 + s * 8;
        const _Float16* gx = a.x + (size_t)rm * a.K + k0 + s * 8;
        const int dst = __builtin_amdgcn_readfirstlane(buf + chunk * 1024);
        __builtin_amdgcn_global_load_lds((gbl_ptr)gw, lds + dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr)gx, lds + dst + OPER, 16, 0, 0);
    }
}

// MFMA fragments of one half K-tile (32 of the 64 k): 8 A-operand (W rows) + 4 B-operand (X rows) ds_read_b128
__device__ inline void read_frags(const unsigned char* smem, int buf, int kk, int wn, int wm, int lane, half8 (&fa)[8],
                                  half8 (&fb)[4]) {
    const int r16 = lane & 15, s = kk * 4 + (lane >> 4);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int R = wn * 128 + i * 16 + r16;
        fa[i] = *reinterpret_cast<const half8*>(smem + buf + R * ROWB + ((s ^ ((R >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int R = wm * 64 + j * 16 + r16;
        fb[j] = *reinterpret_cast<const half8*>(smem + buf + OPER + R * ROWB + ((s ^ ((R >> 1) & 7)) << 4));
    }
}

__device__ inline void mma_half(const half8 (&fa)[8], const half8 (&fb)[4], floatx4 (&acc)[8][4]) {
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
}

template <int EPI, int EXP = 0>
__global__ __launch_bounds__(512) void gemm_f16_nt_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_ptr lds = (lds_ptr)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;
    // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs; give every XCD a contiguous range of
    // tiles (n fastest) so that the tiles sharing an X row panel and the W matrix meet in one L2
    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / a.tiles_n, tn = bid - tm * a.tiles_n;
    const int m0 = tm * GB, n0 = tn * GB;
    const int NT = a.K / GK;

    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 fa0[8], fb0[4], fa1[8], fb1[4];

    stage_tile(a, lds, 0, n0, m0, 0, wave, lane);
    if (NT > 1) stage_tile(a, lds, BUF, n0, m0, GK, wave, lane);
    if (NT > 1) __builtin_amdgcn_s_waitcnt(0x0F78);
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(smem, 0, 0, wn, wm, lane, fa0, fb0);
    __builtin_amdgcn_s_waitcnt(0xC07F);

    for (int t = 0; t < NT; t++) {
        const int cur = (t & 1) * BUF;
        read_frags(smem, cur, 1, wn, wm, lane, fa1, fb1);     // second half of tile t: in flight under the MFMAs below
        __builtin_amdgcn_sched_barrier(0);
        if (EXP & 2) __builtin_amdgcn_s_setprio(1);
        mma_half(fa0, fb0, acc);
        if (EXP & 2) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);    // every LDS read of tile t by this wavefront has returned
        if (t + 1 < NT) {
            if (!(EXP & 1)) __builtin_amdgcn_s_waitcnt(0x0F70);  // ... and its share of tile t + 1 has landed (nothing newer is in flight)
            __builtin_amdgcn_s_barrier();                      // ... for everybody: buffer `cur` is free, the other one is complete
            asm volatile("" ::: "memory");
            if (t + 2 < NT) stage_tile(a, lds, cur, n0, m0, (t + 2) * GK, wave, lane);
            read_frags(smem, cur ^ BUF, 0, wn, wm, lane, fa0, fb0);  // first half of tile t + 1: in flight under the MFMAs below
        }
        __builtin_amdgcn_sched_barrier(0);
        if (EXP & 2) __builtin_amdgcn_s_setprio(1);
        mma_half(fa1, fb1, acc);
        if (EXP & 2) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }

    // ---- epilogue: bias (+ exact GELU) on the f32 accumulators, f16, transpose through LDS, 16-byte stores
    __builtin_amdgcn_s_barrier();   // every wavefront is done with the operand buffers
    asm volatile("" ::: "memory");
    unsigned char* stg = smem + wave * EPI_WAVE;
    const int g4 = (lane >> 4) * 4, c16 = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int nl = i * 16 + g4;                    // 4 consecutive n of this lane, wavefront-local
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        if (a.bias) {
            const int n = min(n0 + wn * 128 + nl, a.N - 4);
            const half4 bv = *reinterpret_cast<const half4*>(a.bias + n);
            b0 = (float)bv[0]; b1 = (float)bv[1]; b2 = (float)bv[2]; b3 = (float)bv[3];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v0 = acc[i][j][0] + b0, v1 = acc[i][j][1] + b1, v2 = acc[i][j][2] + b2, v3 = acc[i][j][3] + b3;
            if (EPI == EPI_BIAS_GELU) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
            const half4 h = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
            *reinterpret_cast<half4*>(stg + (j * 16 + c16) * EPI_ROW + nl * 2) = h;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // wavefront-private region: no workgroup barrier needed
    const int rsub = lane >> 4, chunk = lane & 15;
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const int row = it * 4 + rsub;
        const uint4 v = *reinterpret_cast<const uint4*>(stg + row * EPI_ROW + chunk * 16);
        const int m = m0 + wm * 64 + row, n = n0 + wn * 128 + chunk * 8;
        if (m < a.M && n + 8 <= a.N) *reinterpret_cast<uint4*>(a.c + (size_t)m * a.N + n) = v;
    }
}


// ------------------------------------------------------------------------------------------------ two workgroups per CU
// Same wavefront tile (128 n x 64 m, 8 + 4 fragments, 32 MFMAs per K-step of 32) but a workgroup is 4 wavefronts on a
// 256 (n) x 128 (m) output tile with a K-tile of 32 in a ring of three 24 KB LDS stages: 72 KB, so TWO workgroups share a CU.
// The two wavefronts of a SIMD then belong to different workgroups with unrelated phases: the barrier + staging + fragment
// reads of one, and above all its EPILOGUE (bias, erf-GELU, f16, LDS transpose, stores: ~30 VALU instructions per output
// value in which the one-workgroup form leaves the matrix pipe idle), run under the MFMAs of the other.
//   iteration t:  wait own fragment reads of tile t | request tile t + 2 into the stage tile t - 1 lived in (everybody finished
//   reading it before the previous barrier) | wait tile t + 1 (the 6 newest loads may stay in flight) | barrier | issue the
//   fragment reads of tile t + 1 | 32 MFMAs on tile t.
// Staged rows are 64 B: a bank row holds 4 of them, the slot permutation is [0,3,2,1][(row >> 2) & 3] (conflict-free for the
// four lane groups of ds_read_b128, checked exhaustively).
constexpr int G2_K = 32;
constexpr int G2_ROWB = G2_K * 2;            // 64 B
constexpr int G2_TN = 256, G2_TM = 128;
constexpr int G2_W = G2_TN * G2_ROWB;        // 16 KB
constexpr int G2_STAGE = (G2_TN + G2_TM) * G2_ROWB;   // 24 KB
constexpr int G2_LDS = 3 * G2_STAGE;         // 72 KB (>= 4 epilogue regions of 17 KB)
static_assert(4 * EPI_WAVE <= G2_LDS, "epilogue staging fits the operand ring");

__device__ inline int g2_perm(int r) { return (0x6Cu >> (2 * ((r >> 2) & 3))) & 3; }   // [0,3,2,1]

__device__ inline void g2_stage(const GemmArgs& a, lds_ptr lds, int buf, int n0, int m0, int k0, int wave, int lane) {
    const int sub = lane >> 2, p = lane & 3;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int q = wave * 6 + j;                      // 24 chunks of 16 rows: 16 of W, 8 of X
        const bool is_w = q < 16;                        // uniform
        const int r = (is_w ? q : q - 16) * 16 + sub;
        const int s = p ^ g2_perm(r);
        const int row = is_w ? min(n0 + r, a.N - 1) : min(m0 + r, a.M - 1);
        const _Float16* g = (is_w ? a.w : a.x) + (size_t)row * a.K + k0 + s * 8;
        const int dst = __builtin_amdgcn_readfirstlane(buf + q * 1024);   // W chunks, then X chunks: q * 1 KB either way
        __builtin_amdgcn_global_load_lds((gbl_ptr)g, lds + dst, 16, 0, 0);
    }
}

__device__ inline void g2_read_frags(const unsigned char* smem, int buf, int wn, int wm, int lane, half8 (&fa)[8], half8 (&fb)[4]) {
    const int r16 = lane & 15, s = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int R = wn * 128 + i * 16 + r16;
        fa[i] = *reinterpret_cast<const half8*>(smem + buf + R * G2_ROWB + ((s ^ g2_perm(R)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int R = wm * 64 + j * 16 + r16;
        fb[j] = *reinterpret_cast<const half8*>(smem + buf + G2_W + R * G2_ROWB + ((s ^ g2_perm(R)) << 4));
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f16_nt2_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_ptr lds = (lds_ptr)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {   // every XCD gets a contiguous range of tiles ...
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // ... walked in groups of 8 m-tiles with n slowest inside a group: the 64 tiles an XCD works on at one time are about
    // 8 (m) x 8 (n), i.e. 16 operand panels instead of 24 + 3
    int tm, tn;
    {
        const int per_group = 8 * a.tiles_n, grp = bid / per_group, in = bid - grp * per_group;
        const int rows = min(8, a.tiles_m - grp * 8);
        tn = in / rows;
        tm = grp * 8 + (in - tn * rows);
    }
    const int m0 = tm * G2_TM, n0 = tn * G2_TN;
    const int NT = a.K / G2_K;

    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 fa0[8], fb0[4], fa1[8], fb1[4];

    g2_stage(a, lds, 0, n0, m0, 0, wave, lane);
    if (NT > 1) g2_stage(a, lds, G2_STAGE, n0, m0, G2_K, wave, lane);
    if (NT > 1) __builtin_amdgcn_s_waitcnt(0x0F76);   // vmcnt(6): tile 0 has landed
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    g2_read_frags(smem, 0, wn, wm, lane, fa0, fb0);

    auto step = [&](int t, half8 (&fa)[8], half8 (&fb)[4], half8 (&na)[8], half8 (&nb)[4]) {
        __builtin_amdgcn_s_waitcnt(0xC07F);                                      // fragments of tile t are here
        if (t + 1 < NT) {
            if (t + 2 < NT) {
                g2_stage(a, lds, ((t + 2) % 3) * G2_STAGE, n0, m0, (t + 2) * G2_K, wave, lane);
                __builtin_amdgcn_s_waitcnt(0x0F76);                              // tile t + 1 landed (tile t + 2 may fly)
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            g2_read_frags(smem, ((t + 1) % 3) * G2_STAGE, wn, wm, lane, na, nb);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_half(fa, fb, acc);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t < NT; t += 2) {
        step(t, fa0, fb0, fa1, fb1);
        if (t + 1 < NT) step(t + 1, fa1, fb1, fa0, fb0);
    }

    // ---- epilogue (as above; the wavefront tile is the same)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    unsigned char* stg = smem + wave * EPI_WAVE;
    const int g4 = (lane >> 4) * 4, c16 = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int nl = i * 16 + g4;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        if (a.bias) {
            const int n = min(n0 + wn * 128 + nl, a.N - 4);
            const half4 bv = *reinterpret_cast<const half4*>(a.bias + n);
            b0 = (float)bv[0]; b1 = (float)bv[1]; b2 = (float)bv[2]; b3 = (float)bv[3];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v0 = acc[i][j][0] + b0, v1 = acc[i][j][1] + b1, v2 = acc[i][j][2] + b2, v3 = acc[i][j][3] + b3;
            if (EPI == EPI_BIAS_GELU) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
            const half4 h = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
            *reinterpret_cast<half4*>(stg + (j * 16 + c16) * EPI_ROW + nl * 2) = h;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const int rsub = lane >> 4, chunk = lane & 15;
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const int row = it * 4 + rsub;
        const uint4 v = *reinterpret_cast<const uint4*>(stg + row * EPI_ROW + chunk * 16);
        const int m = m0 + wm * 64 + row, n = n0 + wn * 128 + chunk * 8;
        if (m < a.M && n + 8 <= a.N) *reinterpret_cast<uint4*>(a.c + (size_t)m * a.N + n) = v;
    }
}

}  // namespace vlfm

using namespace vlfm;

// C = epilogue(X . W^T + bias), f16 in / f16 out / f32 accumulate.  epilogue: 0 = bias only, 1 = bias + exact (erf) GELU.
// K must be a multiple of 64, N a multiple of 8; M and N tails are handled.  d_bias may be NULL.
extern "C" int vlfm_gemm_f16_nt(const void* d_x, const void* d_w, const void* d_bias, void* d_c, int m, int n, int k,
                                int epilogue, void* stream) {
    if (m == 0 || n == 0) return VLFM_OK;
    if (!d_x || !d_w || !d_c || m < 0 || n < 0 || k <= 0 || (k % G2_K) != 0 || (n % 8) != 0 || epilogue < 0 || epilogue > 1)
        return fail(VLFM_ERR_INVALID, "gemm_f16_nt: K must be a multiple of 32, N of 8, epilogue 0 or 1");
    GemmArgs a;
    a.x = (const _Float16*)d_x; a.w = (const _Float16*)d_w; a.bias = (const _Float16*)d_bias; a.c = (_Float16*)d_c;
    a.M = m; a.N = n; a.K = k;
    a.tiles_m = (m + GB - 1) / GB; a.tiles_n = (n + GB - 1) / GB;
    static LdsOptIn opt0, opt1;
    const bool ok = epilogue == 0 ? opt0.ensure(reinterpret_cast<const void*>(gemm_f16_nt_kernel<EPI_BIAS>), GEMM_LDS)
                                  : opt1.ensure(reinterpret_cast<const void*>(gemm_f16_nt_kernel<EPI_BIAS_GELU>), GEMM_LDS);
    if (!ok) return fail(VLFM_ERR_HIP, "gemm_f16_nt: cannot opt in to the LDS size");
    const char* ev = getenv("VLFM_GEMM_VARIANT");   // 1: 256 x 256 tiles, one workgroup per CU; 2: 256 x 128, two per CU
    const int variant = ev ? atoi(ev) : ((k % GK) == 0 ? 1 : 2);
    if (variant == 2 && (k % G2_K) == 0) {   // two workgroups per CU (256 x 128 tiles)
        a.tiles_m = (m + G2_TM - 1) / G2_TM; a.tiles_n = (n + G2_TN - 1) / G2_TN;
        static LdsOptIn p0, p1;
        const bool ok2 = epilogue == 0 ? p0.ensure(reinterpret_cast<const void*>(gemm_f16_nt2_kernel<EPI_BIAS>), G2_LDS)
                                       : p1.ensure(reinterpret_cast<const void*>(gemm_f16_nt2_kernel<EPI_BIAS_GELU>), G2_LDS);
        if (!ok2) return fail(VLFM_ERR_HIP, "gemm_f16_nt: cannot opt in to the LDS size");
        const dim3 grid2(a.tiles_m * a.tiles_n), block2(256);
        VLFM_TIMED("gemm_f16_nt2_kernel", stream);
        if (epilogue == 0) VLFM_KLAUNCH(gemm_f16_nt2_kernel<EPI_BIAS>, grid2, block2, G2_LDS, (hipStream_t)stream, a);
        else VLFM_KLAUNCH(gemm_f16_nt2_kernel<EPI_BIAS_GELU>, grid2, block2, G2_LDS, (hipStream_t)stream, a);
        return check_launch("gemm_f16_nt2_kernel");
    }
    if ((k % GK) != 0) return fail(VLFM_ERR_INVALID, "gemm_f16_nt: the one-workgroup-per-CU variant needs K % 64 == 0");
    const dim3 grid(a.tiles_m * a.tiles_n), block(512);
    if (const char* e = getenv("VLFM_GEMM_EXP")) {   // experiments (tools/gemm_f16_probe.py): 1 = no load wait (WRONG results), 2 = setprio
        const int x = atoi(e);
        static LdsOptIn o1, o2, o3;
        if (x == 1 && o1.ensure(reinterpret_cast<const void*>(gemm_f16_nt_kernel<EPI_BIAS, 1>), GEMM_LDS)) {
            VLFM_KLAUNCH((gemm_f16_nt_kernel<EPI_BIAS, 1>), grid, block, GEMM_LDS, (hipStream_t)stream, a); return check_launch("gemm exp1"); }
        if (x == 2 && o2.ensure(reinterpret_cast<const void*>(gemm_f16_nt_kernel<EPI_BIAS, 2>), GEMM_LDS)) {
            VLFM_KLAUNCH((gemm_f16_nt_kernel<EPI_BIAS, 2>), grid, block, GEMM_LDS, (hipStream_t)stream, a); return check_launch("gemm exp2"); }
        if (x == 3 && o3.ensure(reinterpret_cast<const void*>(gemm_f16_nt_kernel<EPI_BIAS, 3>), GEMM_LDS)) {
            VLFM_KLAUNCH((gemm_f16_nt_kernel<EPI_BIAS, 3>), grid, block, GEMM_LDS, (hipStream_t)stream, a); return check_launch("gemm exp3"); }
    }
    VLFM_TIMED("gemm_f16_nt_kernel", stream);
    if (epilogue == 0) VLFM_KLAUNCH(gemm_f16_nt_kernel<EPI_BIAS>, grid, block, GEMM_LDS, (hipStream_t)stream, a);
    else VLFM_KLAUNCH(gemm_f16_nt_kernel<EPI_BIAS_GELU>, grid, block, GEMM_LDS, (hipStream_t)stream, a);
    return check_launch("gemm_f16_nt_kernel");
}
