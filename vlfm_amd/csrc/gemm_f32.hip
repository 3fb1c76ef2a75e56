// gemm_f32.hip -- C[M][N] = act(X[M][K] . W[N][K]^T + bias[N]) (+ residual[M][N]) for f32 operands on the gfx950 matrix cores:
// the Linear layers of GroundingDINO (reference: vlfm/vlm/grounding_dino.py:38-74 runs the network in fp32; 15.7 TFLOP of such
// GEMMs per 64-frame batch) and of MobileSAM's TinyViT (vlfm/vlm/sam.py:40-57).  gfx950 has no TF32-like mode, so "fp32 GEMM"
// has two honest forms here, both with f32 inputs, f32 outputs and f32 accumulation:
//
//   precision 0  EXACT: v_mfma_f32_32x32x2_f32.  Bit for bit a k-ordered f32 fma chain (one rounding per product), at the f32
//                vector rate (157 TFLOP/s peak).  One f32 VGPR per operand per lane, 64 cycles per instruction: the matrix pipe is
//                saturated by one wavefront per SIMD, LDS and HBM are nearly idle -- what matters is never to stall the pipe.
//   precision 1  SPLIT: every f32 operand a is written as hi + 2^-11 lo' with hi = f16(a), lo' = f16((a - hi) 2^11): the
//                remainder a - hi is exact in f32 and has at most 13 significant bits, of which lo' keeps 11, so
//                |a - (hi + 2^-11 lo')| <= 2^-24 |a| -- f32's own unit roundoff.  Then a b = hi_a hi_b + 2^-11 (hi_a lo'_b + lo'_a hi_b)
//                + O(2^-24 |a b|): THREE v_mfma_f32_32x32x16_f16 (exact 22-bit products, f32 accumulation) in place of eight
//                32x32x2_f32, i.e. 16 / 3 = 5.3 x the matrix rate of the exact form (833 TFLOP/s-equivalent peak).  The two cross terms
//                share one accumulator that is scaled by 2^-11 in the epilogue.  |a| >= 65504 cannot be represented: the kernel
//                raises a sticky device flag instead of producing infinities silently and the caller re-runs the exact form
//                (vlfm_gemm_f32_overflow_flag).  tests/test_gemm_f32_gpu.py measures both forms against an f64 reference.
//
// Tile: 128 (m) x 128 (n) per workgroup of 4 wavefronts (2 x 2, each 64 x 64 = 2 x 2 MFMA tiles of 32 x 32), K-tile 32 floats =
// 128-byte rows -- the row geometry of gemm_f16.hip, so staging (global_load_lds, 16 B per lane, slot ^ ((row >> 1) & 7)) and the
// conflict-free ds_read_b128 fragment reads carry over: lane (r = lane & 31, h = lane >> 5) reads the four consecutive k of slot
// 2 kk + h of its row; register q of that read feeds MFMA q, whose k pair is (4 (2 kk) + q, 4 (2 kk + 1) + q) -- the same permutation
// of k for both operands.  Two LDS buffers of 32 KB (+ the epilogue's transposition area): two workgroups per CU.
// Operand roles are swapped (A-operand = W rows) so that a lane holds 4 consecutive n of one m; the epilogue transposes through a
// wavefront-private LDS area (row stride 272 B: conflict-free 16-byte writes) and stores 256-byte row segments, 16 B per lane,
// adding the residual on the way out.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/vlfm_amd.h"
#include "profile.h"
#include "status.h"

namespace vlfm {
namespace g32 {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int TB = 128;            // tile rows (m and n)
constexpr int TK = 32;             // floats of K per tile
constexpr int ROWB = TK * 4;       // 128-byte staged rows
constexpr int OPER = TB * ROWB;    // 16 KB per operand tile
constexpr int BUF = 2 * OPER;      // W tile + X tile
constexpr int EPI_ROW = 272;       // 64 n x 4 B + 16
constexpr int EPI_WAVE = 64 * EPI_ROW;
constexpr int LDS_EXACT = 4 * EPI_WAVE > 2 * BUF ? 4 * EPI_WAVE : 2 * BUF;

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

struct Args {
    const float* x;         // [M][K]
    const float* w;         // [N][K]
    const float* bias;      // [N] or null
    const float* residual;  // [M][N] or null (may alias c)
    float* c;               // [M][N]
    int M, N, K;
    int tiles_m, tiles_n;
    int* overflow;          // split form: sticky flag
    const _Float16* w_hi;   // split form: the weights, split once on the host side of the ABI ([N][K] each)
    const _Float16* w_lo;
};

// exact-form GELU as in gemm_f16.hip (|gelu - f64| <= 2.8e-7)
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

__device__ inline void tile_of_block(const Args& a, int& tm, int& tn) {
    // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs; give every XCD a contiguous range of tiles
    // (n fastest) so that the tiles sharing an X row panel and the (small) W matrix meet in one L2
    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    tm = bid / a.tiles_n;
    tn = bid - tm * a.tiles_n;
}

// one K-tile (32 floats) of both operands -> LDS buffer `buf`: 4 + 4 global_load_lds per wavefront (8 rows x 128 B each)
__device__ inline void stage_tile(const Args& a, lds_ptr lds, int buf, int n0, int m0, int k0, int wave, int lane) {
    const int sub = lane >> 3, p = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int chunk = wave * 4 + j;
        const int r = chunk * 8 + sub;
        const int s = p ^ ((r >> 1) & 7);
        const int rn = min(n0 + r, a.N - 1), rm = min(m0 + r, a.M - 1);
        const float* gw = a.w + (size_t)rn * a.K + k0 + s * 4;
        const float* gx = a.x + (size_t)rm * a.K + k0 + s * 4;
        const int dst = __builtin_amdgcn_readfirstlane(buf + chunk * 1024);
        __builtin_amdgcn_global_load_lds((gbl_ptr)gw, lds + dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr)gx, lds + dst + OPER, 16, 0, 0);
    }
}

// bias + activation on the accumulators, transposition through the wavefront's LDS area, residual, 16-byte stores.
// acc[i][j][reg] = C[m = wm 64 + j 32 + (lane & 31)][n = wn 64 + i 32 + 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3)]
template <bool SPLIT>
__device__ inline void store_tile(const Args& a, unsigned char* smem, const floatx16 (&acc)[2][2], const floatx16 (&cross)[2][2],
                                  int act, int wave, int wn, int wm, int lane, int m0, int n0) {
    unsigned char* stg = smem + wave * EPI_WAVE;
    const int c32 = lane & 31, h4 = (lane >> 5) * 4;
    floatx4 bias4[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            floatx4 b = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
                const int n = n0 + wn * 64 + i * 32 + g * 8 + h4;
                if (n + 4 <= a.N && ((uintptr_t)a.bias & 15) == 0) b = *reinterpret_cast<const floatx4*>(a.bias + n);
                else
                    for (int e = 0; e < 4; e++) b[e] = n + e < a.N ? a.bias[n + e] : 0.f;
            }
            bias4[i][g] = b;
        }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int nl = i * 32 + g * 8 + h4;                       // 4 consecutive n, wavefront-local
            const floatx4 b = bias4[i][g];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                floatx4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float t = acc[i][j][g * 4 + e];
                    if (SPLIT) t = fmaf(cross[i][j][g * 4 + e], 0x1p-11f, t);
                    t += b[e];
                    if (act == ACT_RELU) t = fmaxf(t, 0.f);
                    else if (act == ACT_GELU) t = gelu_erf(t);
                    v[e] = t;
                }
                *reinterpret_cast<floatx4*>(stg + (j * 32 + c32) * EPI_ROW + nl * 4) = v;
            }
        }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // wavefront-private area
    const int rsub = lane >> 4, chunk = lane & 15;
    const bool vec = (a.N & 3) == 0;      // rows of C / residual are 16-byte aligned
    // Round 5 (as in gemm_f16.hip): all 16 residual loads first, then all 16 LDS reads, then the stores -- the loop that loaded,
    // read, added and stored one row at a time (unrolled by four, one register quad per slot) waited for a residual load AND for
    // the previous store before every LDS read.
    const int n = n0 + wn * 64 + chunk * 4;
    const bool fast = vec && n + 4 <= a.N;
    floatx4 r[16], v[16];
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const int m = m0 + wm * 64 + it * 4 + rsub;
        r[it] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (a.residual && fast && m < a.M) r[it] = *reinterpret_cast<const floatx4*>(a.residual + (size_t)m * a.N + n);
    }
#pragma unroll
    for (int it = 0; it < 16; it++) v[it] = *reinterpret_cast<const floatx4*>(stg + (it * 4 + rsub) * EPI_ROW + chunk * 16);
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const int m = m0 + wm * 64 + it * 4 + rsub;
        if (m >= a.M || n >= a.N) continue;
        const size_t at = (size_t)m * a.N + n;
        if (fast) {
            *reinterpret_cast<floatx4*>(a.c + at) = v[it] + r[it];
        } else {
            for (int e = 0; e < 4 && n + e < a.N; e++) a.c[at + e] = v[it][e] + (a.residual ? a.residual[at + e] : 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------ precision 0: exact
// Lock-step over the K-tiles with the next tile's loads issued first; the two workgroups of a CU (and the 16 x 64-cycle MFMAs
// behind every four ds_read_b128) cover the barrier.
__global__ __launch_bounds__(256, 2) void gemm_f32_exact_kernel(Args a, int act) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_ptr lds = (lds_ptr)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    int tm, tn;
    tile_of_block(a, tm, tn);
    const int m0 = tm * TB, n0 = tn * TB;
    const int NT = a.K / TK;

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    stage_tile(a, lds, 0, n0, m0, 0, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const int r32 = lane & 31, h = lane >> 5;
    for (int t = 0; t < NT; t++) {
        const int cur = (t & 1) * BUF;
        if (t + 1 < NT) stage_tile(a, lds, cur ^ BUF, n0, m0, (t + 1) * TK, wave, lane);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int s = kk * 2 + h;
            floatx4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int R = wn * 64 + i * 32 + r32;
                fa[i] = *reinterpret_cast<const floatx4*>(smem + cur + R * ROWB + ((s ^ ((R >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int R = wm * 64 + j * 32 + r32;
                fb[j] = *reinterpret_cast<const floatx4*>(smem + cur + OPER + R * ROWB + ((s ^ ((R >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q], fb[j][q], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // this wavefront's share of tile t + 1 has landed
        __builtin_amdgcn_s_barrier();         // ... everybody's, and everybody is done reading tile t
        asm volatile("" ::: "memory");
    }
    store_tile<false>(a, smem, acc, acc, act, wave, wn, wm, lane, m0, n0);
}

// ------------------------------------------------------------------------------------------------ precision 1: f16 hi/lo split
// LDS per buffer: X hi | X lo | W hi | W lo, each 128 rows x 64 B (32 k as f16) with the 16-byte slot of a row XORed by
// (row >> 2) & 3 (rows are 64 B: four rows per 256-byte bank period).  X is split on the way in (global f32 -> registers ->
// hi / lo -> ds_write_b64 x 2); W arrives already split (w_hi / w_lo, global_load_lds).  A K-tile of 32 is two MFMA k-steps of
// 16; per step and wavefront 8 ds_read_b128 feed 12 MFMAs of 32 cycles.
constexpr int SROWB = TK * 2;          // 64-byte rows
constexpr int SOPER = TB * SROWB;      // 8 KB
constexpr int SBUF = 4 * SOPER;        // 32 KB

__device__ inline int sslot(int row, int slot) { return (slot ^ ((row >> 2) & 3)) << 4; }

__device__ inline void split4(const floatx4 v, half4& hi, half4& lo, bool& over) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float a = v[e];
        over |= !(fabsf(a) < 65504.0f);                        // (also catches NaN / inf inputs)
        const _Float16 hh = (_Float16)a;
        hi[e] = hh;
        lo[e] = (_Float16)((a - (float)hh) * 2048.0f);
    }
}

// NI = 32-row MFMA tiles of W per wavefront: 2 -> the 128 (n) x 128 (m) workgroup tile above, two workgroups per CU;
// 4 -> a 256 (n) x 128 (m) tile (wavefronts of 128 x 64 = 4 x 2 MFMA tiles, 256 accumulator registers, ONE workgroup per CU): every
// X tile is converted and staged for twice as many MFMAs (the conversion of X is repeated by every workgroup along n), and a k-step's
// 12 fragment reads feed 24 MFMAs instead of 8 feeding 12 -- the LDS read rate, not the matrix pipe, bounds the small tile.
template <int NI>
struct SplitGeom {
    static constexpr int TN = 64 * NI;               // workgroup tile along n (two wavefronts)
    static constexpr int WOPER = TN * SROWB;         // bytes of one W plane tile
    static constexpr int XO = 0, WO = 2 * SOPER;     // X hi | X lo | W hi | W lo
    static constexpr int BUF = 2 * SOPER + 2 * WOPER;
    static constexpr int LDS = 4 * EPI_WAVE > 2 * BUF ? 4 * EPI_WAVE : 2 * BUF;
};

template <int NI>
__device__ __forceinline__ void gemm_f32_split_body(const Args& a, int act) {
    using G = SplitGeom<NI>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_ptr lds = (lds_ptr)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    int tm, tn;
    tile_of_block(a, tm, tn);
    const int m0 = tm * TB, n0 = tn * G::TN;
    const int NT = a.K / TK;

    floatx16 acc[NI][2], crs[NI][2];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) { acc[i][j][e] = 0.f; crs[i][j][e] = 0.f; }

    // X: 128 rows x 32 floats per K-tile = 1024 float4; thread tid takes float4 index tid + 256 u (u < 4): row = idx >> 3,
    // piece = idx & 7 (4 floats = 8 bytes of f16 -> half a 16-byte slot)
    floatx4 xr[4];
    bool over = false;
    auto load_x = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = tid + 256 * u, r = idx >> 3, pc = idx & 7;
            const int rm = min(m0 + r, a.M - 1);
            xr[u] = *reinterpret_cast<const floatx4*>(a.x + (size_t)rm * a.K + k0 + pc * 4);
        }
    };
    auto put_x = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = tid + 256 * u, r = idx >> 3, pc = idx & 7;
            half4 hi, lo;
            split4(xr[u], hi, lo, over);
            const int off = r * SROWB + sslot(r, pc >> 1) + (pc & 1) * 8;
            *reinterpret_cast<half4*>(smem + buf + G::XO + off) = hi;
            *reinterpret_cast<half4*>(smem + buf + G::XO + SOPER + off) = lo;
        }
    };
    // W hi / lo: TN rows x 64 B per plane; one global_load_lds instruction covers 16 rows (4 lanes per row), NI per wavefront and plane
    auto stage_w = [&](int buf, int k0) {
        const int sub = lane >> 2, p = lane & 3;
#pragma unroll
        for (int j = 0; j < NI; j++) {
            const int chunk = wave * NI + j;                // 16 rows each
            const int r = chunk * 16 + sub;
            const int s = p ^ ((r >> 2) & 3);
            const int rn = min(n0 + r, a.N - 1);
            const size_t g = (size_t)rn * a.K + k0 + s * 8;
            const int dst = __builtin_amdgcn_readfirstlane(buf + G::WO + chunk * 1024);
            __builtin_amdgcn_global_load_lds((gbl_ptr)(a.w_hi + g), lds + dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr)(a.w_lo + g), lds + dst + G::WOPER, 16, 0, 0);
        }
    };

    load_x(0);
    stage_w(0, 0);
    put_x(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const int r32 = lane & 31, h = lane >> 5;
    for (int t = 0; t < NT; t++) {
        const int cur = (t & 1) * G::BUF;
        if (t + 1 < NT) {
            load_x((t + 1) * TK);
            stage_w(cur ^ G::BUF, (t + 1) * TK);
        }
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int s = kk * 2 + h;                      // 16 k per step: lanes 0-31 take k 0-7, lanes 32-63 k 8-15
            half8 whi[NI], wlo[NI], xhi[2], xlo[2];
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const int R = wn * 32 * NI + i * 32 + r32, o = cur + G::WO + R * SROWB + sslot(R, s);
                whi[i] = *reinterpret_cast<const half8*>(smem + o);
                wlo[i] = *reinterpret_cast<const half8*>(smem + o + G::WOPER);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int R = wm * 64 + j * 32 + r32, o = cur + G::XO + R * SROWB + sslot(R, s);
                xhi[j] = *reinterpret_cast<const half8*>(smem + o);
                xlo[j] = *reinterpret_cast<const half8*>(smem + o + SOPER);
            }
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[i], xhi[j], acc[i][j], 0, 0, 0);
                    crs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[i], xlo[j], crs[i][j], 0, 0, 0);
                    crs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[i], xhi[j], crs[i][j], 0, 0, 0);
                }
        }
        if (t + 1 < NT) put_x(cur ^ G::BUF);      // (the other buffer: everybody left it before the previous barrier)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (over && a.overflow) atomicOr(a.overflow, 1);
    // epilogue: 64 (n) x 64 (m) per wavefront and pass through the wavefront's staging area
#pragma unroll
    for (int half = 0; half < NI / 2; half++)
        store_tile<true>(a, smem, reinterpret_cast<const floatx16(&)[2][2]>(acc[2 * half]),
                         reinterpret_cast<const floatx16(&)[2][2]>(crs[2 * half]), act, wave, 0, wm, lane, m0,
                         n0 + wn * 32 * NI + half * 64);
}

__global__ __launch_bounds__(256, 2) void gemm_f32_split_kernel(Args a, int act) { gemm_f32_split_body<2>(a, act); }

// f32 [rows][cols] -> hi / lo f16 planes (the weights, once per layer)
__global__ void split_f32_kernel(const float* __restrict__ src, _Float16* __restrict__ hi, _Float16* __restrict__ lo, long long n,
                                 int* overflow) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = src[i];
    if (!(fabsf(a) < 65504.0f) && overflow) atomicOr(overflow, 1);
    const _Float16 hh = (_Float16)a;
    hi[i] = hh;
    lo[i] = (_Float16)((a - (float)hh) * 2048.0f);
}

}  // namespace g32
}  // namespace vlfm

using namespace vlfm;
using namespace vlfm::g32;

extern "C" int vlfm_split_f32_to_f16_pair(const float* d_src, void* d_hi, void* d_lo, long long count, int* d_overflow, void* stream) {
    if (count == 0) return VLFM_OK;
    if (!d_src || !d_hi || !d_lo || count < 0) return fail(VLFM_ERR_INVALID, "split_f32_to_f16_pair: bad argument");
    const long long blocks = (count + 255) / 256;
    VLFM_KLAUNCH(split_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_src, (_Float16*)d_hi, (_Float16*)d_lo,
                 count, d_overflow);
    return check_launch("split_f32_kernel");
}

extern "C" int vlfm_gemm_f32_nt(const float* d_x, const float* d_w, const float* d_bias, const float* d_residual, float* d_c, int m,
                                int n, int k, int activation, int precision, const void* d_w_hi, const void* d_w_lo,
                                int* d_overflow, void* stream) {
    if (m == 0 || n == 0) return VLFM_OK;
    if (!d_x || !d_w || !d_c || m < 0 || n < 0 || k <= 0 || (k % TK) != 0 || activation < 0 || activation > 2 || precision < 0 ||
        precision > 1)
        return fail(VLFM_ERR_INVALID, "gemm_f32_nt: K must be a multiple of 32, activation 0..2, precision 0..1");
    if (((uintptr_t)d_x | (uintptr_t)d_w | (uintptr_t)d_c | (uintptr_t)d_residual | (uintptr_t)d_w_hi | (uintptr_t)d_w_lo) & 15)
        return fail(VLFM_ERR_INVALID, "gemm_f32_nt: X, W, C, residual and the split weights must be 16-byte aligned");
    if (precision == 1 && (!d_w_hi || !d_w_lo || !d_overflow))
        return fail(VLFM_ERR_INVALID, "gemm_f32_nt: the split form needs the pre-split weights and the overflow flag");
    Args a;
    a.x = d_x; a.w = d_w; a.bias = d_bias; a.residual = d_residual; a.c = d_c;
    a.M = m; a.N = n; a.K = k;
    a.tiles_m = (m + TB - 1) / TB; a.tiles_n = (n + TB - 1) / TB;
    a.overflow = d_overflow; a.w_hi = (const _Float16*)d_w_hi; a.w_lo = (const _Float16*)d_w_lo;
    if ((long long)a.tiles_m * a.tiles_n > 0x7fffffffLL) return fail(VLFM_ERR_CAPACITY, "gemm_f32_nt: too many tiles");
    const dim3 grid(a.tiles_m * a.tiles_n), block(256);
    static LdsOptIn opt_exact, opt_split;
    if (precision == 0) {
        if (LDS_EXACT > 64 * 1024 && !opt_exact.ensure(reinterpret_cast<const void*>(gemm_f32_exact_kernel), LDS_EXACT))
            return fail(VLFM_ERR_HIP, "gemm_f32_nt: cannot opt in to the LDS size");
        VLFM_TIMED("gemm_f32_exact_kernel", stream);
        VLFM_KLAUNCH(gemm_f32_exact_kernel, grid, block, LDS_EXACT, (hipStream_t)stream, a, activation);
        return check_launch("gemm_f32_exact_kernel");
    }
    // (a 256-wide tile with one workgroup per CU -- gemm_f32_split_body<4>, 512 registers per wavefront -- was measured and is NOT
    // dispatched: with one wavefront per SIMD nothing covers the barrier, the X conversion and the fragment reads between the MFMA
    // groups: 145 / 177 / 141 TFLOP/s-equivalent on the encoder FFN / Swin stage-3 MLP / 256 -> 256 shapes against 186 / 223 / 180 for
    // two 128-wide workgroups per CU, tools/gemm_f32_probe.py, profiles/r04_gemm_f32_probe_wide_tile.txt)
    if (SplitGeom<2>::LDS > 64 * 1024 && !opt_split.ensure(reinterpret_cast<const void*>(gemm_f32_split_kernel), SplitGeom<2>::LDS))
        return fail(VLFM_ERR_HIP, "gemm_f32_nt: cannot opt in to the LDS size");
    VLFM_TIMED("gemm_f32_split_kernel", stream);
    VLFM_KLAUNCH(gemm_f32_split_kernel, grid, block, SplitGeom<2>::LDS, (hipStream_t)stream, a, activation);
    return check_launch("gemm_f32_split_kernel");
}
