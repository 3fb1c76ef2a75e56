// vlm_ops.hip -- the small HIP kernels either side of the BLIP-2 forward pass (gfx950).
//
//   preprocess_rgb   u8 HWC camera frame -> normalised CHW network input.  Restates the LAVIS eval transform the
//                    reference applies per frame on the CPU (vlfm/vlm/blip2itm.py:48-49 -> PIL Resize(224, BICUBIC)
//                    -> ToTensor -> Normalize(CLIP mean/std)) with Pillow's exact 8-bit resampler: separable
//                    antialiased bicubic (a = -0.5), 22-bit fixed-point coefficients, horizontal pass then vertical
//                    pass with a u8 round trip in between (Pillow src/libImaging/Resample.c).  Bit-exact vs Pillow
//                    on the u8 intermediate; checked against the real PIL in tests/.
//   itc_head         the ITC head: vision_proj (768->256) of the 32 Q-Former queries, L2 normalise, dot with the
//                    cached normalised text feature, max over queries (LAVIS Blip2ITM.forward(match_head="itc") [ext],
//                    SURVEY.md 3.4) -- one workgroup per image, no host sync.
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdlib>
#include <vector>

#include "../../include/vlfm_amd.h"
#include "profile.h"
#include "status.h"

namespace vlfm {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Pillow: 8-bit pixels, 22 fractional bits

__device__ inline unsigned char clip8(int v) {
    v >>= PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// horizontal pass: [n][H][W][3] u8 -> [n][H][O][3] u8.  One workgroup per (row, image); the source row is staged in LDS.
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ src, int H, int W, int O,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk,
                                                         int ksize, unsigned char* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int y = blockIdx.x, n = blockIdx.y;
    const unsigned char* row = src + ((size_t)n * H + y) * W * 3;
    const int row_bytes = W * 3;
    // 16-byte staging when the row start is aligned (W*3 multiple of 16 for 640 and 1280), scalar otherwise
    if ((row_bytes & 15) == 0 && ((size_t)row & 15) == 0) {
        for (int i = threadIdx.x; i < row_bytes / 16; i += blockDim.x)
            reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(row)[i];
    } else {
        for (int i = threadIdx.x; i < row_bytes; i += blockDim.x) smem[i] = row[i];
    }
    __syncthreads();
    unsigned char* out = dst + ((size_t)n * H + y) * O * 3;
    for (int o = threadIdx.x; o < O * 3; o += blockDim.x) {
        const int xx = o / 3, c = o - xx * 3;
        const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const int* k = kk + xx * ksize;
        int ss = 1 << (PRECISION_BITS - 1);
        for (int x = 0; x < cnt; x++) ss += (int)smem[(xmin + x) * 3 + c] * k[x];
        out[o] = clip8(ss);
    }
}

template <typename OutT>
__device__ inline OutT cast_out(float v);
template <> __device__ inline float cast_out<float>(float v) { return v; }
template <> __device__ inline __half cast_out<__half>(float v) { return __float2half_rn(v); }
template <> __device__ inline __hip_bfloat16 cast_out<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

struct Norm3 { float mean[3]; float std[3]; };

// vertical pass + ToTensor + Normalize: [n][H][O][3] u8 -> OutT, either [n][3][O][O] (patch == 0) or, for a ViT whose
// patch embedding is a stride-P convolution, directly in im2col order [n][(O/P)^2][3*P*P] (patch == P): the embedding
// then is one plain GEMM with the conv weight viewed as [out][3*P*P], with no unfold pass over the image.
template <typename OutT>
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const unsigned char* __restrict__ src, int H, int O,
                                                              const int* __restrict__ bounds,
                                                              const int* __restrict__ kk, int ksize, Norm3 nrm,
                                                              OutT* __restrict__ dst, int patch) {
    const int yy = blockIdx.x, n = blockIdx.y;
    const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
    const int* k = kk + yy * ksize;
    const unsigned char* img = src + (size_t)n * H * O * 3;
    for (int o = threadIdx.x; o < 3 * O; o += blockDim.x) {
        const int c = o / O, xx = o - c * O;  // channel-major so that the CHW stores are coalesced
        int ss = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < cnt; y++) ss += (int)img[((size_t)(ymin + y) * O + xx) * 3 + c] * k[y];
        const float px = (float)clip8(ss);
        // ToTensor: u8 -> f32 / 255 ; Normalize: (x - mean) / std, all in f32
        const float v = __fdiv_rn(__fsub_rn(__fdiv_rn(px, 255.0f), nrm.mean[c]), nrm.std[c]);
        if (patch == 0) {
            dst[(((size_t)n * 3 + c) * O + yy) * O + xx] = cast_out<OutT>(v);
        } else {
            const int g = O / patch, py = yy / patch, px_ = xx / patch;
            const size_t cell = ((size_t)n * g * g + (size_t)py * g + px_) * (3 * patch * patch);
            dst[cell + (size_t)c * patch * patch + (yy - py * patch) * patch + (xx - px_ * patch)] = cast_out<OutT>(v);
        }
    }
}

// ---------------------------------------------------------------------------------------- fused resampler (BLIP-2 path)
// Both passes in ONE launch, the u8 intermediate never leaves the CU.  A workgroup owns a band of R output rows of one image:
//   * H phase: the input rows the band needs (r0 .. r1, about R * H / O + 2 * support of them) stream through a double-buffered
//     LDS row stage, PF_GROUPS rows per step (16-byte global loads issued one step ahead, one barrier per step).  Thread xx of a
//     256-thread group keeps the coefficients of output column xx in REGISTERS for the whole band and turns its window of the
//     staged row (ND unaligned dwords -> v_alignbyte -> compile-time byte extracts) into the three channel sums; the clipped
//     bytes go to a channel-PLANAR intermediate in LDS.
//   * V phase: a thread owns 4 consecutive columns of one channel (one aligned LDS dword per tap row, the coefficients of the
//     output row are wave-uniform scalars), ToTensor + Normalize is a 3 x 256-entry table built with the reference's two f32
//     divisions, and the 4 results leave as two 2-element stores (a patch row of the im2col layout holds an even number).
// Integer arithmetic identical to the two-launch kernels above (same products, same 32-bit sums), so still Pillow-exact.
// HBM: the frame is read once (+ the band overlap, ~15 %) and the network input written once: 1.2 MB per 640 x 480 frame
// against 2.8 MB for the two launches, which also issued one byte load per tap.
constexpr int PF_THREADS = 768;                 // 3 H groups of 256 = 4 V groups of 192
constexpr int PF_GROUPS = PF_THREADS / 256;
constexpr int PF_VGROUP = 192;
constexpr int PF_ROW_BYTES = 4096 + 128;        // one staged source row (W * 3 <= 4096) + slack for the window over-read

template <typename OutT> struct alignas(2 * sizeof(OutT)) Pair { OutT a, b; };

template <int KS, typename OutT>
__global__ __launch_bounds__(PF_THREADS) void preprocess_fused_kernel(const unsigned char* __restrict__ src, int H, int W, int O, int R,
                                                                      int TR, const int* __restrict__ hbounds,
                                                                      const int* __restrict__ hk, int hksize,
                                                                      const int* __restrict__ vbounds,
                                                                      const int* __restrict__ vk, int vksize, Norm3 nrm,
                                                                      OutT* __restrict__ dst, int patch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int OP = (O + 3) & ~3;                                   // columns of an intermediate row (whole dwords)
    float* lut = reinterpret_cast<float*>(smem);                   // [3][256]
    unsigned char* rowbuf = smem + 3 * 256 * 4;                    // [2][PF_GROUPS][PF_ROW_BYTES]
    unsigned char* tmp = rowbuf + 2 * PF_GROUPS * PF_ROW_BYTES;    // [3][TR][OP]
    const int tid = threadIdx.x, n = blockIdx.y;
    const int yy0 = blockIdx.x * R, yy1 = min(O, yy0 + R);
    const int r0 = vbounds[2 * yy0], r1 = vbounds[2 * (yy1 - 1)] + vbounds[2 * (yy1 - 1) + 1];
    if (r1 - r0 > TR) __builtin_trap();                            // the host sized TR from the same scale: cannot happen
    {   // ToTensor (u8 / 255) and Normalize ((x - mean) / std) exactly as the reference evaluates them, once per value
        const int c = tid >> 8, v = tid & 255;
        lut[tid] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v, 255.0f), nrm.mean[c]), nrm.std[c]);
    }
    // ---- H phase
    const int g = tid >> 8, xx = tid & 255;
    const bool live = xx < O;
    int k[KS];
    int d0 = 0, b0 = 0;
    if (live) {
        const int xmin = hbounds[2 * xx];
        d0 = (xmin * 3) >> 2;
        b0 = (xmin * 3) & 3;
    }
#pragma unroll
    for (int x = 0; x < KS; x++) k[x] = (live && x < hksize) ? hk[xx * hksize + x] : 0;   // zero beyond the window: over-read is harmless
    const int row_bytes = W * 3;
    const unsigned char* img = src + (size_t)n * H * row_bytes;
    const int steps = (r1 - r0 + PF_GROUPS - 1) / PF_GROUPS;
    const bool loader = xx * 16 < row_bytes;
    uint4 pre = make_uint4(0u, 0u, 0u, 0u);
    if (loader && r0 + g < r1) pre = *reinterpret_cast<const uint4*>(img + (size_t)(r0 + g) * row_bytes + xx * 16);
    if (loader) *reinterpret_cast<uint4*>(rowbuf + (size_t)g * PF_ROW_BYTES + xx * 16) = pre;
    for (int st = 0; st < steps; st++) {
        const int r = r0 + st * PF_GROUPS + g, rn = r + PF_GROUPS;
        if (loader && rn < r1) pre = *reinterpret_cast<const uint4*>(img + (size_t)rn * row_bytes + xx * 16);
        __syncthreads();   // stage (st & 1) is complete, and every reader of the other stage (step st - 1) is done with it
        if (live && r < r1) {
            const unsigned* rowdw = reinterpret_cast<const unsigned*>(rowbuf + (size_t)((st & 1) * PF_GROUPS + g) * PF_ROW_BYTES);
            constexpr int NA = (3 * KS + 3) / 4;   // aligned dwords holding the window's 3 * KS bytes
            unsigned w[NA + 1], a[NA];
#pragma unroll
            for (int i = 0; i <= NA; i++) w[i] = rowdw[d0 + i];
#pragma unroll
            for (int i = 0; i < NA; i++) a[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], (unsigned)b0);
            int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll
            for (int x = 0; x < KS; x++) {
                const int j = 3 * x;
                // v_mad_i32_i24: a pixel is 8 bits, a coefficient at most 1.0 in 22 fractional bits (a full 32-bit multiply is quarter rate)
                s0 = __mul24((int)((a[j >> 2] >> (8 * (j & 3))) & 0xFFu), k[x]) + s0;
                s1 = __mul24((int)((a[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xFFu), k[x]) + s1;
                s2 = __mul24((int)((a[(j + 2) >> 2] >> (8 * ((j + 2) & 3))) & 0xFFu), k[x]) + s2;
            }
            unsigned char* t = tmp + (size_t)(r - r0) * OP + xx;
            t[0] = clip8(s0);
            t[(size_t)TR * OP] = clip8(s1);
            t[(size_t)2 * TR * OP] = clip8(s2);
        }
        if (loader) *reinterpret_cast<uint4*>(rowbuf + (size_t)(((st + 1) & 1) * PF_GROUPS + g) * PF_ROW_BYTES + xx * 16) = pre;
    }
    __syncthreads();
    // ---- V phase: thread vt of a 192-thread group owns (channel c, columns 4 dc .. 4 dc + 3) for every row of its group
    const int vg = tid / PF_VGROUP, vt = tid - vg * PF_VGROUP;
    const int opd = OP >> 2;                                        // 3 * opd <= PF_VGROUP (O <= 256, checked by the host)
    const bool vlive = vt < 3 * opd;
    const bool pair_ok = (O & 1) == 0 && (patch & 1) == 0;
    const int c = vlive ? vt / opd : 0, dc = vt - c * opd, x4 = 4 * dc;
    const int cell = 3 * patch * patch, gp = patch ? O / patch : 0;
    int coloff[4];                                                  // where column x4 + i sits inside an output row
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = x4 + i;
        coloff[i] = patch ? (x / patch) * cell + (x % patch) : x;
    }
    const int chan_off = c * (patch ? patch * patch : O * O);
    const unsigned* col0 = reinterpret_cast<const unsigned*>(tmp) + (size_t)c * TR * opd + dc;
    const float* lut_c = lut + c * 256;
    for (int yy = yy0 + vg; yy < yy1; yy += PF_THREADS / PF_VGROUP) {
        const int yu = __builtin_amdgcn_readfirstlane(yy);          // uniform in the wavefront (192 = 3 wavefronts)
        const int ymin = vbounds[2 * yu], cnt = vbounds[2 * yu + 1];
        const int* kv = vk + (size_t)yu * vksize;
        size_t row_off;                                             // scalar: start of this output row (channel 0, column 0)
        if (patch == 0) {
            row_off = ((size_t)n * 3 * O + yu) * O;
        } else {
            const int py = yu / patch;
            row_off = ((size_t)n * gp * gp + (size_t)py * gp) * cell + (size_t)(yu - py * patch) * patch;
        }
        if (!vlive) continue;
        const unsigned* col = col0 + (size_t)(ymin - r0) * opd;
        int s[4];
#pragma unroll
        for (int i = 0; i < 4; i++) s[i] = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < cnt; y++) {
            const unsigned wv = col[(size_t)y * opd];
            const int kq = kv[y];
            s[0] = __mul24((int)(wv & 0xFFu), kq) + s[0];
            s[1] = __mul24((int)((wv >> 8) & 0xFFu), kq) + s[1];
            s[2] = __mul24((int)((wv >> 16) & 0xFFu), kq) + s[2];
            s[3] = __mul24((int)(wv >> 24), kq) + s[3];
        }
        OutT v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = cast_out<OutT>(lut_c[clip8(s[i])]);
        OutT* o = dst + row_off + chan_off;
        if (pair_ok && x4 + 3 < O) {
            *reinterpret_cast<Pair<OutT>*>(o + coloff[0]) = Pair<OutT>{v[0], v[1]};
            *reinterpret_cast<Pair<OutT>*>(o + coloff[2]) = Pair<OutT>{v[2], v[3]};
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (x4 + i < O) o[coloff[i]] = v[i];
        }
    }
}

// SamPredictor.set_image [ext mobile_sam]: vertical pass of the PIL BILINEAR resize to (OH, OW) (longest side 1024), then
// Sam.preprocess: (x - pixel_mean) / pixel_std on the 0..255 scale and zero padding to P x P.  [n][H][OW][3] u8 -> [n][3][P][P]
__global__ __launch_bounds__(256) void resample_v_sam_kernel(const unsigned char* __restrict__ src, int H, int OH, int OW,
                                                             int P, const int* __restrict__ bounds,
                                                             const int* __restrict__ kk, int ksize, Norm3 nrm,
                                                             float* __restrict__ dst) {
    const int yy = blockIdx.x, n = blockIdx.y;
    const unsigned char* img = src + (size_t)n * H * OW * 3;
    const bool row_in = yy < OH;
    const int ymin = row_in ? bounds[2 * yy] : 0, cnt = row_in ? bounds[2 * yy + 1] : 0;
    const int* k = kk + (size_t)(row_in ? yy : 0) * ksize;
    for (int o = threadIdx.x; o < 3 * P; o += blockDim.x) {
        const int c = o / P, xx = o - c * P;
        float v = 0.0f;  // padding is applied AFTER normalisation: zeros
        if (row_in && xx < OW) {
            int ss = 1 << (PRECISION_BITS - 1);
            for (int y = 0; y < cnt; y++) ss += (int)img[((size_t)(ymin + y) * OW + xx) * 3 + c] * k[y];
            v = __fdiv_rn(__fsub_rn((float)clip8(ss), nrm.mean[c]), nrm.std[c]);
        }
        dst[(((size_t)n * 3 + c) * P + yy) * P + xx] = v;
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm(x + c)
// y[r][:] = LayerNorm(x[r][:] + c[:]) * gamma + beta for f16 rows, statistics in f32 (two passes over registers).
// ``c`` (f32, may be null) is a per-channel constant: the ViT's projection / fc2 biases are not added to the residual
// stream by separate elementwise kernels -- the GEMMs accumulate into the stream (beta = 1) and the biases, which are
// constants of the network, are summed once per layer and enter here.  HBM-bound: 2 B in + 2 B out per element.
// One wavefront per row, LN_ROWS consecutive rows per wavefront so that c / gamma / beta stay in registers; the next
// row's loads are issued before the current row is reduced.
constexpr int LN_MAXV = 4;    // 16-byte vectors per lane -> D <= 64 * 8 * 4 = 2048
constexpr int LN_ROWS = 8;    // upper bound of rows per wavefront (fewer when the batch is small)

struct alignas(16) Half8 { __half2 h[4]; };

template <int NV>
__global__ __launch_bounds__(256) void layernorm_bias_f16_kernel(const __half* __restrict__ x, const float* __restrict__ cb,
                                                                 const __half* __restrict__ gamma,
                                                                 const __half* __restrict__ beta, __half* __restrict__ y,
                                                                 int rows, int D, float eps, int rows_per_wave) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nvec = D >> 3;
    const int r0 = wave * rows_per_wave;
    if (r0 >= rows) return;
    // per-channel constants of this lane's columns: 16-byte loads, kept packed (gamma / beta as half2) in registers
    Half8 gm[NV], bt[NV];
    float4 c0[NV], c1[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int v = lane + 64 * i;
        if (v < nvec) {
            gm[i] = reinterpret_cast<const Half8*>(gamma)[v];
            bt[i] = reinterpret_cast<const Half8*>(beta)[v];
            c0[i] = cb ? reinterpret_cast<const float4*>(cb)[2 * v] : make_float4(0.f, 0.f, 0.f, 0.f);
            c1[i] = cb ? reinterpret_cast<const float4*>(cb)[2 * v + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    Half8 cur[NV], nxt[NV], nx2[NV];
    auto load_row = [&](int r, Half8* dst) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int v = lane + 64 * i;
            if (v < nvec) dst[i] = reinterpret_cast<const Half8*>(x + (size_t)r * D)[v];
        }
    };
    // element j (0..7) of vector i as f32, with the channel constant added; 0 for lanes beyond the row
    auto val = [&](const Half8* row, int i, int j) -> float {
        const float2 t = __half22float2(row[i].h[j >> 1]);
        const float cc = j < 4 ? (&c0[i].x)[j] : (&c1[i].x)[j - 4];
        return ((j & 1) ? t.y : t.x) + cc;
    };
    const int r_end = min(rows, r0 + rows_per_wave);
    load_row(r0, cur);
    if (r0 + 1 < r_end) load_row(r0 + 1, nxt);
    const float inv_d = 1.0f / (float)D;
    for (int r = r0; r < r_end; r++) {
        if (r + 2 < r_end) load_row(r + 2, nx2);   // two rows in flight behind the one being reduced
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            if (lane + 64 * i < nvec) {
#pragma unroll
                for (int j = 0; j < 8; j++) sum += val(cur, i, j);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float mean = sum * inv_d;
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            if (lane + 64 * i < nvec) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float dlt = val(cur, i, j) - mean;
                    sq = fmaf(dlt, dlt, sq);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
        const float rstd = rsqrtf(sq * inv_d + eps);
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int v = lane + 64 * i;
            if (v < nvec) {
                Half8 o;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float2 g2 = __half22float2(gm[i].h[j]), b2 = __half22float2(bt[i].h[j]);
                    o.h[j] = __floats2half2_rn((val(cur, i, 2 * j) - mean) * rstd * g2.x + b2.x,
                                               (val(cur, i, 2 * j + 1) - mean) * rstd * g2.y + b2.y);
                }
                reinterpret_cast<Half8*>(y + (size_t)r * D)[v] = o;
            }
        }
#pragma unroll
        for (int i = 0; i < NV; i++) { cur[i] = nxt[i]; nxt[i] = nx2[i]; }
    }
}

// ------------------------------------------------------------------------------------------------ ITC head
// One workgroup per image, one wavefront per query (round-robin): cos_q = <p_q, t> / max(|p_q|, 1e-12); out = max_q.
__global__ __launch_bounds__(512) void itc_head_kernel(const float* __restrict__ proj, int NQ, int P,
                                                       const float* __restrict__ text, float* __restrict__ out) {
    __shared__ float cos_q[64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const float* t = text + (size_t)b * P;
    for (int q = wave; q < NQ; q += n_waves) {
        const float* p = proj + ((size_t)b * NQ + q) * P;
        float n2 = 0.0f, dt = 0.0f;
        for (int j = lane; j < P; j += 64) {
            const float v = p[j];
            n2 = fmaf(v, v, n2);
            dt = fmaf(v, t[j], dt);
        }
        for (int off = 32; off > 0; off >>= 1) {
            n2 += __shfl_down(n2, off, 64);
            dt += __shfl_down(dt, off, 64);
        }
        if (lane == 0) cos_q[q] = dt / fmaxf(sqrtf(n2), 1e-12f);  // F.normalize eps
    }
    __syncthreads();
    if (wave == 0) {
        float c = lane < NQ ? cos_q[lane] : -__builtin_huge_valf();
        for (int off = 32; off > 0; off >>= 1) c = fmaxf(c, __shfl_down(c, off, 64));
        if (lane == 0) out[b] = c;
    }
}

}  // namespace vlfm

using namespace vlfm;

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter (support 2, a = -0.5), box = whole axis.
extern "C" int vlfm_resample_coeffs_filter_host(int in_size, int out_size, int filter, int32_t* h_bounds, int32_t* h_kk,
                                                int kk_capacity, int* ksize_out) {
    if (in_size <= 0 || out_size <= 0 || !h_bounds || !h_kk || !ksize_out || (filter != 0 && filter != 1))
        return fail(VLFM_ERR_INVALID, "resample_coeffs_host: bad argument (filter 0 = bicubic, 1 = bilinear)");
    auto bicubic = [filter](double x) {
        if (x < 0.0) x = -x;
        if (filter == 1) return x < 1.0 ? 1.0 - x : 0.0;  // Pillow bilinear_filter (support 1)
        const double a = -0.5;
        if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
        if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
        return 0.0;
    };
    const float in0 = 0.0f, in1 = (float)in_size;
    double scale = (double)(in1 - in0) / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = (filter == 1 ? 1.0 : 2.0) * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    *ksize_out = ksize;
    if ((long)out_size * ksize > kk_capacity) return fail(VLFM_ERR_CAPACITY, "resample_coeffs_host: kk capacity");
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; xx++) {
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; x++) {
            const double w = bicubic((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; x++)
            if (ww != 0.0) k[x] /= ww;
        for (int x = xmax; x < ksize; x++) k[x] = 0;
        h_bounds[2 * xx] = xmin;
        h_bounds[2 * xx + 1] = xmax;
        for (int x = 0; x < ksize; x++) {
            const double v = k[x];
            h_kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
    }
    return VLFM_OK;
}

extern "C" int vlfm_resample_coeffs_host(int in_size, int out_size, int32_t* h_bounds, int32_t* h_kk, int kk_capacity,
                                         int* ksize_out) {
    return vlfm_resample_coeffs_filter_host(in_size, out_size, 0, h_bounds, h_kk, kk_capacity, ksize_out);
}

extern "C" int vlfm_preprocess_sam_batched(const uint8_t* d_rgb, int n, int height, int width, int out_h, int out_w,
                                           const int32_t* d_hbounds, const int32_t* d_hk, int hksize,
                                           const int32_t* d_vbounds, const int32_t* d_vk, int vksize,
                                           const float* h_mean3, const float* h_std3, int pad_size, uint8_t* d_tmp,
                                           float* d_out, void* stream) {
    if (n == 0) return VLFM_OK;
    if (!d_rgb || !d_hbounds || !d_hk || !d_vbounds || !d_vk || !h_mean3 || !h_std3 || !d_tmp || !d_out || n < 0 ||
        height <= 0 || width <= 0 || out_h <= 0 || out_w <= 0 || pad_size < out_h || pad_size < out_w)
        return fail(VLFM_ERR_INVALID, "preprocess_sam_batched: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = ((size_t)width * 3 + 15) / 16 * 16;
    {
        VLFM_TIMED("resample_h_kernel", s);
        VLFM_KLAUNCH(resample_h_kernel, dim3(height, n), dim3(256), lds, s, d_rgb, height, width, out_w, d_hbounds, d_hk,
                     hksize, d_tmp);
    }
    int rc = check_launch("resample_h_kernel");
    if (rc != VLFM_OK) return rc;
    Norm3 nrm;
    for (int c = 0; c < 3; c++) { nrm.mean[c] = h_mean3[c]; nrm.std[c] = h_std3[c]; }
    VLFM_TIMED("resample_v_sam_kernel", s);
    VLFM_KLAUNCH(resample_v_sam_kernel, dim3(pad_size, n), dim3(256), 0, s, d_tmp, height, out_h, out_w, pad_size,
                 d_vbounds, d_vk, vksize, nrm, d_out);
    return check_launch("resample_v_sam_kernel");
}

extern "C" int vlfm_preprocess_rgb_batched(const uint8_t* d_rgb, int n, int height, int width, int out_size,
                                           const int32_t* d_hbounds, const int32_t* d_hk, int hksize,
                                           const int32_t* d_vbounds, const int32_t* d_vk, int vksize,
                                           const float* h_mean3, const float* h_std3, uint8_t* d_tmp, void* d_out,
                                           int out_dtype, int patch_size, void* stream) {
    if (n == 0) return VLFM_OK;
    if (!d_rgb || !d_hbounds || !d_hk || !d_vbounds || !d_vk || !h_mean3 || !h_std3 || !d_tmp || !d_out || n < 0 ||
        height <= 0 || width <= 0 || out_size <= 0 || patch_size < 0 || (patch_size > 0 && out_size % patch_size != 0))
        return fail(VLFM_ERR_INVALID, "preprocess_rgb_batched: bad argument");
    hipStream_t s = (hipStream_t)stream;
    Norm3 nrm;
    for (int c = 0; c < 3; c++) { nrm.mean[c] = h_mean3[c]; nrm.std[c] = h_std3[c]; }
    if (out_dtype < 0 || out_dtype > 2)
        return fail(VLFM_ERR_INVALID, "preprocess_rgb_batched: out_dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    // one launch (preprocess_fused_kernel) whenever a source row can be staged with 16-byte loads; the two-launch form
    // (u8 intermediate through d_tmp) serves every other geometry
    const int row_bytes = width * 3;
    if (row_bytes % 16 == 0 && row_bytes <= 4096 && ((uintptr_t)d_rgb & 15) == 0 && hksize <= 32 && out_size <= 256 &&
        !getenv("VLFM_PREPROCESS_TWO_PASS")) {
        const int OP = (out_size + 3) & ~3;
        auto span = [&](int R) { return (int)(((long)(R - 1) * height + out_size - 1) / out_size) + vksize + 2; };
        auto lds_of = [&](int R) { return (size_t)3 * 256 * 4 + (size_t)2 * PF_GROUPS * PF_ROW_BYTES + (size_t)3 * span(R) * OP; };
        int R = out_size < 32 ? out_size : 32;
        while (R > 1 && lds_of(R) > 80 * 1024) R--;            // two workgroups per CU
        while (R > 8 && (long)((out_size + R - 1) / R) * n < 512) R = (R + 1) / 2;   // few frames: shorter bands fill the CUs
        const int bands = (out_size + R - 1) / R;
        R = (out_size + bands - 1) / bands;                       // equal bands
        const size_t lds = lds_of(R);
        const int TR = span(R);
        const dim3 grid(bands, n), block(PF_THREADS);
        VLFM_TIMED("preprocess_fused_kernel", s);
#define VLFM_PF_LAUNCH(KS, T)                                                                                          \
    do {                                                                                                               \
        static LdsOptIn opt;                                                                                           \
        if (!opt.ensure(reinterpret_cast<const void*>(preprocess_fused_kernel<KS, T>), 80 * 1024))                     \
            return fail(VLFM_ERR_HIP, "preprocess_rgb_batched: cannot opt in to 80 KB of LDS");                        \
        VLFM_KLAUNCH((preprocess_fused_kernel<KS, T>), grid, block, lds, s, d_rgb, height, width, out_size, R, TR,     \
                     d_hbounds, d_hk, hksize, d_vbounds, d_vk, vksize, nrm, (T*)d_out, patch_size);                    \
    } while (0)
#define VLFM_PF_DTYPE(KS)                                                                                              \
    do {                                                                                                               \
        if (out_dtype == 0) VLFM_PF_LAUNCH(KS, float);                                                                 \
        else if (out_dtype == 1) VLFM_PF_LAUNCH(KS, __half);                                                           \
        else VLFM_PF_LAUNCH(KS, __hip_bfloat16);                                                                       \
    } while (0)
        if (hksize <= 8) VLFM_PF_DTYPE(8);
        else if (hksize <= 13) VLFM_PF_DTYPE(13);
        else if (hksize <= 16) VLFM_PF_DTYPE(16);
        else if (hksize <= 25) VLFM_PF_DTYPE(25);
        else VLFM_PF_DTYPE(32);
#undef VLFM_PF_DTYPE
#undef VLFM_PF_LAUNCH
        return check_launch("preprocess_fused_kernel");
    }
    const size_t lds = ((size_t)width * 3 + 15) / 16 * 16;
    {
        VLFM_TIMED("resample_h_kernel", s);
    VLFM_KLAUNCH(resample_h_kernel, dim3(height, n), dim3(256), lds, s, d_rgb, height, width, out_size,
                       d_hbounds, d_hk, hksize, d_tmp);
    }
    int rc = check_launch("resample_h_kernel");
    if (rc != VLFM_OK) return rc;
    VLFM_TIMED("resample_v_norm_kernel", s);
    if (out_dtype == 0)
        VLFM_KLAUNCH(resample_v_norm_kernel<float>, dim3(out_size, n), dim3(256), 0, s, d_tmp, height, out_size,
                           d_vbounds, d_vk, vksize, nrm, (float*)d_out, patch_size);
    else if (out_dtype == 1)
        VLFM_KLAUNCH(resample_v_norm_kernel<__half>, dim3(out_size, n), dim3(256), 0, s, d_tmp, height,
                           out_size, d_vbounds, d_vk, vksize, nrm, (__half*)d_out, patch_size);
    else
        VLFM_KLAUNCH(resample_v_norm_kernel<__hip_bfloat16>, dim3(out_size, n), dim3(256), 0, s, d_tmp, height,
                           out_size, d_vbounds, d_vk, vksize, nrm, (__hip_bfloat16*)d_out, patch_size);
    return check_launch("resample_v_norm_kernel");
}

extern "C" int vlfm_itc_head_batched(const float* d_proj, int batch, int n_query, int proj_dim, const float* d_text,
                                     float* d_out, void* stream) {
    if (batch == 0) return VLFM_OK;
    if (!d_proj || !d_text || !d_out || batch < 0 || n_query <= 0 || n_query > 64 || proj_dim <= 0)
        return fail(VLFM_ERR_INVALID, "itc_head_batched: bad argument (n_query <= 64)");
    VLFM_TIMED("itc_head_kernel", stream);
    VLFM_KLAUNCH(itc_head_kernel, dim3(batch), dim3(512), 0, (hipStream_t)stream, d_proj, n_query, proj_dim,
                       d_text, d_out);
    return check_launch("itc_head_kernel");
}

extern "C" int vlfm_layernorm_bias_f16(const void* d_x, const float* d_channel_bias, const void* d_gamma, const void* d_beta,
                                       void* d_y, int rows, int dim, float eps, void* stream) {
    if (rows == 0) return VLFM_OK;
    if (!d_x || !d_gamma || !d_beta || !d_y || rows < 0 || dim <= 0 || dim % 8 != 0 || dim > 64 * 8 * LN_MAXV)
        return fail(VLFM_ERR_INVALID, "layernorm_bias_f16: dim must be a multiple of 8 and <= 2048");
    const int nvec = dim / 8, nv = (nvec + 63) / 64;
    // enough wavefronts to fill 256 CUs x 4 SIMDs a few times over before rows are batched per wavefront
    int rpw = rows / 8192;
    rpw = rpw < 1 ? 1 : rpw > LN_ROWS ? LN_ROWS : rpw;
    const int waves = (rows + rpw - 1) / rpw;
    const dim3 grid((waves + 3) / 4), block(256);
    const __half* x = (const __half*)d_x; const __half* g = (const __half*)d_gamma; const __half* b = (const __half*)d_beta;
    __half* y = (__half*)d_y;
    VLFM_TIMED("layernorm_bias_f16_kernel", stream);
    switch (nv) {
        case 1: VLFM_KLAUNCH(layernorm_bias_f16_kernel<1>, grid, block, 0, (hipStream_t)stream, x, d_channel_bias, g, b, y, rows, dim, eps, rpw); break;
        case 2: VLFM_KLAUNCH(layernorm_bias_f16_kernel<2>, grid, block, 0, (hipStream_t)stream, x, d_channel_bias, g, b, y, rows, dim, eps, rpw); break;
        case 3: VLFM_KLAUNCH(layernorm_bias_f16_kernel<3>, grid, block, 0, (hipStream_t)stream, x, d_channel_bias, g, b, y, rows, dim, eps, rpw); break;
        default: VLFM_KLAUNCH(layernorm_bias_f16_kernel<4>, grid, block, 0, (hipStream_t)stream, x, d_channel_bias, g, b, y, rows, dim, eps, rpw); break;
    }
    return check_launch("layernorm_bias_f16_kernel");
}
