// sam_ops.hip -- the memory-bound glue of TinyViT's windowed-attention blocks (the image encoder of MobileSAM behind
// vlfm/vlm/sam.py:54 `predictor.set_image`; mobile_sam/modeling/tiny_vit_sam.py TinyViTBlock [ext]) in f32 on NHWC rows.
//
// Profile of the framework path at 32 frames (tools/sam_probe.py + tools/rocprof_tail.py): LayerNorm 9 % of the
// encoder at 0.7 TB/s (one short row per block), residual / bias adds 10 %, window-partition / reverse / NCHW<->NHWC copies 9 %.  A
// block is: pad -> partition into windows -> LayerNorm -> attention -> reverse -> residual add -> depthwise 3x3 -> LayerNorm -> MLP ->
// residual add.  With the activation kept as [B, H, W, C] rows between the blocks of a stage the glue collapses into three kernels:
//   * layernorm_rows_f32_kernel: LayerNorm of a C-float row per wavefront, written either in place order (the MLP's norm) or straight
//     into the window order [B * nWy * nWx, ws * ws, C] (the attention's norm: pad + partition + norm in one pass; padded positions
//     get LayerNorm(0) = beta, which is what the reference computes for its zero padding);
//   * window_reverse_add_f32_kernel: x[b, y, x, :] += a[window(b, y, x), position, :] (reverse + crop + residual add in one pass);
//   * dwconv3x3_nhwc_f32_kernel: the block's depthwise 3x3 `local_conv` with its folded-BatchNorm bias on NHWC rows (16-byte
//     accesses along the channels, weights as [9][C]), so that the tensor never goes back to NCHW inside a stage.
// All three are HBM-bound by construction: 16 bytes per lane, consecutive lanes on consecutive addresses.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vlfm_amd.h"
#include "profile.h"
#include "status.h"

namespace vlfm {
namespace sam {

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wavefront per OUTPUT row.  window == 0: out row r = in row r.  window > 0: out rows are in window order over the padded
// image (Hp = ceil(H / ws) * ws): r = ((b * nWy + wy) * nWx + wx) * ws * ws + iy * ws + ix  <-  (b, wy * ws + iy, wx * ws + ix).
template <int MAXV>   // float4 per lane: C <= 256 * MAXV
__global__ __launch_bounds__(256) void layernorm_rows_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ out, int B,
                                                                 int H, int W, int C, int ws, float eps, long long out_rows,
                                                                 int shift, int pad_zero) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= out_rows) return;
    const int C4 = C >> 2;
    long long in_row = r;
    bool real = true;
    if (ws > 0) {
        const int nwx = (W + ws - 1) / ws, nwy = (H + ws - 1) / ws, ww = ws * ws;
        const long long win = r / ww;
        const int pos = (int)(r - win * ww), iy = pos / ws, ix = pos - iy * ws;
        const int wx = (int)(win % nwx), wy = (int)((win / nwx) % nwy), b = (int)(win / ((long long)nwx * nwy));
        int y = wy * ws + iy, xx = wx * ws + ix;
        if (shift) {   // Swin's cyclic shift of the PADDED image (torch.roll by -shift): window cell (y, xx) <- padded cell (y + shift, xx + shift)
            const int Hp = nwy * ws, Wp = nwx * ws;
            y = y + shift; if (y >= Hp) y -= Hp;
            xx = xx + shift; if (xx >= Wp) xx -= Wp;
        }
        real = y < H && xx < W;
        in_row = ((long long)b * H + y) * W + xx;
    }
    float4* orow = reinterpret_cast<float4*>(out + r * C);
    if (!real) {   // TinyViT pads BEFORE the norm: LayerNorm(0) = beta.  Swin pads AFTER the norm (pad_zero): 0.
        for (int i = lane; i < C4; i += 64)
            orow[i] = pad_zero ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<const float4*>(beta)[i];
        return;
    }
    const float4* irow = reinterpret_cast<const float4*>(x + in_row * C);
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int i = lane + 64 * k;
        v[k] = i < C4 ? irow[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; k++) {
        if (lane + 64 * k < C4) {
            const float a = v[k].x - mean, b2 = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
            q += (a * a + b2 * b2) + (c * c + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int i = lane + 64 * k;
        if (i < C4) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[i], bb = reinterpret_cast<const float4*>(beta)[i];
            float4 o;
            o.x = (v[k].x - mean) * rstd * g.x + bb.x;
            o.y = (v[k].y - mean) * rstd * g.y + bb.y;
            o.z = (v[k].z - mean) * rstd * g.z + bb.z;
            o.w = (v[k].w - mean) * rstd * g.w + bb.w;
            orow[i] = o;
        }
    }
}

// x[b, y, xx, :] += a[window row of (b, y, xx), :]   (one thread per float4)
__global__ __launch_bounds__(256) void window_reverse_add_f32_kernel(float* __restrict__ x, const float* __restrict__ a, int B, int H,
                                                                     int W, int C, int ws, long long n4, int shift) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int C4 = C >> 2;
    const long long row = i / C4;
    const int c4 = (int)(i - row * C4);
    const int xx = (int)(row % W), y = (int)((row / W) % H), b = (int)(row / ((long long)W * H));
    const int nwx = (W + ws - 1) / ws, nwy = (H + ws - 1) / ws;
    int ys = y, xs = xx;
    if (shift) {   // the windows hold the image rolled by -shift: padded cell (y, xx) sits at shifted cell (y - shift, xx - shift)
        ys = y - shift; if (ys < 0) ys += nwy * ws;
        xs = xx - shift; if (xs < 0) xs += nwx * ws;
    }
    const int wy = ys / ws, iy = ys - wy * ws, wx = xs / ws, ix = xs - wx * ws;
    const long long arow = (((long long)b * nwy + wy) * nwx + wx) * (ws * ws) + iy * ws + ix;
    float4 v = reinterpret_cast<float4*>(x)[i];
    const float4 t = reinterpret_cast<const float4*>(a)[arow * C4 + c4];
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    reinterpret_cast<float4*>(x)[i] = v;
}

// out[b, y, xx, c] = bias[c] + sum_{dy, dx} w9c[(dy * 3 + dx)][c] * x[b, y + dy - 1, xx + dx - 1, c]   (stride 1, zero padding 1)
__global__ __launch_bounds__(256) void dwconv3x3_nhwc_f32_kernel(const float* __restrict__ x, const float* __restrict__ w9c,
                                                                 const float* __restrict__ bias, float* __restrict__ out, int B, int H,
                                                                 int W, int C, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int C4 = C >> 2;
    const long long row = i / C4;
    const int c4 = (int)(i - row * C4);
    const int xx = (int)(row % W), y = (int)((row / W) % H);
    float4 acc = bias ? reinterpret_cast<const float4*>(bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
        const int yi = y + dy - 1;
        if ((unsigned)yi >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const int xi = xx + dx - 1;
            if ((unsigned)xi >= (unsigned)W) continue;
            const float4 v = reinterpret_cast<const float4*>(x)[(row + (long long)(dy - 1) * W + (dx - 1)) * C4 + c4];
            const float4 k = reinterpret_cast<const float4*>(w9c)[(dy * 3 + dx) * C4 + c4];
            acc.x += k.x * v.x; acc.y += k.y * v.y; acc.z += k.z * v.z; acc.w += k.w * v.w;
        }
    }
    reinterpret_cast<float4*>(out)[i] = acc;
}

// Window attention of TinyViT (head width 32, windows of N = ws^2 = 49 or 196 tokens, additive relative-position bias):
//   out[t][h*32 ..] = softmax_j(scale * q_t . k_j + bias[h][i][j]) . v_j      over the N tokens j of token t's window (i = t mod N)
// One workgroup per (window, head), one lane per query; q and the running output stay in registers; the softmax is the online form
// over chunks of four keys (one rescale per chunk); explicit fmaf (the library is built with -ffp-contract=off for the bit-exact map
// kernels).  Every lane needs the SAME key / value row at the same time, i.e. the row address is wave-uniform: the compiler fetches the
// rows with scalar loads (s_load_dwordx16 through the scalar cache) into SGPRs, which the FMAs take as their scalar operand -- no
// staging pass, no barrier, no LDS.  qkv rows are [heads][q | k | v][32] as TinyViT's qkv Linear emits them; bias_t is the bias
// TRANSPOSED, [heads][j][i], so that consecutive lanes read consecutive addresses.
// Measured per block at 32 frames (tools/sam_attn_probe.py; stage 1 / 2 / 3 = 11 552 x 4 x 49, 800 x 5 x 196, 3 200 x 10 x 49 windows x
// heads x tokens): the library flash kernel the framework dispatches 0.76 / 0.80 / 0.50 ms (its tiles are built for long sequences; plus
// q / k / v transposes); K / V staged in LDS and read as broadcasts 0.57 / 0.63 / 0.40 (64 ds_read_b128 per four keys and wavefront
// keep the LDS ~85 % busy beside 300 VALU instructions); this form 0.49 / 0.58 / 0.37; the same with channel pairs in v_pk_fma_f32
// 0.57 / 0.70 / 0.40 (the packed form cannot take the scalar operands directly).  What bounds it now is the f32 vector ALU.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void window_attention_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ bias_t,
                                                                      float* __restrict__ out, int N, int heads, float scale_log2e,
                                                                      const float* __restrict__ mask_t, int wins_per_image) {
    const int h = blockIdx.x % heads;
    const long long win = blockIdx.x / heads;
    const int i = threadIdx.x;
    const size_t row = (size_t)heads * 96;                      // floats per token in qkv
    const float* base = qkv + (size_t)win * N * row + (size_t)h * 96;
    if (i >= N) return;
    float q[32];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const float4 t = reinterpret_cast<const float4*>(base + (size_t)i * row)[c];
        q[4 * c] = t.x * scale_log2e; q[4 * c + 1] = t.y * scale_log2e; q[4 * c + 2] = t.z * scale_log2e; q[4 * c + 3] = t.w * scale_log2e;
    }
    const float LOG2E = 1.4426950408889634f;
    const float* bcol = bias_t + ((size_t)h * N) * N + i;       // bias[h][i][j] at bcol[j * N]
    // Swin's shifted windows: an additive mask per window POSITION in the image, [wins_per_image][j][i] (0 / -100), or null
    const float* mcol = mask_t ? mask_t + ((size_t)(win % wins_per_image) * N) * N + i : nullptr;
    float o[32];
#pragma unroll
    for (int c = 0; c < 32; c++) o[c] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < N; j0 += 4) {
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u;
            if (j < N) {
                const float* kj = base + (size_t)j * row + 32;       // wave-uniform address
                float acc = (mcol ? bcol[(size_t)j * N] + mcol[(size_t)j * N] : bcol[(size_t)j * N]) * LOG2E;
#pragma unroll
                for (int c = 0; c < 32; c++) acc = fmaf(q[c], kj[c], acc);
                sc[u] = acc;
            } else {
                sc[u] = -INFINITY;
            }
        }
        const float mn = fmaxf(fmaxf(m, fmaxf(sc[0], sc[1])), fmaxf(sc[2], sc[3]));
        const float alpha = exp2f(m - mn);                       // first chunk: exp2(-inf) = 0
        m = mn;
        l *= alpha;
#pragma unroll
        for (int c = 0; c < 32; c++) o[c] *= alpha;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u;
            if (j < N) {
                const float* vj = base + (size_t)j * row + 64;       // wave-uniform address
                const float pj = exp2f(sc[u] - mn);
                l += pj;
#pragma unroll
                for (int c = 0; c < 32; c++) o[c] = fmaf(pj, vj[c], o[c]);
            }
        }
    }
    const float inv = 1.0f / l;
    float4* orow = reinterpret_cast<float4*>(out + ((size_t)win * N + i) * ((size_t)heads * 32) + (size_t)h * 32);
#pragma unroll
    for (int c = 0; c < 8; c++) orow[c] = make_float4(o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv);
}


// The same attention on the matrix cores, in exact f32 (v_mfma_f32_32x32x2_f32: every product and every accumulation is an f32
// operation, as on the vector ALU -- the vector form above is bound by the f32 VALU at ~34 TFLOP/s on the 196-token windows).
// One workgroup per (window, head); K (rows padded to 33 floats: 32 lanes read one column of 32 rows) and V (row-major) of the
// window in LDS; a wavefront owns 32 queries at a time.  S^T = K Q^T with the keys as the M index: in the 32 x 32 accumulator
// layout a lane then holds, for ONE query (column lane & 31), the scores of 16 keys per 32-key tile -- the softmax is lane-local
// plus one exchange with lane ^ 32, and the probabilities are already the B operand of O^T = V^T P^T: accumulator register r of a
// lane in half g belongs to key 32 t + (r & 3) + 8 (r >> 2) + 4 g, so MFMA step r takes V[that key][channel lane & 31] as its A
// operand -- a row of V, contiguous over the lanes, no transposition anywhere.  The additive bias (and Swin's window mask) is the
// accumulators' initial value (-inf for the padding keys), so the scores leave the matrix pipe complete.  KT = key / query tiles.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KT>
__global__ __launch_bounds__(KT >= 4 ? 256 : 128) void window_attention_mfma_f32_kernel(
    const float* __restrict__ qkv, const float* __restrict__ bias_t, float* __restrict__ out, int N, int heads, float scale_log2e,
    const float* __restrict__ mask_t, int wins_per_image) {
    constexpr int NP = 32 * KT, WAVES = KT >= 4 ? 4 : 2, KS = 33;
    __shared__ float Ks[NP * KS];
    __shared__ __attribute__((aligned(16))) float Vs[NP * 32];
    const int h = blockIdx.x % heads;
    const long long win = blockIdx.x / heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, grp = lane >> 5;
    const size_t row = (size_t)heads * 96;                      // floats per token in qkv
    const float* base = qkv + (size_t)win * N * row + (size_t)h * 96;
    for (int idx = tid; idx < NP * 8; idx += 64 * WAVES) {
        const int j = idx >> 3, c4 = idx & 7;
        float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
        if (j < N) {
            kk = reinterpret_cast<const float4*>(base + (size_t)j * row + 32)[c4];
            vv = reinterpret_cast<const float4*>(base + (size_t)j * row + 64)[c4];
        }
        float* kd = Ks + j * KS + 4 * c4;
        kd[0] = kk.x; kd[1] = kk.y; kd[2] = kk.z; kd[3] = kk.w;
        reinterpret_cast<float4*>(Vs + j * 32)[c4] = vv;
    }
    __syncthreads();
    const float LOG2E = 1.4426950408889634f;
    const float* bh = bias_t + (size_t)h * N * N;                              // bias[h][i][j] at bh[j * N + i]
    const float* mh = mask_t ? mask_t + (size_t)(win % wins_per_image) * N * N : nullptr;
    for (int qt = wave; qt < KT; qt += WAVES) {
        const int i = 32 * qt + col;                                            // this lane's query
        const bool qlive = i < N;
        // Q^T fragments: step s contracts channels 2 s and 2 s + 1; half g of the wavefront holds channel 2 s + g
        float qf[16];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float4 t4 = qlive ? reinterpret_cast<const float4*>(base + (size_t)i * row)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            qf[2 * c] = (grp ? t4.y : t4.x) * scale_log2e;
            qf[2 * c + 1] = (grp ? t4.w : t4.z) * scale_log2e;
        }
        // one key tile at a time, online softmax across the tiles (the whole score row in registers -- 16 KT of them -- spilled)
        float m = -INFINITY, l = 0.f;
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] = 0.f;
        for (int t = 0; t < KT; t++) {
            // scores start from the bias (padding keys: -inf; padding queries: 0, never stored).  (Requesting the next tile's bias
            // one tile ahead was measured: with the loop unrolled the 7-tile form spills, rolled it needs 256 registers -- 1.05 to
            // 1.86 ms per block against 0.75.)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int j = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * grp;
                float b0 = 0.f;
                if (j >= N) b0 = -INFINITY;
                else if (qlive) b0 = (mh ? bh[(size_t)j * N + i] + mh[(size_t)j * N + i] : bh[(size_t)j * N + i]) * LOG2E;
                acc[r] = b0;
            }
            const float* kr = Ks + (32 * t + col) * KS + grp;
#pragma unroll
            for (int sidx = 0; sidx < 16; sidx++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * sidx], qf[sidx], acc, 0, 0, 0);
            float tm = acc[0];
#pragma unroll
            for (int r = 1; r < 16; r++) tm = fmaxf(tm, acc[r]);
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
            const float mn = fmaxf(m, tm);                    // finite from the first tile on (key 0 exists)
            const float alpha = __builtin_amdgcn_exp2f(m - mn);   // first tile: exp2(-inf) = 0 (v_exp_f32: arguments <= 0)
            m = mn;
            l *= alpha;
#pragma unroll
            for (int r = 0; r < 16; r++) o[r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; r++) { const float pj = __builtin_amdgcn_exp2f(acc[r] - mn); l += pj; acc[r] = pj; }
            const float* vr = Vs + (32 * t + 4 * grp) * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; r++) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[((r & 3) + 8 * (r >> 2)) * 32], acc[r], o, 0, 0, 0);
        }
        l += __shfl_xor(l, 32, 64);
        if (qlive) {
            const float inv = 1.0f / l;
            // o[r] = channel (r & 3) + 8 (r >> 2) + 4 g of query i: four consecutive channels per register quad
            float* orow = out + ((size_t)win * N + i) * ((size_t)heads * 32) + (size_t)h * 32 + 4 * grp;
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++)
                *reinterpret_cast<float4*>(orow + 8 * q4) =
                    make_float4(o[4 * q4] * inv, o[4 * q4 + 1] * inv, o[4 * q4 + 2] * inv, o[4 * q4 + 3] * inv);
        }
    }
}

}  // namespace sam
}  // namespace vlfm

using namespace vlfm;

// TinyViT window attention (head width 32): d_qkv [windows * tokens][heads][q | k | v][32] f32 (the qkv Linear's output), d_bias_t
// [heads][tokens][tokens] = the additive attention bias TRANSPOSED (bias_t[h][j][i] = bias[h][i][j]), d_out [windows * tokens][heads * 32].
// out = softmax(scale * q k^T + bias) v per (window, head); tokens <= 256.
extern "C" int vlfm_window_attention_f32(const float* d_qkv, const float* d_bias_t, float* d_out, long long windows, int tokens, int heads,
                                         float scale, void* stream) {
    return vlfm_window_attention_masked_f32(d_qkv, d_bias_t, nullptr, 1, d_out, windows, tokens, heads, scale, stream);
}

// The same with Swin's shifted-window mask (the detector backbone behind vlfm/vlm/grounding_dino.py:38-74): d_mask_t
// [windows_per_image][tokens][tokens] is added to the bias of window w's scores as d_mask_t[w % windows_per_image] (transposed like
// d_bias_t; the mask is symmetric); NULL = no mask.
extern "C" int vlfm_window_attention_masked_f32(const float* d_qkv, const float* d_bias_t, const float* d_mask_t, int windows_per_image,
                                                float* d_out, long long windows, int tokens, int heads, float scale, void* stream) {
    if (windows == 0) return VLFM_OK;
    if (!d_qkv || !d_bias_t || !d_out || windows < 0 || tokens <= 0 || tokens > 256 || heads <= 0 || windows * heads > 0x7fffffffLL ||
        windows_per_image <= 0)
        return fail(VLFM_ERR_INVALID, "window_attention_f32: 1 <= tokens <= 256, head width 32");
    const dim3 grid((unsigned)(windows * heads));
    const float sl = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    // the matrix-core form for the window sizes of TinyViT and Swin (7 x 7 and 14 x 14 tokens: 2 and 7 key tiles);
    // VLFM_WINDOW_ATTENTION=valu keeps the vector form (A/B, tests)
    static const bool use_mfma = [] { const char* e = getenv("VLFM_WINDOW_ATTENTION"); return !(e && e[0] == 'v'); }();
    const int kt = (tokens + 31) / 32;
    if (use_mfma && (kt == 2 || kt == 7)) {
        VLFM_TIMED("window_attention_mfma_f32_kernel", stream);
        if (kt == 2) VLFM_KLAUNCH((sam::window_attention_mfma_f32_kernel<2>), grid, dim3(128), 0, s, d_qkv, d_bias_t, d_out, tokens, heads, sl, d_mask_t, windows_per_image);
        else VLFM_KLAUNCH((sam::window_attention_mfma_f32_kernel<7>), grid, dim3(256), 0, s, d_qkv, d_bias_t, d_out, tokens, heads, sl, d_mask_t, windows_per_image);
        return check_launch("window_attention_mfma_f32_kernel");
    }
    VLFM_TIMED("window_attention_f32_kernel", stream);
    if (tokens <= 64) VLFM_KLAUNCH((sam::window_attention_f32_kernel<64>), grid, dim3(64), 0, s, d_qkv, d_bias_t, d_out, tokens, heads, sl, d_mask_t, windows_per_image);
    else if (tokens <= 128) VLFM_KLAUNCH((sam::window_attention_f32_kernel<128>), grid, dim3(128), 0, s, d_qkv, d_bias_t, d_out, tokens, heads, sl, d_mask_t, windows_per_image);
    else VLFM_KLAUNCH((sam::window_attention_f32_kernel<256>), grid, dim3(256), 0, s, d_qkv, d_bias_t, d_out, tokens, heads, sl, d_mask_t, windows_per_image);
    return check_launch("window_attention_f32_kernel");
}

// LayerNorm over the last dimension of [batch, height, width, channels] f32 rows (channels % 4 == 0, <= 1024).  window == 0: d_out has
// the same row order.  window > 0: d_out is [batch * ceil(H / window) * ceil(W / window), window * window, channels] -- the rows of
// the zero-padded image in window order (TinyViTBlock's pad + window partition + attn.norm); padded positions receive beta.
extern "C" int vlfm_layernorm_rows_f32(const float* d_x, const float* d_gamma, const float* d_beta, float* d_out, int batch, int height,
                                       int width, int channels, int window, float eps, void* stream) {
    return vlfm_layernorm_rows_shifted_f32(d_x, d_gamma, d_beta, d_out, batch, height, width, channels, window, eps, 0, 0, stream);
}

// The same for Swin's (shifted) windows: the window rows are those of the zero-padded image ROLLED by -shift along both axes
// (SwinLayer.cyclic_shift), 0 <= shift < window; pad_zero != 0: padded positions receive 0 (Swin pads after the norm) instead of beta.
extern "C" int vlfm_layernorm_rows_shifted_f32(const float* d_x, const float* d_gamma, const float* d_beta, float* d_out, int batch,
                                               int height, int width, int channels, int window, float eps, int shift, int pad_zero,
                                               void* stream) {
    if (batch == 0) return VLFM_OK;
    if (!d_x || !d_gamma || !d_beta || !d_out || batch < 0 || height <= 0 || width <= 0 || channels <= 0 || (channels & 3) ||
        channels > 1024 || window < 0 || shift < 0 || (shift > 0 && shift >= window))
        return fail(VLFM_ERR_INVALID, "layernorm_rows_f32: channels must be a multiple of 4 and <= 1024, 0 <= shift < window");
    long long rows = (long long)batch * height * width;
    if (window > 0) {
        const long long nwy = (height + window - 1) / window, nwx = (width + window - 1) / window;
        rows = (long long)batch * nwy * nwx * window * window;
    }
    const long long blocks = (rows + 3) / 4;
    if (blocks > 0x7fffffffLL) return fail(VLFM_ERR_CAPACITY, "layernorm_rows_f32: too many rows");
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
    VLFM_TIMED("layernorm_rows_f32_kernel", stream);
    const int c4 = channels / 4;
    if (c4 <= 64) VLFM_KLAUNCH((sam::layernorm_rows_f32_kernel<1>), grid, block, 0, s, d_x, d_gamma, d_beta, d_out, batch, height, width, channels, window, eps, rows, shift, pad_zero);
    else if (c4 <= 128) VLFM_KLAUNCH((sam::layernorm_rows_f32_kernel<2>), grid, block, 0, s, d_x, d_gamma, d_beta, d_out, batch, height, width, channels, window, eps, rows, shift, pad_zero);
    else VLFM_KLAUNCH((sam::layernorm_rows_f32_kernel<4>), grid, block, 0, s, d_x, d_gamma, d_beta, d_out, batch, height, width, channels, window, eps, rows, shift, pad_zero);
    return check_launch("layernorm_rows_f32_kernel");
}

// d_x[b, y, x, :] += d_windows[row of (b, y, x) in window order, :]: TinyViTBlock's window reverse + crop + residual add, in place.
extern "C" int vlfm_window_reverse_add_f32(float* d_x, const float* d_windows, int batch, int height, int width, int channels, int window,
                                           void* stream) {
    return vlfm_window_reverse_add_shifted_f32(d_x, d_windows, batch, height, width, channels, window, 0, stream);
}

// The same when the windows hold the image rolled by -shift (Swin): reverse + roll back + crop + residual add, in place.
extern "C" int vlfm_window_reverse_add_shifted_f32(float* d_x, const float* d_windows, int batch, int height, int width, int channels,
                                                   int window, int shift, void* stream) {
    if (batch == 0) return VLFM_OK;
    if (!d_x || !d_windows || batch < 0 || height <= 0 || width <= 0 || channels <= 0 || (channels & 3) || window <= 0 || shift < 0 ||
        shift >= window)
        return fail(VLFM_ERR_INVALID, "window_reverse_add_f32: channels must be a multiple of 4, window > 0, 0 <= shift < window");
    const long long n4 = (long long)batch * height * width * (channels / 4);
    const long long blocks = (n4 + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(VLFM_ERR_CAPACITY, "window_reverse_add_f32: tensor too large");
    VLFM_TIMED("window_reverse_add_f32_kernel", stream);
    VLFM_KLAUNCH(sam::window_reverse_add_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_x, d_windows, batch,
                 height, width, channels, window, n4, shift);
    return check_launch("window_reverse_add_f32_kernel");
}

// Depthwise 3x3 convolution (stride 1, zero padding 1) + bias on NHWC f32 rows; d_w9c is [9][channels] (tap-major).
extern "C" int vlfm_dwconv3x3_nhwc_f32(const float* d_x, const float* d_w9c, const float* d_bias, float* d_out, int batch, int height,
                                       int width, int channels, void* stream) {
    if (batch == 0) return VLFM_OK;
    if (!d_x || !d_w9c || !d_out || batch < 0 || height <= 0 || width <= 0 || channels <= 0 || (channels & 3))
        return fail(VLFM_ERR_INVALID, "dwconv3x3_nhwc_f32: channels must be a multiple of 4");
    const long long n4 = (long long)batch * height * width * (channels / 4);
    const long long blocks = (n4 + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(VLFM_ERR_CAPACITY, "dwconv3x3_nhwc_f32: tensor too large");
    VLFM_TIMED("dwconv3x3_nhwc_f32_kernel", stream);
    VLFM_KLAUNCH(sam::dwconv3x3_nhwc_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_x, d_w9c, d_bias, d_out,
                 batch, height, width, channels, n4);
    return check_launch("dwconv3x3_nhwc_f32_kernel");
}
