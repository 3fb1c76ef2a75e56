// vit_attention.hip -- self-attention of the BLIP-2 ViT-g blocks (S = 257 tokens, 16 heads of 88), gfx950.
//
// Replaces F.scaled_dot_product_attention in vlfm_amd/vlm/blip2itm.py:_VitBlock (the reference reaches the same maths
// through LAVIS' eva_vit Attention, vlfm/vlm/blip2itm.py:29-34,52 [ext]).  The library flash kernel spends 290 us per
// block at 128 images on this shape (257 = 8 x 32 + 1 tokens, head 88); this kernel is specialised for it:
//
//   * one workgroup per (image, head), 8 wavefronts; the head's whole K (257 x 96) and V^T (96 x 257) live in LDS
//     (117 KB of the 160 KB), so nothing is re-read and there is no online-softmax rescaling;
//   * wavefront w owns the 32 queries of tokens 1+32w .. 32+32w.  It computes S^T = K Q^T with
//     v_mfma_f32_32x32x16_f16 (A = K rows from LDS, B = Q^T held in registers): in the 32x32 accumulator layout every
//     lane then holds 16 keys of ONE query column per key tile, so the softmax statistics are lane-local apart from one
//     exchange with lane^32, and the probabilities are ALREADY in B-operand order for O^T = V^T P^T -- no cross-lane
//     movement between the two GEMMs (the pairing of accumulator registers with key indices is mirrored in the V^T loads);
//   * the odd token (the CLS query) is a ninth query tile with one live column: its keys are split over the 8
//     wavefronts (12 MFMAs each), partial (max, sum, O) are merged through LDS by wavefront 0;
//   * the MFMA-friendly head width 96 exists only in LDS / registers (channels 88..95 are zeros written while staging);
//     global memory holds the native 88-wide rows, so the qkv and projection GEMMs keep their sizes (DH = 96 serves
//     weights that were zero-padded instead);
//   * output goes straight to the [B, S, H, DH] layout the projection GEMM consumes (the library path needs a
//     transpose copy);
//   * LDS operand fetches run one contraction step ahead of the MFMAs; key tiles are processed in groups of three so
//     that consecutive MFMAs write different accumulators.
//
// 4 x 257^2 x 88 flop per (image, head); HBM traffic = qkv read once + output written once (404 MB at 128 images), which
// is what bounds it: one workgroup per CU (LDS), so staging cannot overlap another workgroup's MFMA phase.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/vlfm_amd.h"
#include "profile.h"
#include "status.h"

namespace vlfm {

using half4_t = __attribute__((ext_vector_type(4))) _Float16;
using half8_t = __attribute__((ext_vector_type(8))) _Float16;
using f32x16_t = __attribute__((ext_vector_type(16))) float;

constexpr int AT_D = 96;                  // padded head width (88 real)
constexpr int AT_NT = 8;                  // full 32-token tiles
constexpr int AT_S = 32 * AT_NT + 1;      // 257 tokens
constexpr int AT_KT = AT_NT + 1;          // key tiles incl. the one-token tail
// LDS row strides (halfs).  Both operands are fetched with ds_read_b128, which the LDS serves in four fixed 16-lane
// groups; the lanes of a group read 16 different rows at the same column, so a row stride of an ODD number of 16-byte
// slots (13 and 37) puts them on 16 different slots: conflict-free.
constexpr int AT_KS = AT_D + 8;           // 208 B = 13 slots
constexpr int AT_VS = 32 * AT_KT + 8;     // 592 B = 37 slots
constexpr int AT_KROWS = 32 * AT_KT;      // 288 (rows >= 257 are zero)
constexpr int AT_WAVES = 8;
constexpr size_t AT_LDS_K = (size_t)AT_KROWS * AT_KS * 2;        // 59 904 B
constexpr size_t AT_LDS_V = (size_t)AT_D * AT_VS * 2;            // 56 832 B
constexpr size_t AT_LDS_PART = (size_t)AT_WAVES * (AT_D + 2) * 4;  // partials of the CLS query
constexpr size_t AT_LDS_QCLS = (size_t)AT_D * 2;                   // the CLS query row
constexpr size_t AT_LDS_BYTES = AT_LDS_K + AT_LDS_V + AT_LDS_PART + AT_LDS_QCLS;

// Diagnostic builds only (tools/vit_attn_stub_probe.py): -DVLFM_ATT_STUB=1 memory side only (loads, staging, stores of zeros; no
// attention), 2 no global loads (constants are staged instead), 3 no CLS share, 4 no stores.  Undefined (0) in the product.
#ifndef VLFM_ATT_STUB
#define VLFM_ATT_STUB 0
#endif

// Optional phase timing (diagnostic build with -DVLFM_PHASE_TIMING; tools/vit_attn_phase_probe.py): lane 0 of wavefront 0
// of the LAST workgroup stamps the 100 MHz wall clock at phase boundaries.
#ifdef VLFM_PHASE_TIMING
__device__ long long g_att_clk[16];
#define AT_PHASE(k)                                                                              \
    do {                                                                                         \
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) g_att_clk[k] = wall_clock64();      \
    } while (0)
#else
#define AT_PHASE(k) do {} while (0)
#endif

// Column of token ``key`` in a V^T row.  Accumulator register r of a 32x32 MFMA tile holds row (r & 3) + 8 (r >> 2) +
// 4 (lane >> 5): the 8 probabilities a lane feeds to one PV MFMA belong to keys {4g..4g+3, 8+4g..8+4g+3} of a 16-key
// chunk (g = lane >> 5).  V^T is stored with exactly those 8 keys adjacent, so the matching A operand is ONE 16-byte read.
__device__ inline int vt_col(int key) {
    const int e = key & 15;
    return (key & ~15) + 8 * ((e >> 2) & 1) + 4 * (e >> 3) + (e & 3);
}

// Attention of the 32 query columns in ``qf`` against key tiles kt0 .. kt0+KT-1.  Returns the UNNORMALISED O^T
// accumulators (3 tiles of 32 head channels), the column maximum of the raw scores and the column sum of
// exp2((s - max) * c), all per lane for query column lane&31 (combined over both lane halves).
template <int KT>
__device__ inline void attend(const _Float16* __restrict__ Kl, const _Float16* __restrict__ Vl, const half8_t (&qf)[AT_D / 16],
                              int kt0, float c, f32x16_t (&o)[AT_D / 32], float& m_out, float& l_out) {
    const int lane = threadIdx.x & 63, col = lane & 31, grp = lane >> 5;
    f32x16_t acc[KT];
#pragma unroll
    for (int t = 0; t < KT; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    }
    // S^T = K Q^T.  Key tiles are taken in groups of (up to) 3 with the contraction step as the outer loop: consecutive
    // MFMAs then write DIFFERENT accumulators (a dependent MFMA is three issue slots away), and the K operands of step
    // kk + 1 are fetched from LDS while step kk is in the matrix pipe.
    constexpr int G = KT >= 3 ? 3 : KT;
    static_assert(KT % G == 0, "key tiles come in whole groups");
    const _Float16* kbase = Kl + (size_t)(32 * kt0 + col) * AT_KS + 8 * grp;
#pragma unroll
    for (int g0 = 0; g0 < KT; g0 += G) {
        half8_t a_cur[G], a_nxt[G];
#pragma unroll
        for (int t = 0; t < G; t++) a_cur[t] = *reinterpret_cast<const half8_t*>(kbase + (size_t)(32 * (g0 + t)) * AT_KS);
#pragma unroll
        for (int kk = 0; kk < AT_D / 16; kk++) {
            if (kk + 1 < AT_D / 16) {
#pragma unroll
                for (int t = 0; t < G; t++)
                    a_nxt[t] = *reinterpret_cast<const half8_t*>(kbase + (size_t)(32 * (g0 + t)) * AT_KS + 16 * (kk + 1));
            }
#pragma unroll
            for (int t = 0; t < G; t++)
                acc[g0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[t], qf[kk], acc[g0 + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < G; t++) a_cur[t] = a_nxt[t];
        }
    }
    if (KT == AT_KT) AT_PHASE(4);
    // accumulator register r of tile t holds key 32 (kt0 + t) + (r & 3) + 8 (r >> 2) + 4 grp of query column ``col``
    float m = -__builtin_huge_valf();
#pragma unroll
    for (int t = 0; t < KT; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int key = 32 * (kt0 + t) + (r & 3) + 8 * (r >> 2) + 4 * grp;
            if (key >= AT_S) acc[t][r] = -__builtin_huge_valf();
            m = fmaxf(m, acc[t][r]);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.0f;
    const float mc = -m * c;
#pragma unroll
    for (int t = 0; t < KT; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float p = __builtin_amdgcn_exp2f(fmaf(acc[t][r], c, mc));   // v_exp_f32: argument <= 0, denormals may flush
            l += p;
            acc[t][r] = p;
        }
    }
    l += __shfl_xor(l, 32, 64);
    if (KT == AT_KT) AT_PHASE(5);
#pragma unroll
    for (int dt = 0; dt < AT_D / 32; dt++) {
#pragma unroll
        for (int r = 0; r < 16; r++) o[dt][r] = 0.0f;
    }
    // O^T = V^T P^T: three channel tiles per 16-key chunk (independent accumulators), the next chunk's V^T operands in flight
    const _Float16* vbase = Vl + (size_t)col * AT_VS + 32 * kt0 + 8 * grp;
    half8_t v_cur[AT_D / 32], v_nxt[AT_D / 32];
#pragma unroll
    for (int dt = 0; dt < AT_D / 32; dt++) v_cur[dt] = *reinterpret_cast<const half8_t*>(vbase + (size_t)(32 * dt) * AT_VS);
#pragma unroll
    for (int ch = 0; ch < 2 * KT; ch++) {       // chunk ch = 16 keys: tile ch / 2, half ch & 1
        if (ch + 1 < 2 * KT) {
#pragma unroll
            for (int dt = 0; dt < AT_D / 32; dt++)
                v_nxt[dt] = *reinterpret_cast<const half8_t*>(vbase + (size_t)(32 * dt) * AT_VS + 16 * (ch + 1));
        }
        half8_t pb;
        // keys behind pb[0..3]: 16 ch + 4 grp + (0..3); behind pb[4..7]: the same + 8 -- the order vt_col() stores
#pragma unroll
        for (int j = 0; j < 8; j++) pb[j] = (_Float16)acc[ch >> 1][8 * (ch & 1) + j];
#pragma unroll
        for (int dt = 0; dt < AT_D / 32; dt++) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_cur[dt], pb, o[dt], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < AT_D / 32; dt++) v_cur[dt] = v_nxt[dt];
    }
    m_out = m;
    l_out = l;
}

// DH = head width as stored in global memory: 96 (heads zero-padded inside the qkv / projection weights) or 88 (ViT-g's
// native width: the missing 8 channels are zeros that exist only in LDS / registers, so the GEMMs on either side
// keep their original sizes).
template <int DH>
__global__ __launch_bounds__(64 * AT_WAVES, 2) void vit_attention_kernel(const _Float16* __restrict__ qkv,
                                                                         _Float16* __restrict__ out, int B, int H,
                                                                         float scale, int stagger) {
    extern __shared__ __attribute__((aligned(16))) unsigned char at_lds[];
    _Float16* Kl = reinterpret_cast<_Float16*>(at_lds);
    _Float16* Vl = reinterpret_cast<_Float16*>(at_lds + AT_LDS_K);
    float* part = reinterpret_cast<float*>(at_lds + AT_LDS_K + AT_LDS_V);
    _Float16* qcls = reinterpret_cast<_Float16*>(at_lds + AT_LDS_K + AT_LDS_V + AT_LDS_PART);
    // XCD-aware placement: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2).  A head's K / V
    // rows are 192-byte pieces of 9 KB token rows, so neighbouring heads share cache lines: all 16 heads of an image are
    // given to ONE XCD, back to back, and each line is fetched from HBM once instead of once per XCD that touches it.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = xcd + 8 * (slot / H), h = slot % H;
    if (b >= B) return;
    const int tid = threadIdx.x, nth = blockDim.x;
    AT_PHASE(0);
    const int lane = tid & 63, wave = tid >> 6, col = lane & 31, grp = lane >> 5;
    static_assert(DH % 8 == 0 && DH <= AT_D, "head width in whole 16-byte chunks");
    const size_t row_halfs = (size_t)3 * H * DH;                         // one token of qkv: [3][H][DH]
    const _Float16* base = qkv + (size_t)b * AT_S * row_halfs + (size_t)h * DH;
    // ---- stage K (row-major) and V^T into LDS; padding rows / columns are zero
    for (int i = tid; i < (AT_KROWS - AT_S) * AT_KS; i += nth) Kl[(size_t)AT_S * AT_KS + i] = (_Float16)0.0f;
    for (int i = tid; i < AT_D * (AT_KROWS - AT_S); i += nth) {   // keys 257 .. 287 of every V^T row
        const int d = i / (AT_KROWS - AT_S), k = AT_S + i % (AT_KROWS - AT_S);
        Vl[(size_t)d * AT_VS + vt_col(k)] = (_Float16)0.0f;
    }
    // All global loads of the staging phase are issued before the first LDS write (one exposure to HBM latency, not
    // one per iteration); a work item is (token pair, 8-channel chunk) so that V^T is written two tokens (4 bytes) at
    // a time -- vt_col() keeps an even token and its successor adjacent.
    constexpr int kPairs = (AT_S + 1) / 2, kItems = kPairs * (AT_D / 8), kIters = (kItems + 64 * AT_WAVES - 1) / (64 * AT_WAVES);
    half8_t kreg[kIters][2], vreg[kIters][2];
#pragma unroll
    for (int it = 0; it < kIters; it++) {
        const int i = tid + it * 64 * AT_WAVES;
        const int s0 = 2 * (i / (AT_D / 8)), ch = i % (AT_D / 8);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int s = s0 + u;
            if (VLFM_ATT_STUB != 2 && i < kItems && s < AT_S && ch < DH / 8) {
                const _Float16* src = base + (size_t)s * row_halfs + (size_t)H * DH + 8 * ch;
                kreg[it][u] = *reinterpret_cast<const half8_t*>(src);                      // K
                vreg[it][u] = *reinterpret_cast<const half8_t*>(src + (size_t)H * DH);     // V
            } else {   // token 257 of the last pair, and channels DH .. 95: zeros
#pragma unroll
                for (int j = 0; j < 8; j++) { kreg[it][u][j] = (_Float16)0.0f; vreg[it][u][j] = (_Float16)0.0f; }
            }
        }
    }
    AT_PHASE(1);
    // the Q fragments of this wavefront's 32 queries travel while K / V are being staged
    const int tq = 1 + 32 * wave + col;
    half8_t qmain[AT_D / 16];
#pragma unroll
    for (int kk = 0; kk < AT_D / 16; kk++) {
        if (VLFM_ATT_STUB != 2 && 2 * kk + grp < DH / 8) {
            qmain[kk] = *reinterpret_cast<const half8_t*>(base + (size_t)tq * row_halfs + 8 * grp + 16 * kk);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) qmain[kk][j] = (_Float16)0.0f;
        }
    }
    if (tid < AT_D / 8) {
        half8_t qv;
#pragma unroll
        for (int j = 0; j < 8; j++) qv[j] = (_Float16)0.0f;
        if (tid < DH / 8) qv = *reinterpret_cast<const half8_t*>(base + 8 * tid);
        *reinterpret_cast<half8_t*>(qcls + 8 * tid) = qv;
    }
#pragma unroll
    for (int it = 0; it < kIters; it++) {
        const int i = tid + it * 64 * AT_WAVES;
        const int s0 = 2 * (i / (AT_D / 8)), ch = i % (AT_D / 8);
        if (i < kItems) {
            *reinterpret_cast<half8_t*>(Kl + (size_t)s0 * AT_KS + 8 * ch) = kreg[it][0];
            if (s0 + 1 < AT_S) *reinterpret_cast<half8_t*>(Kl + (size_t)(s0 + 1) * AT_KS + 8 * ch) = kreg[it][1];
            const int vc = vt_col(s0);   // s0 is even: vt_col(s0 + 1) == vc + 1 (token 257 does not exist: zero)
            using half2_t = __attribute__((ext_vector_type(2))) _Float16;
#pragma unroll
            for (int j = 0; j < 8; j++)
                *reinterpret_cast<half2_t*>(Vl + (size_t)(8 * ch + j) * AT_VS + vc) = half2_t{vreg[it][0][j], vreg[it][1][j]};
        }
    }
    AT_PHASE(2);
    __syncthreads();
    AT_PHASE(3);
    const float c = scale * 1.4426950408889634f;   // exp(x * scale) = exp2(x * c)
    // ---- the CLS query (token 0): column 0 of a ninth query tile; its keys are split over the wavefronts (tile w for
    // wavefront w, tiles 7 and 8 for the last), partial (max, sum, O) go to LDS and are merged below
    auto cls_part = [&]() {
        half8_t qf[AT_D / 16];
#pragma unroll
        for (int kk = 0; kk < AT_D / 16; kk++) {
            half8_t z;
#pragma unroll
            for (int j = 0; j < 8; j++) z[j] = (_Float16)0.0f;
            qf[kk] = col == 0 ? *reinterpret_cast<const half8_t*>(qcls + 8 * grp + 16 * kk) : z;
        }
        f32x16_t o[AT_D / 32];
        float m, l;
        if (wave < AT_WAVES - 1) attend<1>(Kl, Vl, qf, wave, c, o, m, l);
        else attend<2>(Kl, Vl, qf, AT_WAVES - 1, c, o, m, l);
        float* mine = part + (size_t)wave * (AT_D + 2);
        if (col == 0) {
#pragma unroll
            for (int dt = 0; dt < AT_D / 32; dt++) {
#pragma unroll
                for (int r = 0; r < 16; r++) mine[32 * dt + (r & 3) + 8 * (r >> 2) + 4 * grp] = o[dt][r];
            }
            if (grp == 0) { mine[AT_D] = m; mine[AT_D + 1] = l; }
        }
    };
    // Wavefronts w and w + 4 share a SIMD: the upper four do their (short) CLS share first, so that the pair is out of
    // phase -- one in the matrix pipe while the other does softmax VALU work.
    const bool cls_first = stagger && wave >= AT_WAVES / 2;
    if (VLFM_ATT_STUB != 1 && VLFM_ATT_STUB != 3 && cls_first) cls_part();
    // ---- the 32 queries of tokens 1 + 32 wave .. 32 + 32 wave
    {
        f32x16_t o[AT_D / 32];
        float m, l;
        if (VLFM_ATT_STUB != 1) {
            attend<AT_KT>(Kl, Vl, qmain, 0, c, o, m, l);
        } else {
            l = 1.0f;
#pragma unroll
            for (int dt = 0; dt < AT_D / 32; dt++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[dt][r] = (float)qmain[dt][r & 7];
        }
        AT_PHASE(7);
        const float inv = 1.0f / l;
        _Float16* dst = out + ((size_t)(b * AT_S + tq) * H + h) * DH + 4 * grp;
#pragma unroll
        for (int dt = 0; dt < AT_D / 32; dt++) {
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {
                const half4_t v = half4_t{(_Float16)(o[dt][4 * q4] * inv), (_Float16)(o[dt][4 * q4 + 1] * inv),
                                          (_Float16)(o[dt][4 * q4 + 2] * inv), (_Float16)(o[dt][4 * q4 + 3] * inv)};
                if ((VLFM_ATT_STUB != 4 || v[0] == (_Float16)12345.0f) && 32 * dt + 8 * q4 + 4 * grp < DH)
                    *reinterpret_cast<half4_t*>(dst + 32 * dt + 8 * q4) = v;
            }
        }
    }
    AT_PHASE(8);
    if (VLFM_ATT_STUB != 1 && VLFM_ATT_STUB != 3 && !cls_first) cls_part();
    AT_PHASE(9);
    __syncthreads();
    AT_PHASE(10);
    if (wave == 0) {
        float mx = -__builtin_huge_valf();
#pragma unroll
        for (int w = 0; w < AT_WAVES; w++) mx = fmaxf(mx, part[(size_t)w * (AT_D + 2) + AT_D]);
        float wgt[AT_WAVES], lsum = 0.0f;
#pragma unroll
        for (int w = 0; w < AT_WAVES; w++) {
            wgt[w] = exp2f((part[(size_t)w * (AT_D + 2) + AT_D] - mx) * c);
            lsum += wgt[w] * part[(size_t)w * (AT_D + 2) + AT_D + 1];
        }
        const float inv = 1.0f / lsum;
        _Float16* dst = out + ((size_t)(b * AT_S) * H + h) * DH;
        for (int d = lane; d < DH; d += 64) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < AT_WAVES; w++) v += wgt[w] * part[(size_t)w * (AT_D + 2) + d];
            dst[d] = (_Float16)(v * inv);
        }
    }
    AT_PHASE(11);
}

}  // namespace vlfm

using namespace vlfm;

extern "C" int vlfm_vit_attention_f16(const void* d_qkv, void* d_out, int batch, int tokens, int heads, int head_dim,
                                      float scale, void* stream) {
    if (batch == 0) return VLFM_OK;
    if (!d_qkv || !d_out || batch < 0 || heads <= 0)
        return fail(VLFM_ERR_INVALID, "vit_attention_f16: bad argument");
    if (tokens != AT_S || (head_dim != AT_D && head_dim != 88))
        return fail(VLFM_ERR_INVALID, "vit_attention_f16: specialised for 257 tokens and a head width of 88 or 96");
    static LdsOptIn opt_in96, opt_in88;
    if (!(head_dim == 88 ? opt_in88.ensure(reinterpret_cast<const void*>(vit_attention_kernel<88>), AT_LDS_BYTES)
                         : opt_in96.ensure(reinterpret_cast<const void*>(vit_attention_kernel<96>), AT_LDS_BYTES)))
        return fail(VLFM_ERR_HIP, "vit_attention_f16: cannot opt in to 117 KB of LDS");
    // the upper four wavefronts do their CLS share first (VLFM_ATT_STAGGER=0, diagnostic: all do it last): 269 against 277 us at 256
    // images; delaying them further (up to 3 000 cycles) changes nothing (round 5, profiles/r05_vit_attention_stub_probe.txt)
    const char* es = getenv("VLFM_ATT_STAGGER");
    const int stagger = es ? (atoi(es) != 0) : 1;
    VLFM_TIMED("vit_attention_kernel", stream);
    const dim3 grid(8 * ((batch + 7) / 8) * heads), block(64 * AT_WAVES);
    if (head_dim == 88) {
        VLFM_KLAUNCH(vit_attention_kernel<88>, grid, block, AT_LDS_BYTES, (hipStream_t)stream, (const _Float16*)d_qkv,
                     (_Float16*)d_out, batch, heads, scale, stagger);
    } else {
        VLFM_KLAUNCH(vit_attention_kernel<96>, grid, block, AT_LDS_BYTES, (hipStream_t)stream, (const _Float16*)d_qkv,
                     (_Float16*)d_out, batch, heads, scale, stagger);
    }
    return check_launch("vit_attention_kernel");
}

#ifdef VLFM_PHASE_TIMING
extern "C" int vlfm_debug_attention_clocks(long long* h_out16) {
    return hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(g_att_clk), sizeof(long long) * 16) == hipSuccess ? VLFM_OK : VLFM_ERR_HIP;
}
#endif
