// conv_nhwc.hip -- 1x1 and 3x3 convolutions of the YOLOv7-E6E deploy graph (reference: vlfm/vlm/yolov7.py:35-48 loads
// `yolov7-e6e.pt`, :89 runs it in fp16) as IMPLICIT GEMMs on the gfx950 matrix cores, bias + SiLU in the epilogue.
//
// Why: after BatchNorm folding the network is 244 convolutions, 147 GFLOP per 448x640 frame; 64 % of it are 3x3 stride-1 layers with
// Cin = Cout in {64 .. 512}, 29 % are 1x1 layers.  MIOpen serves the fp16 3x3 layers with `gfx9_fp16_dot2` Winograd kernels
// (v_dot2 on the vector ALU, ~185 TFLOP/s measured over the whole net, profiles/r03_yolo_*) and every layer is followed by a
// separate bias + SiLU pass over the activation.
//
// Formulation: activations are NHWC (torch channels_last), weights [Cout][kh][kw][Cin] (the channels_last memory of the folded
// weight), so that for one filter tap a pixel's Cin channels and a filter's Cin weights are both K-contiguous:
//     out[m][n] = act(bias[n] + sum_tap sum_c  x[pixel(m) + tap][c] . w[n][tap][c]),      m = (b, oy, ox) linear, n = output channel
// is an "NT" GEMM with K = taps * Cin in which the A-row of pixel m moves by a constant (dy * W + dx) * pixel_stride per tap.
// A K-tile is 64 channels of ONE tap (Cin must be a multiple of 64), staged by global_load_lds with one address per lane: rows whose
// tap falls into the zero padding (or beyond M) fetch from a 16-byte zero page instead -- no im2col buffer, no padded copy.
// LDS layout, the XOR slot permutation that makes the MFMA fragment reads conflict-free, the operand roles (A-operand = filter rows,
// so a lane ends up with 4 consecutive output channels of one pixel) and the transposing epilogue are those of gemm_f16.hip.
// Tile: BM pixels x BN channels from seven shapes between 256 x 256 and 64 x 64, chosen per layer by a cost estimate (pick_cfg);
// 8 wavefronts; two LDS buffers; the lock-step schedule.  Pixel strides of input and output are arguments: a layer can read a channel slice of a concatenation buffer
// and write into one.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/vlfm_amd.h"
#include "profile.h"
#include "status.h"

namespace vlfm {
namespace conv {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
using lds_ptr = __attribute__((address_space(3))) unsigned char*;
using gbl_ptr = const __attribute__((address_space(1))) unsigned char*;

constexpr int GK = 64;             // channels per K-tile
constexpr int ROWB = GK * 2;       // bytes per staged row

struct ConvArgs {
    const _Float16* x;      // [B][H][W] pixels of x_pix elements, the first Cin of which are read
    const _Float16* w;      // [Cout][taps][Cin]
    const _Float16* bias;   // [Cout] or null
    _Float16* out;          // [B][Ho][Wo] pixels of out_pix elements, the first Cout of which are written
    const _Float16* zero;   // >= 16 bytes of zeros
    int B, H, W, Ho, Wo, Cin, Cout;
    int taps, stride, pad;
    int x_pix, out_pix;
    int M, tiles_m, tiles_n, ktiles_per_tap;
};

enum { ACT_NONE = 0, ACT_SILU = 1 };

__device__ inline float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// GEN = false: Cin % 64 == 0, a K-tile is 64 channels of one tap (one tap decode per K-tile, scalar).
// GEN = true : Cin % 8 == 0; k runs over (tap, channel) continuously, every 16-byte slot (8 channels) decodes its own tap, and K is
//              padded to a multiple of 64 with zero weights (the filter rows are Kpad = round_up(taps * Cin, 64) elements apart; slots
//              past the last tap fetch the zero page) -- the 80 / 160 / 480-channel layers and the 12(16)-channel stem.
template <int BM, int BN, int WM, int WN, int ACT, bool GEN>
__global__ __launch_bounds__(512) void conv_nhwc_kernel(ConvArgs a) {
    constexpr int FA = WN / 16, FB = WM / 16;       // fragments per wavefront: filter rows, pixel rows
    constexpr int WAVES_M = BM / WM;
    constexpr int WOPER = BN * ROWB, BUF = WOPER + BM * ROWB;
    constexpr int WCH = BN / 64;                    // 8-row chunks of the filter tile each wavefront stages
    constexpr int XCH = BM / 64;                    // ... and of the pixel tile
    constexpr int EPI_ROW = WN * 2 + 16;            // epilogue staging row stride (conflict-free 8-byte writes, 16-byte reads)
    static_assert((BM / WM) * (BN / WN) == 8, "8 wavefronts");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_ptr lds = (lds_ptr)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    // XCD-aware tile order (as in gemm_f16.hip): every XCD works on a contiguous range of tiles, n fastest, so that the tiles sharing
    // an activation panel (and, for 3x3 layers, the image rows above and below it) meet in one L2
    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / a.tiles_n, tn = bid - tm * a.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Ktot = GEN ? (a.taps * a.Cin + GK - 1) / GK * GK : a.taps * a.Cin;   // filter row stride
    const int NT = Ktot / GK;

    // ---- the four pixel rows and WCH filter rows this lane stages in every K-tile
    const int sub = lane >> 3, p = lane & 7;
    int x_off[XCH], x_iy[XCH], x_ix[XCH];
    int slot2[2];                       // the logical 16-byte slot this lane fetches in rows of an even / odd chunk
    slot2[0] = p ^ (sub >> 1);
    slot2[1] = p ^ (4 + (sub >> 1));
#pragma unroll
    for (int j = 0; j < XCH; j++) {
        const int r = (wave * XCH + j) * 8 + sub, m = m0 + r;     // (r >> 1) & 7 == (chunk & 1) * 4 + (sub >> 1)
        if (m < a.M) {
            const int hw = a.Ho * a.Wo, b = m / hw, rem = m - b * hw, oy = rem / a.Wo, ox = rem - oy * a.Wo;
            x_iy[j] = oy * a.stride - a.pad;
            x_ix[j] = ox * a.stride - a.pad;
            x_off[j] = ((b * a.H + x_iy[j]) * a.W + x_ix[j]) * a.x_pix + (GEN ? 0 : slot2[(wave * XCH + j) & 1] * 8);
        } else {
            x_iy[j] = -(1 << 20); x_ix[j] = 0; x_off[j] = 0;
        }
    }
    int w_off[WCH];
#pragma unroll
    for (int j = 0; j < WCH; j++) {
        const int r = (wave * WCH + j) * 8 + sub;
        w_off[j] = min(n0 + r, a.Cout - 1) * Ktot + (p ^ ((r >> 1) & 7)) * 8;
    }

    const int cin8 = a.Cin >> 3;
    auto stage = [&](int buf, int kt) {
        int ty[2] = {0, 0}, tx[2] = {0, 0}, tap_off[2];
        bool tap_ok[2] = {true, true};
        if (GEN) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int q = kt * 8 + slot2[h], tap = q / cin8, c = (q - tap * cin8) * 8;
                tap_ok[h] = tap < a.taps;
                if (a.taps == 9) { ty[h] = tap / 3; tx[h] = tap - ty[h] * 3; }
                tap_off[h] = (ty[h] * a.W + tx[h]) * a.x_pix + c;
            }
        } else {
            const int tap = kt / a.ktiles_per_tap, c0 = (kt - tap * a.ktiles_per_tap) * GK;
            if (a.taps == 9) { ty[0] = tap / 3; tx[0] = tap - ty[0] * 3; }
            ty[1] = ty[0]; tx[1] = tx[0];
            tap_off[0] = tap_off[1] = (ty[0] * a.W + tx[0]) * a.x_pix + c0;
        }
#pragma unroll
        for (int j = 0; j < XCH; j++) {
            const int h = (wave * XCH + j) & 1;
            const bool ok = tap_ok[h] && (unsigned)(x_iy[j] + ty[h]) < (unsigned)a.H && (unsigned)(x_ix[j] + tx[h]) < (unsigned)a.W;
            const _Float16* gx = ok ? a.x + (x_off[j] + tap_off[h]) : a.zero;
            const int dst = __builtin_amdgcn_readfirstlane(buf + WOPER + (wave * XCH + j) * 1024);
            __builtin_amdgcn_global_load_lds((gbl_ptr)gx, lds + dst, 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WCH; j++) {
            const _Float16* gw = a.w + (w_off[j] + kt * GK);
            const int dst = __builtin_amdgcn_readfirstlane(buf + (wave * WCH + j) * 1024);
            __builtin_amdgcn_global_load_lds((gbl_ptr)gw, lds + dst, 16, 0, 0);
        }
    };
    // MFMA fragments of one half K-tile (32 of the 64 channels)
    auto read_frags = [&](int buf, int kk, half8 (&fa)[FA], half8 (&fb)[FB]) {
        const int r16 = lane & 15, s = kk * 4 + (lane >> 4);
#pragma unroll
        for (int i = 0; i < FA; i++) {
            const int R = wn * WN + i * 16 + r16;
            fa[i] = *reinterpret_cast<const half8*>(smem + buf + R * ROWB + ((s ^ ((R >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < FB; j++) {
            const int R = wm * WM + j * 16 + r16;
            fb[j] = *reinterpret_cast<const half8*>(smem + buf + WOPER + R * ROWB + ((s ^ ((R >> 1) & 7)) << 4));
        }
    };

    floatx4 acc[FA][FB];
#pragma unroll
    for (int i = 0; i < FA; i++)
#pragma unroll
        for (int j = 0; j < FB; j++) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 fa0[FA], fb0[FB], fa1[FA], fb1[FB];
    auto mma = [&](const half8 (&fa)[FA], const half8 (&fb)[FB]) {
#pragma unroll
        for (int i = 0; i < FA; i++)
#pragma unroll
            for (int j = 0; j < FB; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };

    stage(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (NT > 1) stage(BUF, 1);
    read_frags(0, 0, fa0, fb0);
    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0)

    for (int t = 0; t < NT; t++) {
        const int cur = (t & 1) * BUF;
        read_frags(cur, 1, fa1, fb1);                      // second half of tile t: in flight under the MFMAs below
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                // every LDS read of tile t by this wavefront has returned
        if (t + 1 < NT) {
            __builtin_amdgcn_s_waitcnt(0x0F70);            // ... and its share of tile t + 1 has landed
            __builtin_amdgcn_s_barrier();                  // ... for everybody: buffer `cur` is free, the other one is complete
            asm volatile("" ::: "memory");
            if (t + 2 < NT) stage(cur, t + 2);
            read_frags(cur ^ BUF, 0, fa0, fb0);            // first half of tile t + 1
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    __builtin_amdgcn_s_barrier();   // every wavefront is done with the operand buffers: the epilogue staging aliases them
    asm volatile("" ::: "memory");

    // ---- epilogue: bias (+ SiLU) on the f32 accumulators, f16, transpose through a wavefront-private LDS region, 16-byte stores.
    // Round 5 (as in gemm_f16.hip's store_tile): the FA bias quads are loaded before the arithmetic instead of one dependent L2 round
    // trip per fragment row, and all LDS reads of the second half precede the first store -- with one register quad re-used for every
    // ds_read_b128 / global_store pair hipcc waits vmcnt(0), i.e. for the previous STORE to leave, before every read.  A 1x1 layer with
    // 256 input channels is four K-tiles: the epilogue is a large share of such a tile.
    unsigned char* stg = smem + wave * (WM * EPI_ROW);
    const int g4 = (lane >> 4) * 4, c16 = lane & 15;
    half4 bias4[FA];
#pragma unroll
    for (int i = 0; i < FA; i++) {
        bias4[i] = half4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (a.bias) bias4[i] = *reinterpret_cast<const half4*>(a.bias + min(n0 + wn * WN + i * 16 + g4, a.Cout - 4));
    }
#pragma unroll
    for (int i = 0; i < FA; i++) {
        const int nl = i * 16 + g4;                    // 4 consecutive output channels of this lane, wavefront-local
        const float b0 = (float)bias4[i][0], b1 = (float)bias4[i][1], b2 = (float)bias4[i][2], b3 = (float)bias4[i][3];
#pragma unroll
        for (int j = 0; j < FB; j++) {
            float v0 = acc[i][j][0] + b0, v1 = acc[i][j][1] + b1, v2 = acc[i][j][2] + b2, v3 = acc[i][j][3] + b3;
            if (ACT == ACT_SILU) { v0 = silu(v0); v1 = silu(v1); v2 = silu(v2); v3 = silu(v3); }
            const half4 h = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
            *reinterpret_cast<half4*>(stg + (j * 16 + c16) * EPI_ROW + nl * 2) = h;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // wavefront-private region: no workgroup barrier needed
    constexpr int CHUNKS = WN / 8, ROWS_PER_IT = 64 / CHUNKS, ITS = WM / ROWS_PER_IT;
    const int rsub = lane / CHUNKS, chunk = lane % CHUNKS;
    uint4 v[ITS];
#pragma unroll
    for (int it = 0; it < ITS; it++) v[it] = *reinterpret_cast<const uint4*>(stg + (it * ROWS_PER_IT + rsub) * EPI_ROW + chunk * 16);
    const int n = n0 + wn * WN + chunk * 8;
    if (n + 8 <= a.Cout) {
#pragma unroll
        for (int it = 0; it < ITS; it++) {
            const int m = m0 + wm * WM + it * ROWS_PER_IT + rsub;
            if (m < a.M) *reinterpret_cast<uint4*>(a.out + (size_t)m * a.out_pix + n) = v[it];
        }
    }
}

template <int BM, int BN, int WM, int WN>
constexpr int lds_bytes() {
    constexpr int ops = 2 * (BN + BM) * ROWB, epi = 8 * WM * (WN * 2 + 16);
    return ops > epi ? ops : epi;
}

template <int BM, int BN, int WM, int WN, bool GEN>
static int launch(ConvArgs a, int act, hipStream_t stream) {
    constexpr int LDS = lds_bytes<BM, BN, WM, WN>();
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    void (*const k_silu)(ConvArgs) = conv_nhwc_kernel<BM, BN, WM, WN, ACT_SILU, GEN>;
    void (*const k_none)(ConvArgs) = conv_nhwc_kernel<BM, BN, WM, WN, ACT_NONE, GEN>;
    static LdsOptIn opt[2];
    if (!opt[act].ensure(reinterpret_cast<const void*>(act == ACT_SILU ? k_silu : k_none), LDS))
        return fail(VLFM_ERR_HIP, "conv_nhwc_f16: cannot opt in to the LDS size");
    const dim3 grid(a.tiles_m * a.tiles_n), block(512);
    VLFM_TIMED("conv_nhwc_kernel", stream);
    if (act == ACT_SILU) VLFM_KLAUNCH(k_silu, grid, block, LDS, stream, a);
    else VLFM_KLAUNCH(k_none, grid, block, LDS, stream, a);
    return check_launch("conv_nhwc_kernel");
}

// The tile shapes (pixels x channels; 8 wavefronts each): the big ones have the best ratio of MFMA work to LDS traffic, the small ones
// fill the chip when a layer has few pixels (7x10 and 14x20 feature maps) and waste less on narrow layers.
struct TileCfg { int bm, bn; double rate; };      // rate: relative throughput of a CU running this shape (tools/conv_nhwc_probe.py sweep:
                                                  // with these the pick is within 1.3 % of the best shape summed over the probe layers)
static const TileCfg kCfg[] = {{256, 256, 0.85}, {256, 128, 1.00}, {256, 64, 0.95}, {128, 128, 1.05},
                               {128, 64, 0.85},  {64, 128, 0.82},  {64, 64, 0.62}};
constexpr int kNumCfg = 7;   // (a 512 x 64 shape -- a third less LDS traffic per MFMA, but one workgroup per CU -- measured 24 % slower
                             //  than 256 x 64 on the 64-channel layers and removed)

template <bool GEN>
static int launch_cfg(const ConvArgs& a, int cfg, int act, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch<256, 256, 64, 128, GEN>(a, act, stream);
        case 1: return launch<256, 128, 64, 64, GEN>(a, act, stream);
        case 2: return launch<256, 64, 32, 64, GEN>(a, act, stream);
        case 3: return launch<128, 128, 32, 64, GEN>(a, act, stream);
        case 4: return launch<128, 64, 32, 32, GEN>(a, act, stream);
        case 5: return launch<64, 128, 16, 64, GEN>(a, act, stream);
        default: return launch<64, 64, 16, 32, GEN>(a, act, stream);
    }
}

// Estimated time of a layer under a tile shape, in units of (256 x 256 x 64) MFMA tiles at the big shape's rate: the busiest CU works
// through ceil(tiles / 256) tiles (workgroups are dealt round-robin), each costing its padded volume over the shape's rate.
static int pick_cfg(int M, int cout, int ktiles) {
    int best = 0;
    double best_t = 1e300;
    for (int c = 0; c < kNumCfg; c++) {
        const long long tiles = (long long)((M + kCfg[c].bm - 1) / kCfg[c].bm) * ((cout + kCfg[c].bn - 1) / kCfg[c].bn);
        const double per_cu = (double)((tiles + 255) / 256);
        // + 1.5 K-tiles of fixed cost per tile: prologue (first loads in flight) and the epilogue's store
        const double t = per_cu * (kCfg[c].bm / 256.0) * (kCfg[c].bn / 256.0) * (ktiles + 1.5) / kCfg[c].rate;
        if (t < best_t) { best_t = t; best = c; }
    }
    return best;
}

// 2x2 / stride-2 max pooling on NHWC f16 rows (DownC's `mp` in front of its 1x1 convolution): 16 bytes = 8 channels per lane,
// consecutive lanes on consecutive channels.  The framework's NHWC pooling kernel moves ~1.5 TB/s on the 80-channel 224x320 map.
__global__ __launch_bounds__(256) void maxpool2x2_nhwc_f16_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ out, int H,
                                                                 int W, int C8, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int Ho = H >> 1, Wo = W >> 1;
    const int c = (int)(i % C8);
    const long long pix = i / C8;
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
    const long long b = pix / ((long long)Wo * Ho);
    const half8* p = reinterpret_cast<const half8*>(x) + (((b * H + 2 * oy) * W + 2 * ox) * C8 + c);
    const half8 v00 = p[0], v01 = p[C8], v10 = p[(long long)W * C8], v11 = p[(long long)W * C8 + C8];
    half8 r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const _Float16 a = v00[k] > v01[k] ? v00[k] : v01[k], bb = v10[k] > v11[k] ? v10[k] : v11[k];
        r[k] = a > bb ? a : bb;
    }
    reinterpret_cast<half8*>(out)[i] = r;
}

}  // namespace conv
}  // namespace vlfm

using namespace vlfm;

// out[b][oy][ox][:] = max over the 2x2 block of x (stride 2, no padding; height and width even): contiguous NHWC f16, channels % 8 == 0.
extern "C" int vlfm_maxpool2x2_nhwc_f16(const void* d_x, void* d_out, int batch, int height, int width, int channels, void* stream) {
    if (batch == 0) return VLFM_OK;
    if (!d_x || !d_out || batch < 0 || height <= 0 || width <= 0 || (height & 1) || (width & 1) || channels <= 0 || (channels & 7))
        return fail(VLFM_ERR_INVALID, "maxpool2x2_nhwc_f16: even height and width, channels % 8 == 0");
    const long long n = (long long)batch * (height / 2) * (width / 2) * (channels / 8);
    const long long blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(VLFM_ERR_CAPACITY, "maxpool2x2_nhwc_f16: tensor too large");
    VLFM_TIMED("maxpool2x2_nhwc_f16_kernel", stream);
    VLFM_KLAUNCH(conv::maxpool2x2_nhwc_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)d_x,
                 (_Float16*)d_out, height, width, channels / 8, n);
    return check_launch("maxpool2x2_nhwc_f16_kernel");
}

// The tile shape vlfm_conv_nhwc_f16 uses for a layer of `pixels` output pixels, `cout` channels and K = ksize^2 * cin (host logic only:
// no device needed; tests/test_host_logic.py pins the properties the estimate must have).
extern "C" int vlfm_conv_nhwc_tile(int pixels, int cin, int cout, int ksize, int* tile_pixels, int* tile_channels) {
    if (pixels <= 0 || cin <= 0 || cout <= 0 || (ksize != 1 && ksize != 3) || !tile_pixels || !tile_channels)
        return fail(VLFM_ERR_INVALID, "conv_nhwc_tile: bad argument");
    const int c = conv::pick_cfg(pixels, cout, (ksize * ksize * cin + conv::GK - 1) / conv::GK);
    *tile_pixels = conv::kCfg[c].bm;
    *tile_channels = conv::kCfg[c].bn;
    return VLFM_OK;
}

// out = act(conv(x, w) + bias) for a 1x1 (pad 0) or 3x3 (pad 1) filter, stride 1 or 2, NHWC f16 with f32 accumulation.
//   d_x   [batch][height][width] pixels, x_pix_stride elements apart, the first `cin` channels of each are read
//   d_w   [cout][Kpad]: a filter's [ksize][ksize][cin] weights followed by zeros up to Kpad = round_up(ksize * ksize * cin, 64)
//         (no padding when cin % 64 == 0)       d_bias [cout] or NULL          d_zero: at least 16 zero bytes
//   d_out [batch][Ho][Wo] pixels, out_pix_stride elements apart, `cout` channels written;  Ho = (height + 2 pad - ksize) / stride + 1
//   act   0 = none, 1 = SiLU
// cin, cout and both pixel strides must be multiples of 8 (cin % 64 == 0 takes the faster staging path), every tensor below 2^31
// elements.
extern "C" int vlfm_conv_nhwc_f16(const void* d_x, const void* d_w, const void* d_bias, void* d_out, const void* d_zero, int batch,
                                  int height, int width, int cin, int cout, int ksize, int stride, int x_pix_stride,
                                  int out_pix_stride, int act, void* stream) {
    if (batch == 0) return VLFM_OK;
    if (!d_x || !d_w || !d_out || !d_zero || batch < 0 || height <= 0 || width <= 0 || cin <= 0 || cout <= 0 ||
        (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || (cin % 8) != 0 || (cout % 8) != 0 ||
        x_pix_stride < cin || out_pix_stride < cout || (x_pix_stride % 8) != 0 || (out_pix_stride % 8) != 0 || act < 0 || act > 1)
        return fail(VLFM_ERR_INVALID, "conv_nhwc_f16: ksize 1|3, stride 1|2, cin % 8 == 0, cout % 8 == 0, pixel strides % 8 == 0");
    conv::ConvArgs a;
    a.x = (const _Float16*)d_x; a.w = (const _Float16*)d_w; a.bias = (const _Float16*)d_bias; a.out = (_Float16*)d_out;
    a.zero = (const _Float16*)d_zero;
    a.B = batch; a.H = height; a.W = width; a.Cin = cin; a.Cout = cout;
    a.taps = ksize * ksize; a.stride = stride; a.pad = ksize / 2;
    a.Ho = (height + 2 * a.pad - ksize) / stride + 1;
    a.Wo = (width + 2 * a.pad - ksize) / stride + 1;
    a.x_pix = x_pix_stride; a.out_pix = out_pix_stride;
    const long long in_elems = (long long)batch * height * width * x_pix_stride;
    const long long out_elems = (long long)batch * a.Ho * a.Wo * out_pix_stride;
    if (in_elems >= (1LL << 31) || out_elems >= (1LL << 31) || (long long)cout * (a.taps * cin + 63) >= (1LL << 31))
        return fail(VLFM_ERR_INVALID, "conv_nhwc_f16: tensors of 2^31 elements or more are not supported (split the batch)");
    a.M = batch * a.Ho * a.Wo;
    a.ktiles_per_tap = cin / conv::GK;
    const int ktiles = (a.taps * cin + conv::GK - 1) / conv::GK;
    int cfg = conv::pick_cfg(a.M, cout, ktiles);
    if (const char* e = getenv("VLFM_CONV_CFG")) { const int v = atoi(e); if (v >= 0 && v < conv::kNumCfg) cfg = v; }   // tuning aid
    return (cin % conv::GK) == 0 ? conv::launch_cfg<false>(a, cfg, act, (hipStream_t)stream)
                                 : conv::launch_cfg<true>(a, cfg, act, (hipStream_t)stream);
}
