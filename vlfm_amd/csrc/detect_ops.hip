// detect_ops.hip -- HIP kernels either side of the detector forward passes (gfx950).
//
//   resize_area      YOLOv7.predict's preprocessing (vlfm/vlm/yolov7.py:70-83): cv2.resize(image, (640, 448), INTER_AREA)
//                    -> letterbox (a no-op at that size) -> HWC->CHW -> half()/float() -> /255, one kernel, u8 in, network
//                    tensor out.  Restates cv::resize's general area path (computeResizeAreaTab + ResizeArea_<uchar,float>:
//                    float accumulation row by row, cvRound saturate) [ext OpenCV 4.5.5].
//   to_tensor_norm   GroundingDINO.predict's preprocessing (vlfm/vlm/grounding_dino.py:52-54): to_tensor + normalize at
//                    native resolution, u8 HWC -> f32 CHW.
//   nms              torchvision.ops.nms [ext] as used by yolov7's non_max_suppression (yolov7.py:91-99): 64x64 IoU
//                    bit-mask tiles (one 64-bit word per row box and column block -- a wavefront-native layout), then an
//                    in-order reduction by a single wavefront ON THE DEVICE (torchvision copies the mask to the host).
// Memory-bound / latency-bound integer+float work; no MFMA.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vlfm_amd.h"
#include "profile.h"
#include "status.h"

namespace vlfm {

// ------------------------------------------------------------------------------------------------ INTER_AREA resize
struct AreaTab {       // per destination index: first source index, tap count, taps (alpha)
    const int* first;  // [dsize]
    const int* count;  // [dsize]
    const float* w;    // [dsize][ktaps]
    int ktaps;
};

template <typename OutT> __device__ inline OutT to_out(float v);
template <> __device__ inline float to_out<float>(float v) { return v; }
template <> __device__ inline __half to_out<__half>(float v) { return __float2half_rn(v); }

// one workgroup per (dst row, image); threads over dst x * channel.  Accumulation order follows ResizeArea_: for every
// contributing source row (top to bottom) the horizontal weighted sum is formed first (left to right), then scaled by the
// row weight and added.
template <typename OutT>
__global__ __launch_bounds__(256) void resize_area_kernel(const unsigned char* __restrict__ src, int H, int W, int OH,
                                                          int OW, AreaTab xt, AreaTab yt, OutT* __restrict__ dst) {
    const int dy = blockIdx.x, n = blockIdx.y;
    const unsigned char* img = src + (size_t)n * H * W * 3;
    const int y0 = yt.first[dy], yc = yt.count[dy];
    const float* yw = yt.w + (size_t)dy * yt.ktaps;
    for (int o = threadIdx.x; o < OW * 3; o += blockDim.x) {
        const int c = o / OW, dx = o - c * OW;  // channel-major: CHW stores are coalesced
        const int x0 = xt.first[dx], xc = xt.count[dx];
        const float* xw = xt.w + (size_t)dx * xt.ktaps;
        float sum = 0.0f;
        for (int j = 0; j < yc; j++) {
            const unsigned char* row = img + ((size_t)(y0 + j) * W) * 3 + c;
            float buf = 0.0f;
            for (int i = 0; i < xc; i++) buf = __fadd_rn(buf, __fmul_rn((float)row[(size_t)(x0 + i) * 3], xw[i]));
            sum = j == 0 ? __fmul_rn(yw[0], buf) : __fadd_rn(sum, __fmul_rn(yw[j], buf));
        }
        // saturate_cast<uchar>(float) = cvRound (half-even) + clamp; then uint8 -> half/float, /= 255 (yolov7.py:81-82)
        int q = __float2int_rn(sum);
        q = q < 0 ? 0 : q > 255 ? 255 : q;
        dst[(((size_t)n * 3 + c) * OH + dy) * OW + dx] = to_out<OutT>(__fdiv_rn((float)q, 255.0f));
    }
}

// ------------------------------------------------------------------------------------------------ to_tensor + normalize
struct Norm3f { float mean[3]; float std[3]; };
__global__ __launch_bounds__(256) void to_tensor_norm_kernel(const unsigned char* __restrict__ src, int HW, Norm3f nrm,
                                                             float* __restrict__ dst) {
    const int n = blockIdx.y;
    const unsigned char* img = src + (size_t)n * HW * 3;
    float* out = dst + (size_t)n * HW * 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = __fdiv_rn((float)img[(size_t)i * 3 + c], 255.0f);                       // to_tensor
            out[(size_t)c * HW + i] = __fdiv_rn(__fsub_rn(v, nrm.mean[c]), nrm.std[c]);            // sub_(mean).div_(std)
        }
    }
}

// ------------------------------------------------------------------------------------------------ NMS
__device__ inline bool iou_over(const float4 a, const float4 b, float thr) {
    const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z), top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    const float width = fmaxf(__fsub_rn(right, left), 0.f), height = fmaxf(__fsub_rn(bottom, top), 0.f);
    const float inter = __fmul_rn(width, height);
    const float sa = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y)), sb = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
    return __fdiv_rn(inter, __fsub_rn(__fadd_rn(sa, sb), inter)) > thr;
}

// grid (col_blocks, row_blocks), 64 threads: thread t of block (cb, rb) owns sorted box rb*64+t and tests it against the
// 64 boxes of column block cb (staged in LDS); only cb >= rb matters (a box can only suppress lower-ranked ones).
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes, const int* __restrict__ order,
                                                      int n, float thr, unsigned long long* __restrict__ mask,
                                                      int col_blocks) {
    const int cb = blockIdx.x, rb = blockIdx.y, t = threadIdx.x;
    if (cb < rb) return;
    __shared__ float4 cbox[64];
    const int cj = cb * 64 + t;
    if (cj < n) cbox[t] = boxes[order[cj]];
    __syncthreads();
    const int ri = rb * 64 + t;
    if (ri >= n) return;
    const float4 me = boxes[order[ri]];
    const int ncol = min(64, n - cb * 64);
    unsigned long long bits = 0ull;
    for (int j = (cb == rb ? t + 1 : 0); j < ncol; j++)
        if (iou_over(me, cbox[j], thr)) bits |= 1ull << j;
    mask[(size_t)ri * col_blocks + cb] = bits;
}

// one wavefront: walk the sorted boxes in order; lane l holds the "removed" words l, l+64, ...
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                      const int* __restrict__ order, int n, int col_blocks,
                                                      int* __restrict__ keep, int* __restrict__ num_keep, int max_keep) {
    constexpr int MAXW = 8;  // 64 lanes x 8 words x 64 bits = 32768 boxes
    unsigned long long remv[MAXW];
#pragma unroll
    for (int k = 0; k < MAXW; k++) remv[k] = 0ull;
    const int lane = threadIdx.x;
    int kept = 0;
    for (int i = 0; i < n; i++) {
        const int w = i >> 6, owner = w & 63, slot = w >> 6;
        unsigned long long word = 0ull;
#pragma unroll
        for (int k = 0; k < MAXW; k++) if (k == slot) word = remv[k];
        const unsigned lo = __shfl((unsigned)word, owner, 64), hi = __shfl((unsigned)(word >> 32), owner, 64);
        const unsigned long long rw = ((unsigned long long)hi << 32) | lo;
        if (!((rw >> (i & 63)) & 1ull)) {
            if (lane == 0 && kept < max_keep) keep[kept] = order[i];
            kept++;
            if (kept >= max_keep) break;  // yolov7: i = i[:max_det]
            const unsigned long long* row = mask + (size_t)i * col_blocks;
#pragma unroll
            for (int k = 0; k < MAXW; k++) {
                const int cw = lane + 64 * k;
                if (cw < col_blocks && cw >= w) remv[k] |= row[cw];
            }
        }
    }
    if (lane == 0) *num_keep = kept < max_keep ? kept : max_keep;
}

}  // namespace vlfm

using namespace vlfm;

// cv::computeResizeAreaTab for one axis.  h_first/h_count [dsize], h_w [dsize][ktaps].  Returns ktaps needed (>0) or <0.
extern "C" int vlfm_resize_area_tab_host(int ssize, int dsize, int32_t* h_first, int32_t* h_count, float* h_w,
                                         int ktaps_capacity) {
    if (ssize <= 0 || dsize <= 0 || dsize > ssize || !h_first || !h_count || !h_w)
        return fail(VLFM_ERR_INVALID, "resize_area_tab_host: bad argument (only shrinking or identity)");
    const double scale = (double)ssize / dsize;
    int need = 0;
    for (int dx = 0; dx < dsize; dx++) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)__builtin_ceil(fsx1), sx2 = (int)__builtin_floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        int k = 0, first = -1;
        float* w = h_w + (size_t)dx * ktaps_capacity;
        auto put = [&](int si, float a) {
            if (first < 0) first = si;
            if (k < ktaps_capacity) w[k] = a;
            k++;
        };
        if (sx1 - fsx1 > 1e-3) put(sx1 - 1, (float)((sx1 - fsx1) / cell));
        for (int sx = sx1; sx < sx2; sx++) put(sx, (float)(1.0 / cell));
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2;
            if (a > 1.0) a = 1.0;
            if (a > cell) a = cell;
            put(sx2, (float)(a / cell));
        }
        h_first[dx] = first < 0 ? 0 : first;
        h_count[dx] = k;
        if (k > need) need = k;
    }
    if (need > ktaps_capacity) return fail(VLFM_ERR_CAPACITY, "resize_area_tab_host: ktaps capacity");
    return need;
}

extern "C" int vlfm_resize_area_batched(const uint8_t* d_rgb, int n, int height, int width, int out_h, int out_w,
                                        const int32_t* d_xfirst, const int32_t* d_xcount, const float* d_xw, int xktaps,
                                        const int32_t* d_yfirst, const int32_t* d_ycount, const float* d_yw, int yktaps,
                                        void* d_out, int out_dtype, void* stream) {
    if (n == 0) return VLFM_OK;
    if (!d_rgb || !d_xfirst || !d_xcount || !d_xw || !d_yfirst || !d_ycount || !d_yw || !d_out || n < 0 || height <= 0 ||
        width <= 0 || out_h <= 0 || out_w <= 0 || (out_dtype != 0 && out_dtype != 1))
        return fail(VLFM_ERR_INVALID, "resize_area_batched: bad argument (out_dtype 0 = f32, 1 = f16)");
    AreaTab xt{d_xfirst, d_xcount, d_xw, xktaps}, yt{d_yfirst, d_ycount, d_yw, yktaps};
    VLFM_TIMED("resize_area_kernel", stream);
    if (out_dtype == 0)
        VLFM_KLAUNCH(resize_area_kernel<float>, dim3(out_h, n), dim3(256), 0, stream, d_rgb, height, width, out_h, out_w,
                     xt, yt, (float*)d_out);
    else
        VLFM_KLAUNCH(resize_area_kernel<__half>, dim3(out_h, n), dim3(256), 0, stream, d_rgb, height, width, out_h, out_w,
                     xt, yt, (__half*)d_out);
    return check_launch("resize_area_kernel");
}

extern "C" int vlfm_to_tensor_normalize_batched(const uint8_t* d_rgb, int n, int height, int width, const float* h_mean3,
                                                const float* h_std3, float* d_out, void* stream) {
    if (n == 0) return VLFM_OK;
    if (!d_rgb || !h_mean3 || !h_std3 || !d_out || n < 0 || height <= 0 || width <= 0)
        return fail(VLFM_ERR_INVALID, "to_tensor_normalize_batched: bad argument");
    Norm3f nrm;
    for (int c = 0; c < 3; c++) { nrm.mean[c] = h_mean3[c]; nrm.std[c] = h_std3[c]; }
    const int HW = height * width;
    int bx = (HW + 255) / 256;
    if (bx > 1024) bx = 1024;
    VLFM_TIMED("to_tensor_norm_kernel", stream);
    VLFM_KLAUNCH(to_tensor_norm_kernel, dim3(bx, n), dim3(256), 0, stream, d_rgb, HW, nrm, d_out);
    return check_launch("to_tensor_norm_kernel");
}

extern "C" size_t vlfm_nms_scratch_bytes(int n) {
    if (n <= 0) return 0;
    const size_t cb = ((size_t)n + 63) / 64;
    return (size_t)n * cb * sizeof(unsigned long long);
}

extern "C" int vlfm_nms(const float* d_boxes_xyxy, const int32_t* d_order, int n, float iou_threshold, void* d_scratch,
                        size_t scratch_bytes, int32_t* d_keep, int32_t* d_num_keep, int max_keep, void* stream) {
    if (!d_num_keep) return fail(VLFM_ERR_INVALID, "nms: d_num_keep is null");
    if (n == 0) return hipMemsetAsync(d_num_keep, 0, sizeof(int32_t), (hipStream_t)stream) == hipSuccess ? VLFM_OK : VLFM_ERR_HIP;
    if (!d_boxes_xyxy || !d_order || !d_scratch || !d_keep || n < 0 || n > 32768 || max_keep <= 0)
        return fail(VLFM_ERR_INVALID, "nms: bad argument (n <= 32768)");
    if (scratch_bytes < vlfm_nms_scratch_bytes(n)) return fail(VLFM_ERR_CAPACITY, "nms: scratch too small");
    const int cb = (n + 63) / 64;
    {
        VLFM_TIMED("nms_mask_kernel", stream);
        VLFM_KLAUNCH(nms_mask_kernel, dim3(cb, cb), dim3(64), 0, stream, reinterpret_cast<const float4*>(d_boxes_xyxy),
                     d_order, n, iou_threshold, reinterpret_cast<unsigned long long*>(d_scratch), cb);
    }
    VLFM_TIMED("nms_scan_kernel", stream);
    VLFM_KLAUNCH(nms_scan_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<const unsigned long long*>(d_scratch),
                 d_order, n, cb, d_keep, d_num_keep, max_keep);
    return check_launch("nms_scan_kernel");
}

// ------------------------------------------------------------------------------------------------ MsDeformAttn
// Multi-scale deformable attention sampling (GroundingDINO encoder/decoder; the reference's package ships a CUDA extension
// for it -- groundingdino/models/GroundingDINO/csrc/MsDeformAttn [ext] -- with a grid_sample fallback).
//   value   [B][sum(H_l W_l)][heads][D] f32        loc [B][Q][heads][L][P][2] in [0,1]       w [B][Q][heads][L][P]
//   out     [B][Q][heads * D]:  sum_{l,p} w * bilinear(value_l, loc)   (grid_sample: align_corners = False, zero padding)
// One lane per output channel: the D channels of a (location, head) are contiguous, so every tap is one coalesced
// D*4-byte read; locations and weights are wave-broadcast loads.  Gather-bound; no MFMA.
namespace vlfm {
__global__ __launch_bounds__(256) void ms_deform_attn_kernel(const float* __restrict__ value, const int* __restrict__ shapes,
                                                             const int* __restrict__ level_start, const float* __restrict__ loc,
                                                             const float* __restrict__ weight, int B, int Q, int heads, int D,
                                                             int L, int P, int total, float* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n_out = (long long)B * Q * heads * D;
    if (gid >= n_out) return;
    const int d = (int)(gid % D);
    const int h = (int)((gid / D) % heads);
    const int q = (int)((gid / ((long long)D * heads)) % Q);
    const int b = (int)(gid / ((long long)D * heads * Q));
    const float* lq = loc + ((((size_t)b * Q + q) * heads + h) * L) * P * 2;
    const float* wq = weight + ((((size_t)b * Q + q) * heads + h) * L) * P;
    const float* vb = value + (size_t)b * total * heads * D + (size_t)h * D + d;
    float acc = 0.0f;
    for (int l = 0; l < L; l++) {
        const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
        const float* vl = vb + (size_t)level_start[l] * heads * D;
        for (int p = 0; p < P; p++) {
            const float x = lq[(l * P + p) * 2] * (float)Wl - 0.5f, y = lq[(l * P + p) * 2 + 1] * (float)Hl - 0.5f;
            const float xf = floorf(x), yf = floorf(y);
            const int x0 = (int)xf, y0 = (int)yf;
            const float tx = x - xf, ty = y - yf;
            float s = 0.0f;
            const bool xin0 = (unsigned)x0 < (unsigned)Wl, xin1 = (unsigned)(x0 + 1) < (unsigned)Wl;
            const bool yin0 = (unsigned)y0 < (unsigned)Hl, yin1 = (unsigned)(y0 + 1) < (unsigned)Hl;
            if (yin0 && xin0) s += vl[((size_t)y0 * Wl + x0) * heads * D] * (1.f - tx) * (1.f - ty);
            if (yin0 && xin1) s += vl[((size_t)y0 * Wl + x0 + 1) * heads * D] * tx * (1.f - ty);
            if (yin1 && xin0) s += vl[((size_t)(y0 + 1) * Wl + x0) * heads * D] * (1.f - tx) * ty;
            if (yin1 && xin1) s += vl[((size_t)(y0 + 1) * Wl + x0 + 1) * heads * D] * tx * ty;
            acc += s * wq[l * P + p];
        }
    }
    out[gid] = acc;
}

// ------------------------------------------------------------------------------------------------ depthwise 3x3 convolution
// MobileSAM's TinyViT (behind vlfm/vlm/sam.py:54, SamPredictor.set_image [ext mobile_sam]) is full of depthwise 3x3 convolutions
// in f32 -- MBConv's conv2 on a [B, 256, 256, 256] activation (2.1 GB at 32 frames), the local_conv of every attention block,
// the patch-merging convs -- and MIOpen serves them with its naive fallback (2.2 TFLOP/s: 4.4 ms for the big one, 13 ms of the
// 128-env full step, tools/conv_probe.py).  The op is memory-bound (18 flops per 8 bytes): one thread per 4 consecutive outputs,
// 16-byte loads and stores, the three input rows of a quad come from L1 when the neighbouring row lanes fetched them, bias (the
// folded BatchNorm) and the exact GELU that follows conv2 applied in registers.  NCHW, padding 1, stride 1 or 2.
template <int STRIDE, bool GELU>
__global__ __launch_bounds__(256) void dwconv3x3_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int C, int H,
                                                            int W, int HO, int WO, int rows_per_block) {
    const int Q = WO >> 2;                       // output quads per row
    const int tq = threadIdx.x % Q, ty = threadIdx.x / Q, RL = blockDim.x / Q;
    if (ty >= RL) return;
    const size_t plane = blockIdx.y;             // n * C + c
    const int c = (int)(plane % C);
    const float* xp = x + plane * H * W;
    float* yp = y + plane * HO * WO;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; i++) k[i] = w[c * 9 + i];
    const float b = bias ? bias[c] : 0.0f;
    const int y_end = min(HO, (int)(blockIdx.x + 1) * rows_per_block);
    for (int yo = blockIdx.x * rows_per_block + ty; yo < y_end; yo += RL) {
        float4 acc = make_float4(b, b, b, b);
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            const int yi = yo * STRIDE + dy - 1;
            if ((unsigned)yi >= (unsigned)H) continue;
            const float* row = xp + (size_t)yi * W;
            const float k0 = k[3 * dy], k1 = k[3 * dy + 1], k2 = k[3 * dy + 2];
            if (STRIDE == 1) {
                const int x0 = 4 * tq;
                const float4 v = *reinterpret_cast<const float4*>(row + x0);
                const float l = x0 > 0 ? row[x0 - 1] : 0.0f, r = x0 + 4 < W ? row[x0 + 4] : 0.0f;
                acc.x += k0 * l + k1 * v.x + k2 * v.y;
                acc.y += k0 * v.x + k1 * v.y + k2 * v.z;
                acc.z += k0 * v.y + k1 * v.z + k2 * v.w;
                acc.w += k0 * v.z + k1 * v.w + k2 * r;
            } else {
                const int x0 = 8 * tq;
                const float4 v = *reinterpret_cast<const float4*>(row + x0);
                const float4 u = *reinterpret_cast<const float4*>(row + x0 + 4);
                const float l = x0 > 0 ? row[x0 - 1] : 0.0f;
                acc.x += k0 * l + k1 * v.x + k2 * v.y;
                acc.y += k0 * v.y + k1 * v.z + k2 * v.w;
                acc.z += k0 * v.w + k1 * u.x + k2 * u.y;
                acc.w += k0 * u.y + k1 * u.z + k2 * u.w;
            }
        }
        if (GELU) {
            acc.x = 0.5f * acc.x * (1.0f + erff(acc.x * 0.70710678118654752440f));
            acc.y = 0.5f * acc.y * (1.0f + erff(acc.y * 0.70710678118654752440f));
            acc.z = 0.5f * acc.z * (1.0f + erff(acc.z * 0.70710678118654752440f));
            acc.w = 0.5f * acc.w * (1.0f + erff(acc.w * 0.70710678118654752440f));
        }
        *reinterpret_cast<float4*>(yp + (size_t)yo * WO + 4 * tq) = acc;
    }
}


// ------------------------------------------------------------------------------------------------ bias + activation, in place
// y[n][c][:] = act(y[n][c][:] + bias[c]) on an NCHW tensor: what remains of a conv + BatchNorm + activation triple once the
// BatchNorm is folded into the weights.  The framework runs the bias add and the activation as two passes over the
// activation (MIOpen convolutions do not take a bias); this is one, with 16-byte accesses.  act: 0 none, 1 exact GELU, 2 SiLU.
template <typename T> struct Vec16;
template <> struct Vec16<float> { static constexpr int N = 4; };
template <> struct Vec16<__half> { static constexpr int N = 8; };

template <int ACT> __device__ inline float apply_act(float v) {
    if (ACT == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (ACT == 2) return v / (1.0f + __expf(-v));
    return v;
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void bias_act_nchw_kernel(T* __restrict__ y, const T* __restrict__ bias, int C, long long HW) {
    constexpr int N = Vec16<T>::N;
    const size_t plane = blockIdx.y;
    const float b = (float)bias[plane % C];
    T* p = y + plane * HW;
    const long long nvec = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? HW / N : 0;   // aligned planes only (else scalar)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        uint4 raw = reinterpret_cast<uint4*>(p)[i];
        T* e = reinterpret_cast<T*>(&raw);
#pragma unroll
        for (int k = 0; k < N; k++) e[k] = (T)apply_act<ACT>((float)e[k] + b);
        reinterpret_cast<uint4*>(p)[i] = raw;
    }
    for (long long i = nvec * N + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x)
        p[i] = (T)apply_act<ACT>((float)p[i] + b);
}


// The same sampling with everything in front of it fused in (round 4): the softmax over a head's L * P logits and the sampling
// locations (GroundingDinoMultiscaleDeformableAttention.forward [ext]: reference point + offset / (W_l, H_l), or for 4-coordinate
// reference boxes + offset / P * box size * 0.5) are evaluated in the kernel from the RAW output of the fused offsets|logits GEMM, so
// the six elementwise passes over [B, Q, heads, L, P, 2] tensors (0.4 GB each at 64 frames) and the softmax launch disappear.  A
// wavefront owns one query: lane = head * 8 + channel group, 4 channels (16 bytes) per lane -- the location arithmetic runs once per 8
// lanes instead of once per channel, every gather is a dwordx4 and the 8 lanes of a head read one 128-byte segment, the result row
// (heads * 32 floats = 1 KB) is written by one wavefront.  heads == 8, D == 32 (the shipped geometry; others use the plain kernel).
//   ow   [B][Q][heads*L*P*2 + heads*L*P]: offsets [h][l][p][2], then logits [h][l][p]
//   ref  [B][Q][L][C], C = 2 or 4
__global__ __launch_bounds__(256) void ms_deform_attn_fused_kernel(const float* __restrict__ value, const int* __restrict__ shapes,
                                                                   const int* __restrict__ level_start, const float* __restrict__ ow,
                                                                   const float* __restrict__ ref, int B, int Q, int L, int P, int C,
                                                                   int total, float* __restrict__ out) {
    constexpr int HEADS = 8, D = 32;
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);     // (b, q)
    if (item >= (long long)B * Q) return;
    const int b = (int)(item / Q);
    const int h = lane >> 3, c4 = lane & 7;
    const int LP = L * P;
    const float* row = ow + (size_t)item * (HEADS * LP * 3);
    const float* off = row + (size_t)h * LP * 2;
    const float* lg = row + (size_t)HEADS * LP * 2 + (size_t)h * LP;
    const float* rq = ref + (size_t)item * L * C;
    // softmax over the head's L * P logits
    float mx = -INFINITY;
    for (int i = 0; i < LP; i++) mx = fmaxf(mx, lg[i]);
    float den = 0.f;
    for (int i = 0; i < LP; i++) den += expf(lg[i] - mx);
    const float inv = 1.0f / den;
    const float* vb = value + (size_t)b * total * HEADS * D + (size_t)h * D + c4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < L; l++) {
        const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
        const float* vl = vb + (size_t)level_start[l] * HEADS * D;
        const float rx = rq[l * C], ry = rq[l * C + 1];
        float sx, sy;
        if (C == 2) { sx = 1.0f / (float)Wl; sy = 1.0f / (float)Hl; }
        else { sx = rq[l * C + 2]; sy = rq[l * C + 3]; }
        for (int p = 0; p < P; p++) {
            const float ox = off[(l * P + p) * 2], oy = off[(l * P + p) * 2 + 1];
            float lx, ly;
            if (C == 2) { lx = rx + ox / (float)Wl; ly = ry + oy / (float)Hl; }        // reference + offset / (W, H)
            else { lx = rx + ox / (float)P * sx * 0.5f; ly = ry + oy / (float)P * sy * 0.5f; }
            const float w = expf(lg[l * P + p] - mx) * inv;
            const float x = lx * (float)Wl - 0.5f, y = ly * (float)Hl - 0.5f;
            const float xf = floorf(x), yf = floorf(y);
            const int x0 = (int)xf, y0 = (int)yf;
            const float tx = x - xf, ty = y - yf;
            const bool xin0 = (unsigned)x0 < (unsigned)Wl, xin1 = (unsigned)(x0 + 1) < (unsigned)Wl;
            const bool yin0 = (unsigned)y0 < (unsigned)Hl, yin1 = (unsigned)(y0 + 1) < (unsigned)Hl;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            auto tap = [&](int yy, int xx, float k) {
                const float4 v = *reinterpret_cast<const float4*>(vl + ((size_t)yy * Wl + xx) * HEADS * D);
                s.x += v.x * k; s.y += v.y * k; s.z += v.z * k; s.w += v.w * k;
            };
            if (yin0 && xin0) tap(y0, x0, (1.f - tx) * (1.f - ty));
            if (yin0 && xin1) tap(y0, x0 + 1, tx * (1.f - ty));
            if (yin1 && xin0) tap(y0 + 1, x0, (1.f - tx) * ty);
            if (yin1 && xin1) tap(y0 + 1, x0 + 1, tx * ty);
            acc.x += s.x * w; acc.y += s.y * w; acc.z += s.z * w; acc.w += s.w * w;
        }
    }
    *reinterpret_cast<float4*>(out + (size_t)item * HEADS * D + h * D + c4 * 4) = acc;
}

}  // namespace vlfm

extern "C" int vlfm_ms_deform_attn_fused(const float* d_value, const int32_t* d_spatial_shapes, const int32_t* d_level_start,
                                         const float* d_offsets_logits, const float* d_reference, int batch, int n_query, int n_heads,
                                         int head_dim, int n_levels, int n_points, int ref_coords, int total_len, float* d_out,
                                         void* stream) {
    if (batch == 0 || n_query == 0) return VLFM_OK;
    if (!d_value || !d_spatial_shapes || !d_level_start || !d_offsets_logits || !d_reference || !d_out || batch < 0 || n_query < 0 ||
        n_heads != 8 || head_dim != 32 || n_levels <= 0 || n_points <= 0 || total_len <= 0 || (ref_coords != 2 && ref_coords != 4))
        return fail(VLFM_ERR_INVALID, "ms_deform_attn_fused: 8 heads of width 32, reference points with 2 or 4 coordinates");
    const long long items = (long long)batch * n_query;
    VLFM_TIMED("ms_deform_attn_fused_kernel", stream);
    VLFM_KLAUNCH(vlfm::ms_deform_attn_fused_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, d_value, d_spatial_shapes,
                 d_level_start, d_offsets_logits, d_reference, batch, n_query, n_levels, n_points, ref_coords, total_len, d_out);
    return check_launch("ms_deform_attn_fused_kernel");
}

extern "C" int vlfm_ms_deform_attn(const float* d_value, const int32_t* d_spatial_shapes, const int32_t* d_level_start,
                                   const float* d_sampling_loc, const float* d_attn_weight, int batch, int n_query,
                                   int n_heads, int head_dim, int n_levels, int n_points, int total_len, float* d_out,
                                   void* stream) {
    if (batch == 0 || n_query == 0) return VLFM_OK;
    if (!d_value || !d_spatial_shapes || !d_level_start || !d_sampling_loc || !d_attn_weight || !d_out || batch < 0 ||
        n_query < 0 || n_heads <= 0 || head_dim <= 0 || n_levels <= 0 || n_points <= 0 || total_len <= 0)
        return fail(VLFM_ERR_INVALID, "ms_deform_attn: bad argument");
    const long long n_out = (long long)batch * n_query * n_heads * head_dim;
    VLFM_TIMED("ms_deform_attn_kernel", stream);
    VLFM_KLAUNCH(vlfm::ms_deform_attn_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, stream, d_value,
                 d_spatial_shapes, d_level_start, d_sampling_loc, d_attn_weight, batch, n_query, n_heads, head_dim, n_levels,
                 n_points, total_len, d_out);
    return check_launch("ms_deform_attn_kernel");
}

extern "C" int vlfm_dwconv3x3_f32(const float* d_x, const float* d_w, const float* d_bias, float* d_y, int n, int channels,
                                  int height, int width, int stride, int gelu, void* stream) {
    if (n == 0) return VLFM_OK;
    const int ho = (height + 2 - 3) / (stride > 0 ? stride : 1) + 1, wo = (width + 2 - 3) / (stride > 0 ? stride : 1) + 1;
    if (!d_x || !d_w || !d_y || n < 0 || channels <= 0 || height <= 0 || width <= 0 || (stride != 1 && stride != 2) ||
        (width & 3) || (wo & 3) || wo > 1024 || (stride == 2 && width != 2 * wo))
        return fail(VLFM_ERR_INVALID, "dwconv3x3_f32: stride 1 or 2, width and output width multiples of 4 (<= 1024 outputs per row)");
    const int q = wo / 4, rl = 256 / q > 0 ? 256 / q : 1;
    const int threads = q * rl;
    int rows_per_block = rl * 8;
    if (rows_per_block > ho) rows_per_block = (ho + rl - 1) / rl * rl;
    const dim3 grid((ho + rows_per_block - 1) / rows_per_block, (unsigned)n * channels), block(threads);
    if (grid.y > 65535u * 16u) return fail(VLFM_ERR_CAPACITY, "dwconv3x3_f32: too many planes");
    VLFM_TIMED("dwconv3x3_f32_kernel", stream);
    hipStream_t s = (hipStream_t)stream;
    if (stride == 1) {
        if (gelu) VLFM_KLAUNCH((vlfm::dwconv3x3_f32_kernel<1, true>), grid, block, 0, s, d_x, d_w, d_bias, d_y, channels, height, width, ho, wo, rows_per_block);
        else VLFM_KLAUNCH((vlfm::dwconv3x3_f32_kernel<1, false>), grid, block, 0, s, d_x, d_w, d_bias, d_y, channels, height, width, ho, wo, rows_per_block);
    } else {
        if (gelu) VLFM_KLAUNCH((vlfm::dwconv3x3_f32_kernel<2, true>), grid, block, 0, s, d_x, d_w, d_bias, d_y, channels, height, width, ho, wo, rows_per_block);
        else VLFM_KLAUNCH((vlfm::dwconv3x3_f32_kernel<2, false>), grid, block, 0, s, d_x, d_w, d_bias, d_y, channels, height, width, ho, wo, rows_per_block);
    }
    return check_launch("dwconv3x3_f32_kernel");
}

extern "C" int vlfm_bias_act_nchw(void* d_y, const void* d_bias, int n, int channels, long long hw, int dtype, int act,
                                  void* stream) {
    if (n == 0 || hw == 0) return VLFM_OK;
    if (!d_y || !d_bias || n < 0 || channels <= 0 || hw < 0 || (dtype != 0 && dtype != 1) || act < 0 || act > 2 ||
        (long long)n * channels > 65535LL * 16)
        return fail(VLFM_ERR_INVALID, "bias_act_nchw: dtype 0 (f32) / 1 (f16), act 0 (none) / 1 (GELU) / 2 (SiLU)");
    const int per = dtype == 0 ? 4 : 8;
    long long blocks = (hw / per + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;
    const dim3 grid((unsigned)blocks, (unsigned)(n * channels)), block(256);
    hipStream_t s = (hipStream_t)stream;
    VLFM_TIMED("bias_act_nchw_kernel", stream);
#define VLFM_BA(T, A) VLFM_KLAUNCH((vlfm::bias_act_nchw_kernel<T, A>), grid, block, 0, s, (T*)d_y, (const T*)d_bias, channels, hw)
    if (dtype == 0) { if (act == 0) VLFM_BA(float, 0); else if (act == 1) VLFM_BA(float, 1); else VLFM_BA(float, 2); }
    else { if (act == 0) VLFM_BA(__half, 0); else if (act == 1) VLFM_BA(__half, 1); else VLFM_BA(__half, 2); }
#undef VLFM_BA
    return check_launch("bias_act_nchw_kernel");
}
