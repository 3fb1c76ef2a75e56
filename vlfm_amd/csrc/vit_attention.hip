// vit_attention.hip -- self-attention of the BLIP-2 ViT-g blocks (S = 257 tokens, 16 heads of 88), gfx950.
//
// Replaces F.scaled_dot_product_attention in vlfm_amd/vlm/blip2itm.py:_VitBlock (the reference reaches the same maths
// through LAVIS' eva_vit Attention, vlfm/vlm/blip2itm.py:29-34,52 [ext]).  The library flash kernel needs the heads padded to 96
// and a transpose copy on this shape (257 = 8 x 32 + 1 tokens, head 88): 1 050 us per block at 256 images; this kernel is
// specialised for the shape and reads / writes the GEMMs' own layouts (qkv [B, S, 3, H, 88] in, [B, S, H, 88] out):
//
//   * a PERSISTENT kernel (round 6): one workgroup of 8 wavefronts per CU walks its (image, head) items; K, V and Q of the next
//     item travel by LDS-DMA while the current one is in the matrix pipe, output rows leave through LDS as whole rows;
//   * the head's whole K and V are resident in LDS for the item, so there is no online-softmax rescaling;
//   * wavefront w owns the 32 queries of tokens 1 + 32 w .. 32 + 32 w.  It computes S^T = K Q^T with v_mfma_f32_32x32x16_f16
//     (A = K rows from LDS, B = Q^T held in registers): in the 32 x 32 accumulator layout every lane then holds 16 keys of ONE
//     query column per key tile, so the softmax statistics are lane-local apart from one exchange with lane ^ 32, and the
//     probabilities are ALREADY in B-operand order for O^T = V^T P^T -- no cross-lane movement between the two GEMMs;
//   * V stays row-major in LDS (as the DMA delivers it); the A operand of the PV MFMAs comes out of it through
//     ds_read_b64_tr_b16, the hardware transpose read (semantics pinned by tools/native/tr_read_probe.hip);
//   * the odd token (the CLS query) runs on the VALU (v_dot2_f32_f16 scores against the wavefront's 32 keys, DPP reductions,
//     partial (max, sum, O) per wavefront merged by wavefront 0 behind the next barrier) instead of a ninth query tile of MFMAs;
//   * the MFMA-friendly head width 96 exists only as zeros in the Q registers; global memory and LDS hold the native 88 channels.
//
// 4 x 257^2 x 88 flop and 180 KB of traffic per (image, head): 95 GFLOP and 741 MB at 256 images.  What bounds it is the memory
// side: the rows of a head are 176-byte pieces at a stride of 8 448 B (in) / 2 816 B (out).  With everything but the memory traffic
// compiled out the kernel runs 190 us (220 with no waits or barriers at all: it is the issue rate of this access pattern, 556 MB of
// loads at 4.5 TB/s + 185 MB of stores), with the traffic compiled out 105 us, the product 231 us -- round 5's kernel: 299.  Reading
// qkv head-major would stream the loads at 5.9 TB/s (memory side 139 us with a head-major output too), but the GEMMs on either side
// are the library's and write / read token-major (profiles/r06_vit_attention_probes.txt).
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

constexpr int AT_D = 96;                  // head width as the MFMAs see it (88 real channels + zeros in the Q registers)
constexpr int AT_NT = 8;                  // full 32-token tiles
constexpr int AT_S = 32 * AT_NT + 1;      // 257 tokens
constexpr int AT_KT = AT_NT + 1;          // key tiles incl. the one-token tail
constexpr int AT_WAVES = 8;

// One workgroup per CU walks its (image, head) items.  Every byte enters and leaves through LDS in
// whole 176-byte rows: K, V and Q arrive by LDS-DMA (global_load_lds, no registers) while the previous item is in the matrix pipe,
// the output rows are staged in LDS and leave as 16 bytes per lane.  (The first persistent form loaded the Q fragments and stored the
// output straight in MFMA operand layout: 32 cache lines per instruction -- those 12 instructions per wavefront took longer to
// issue than the 14 DMA pieces that move twice the bytes, and the memory side alone ran 211 us at 256 images; 139 with the stores
// staged, profiles/r06_vit_attention_probes.txt.)  The odd (CLS) query runs on the VALU instead of a ninth MFMA tile; V stays
// row-major in LDS and is transposed by ds_read_b64_tr_b16 on the way to the PV MFMAs.
// LDS (bytes):  buffer 0 | buffer 1 | V | partials of the odd query x 2 | its probabilities.  Item n (kb = n & 1):
//   buffer kb      K_n (257 rows x 176 B, the native 88 channels, no padding; a row stride of 11 sixteen-byte slots is odd, so the 16
//                  lanes of a ds_read_b128 group hit 16 different slots) -> behind barrier B: Q_{n+1} (same image; row 0 is the odd query)
//   buffer 1 - kb  Q_n -> read into registers behind barrier A, then the wavefront's rows are overwritten with O_{n-1} (f16) and
//                  leave during the softmax -> behind barrier B: K_{n+1}
//   V              288 rows x 192 B.  Channels 88..95 and rows 257.. hold finite junk (duplicates written by run-over lanes): junk
//                  channels only reach rows 88..95 of O^T, which are never stored; junk rows meet P = 0.
// The sixth contraction step of QK^T reads 16 B past a K row (channels 88..95 = the next row's first 8): they meet zeros in the Q
// registers, and LDS is zeroed once at kernel start so that such bytes are always finite.  Rows 257..287 of the ninth key tile
// are whatever follows the buffer: their scores are replaced by -inf (a select, not arithmetic).
// DMA pieces: one wavefront instruction moves 64 x 16 B to 1 KB of consecutive LDS.  K = 2 827 slots as 48 pieces at a pitch of 59,
// V = 3 084 slots (12 per row, the 12th a duplicate of the 11th) as 56 pieces at a pitch of 56, Q = the wavefront's own 33 rows as 6
// pieces: every wavefront issues exactly 7 (V, between the MFMAs of QK^T) + 6 + 6 (K, Q of the next item, between the MFMAs of PV)
// pieces per item; overlapping lanes rewrite identical bytes.  vmcnt is waited by COUNT (gfx9 retires vector memory operations in
// issue order): the six row stores of the previous item stay in flight across barrier B.
constexpr int PA_KROW = 176, PA_VROW = 192;
constexpr int PA_KB = 45440;                          // one K / Q / O buffer incl. the run-over of the last DMA piece
constexpr int PA_V_OFF = 2 * PA_KB;                   // 90 880
constexpr int PA_V_BYTES = 288 * PA_VROW;             // 55 296
constexpr int PA_PART_OFF = PA_V_OFF + PA_V_BYTES;    // 146 176
constexpr int PA_PART = 100;                          // floats per wavefront: O (96) + max + sum + p of key 256; two sets (item parity)
constexpr int PA_PS_OFF = PA_PART_OFF + 2 * AT_WAVES * PA_PART * 4;   // 152 576: the odd query's 256 probabilities (f32)
constexpr int PA_LDS = PA_PS_OFF + AT_WAVES * 32 * 4;                 // 153 600
constexpr int PA_KPIECES = 6, PA_VPIECES = 7, PA_QPIECES = 6, PA_KPITCH = 59, PA_VPITCH = 56;
constexpr int PA_KSLOTS = AT_S * 11, PA_VSLOTS = AT_S * 12;

using lds_ptr = __attribute__((address_space(3))) unsigned char*;
using gbl_ptr = const __attribute__((address_space(1))) unsigned char*;
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef short short8v __attribute__((__vector_size__(8 * sizeof(short))));
using half2_t = __attribute__((ext_vector_type(2))) _Float16;

#ifdef PA_FREE_RUN     // timing experiment only (with PA_STUB=4): no waits, no barriers inside the item loop -- the memory side's issue rate alone
#define PA_WAIT_VM(n) do {} while (0)
#else
#define PA_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))   /* vmcnt(n), n < 64 */
#endif
#define PA_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#ifdef PA_FREE_RUN
#define PA_BARRIER() asm volatile("" ::: "memory")
#else
#define PA_BARRIER()                               \
    do {                                           \
        asm volatile("" ::: "memory");             \
        __builtin_amdgcn_s_barrier();              \
        asm volatile("" ::: "memory");             \
    } while (0)
#endif

// reductions over the 16 lanes of a DPP row (quad swaps, then the two mirrors): every lane ends with the row's result
#define PA_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true))
__device__ inline float row16_max(float x) {
    x = fmaxf(x, PA_DPP(x, 0xB1));     // quad_perm [1,0,3,2]
    x = fmaxf(x, PA_DPP(x, 0x4E));     // quad_perm [2,3,0,1]
    x = fmaxf(x, PA_DPP(x, 0x141));    // row_half_mirror
    return fmaxf(x, PA_DPP(x, 0x140)); // row_mirror
}
__device__ inline float row16_sum(float x) {
    x += PA_DPP(x, 0xB1);
    x += PA_DPP(x, 0x4E);
    x += PA_DPP(x, 0x141);
    return x + PA_DPP(x, 0x140);
}

__device__ inline half8_t tr_pair(const unsigned char* p0, const unsigned char* p1) {
    const short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds_ptr)p0);
    const short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds_ptr)p1);
    const short8v ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(half8_t, ab);
}

#ifndef PA_NO_STORES
#define PA_NO_STORES 0   // diagnostic: the product without the row stores
#endif
#ifndef PA_STUB
#define PA_STUB 0      // diagnostic builds (tools/vit_attn_stub_probe.py): 1 no MFMAs, 2 + no softmax, 3 + no odd-query passes, 4 + no sleeps
#endif                 // (memory traffic and barriers only), 5 + no stores; 6 = the product without any vector memory operation
// Optional stamps (diagnostic build -DVLFM_PHASE_TIMING, tools/vit_attn_phase_probe.py): wavefronts 0 and 4 of workgroup 0 record the
// shader clock at the phase boundaries of every item they walk.
#ifdef VLFM_PHASE_TIMING
__device__ long long g_pa_clk[2][20][12];
#define PA_STAMP(k)                                                                                              \
    do {                                                                                                         \
        if (blockIdx.x == 0 && (tid & 255) == 0 && n < 20) g_pa_clk[tid >> 8][n][k] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define PA_STAMP(k) do {} while (0)
#endif

__global__ __launch_bounds__(64 * AT_WAVES, 2) void vit_attention_kernel(const _Float16* __restrict__ qkv,
                                                                              _Float16* __restrict__ out, int B, int H,
                                                                              float scale) {
    constexpr int DH = 88;
    extern __shared__ __attribute__((aligned(16))) unsigned char at_lds[];
    const lds_ptr lds = (lds_ptr)at_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), col = lane & 31, grp = lane >> 5;
    // the 16 heads of an image go to ONE XCD back to back (their K / V rows share cache lines): workgroup id -> (XCD, lane j of
    // the XCD); the XCD's items are (image x + 8 i, head) in head-fastest order, lane j takes items j, j + per, ...
    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    int t = blockIdx.x >> 3;
    const int nimg = B > xcd ? (B - xcd + 7) >> 3 : 0;      // images of this XCD: xcd, xcd + 8, ...
    if (t / H >= nimg) return;
    auto image_of = [&](int tt) { return xcd + 8 * (tt / H); };
#ifdef PA_HEAD_MAJOR   // timing experiment only: qkv read as [B][H][3][S][88] (every K / V / Q block contiguous)
    const int row_halfs = 3 * H * DH, row_bytes = 2 * DH, part_bytes = AT_S * 2 * DH;
#else
    const int row_halfs = 3 * H * DH, row_bytes = 2 * row_halfs, part_bytes = 2 * H * DH;
#endif
    // ---- zero the whole LDS once (over-read bytes must be finite)
    for (int i = tid; i < PA_LDS / 16; i += 64 * AT_WAVES) *reinterpret_cast<uint4*>(at_lds + 16 * i) = uint4{0, 0, 0, 0};
    PA_WAIT_LGKM0();
    PA_BARRIER();
    auto item_base = [&](int tt) {
        const int b = image_of(tt), h = tt % H;
#ifdef PA_HEAD_MAJOR
        return reinterpret_cast<const unsigned char*>(qkv) + (size_t)(b * H + h) * 3 * AT_S * DH * 2;
#else
        return reinterpret_cast<const unsigned char*>(qkv) + ((size_t)b * AT_S * row_halfs + (size_t)h * DH) * 2;
#endif
    };
    // LDS-DMA pieces go through inline asm: hipcc books a global_load_lds builtin as a FLAT operation that touches both memories and,
    // while one is pending, turns every LDS or vector-memory dependency into a wait for ZERO (the operand prefetch of the MFMA loops
    // would wait for the pieces just issued).  Hidden from its scoreboard, the pieces are waited by count below; M0 (the
    // destination) is saved and restored around each piece.  The lane index is made opaque per call, so the per-lane source
    // offsets are recomputed (a few VALU per piece) instead of being hoisted into 19 registers the main pass has no room for.
    // A vector-memory instruction blocks its wavefront at ISSUE while the CU's queue is full -- 104 KB issued back to back cost 6 000
    // cycles during which nothing else ran -- so the pieces are spread between the MFMAs at about the rate HBM drains them.
    auto dma16 = [&](const unsigned char* sbase, uint32_t voff, int dst) {
        if (PA_STUB == 6) return;
        unsigned keep;
        const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds + dst));
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(sbase) : "memory");
    };
    auto k_piece = [&](const unsigned char* base, int bo, int j) {
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const int p = wave + AT_WAVES * j;
        const int sg = min(PA_KPITCH * p + lane, PA_KSLOTS - 1), row = sg / 11, s = sg - 11 * row;
        dma16(base + part_bytes, (uint32_t)(row * row_bytes + 16 * s), bo + 16 * PA_KPITCH * p);
    };
    auto v_piece = [&](const unsigned char* base, int j) {
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const int p = wave + AT_WAVES * j;
        const int sg = PA_VPITCH * p + lane, row = min(sg / 12, AT_S - 1), s = min(sg % 12, 10);
        dma16(base + 2 * part_bytes, (uint32_t)(row * row_bytes + 16 * s), PA_V_OFF + 16 * PA_VPITCH * p);
    };
    // Q: rows 32 wave .. 32 wave + 33 of the [257][88] image (the wavefront's 32 queries are rows 1 + 32 wave ..; wavefront 0 brings the
    // odd query's row 0 with it); the run-over lanes fetch the neighbour's rows (identical bytes), none past the image
    auto q_piece = [&](const unsigned char* base, int bo, int j) {
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const int sg = 32 * 11 * wave + 64 * j + lane, row = sg / 11, s = sg - 11 * row;
        if (sg < PA_KSLOTS) dma16(base, (uint32_t)(row * row_bytes + 16 * s), bo + 16 * (32 * 11 * wave + 64 * j));
    };
    // store-out of the staged rows of item tt (buffer offset bo): instruction i = slots 64 i .. of the wavefront's 352 (32 rows x 11)
    auto store_out = [&](int tt, int bo, int i) {
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const int sg = 64 * i + lane, row = sg / 11, s = sg - 11 * row;
        if (PA_STUB < 5 && !PA_NO_STORES && sg < 352) {
            const int b = image_of(tt), h = tt % H;
            const uint4 v = *reinterpret_cast<const uint4*>(at_lds + bo + (1 + 32 * wave) * PA_KROW + 16 * sg);
#ifdef PA_CONTIG_OUT   // timing experiment only: head-major output (each (image, head) block of 257 rows contiguous)
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(out) + ((size_t)(b * H + h) * AT_S + 1 + 32 * wave) * (2 * DH) + 16 * sg) = v;
#else
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(out) + ((size_t)(b * AT_S + 1 + 32 * wave + row) * H + h) * (2 * DH) + 16 * s) = v;
#endif
        }
    };
    const unsigned char* base = item_base(t);
#pragma unroll
    for (int j = 0; j < PA_KPIECES; j++) k_piece(base, 0, j);
#pragma unroll
    for (int j = 0; j < PA_QPIECES; j++) q_piece(base, PA_KB, j);
    const float c = scale * 1.4426950408889634f;
    // merge of the odd query's partials (wavefront 0): written during an item (set = item parity), read behind barrier A of the next
    auto merge_odd = [&](int tt, int set) {
        if (wave == 0) {
            const float* part = reinterpret_cast<const float*>(at_lds + PA_PART_OFF) + set * AT_WAVES * PA_PART;
            float mx = -__builtin_huge_valf();
#pragma unroll
            for (int w = 0; w < AT_WAVES; w++) mx = fmaxf(mx, part[w * PA_PART + 96]);
            float o0 = 0.0f, o1 = 0.0f, lsum = 0.0f;
            const int cp = min(lane, 47);
#pragma unroll
            for (int w = 0; w < AT_WAVES; w++) {
                const float wgt = __builtin_amdgcn_exp2f((part[w * PA_PART + 96] - mx) * c);
                lsum = fmaf(wgt, part[w * PA_PART + 97], lsum);
                const float2 ov = *reinterpret_cast<const float2*>(part + w * PA_PART + 2 * cp);
                o0 = fmaf(wgt, ov.x, o0);
                o1 = fmaf(wgt, ov.y, o1);
            }
            const float inv = 1.0f / lsum;
            const int b = image_of(tt), h = tt % H;
            if (PA_STUB < 5 && lane < DH / 2)
                *reinterpret_cast<half2_t*>(out + ((size_t)(b * AT_S) * H + h) * DH + 2 * lane) = half2_t{(_Float16)(o0 * inv), (_Float16)(o1 * inv)};
        }
    };
    half4_t of[11];          // the wavefront's 32 output rows of the previous item (f16, accumulator layout) until they are staged
    // rows of the accumulator layout -> LDS: lane (col, grp) holds channels 32 dt + 8 q4 + 4 grp .. + 3 of row col
    auto stage_rows = [&](int bo) {
        unsigned char* st = at_lds + bo + (1 + 32 * wave + col) * PA_KROW + 8 * grp;
#pragma unroll
        for (int i = 0; i < 11; i++) *reinterpret_cast<half4_t*>(st + 16 * i) = of[i];
    };
    int n = 0, t_prev = -1;
    for (;;) {
        const int tn = t + per;
        const bool has_next = tn / H < nimg;
        // (the last item fetches itself again as its "next": the DMA pieces between the MFMAs stay unconditional -- a branch around each
        // would cut the MFMA loops into separate scheduling regions -- at the price of 90 KB of extra L2 traffic per workgroup)
        const unsigned char* base_n = has_next ? item_base(tn) : base;
        const int kb = (n & 1) * PA_KB, ob = PA_KB - kb;
        PA_STAMP(0);
        // ---- A: K and Q of this item are in LDS (this wavefront's pieces: the last vector-memory operations of the previous item)
        PA_WAIT_VM(0);
        PA_STAMP(1);
        PA_BARRIER();
        PA_STAMP(2);
        // the B operand (Q^T) of the wavefront's 32 queries: lane (col, grp) = channels 16 kk + 8 grp .. of row 1 + 32 wave + col;
        // channels 88..95 (kk = 5, upper half) read 16 B of the next row: zeroed
        half8_t qf[AT_D / 16];
        {
            const unsigned char* qr = at_lds + ob + (1 + 32 * wave + col) * PA_KROW + 16 * grp;
#pragma unroll
            for (int kk = 0; kk < AT_D / 16; kk++) qf[kk] = *reinterpret_cast<const half8_t*>(qr + 32 * kk);
            if (grp) {
#pragma unroll
                for (int j = 0; j < 8; j++) qf[AT_D / 16 - 1][j] = (_Float16)0.0f;
            }
        }
        if (n > 0) {       // ... and the same rows take the previous item's output (this wavefront's LDS operations retire in order), which
            stage_rows(ob);                                                   // leaves at once as whole rows: nothing else is in the
#pragma unroll                                                                // memory queue right behind barrier A
            for (int i = 0; i < 6; i++) store_out(t_prev, ob, i);
        }
        if (PA_STUB < 3 && t_prev >= 0) merge_odd(t_prev, (n + 1) & 1);
        PA_STAMP(3);
        // ---- the odd query (row 0 of the Q image) against keys 32 wave .. 32 wave + 31 (+ key 256: every wavefront computes it, the last
        // one uses it) on the VALU, BETWEEN the MFMAs of QK^T below (one 16-byte slot per three contraction steps): lane (col, grp)
        // takes channels 48 grp .. of key 32 wave + col (the upper half has 40: its 6th slot is masked)
        float* mine = reinterpret_cast<float*>(at_lds + PA_PART_OFF) + ((n & 1) * AT_WAVES + wave) * PA_PART;
        const unsigned char* oq0 = at_lds + ob + 96 * grp;
        const unsigned char* okr = at_lds + kb + (32 * wave + col) * PA_KROW + 96 * grp;
        const unsigned char* okl = at_lds + kb + 256 * PA_KROW + 96 * grp;
        float os = 0.0f, os2 = 0.0f;
        auto odd_score_slot = [&](int i) {
            half8_t qv = *reinterpret_cast<const half8_t*>(oq0 + 16 * i);
            if (i == 5 && grp) {
#pragma unroll
                for (int j = 0; j < 8; j++) qv[j] = (_Float16)0.0f;
            }
            const half8_t kv = *reinterpret_cast<const half8_t*>(okr + 16 * i);
            const half8_t k2 = *reinterpret_cast<const half8_t*>(okl + 16 * i);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                os = __builtin_amdgcn_fdot2(half2_t{qv[2 * e], qv[2 * e + 1]}, half2_t{kv[2 * e], kv[2 * e + 1]}, os, false);
                os2 = __builtin_amdgcn_fdot2(half2_t{qv[2 * e], qv[2 * e + 1]}, half2_t{k2[2 * e], k2[2 * e + 1]}, os2, false);
            }
        };
        PA_STAMP(4);
        // ---- S^T = K Q^T for the wavefront's 32 queries: 9 key tiles x 6 contraction steps; the 7 V pieces of this item in between
        f32x16_t acc[AT_KT];
#pragma unroll
        for (int tt = 0; tt < AT_KT; tt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tt][r] = 0.0f;
        {
            const unsigned char* kbase = at_lds + kb + col * PA_KROW + 16 * grp;
#pragma unroll
            for (int g0 = 0; g0 < AT_KT; g0 += 3) {
                half8_t a_cur[3], a_nxt[3];
#pragma unroll
                for (int u = 0; u < 3; u++) a_cur[u] = *reinterpret_cast<const half8_t*>(kbase + 32 * (g0 + u) * PA_KROW);
#pragma unroll
                for (int kk = 0; kk < AT_D / 16; kk++) {
                    if (kk + 1 < AT_D / 16) {
#pragma unroll
                        for (int u = 0; u < 3; u++) a_nxt[u] = *reinterpret_cast<const half8_t*>(kbase + 32 * (g0 + u) * PA_KROW + 32 * (kk + 1));
                    }
#if PA_STUB == 0 || PA_STUB == 6
#pragma unroll
                    for (int u = 0; u < 3; u++) acc[g0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[u], qf[kk], acc[g0 + u], 0, 0, 0);
#else
                    if (g0 == 0 && kk == 0) acc[0][0] = (float)qf[0][0] + (float)qf[1][0] + (float)qf[2][0] + (float)qf[3][0] + (float)qf[4][0] + (float)qf[5][0];
                    if (PA_STUB < 4) __builtin_amdgcn_s_sleep(4);
#endif
#pragma unroll
                    for (int u = 0; u < 3; u++) a_cur[u] = a_nxt[u];
                    const int step = 2 * g0 + kk;          // 0 .. 17
                    if (step >= 1 && step <= 13 && (step & 1)) v_piece(base, step >> 1);     // 7 pieces at steps 1, 3, .., 13
                    if (PA_STUB < 3 && step % 3 == 0) odd_score_slot(step / 3);              // 6 slots at steps 0, 3, .., 15
                }
            }
        }
        if (PA_STUB < 3) {      // the odd query's 32 scores -> probabilities against the wavefront's own maximum (merged later)
            float s = os + __shfl_xor(os, 32, 64);
            const float s2 = os2 + __shfl_xor(os2, 32, 64);
            const bool last = wave == AT_WAVES - 1;
            float m = row16_max(s);
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, last ? s2 : m);
            const float p2 = last ? __builtin_amdgcn_exp2f((s2 - m) * c) : 0.0f;
            const float ps = __builtin_amdgcn_exp2f((s - m) * c);
            float l = row16_sum(ps);
            l += __shfl_xor(l, 16, 64);
            if (grp == 0) reinterpret_cast<float*>(at_lds + PA_PS_OFF)[32 * wave + col] = ps;    // read back (broadcast) by the odd query's P V
            if (lane == 0) { mine[96] = m; mine[97] = l + p2; mine[98] = p2; }
        }
        PA_STAMP(5);
        // ---- C: every wavefront is done with K (and with the odd query's row, and has read its staged rows out of the other buffer)
        PA_BARRIER();
        // ---- softmax of the 32 main queries: accumulator register r of tile tt = key 32 tt + (r & 3) + 8 (r >> 2) + 4 grp
        float m = -__builtin_huge_valf();
#pragma unroll
        for (int tt = 0; tt < (PA_STUB >= 2 && PA_STUB < 6 ? 1 : AT_KT); tt++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (32 * tt + (r & 3) + 8 * (r >> 2) + 4 * grp >= AT_S) acc[tt][r] = -__builtin_huge_valf();
                m = fmaxf(m, acc[tt][r]);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.0f;
        const float mc = -m * c;
        // probabilities go to f16 at once (the B operands of the PV MFMAs: chunk ch = 16 keys = registers 8 (ch & 1) .. + 7 of tile
        // ch >> 1): 72 registers instead of 144; the previous item's rows leave LDS in between (6 x 16 B per lane)
        constexpr int NCH = 2 * AT_KT - 1;      // 17 chunks of 16 keys: keys 272.. do not exist
        half8_t pb[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            if (PA_STUB < 2 || PA_STUB == 6 || ch == 0) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(acc[ch >> 1][8 * (ch & 1) + j], c, mc));
                    l += p;
                    pb[ch][j] = (_Float16)p;
                }
            } else {
                pb[ch] = pb[0];
            }
            if (ch % 3 == 1) q_piece(base_n, kb, ch / 3);       // the 6 Q pieces of the next item -> this item's K buffer
        }
        l += __shfl_xor(l, 32, 64);
        // ---- B: V of this item is in LDS (this wavefront's pieces: older than the 6 Q pieces)
        PA_STAMP(6);
        PA_WAIT_VM(6);
        PA_WAIT_LGKM0();
        PA_STAMP(7);
        PA_BARRIER();
        PA_STAMP(8);
        // ---- the odd query's share of P V runs BETWEEN the PV MFMAs below (two keys per 16-key chunk): lane = channel pair, its 32
        // probabilities are read back from LDS (one address per instruction: a broadcast)
        const int cp = min(lane, 47);
        const unsigned char* ovr = at_lds + PA_V_OFF + (32 * wave) * PA_VROW + 4 * cp;
        const float* opr = reinterpret_cast<const float*>(at_lds + PA_PS_OFF) + 32 * wave;
        float oo0[2] = {0.0f, 0.0f}, oo1[2] = {0.0f, 0.0f};
        auto odd_pv_key = [&](int k) {
            const float pk = opr[k];
            const half2_t vv = *reinterpret_cast<const half2_t*>(ovr + k * PA_VROW);
            oo0[k & 1] = fmaf(pk, (float)vv[0], oo0[k & 1]);
            oo1[k & 1] = fmaf(pk, (float)vv[1], oo1[k & 1]);
        };
        PA_STAMP(9);
        // ---- O^T = V^T P^T: per 16-key chunk three channel tiles; the A operand (32 channels x 16 keys) comes out of the row-major V
        // by two ds_read_b64_tr_b16: a 16-lane group addresses a [4 keys][16 channels] block (lane j: key j >> 2, channels 4 (j & 3) ..)
        // and lane j receives channel j of the 4 keys.  Keys behind the 8 operand elements: 16 ch + 4 grp + (0..3), then + 8.
        // In between: the 6 K pieces (-> the other buffer) and the 6 Q pieces (-> this item's K buffer) of the next item.
        f32x16_t o[AT_D / 32];
#pragma unroll
        for (int dt = 0; dt < AT_D / 32; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[dt][r] = 0.0f;
        {
            const int j16 = lane & 15, g16 = (lane >> 4) & 1;
            const unsigned char* vb = at_lds + PA_V_OFF + (4 * grp + (j16 >> 2)) * PA_VROW + 2 * (16 * g16 + 4 * (j16 & 3));
            half8_t v_cur[AT_D / 32], v_nxt[AT_D / 32];
#pragma unroll
            for (int dt = 0; dt < AT_D / 32; dt++) v_cur[dt] = tr_pair(vb + 64 * dt, vb + 64 * dt + 8 * PA_VROW);
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                if (ch + 1 < NCH) {
#pragma unroll
                    for (int dt = 0; dt < AT_D / 32; dt++)
                        v_nxt[dt] = tr_pair(vb + 16 * (ch + 1) * PA_VROW + 64 * dt, vb + 16 * (ch + 1) * PA_VROW + 64 * dt + 8 * PA_VROW);
                }
#if PA_STUB == 0 || PA_STUB == 6
#pragma unroll
                for (int dt = 0; dt < AT_D / 32; dt++) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_cur[dt], pb[ch], o[dt], 0, 0, 0);
#else
                if (ch == 0) { o[0][0] = (float)pb[0][0]; }
                if (PA_STUB < 4) __builtin_amdgcn_s_sleep(4);
#endif
#pragma unroll
                for (int dt = 0; dt < AT_D / 32; dt++) v_cur[dt] = v_nxt[dt];
                if (ch >= 2 && ch <= 12 && !(ch & 1)) k_piece(base_n, ob, (ch - 2) >> 1);     // the 6 K pieces of the next item
                if (PA_STUB < 3 || PA_STUB == 6) {
                    if (ch < 16) { odd_pv_key(2 * ch); odd_pv_key(2 * ch + 1); }
                }
            }
        }
        if (PA_STUB < 3 || PA_STUB == 6) {       // key 256 (p2 is zero except in the last wavefront), then the partial sums
            const half2_t vv = *reinterpret_cast<const half2_t*>(at_lds + PA_V_OFF + 256 * PA_VROW + 4 * cp);
            const float p2 = mine[98];
            oo0[0] = fmaf(p2, (float)vv[0], oo0[0]);
            oo1[0] = fmaf(p2, (float)vv[1], oo1[0]);
            if (lane < 48) *reinterpret_cast<float2*>(mine + 2 * lane) = float2{oo0[0] + oo0[1], oo1[0] + oo1[1]};
        }
        PA_STAMP(10);
        // ---- normalise; the rows stay in registers (f16) until the next item's Q has been read out of the buffer that stages them
        {
            const float inv = 1.0f / l;
            int i = 0;
#pragma unroll
            for (int dt = 0; dt < AT_D / 32; dt++) {
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    if (32 * dt + 8 * q4 + 8 <= DH) {      // 11 x 4 channels per lane
                        of[i] = half4_t{(_Float16)(o[dt][4 * q4] * inv), (_Float16)(o[dt][4 * q4 + 1] * inv),
                                        (_Float16)(o[dt][4 * q4 + 2] * inv), (_Float16)(o[dt][4 * q4 + 3] * inv)};
                        i++;
                    }
                }
            }
        }
        PA_WAIT_LGKM0();     // the partials are written before the next barrier A
        PA_STAMP(11);
        t_prev = t;
        if (!has_next) break;
        t = tn;
        base = base_n;
        n++;
    }
    // the last item's rows: through this item's K buffer, once the (unused) pieces that were still aimed at it have landed
    PA_WAIT_VM(0);
    PA_BARRIER();
    stage_rows((n & 1) * PA_KB);
#pragma unroll
    for (int i = 0; i < 6; i++) store_out(t_prev, (n & 1) * PA_KB, i);
    PA_BARRIER();
    if (PA_STUB < 3) merge_odd(t_prev, n & 1);
}

}  // namespace vlfm

using namespace vlfm;

extern "C" int vlfm_vit_attention_f16(const void* d_qkv, void* d_out, int batch, int tokens, int heads, int head_dim,
                                      float scale, void* stream) {
    if (batch == 0) return VLFM_OK;
    if (!d_qkv || !d_out || batch < 0 || heads <= 0)
        return fail(VLFM_ERR_INVALID, "vit_attention_f16: bad argument");
    if (tokens != AT_S || head_dim != 88)
        return fail(VLFM_ERR_INVALID, "vit_attention_f16: specialised for 257 tokens and a head width of 88");
    if (heads > 1024)      // (per-lane source offsets within an image are 32-bit: 257 rows of 6 * heads * 88 bytes)
        return fail(VLFM_ERR_INVALID, "vit_attention_f16: at most 1024 heads");
    static LdsOptIn opt_in;
    if (!opt_in.ensure(reinterpret_cast<const void*>(vit_attention_kernel), PA_LDS))
        return fail(VLFM_ERR_HIP, "vit_attention_f16: cannot opt in to 150 KB of LDS");
    // one workgroup per CU, in whole XCDs (workgroup id mod 8 = XCD); fewer when the batch has fewer items than that
    const int cus = device_cu_count() & ~7, images_per_xcd = (batch + 7) / 8;
    const int per = min(cus / 8, images_per_xcd * heads);
    VLFM_TIMED("vit_attention_kernel", stream);
    VLFM_KLAUNCH(vit_attention_kernel, dim3(8 * per), dim3(64 * AT_WAVES), PA_LDS, (hipStream_t)stream, (const _Float16*)d_qkv,
                 (_Float16*)d_out, batch, heads, scale);
    return check_launch("vit_attention_kernel");
}

#ifdef VLFM_PHASE_TIMING
extern "C" int vlfm_debug_attention_pers_clocks(long long* h_out480) {
    return hipMemcpyFromSymbol(h_out480, HIP_SYMBOL(g_pa_clk), sizeof(long long) * 480) == hipSuccess ? VLFM_OK : VLFM_ERR_HIP;
}
#endif
