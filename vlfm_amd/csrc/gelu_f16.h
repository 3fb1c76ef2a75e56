// GELU of the f16 GEMM epilogues (csrc/gemm_f16.hip): scalar form and the packed-f32 form the kernels use.
#pragma once
#include <hip/hip_runtime.h>
namespace vlfm {

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

// The same arithmetic on two values per lane in packed f32 (v_pk_fma_f32 / v_pk_mul_f32: one issue slot for both): 8 VALU
// instructions per value instead of 13 -- the epilogue of fc1 is VALU-bound (128 values per lane, two wavefronts per SIMD, nothing
// to overlap with: ~5.9 us of a ~46 us tile before, round 5's stamp probe).  Every operation is the scalar form's, in the same
// order (max(v, 0) as 0.5 (v + |v|), exact below 2^127): bit-identical results for every f32 input below 2^127
// (tools/native/gelu_pk_check.hip sweeps all 2^32 bit patterns).
typedef float floatx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ floatx2 gelu_erf2(floatx2 v) {
    const floatx2 u = __builtin_elementwise_abs(v);
    floatx2 q = floatx2{2.834908400e-06f, 2.834908400e-06f};
    q = __builtin_elementwise_fma(q, u, floatx2{-3.937762449e-05f, -3.937762449e-05f});
    q = __builtin_elementwise_fma(q, u, floatx2{1.861798810e-04f, 1.861798810e-04f});
    q = __builtin_elementwise_fma(q, u, floatx2{1.369373058e-04f, 1.369373058e-04f});
    q = __builtin_elementwise_fma(q, u, floatx2{-7.063421421e-03f, -7.063421421e-03f});
    q = __builtin_elementwise_fma(q, u, floatx2{5.249617994e-02f, 5.249617994e-02f});
    q = __builtin_elementwise_fma(q, u, floatx2{4.592081904e-01f, 4.592081904e-01f});
    q = __builtin_elementwise_fma(q, u, floatx2{1.151105165e+00f, 1.151105165e+00f});
    const floatx2 p = q * u;
    const floatx2 e = floatx2{__builtin_amdgcn_exp2f(-p[0]), __builtin_amdgcn_exp2f(-p[1])};
    const floatx2 h = u * floatx2{-0.5f, -0.5f};
    const floatx2 mx = (v + u) * floatx2{0.5f, 0.5f};
    return __builtin_elementwise_fma(h, e, mx);
}
__device__ __forceinline__ void gelu_erf4(float& v0, float& v1, float& v2, float& v3) {
    const floatx2 a = gelu_erf2(floatx2{v0, v1}), b = gelu_erf2(floatx2{v2, v3});
    v0 = a[0]; v1 = a[1]; v2 = b[0]; v3 = b[1];
}

}  // namespace vlfm
