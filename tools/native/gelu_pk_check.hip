// Diagnostic (not product): the packed-f32 GELU of csrc/gemm_f16.hip against the scalar form it replaced, over EVERY f32 bit
// pattern, in both packed positions (inputs >= 2^127 excepted, see below).  Prints the number of differing results (NaN == NaN).  Build: hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off -I vlfm_amd/csrc tools/native/gelu_pk_check.hip -o tools/native/gelu_pk_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "gelu_f16.h"

__global__ void sweep(unsigned long long* bad) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const uint32_t b0 = (uint32_t)b, b1 = b0 * 2654435761u + 12345u;      // a second, unrelated value for the other position
        const float x0 = __uint_as_float(b0), x1 = __uint_as_float(b1);
        const float s0 = vlfm::gelu_erf(x0), s1 = vlfm::gelu_erf(x1);
        const vlfm::floatx2 p = vlfm::gelu_erf2(vlfm::floatx2{x0, x1});
        const bool ok0 = __float_as_uint(s0) == __float_as_uint(p[0]) || (s0 != s0 && p[0] != p[0]);
        const bool ok1 = __float_as_uint(s1) == __float_as_uint(p[1]) || (s1 != s1 && p[1] != p[1]);
        // (v + |v| overflows for v >= 2^127: the packed form returns inf there, the scalar form v -- f32 accumulators of f16
        //  products over K <= 2^16 cannot get near it)
        local += ((ok0 || x0 >= 0x1p127f) ? 0 : 1) + ((ok1 || x1 >= 0x1p127f) ? 0 : 1);
    }
    if (local) atomicAdd(bad, local);
}
int main() {
    unsigned long long* d; unsigned long long h = 0;
    if (hipMalloc(&d, 8) != hipSuccess || hipMemset(d, 0, 8) != hipSuccess) return 2;
    sweep<<<4096, 256>>>(d);
    if (hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("gelu packed vs scalar over every f32 bit pattern below 2^127, both packed positions: %llu differing\n", h);
    return h == 0 ? 0 : 1;
}
