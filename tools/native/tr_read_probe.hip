// ds_read_b64_tr_b16 semantics probe (gfx950): LDS holds the u16 value i at halfword i; lane l reads at byte address 8*l
// (pattern 0: four consecutive halfwords per lane) and prints which halfwords come back.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/native/tr_read_probe.hip -o gpurun_out/tr_read_probe && gpurun_out/tr_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(short* out, int pattern) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int byte = 8 * l;                                   // pattern 0
    if (pattern == 1) {                                 // [4 key rows of 192 B][16 columns]: lane j of a 16-lane group -> row j / 4, 4 (j % 4) columns
        const int g = l >> 4, j = l & 15;
        byte = (j >> 2) * 192 + 8 * (j & 3) + 32 * (g & 1) + 768 * (g >> 1);
    }
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)((__attribute__((address_space(3))) char*)lds + byte));
    for (int e = 0; e < 4; e++) out[4 * l + e] = v[e];
}
int main() {
    short* d;
    hipMalloc(&d, 64 * 4 * 2);
    for (int p = 0; p < 2; p++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, p);
        short h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", p);
        for (int l = 0; l < 64; l++) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
