// lds_dma_bytes.hip -- where does global_load_lds_ubyte put the byte of lane i?  (stride 1 or stride 4 per lane)
// hipcc --offload-arch=gfx950 -O3 lds_dma_bytes.hip -o bin/lds_dma_bytes && ./bin/lds_dma_bytes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const uint8_t *g, uint8_t *o, int dst0, int nact) {
    __shared__ uint8_t ring[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) ring[i] = 0xEE;
    __syncthreads();
    const uint8_t *src = g + threadIdx.x;
    uint32_t dst = __builtin_amdgcn_readfirstlane(dst0);
    if ((int)threadIdx.x < nact)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(ring + dst), 1, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) o[i] = ring[i];
}
int main() {
    uint8_t h[256], *g, *o, r[1024];
    for (int i = 0; i < 256; ++i) h[i] = (uint8_t)i;
    hipMalloc(&g, 256); hipMalloc(&o, 1024);
    hipMemcpy(g, h, 256, hipMemcpyHostToDevice);
    for (int dst0 : {0, 5, 130}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, dst0, 50);
        hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        printf("dst0=%d nact=50:", dst0);
        for (int i = 0; i < 1024; ++i) if (r[i] != 0xEE) printf(" [%d]=%d", i, r[i]);
        printf("\n");
    }
    return 0;
}
