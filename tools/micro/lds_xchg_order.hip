#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t* out, int pattern) {
  __shared__ uint32_t tab[256];
  int t = threadIdx.x;
  for (int i=t;i<256;i+=blockDim.x) tab[i]=1000;
  __syncthreads();
  uint32_t slot = pattern==0 ? 0 : pattern==1 ? (t&3) : pattern==2 ? ((t*7)&15) : (t>>4);
  uint32_t old = atomicExch(&tab[slot], (uint32_t)t);
  out[t] = old;
}
int main(){
  uint32_t* d; hipMalloc(&d, 64*4); uint32_t h[64];
  for (int pat=0; pat<4; ++pat){
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, pat); hipMemcpy(h,d,256,hipMemcpyDeviceToHost);
    int ok=1;
    for (int t=0;t<64;++t){
      uint32_t slot = pat==0 ? 0 : pat==1 ? (t&3) : pat==2 ? ((t*7)&15) : (t>>4);
      int pred=-1; for (int j=0;j<t;++j){ uint32_t sj = pat==0 ? 0 : pat==1 ? (j&3) : pat==2 ? ((j*7)&15) : (j>>4); if (sj==slot) pred=j; }
      uint32_t want = pred<0 ? 1000 : (uint32_t)pred;
      if (h[t]!=want) { ok=0; if (t<8) printf("pat %d lane %d got %u want %u\n", pat,t,h[t],want);} }
    printf("pattern %d ordered=%d\n", pat, ok);
  }
  return 0;
}
