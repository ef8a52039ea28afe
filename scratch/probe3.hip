#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out) {
  int lane = threadIdx.x;
  unsigned a = 100 + lane, b = 200 + lane;
  asm volatile("" : "+v"(a)); asm volatile("" : "+v"(b));
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane*2] = (float)r[0]; out[lane*2+1] = (float)r[1];
}
int main(){ float* d; hipMalloc(&d, 4*128); float h[128]; hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h,d,4*128,hipMemcpyDeviceToHost);
 for (int l : {0,1,31,32,33,63}) printf("lane %d: r0=%g r1=%g\n", l, h[2*l], h[2*l+1]); return 0; }
