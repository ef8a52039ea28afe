#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out) {
  int lane = threadIdx.x;
  float a = 100 + lane, b = 200 + lane;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  out[lane*2] = a; out[lane*2+1] = b;
}
int main(){ float* d; hipMalloc(&d, 4*128); float h[128]; hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h,d,4*128,hipMemcpyDeviceToHost);
 for (int l : {0,1,15,16,17,31,32,33,47,48,63}) printf("lane %d: a=%g b=%g\n", l, h[2*l], h[2*l+1]); return 0; }
