#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ int tile_off(int row, int ch) { return 256*row + 16*(ch ^ (((row&3)<<2) | ((row>>2)&3))); }
__global__ void probe(short* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  short* s = (short*)smem;
  for (int c = threadIdx.x; c < 1024; c += 64) { int row = c>>4, ch = c&15; for (int e=0;e<8;++e) s[tile_off(row,ch)/2+e] = (short)(row*128 + ch*8+e); }
  __syncthreads();
  lds_u8* vb = (lds_u8*)smem;
  int lane = threadIdx.x; int h = lane>>5; int q4=(lane>>2)&3, p4=lane&3, g1=(lane>>4)&1;
  int idx = 0;
  for (int kk=0;kk<4;++kk) for (int d=0;d<4;++d) for (int half=0;half<2;++half) {
    int v_base = 256*(4*h+8*half+q4)+8*(p4&1); int v_low=(2*g1+(p4>>1))^(h+2*half);
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + v_base + 4096*kk + 64*(d^q4) + 16*v_low));
    for (int e=0;e<4;++e) out[(idx*64+lane)*4+e] = t[e];
    ++idx;
  }
}
int main() {
  short* d; hipMalloc(&d, 32*64*4*2); static short h[32*64*4];
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 16384, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad=0, idx=0;
  for (int kk=0;kk<4;++kk) for (int dd=0;dd<4;++dd) for (int half=0;half<2;++half) {
    for (int lane=0;lane<64;++lane) for (int e=0;e<4;++e) {
      int hh=lane>>5, r=lane&31; int key=16*kk+4*hh+8*half+e; int col=32*dd+r; int want=key*128+col; int got=h[(idx*64+lane)*4+e];
      if (got!=want) { if (bad<12) printf("kk %d d %d half %d lane %d e %d: got key %d col %d want key %d col %d\n",kk,dd,half,lane,e,got/128,got%128,key,col); ++bad; }
    }
    ++idx;
  }
  printf("bad %d\n", bad);
  return 0;
}
