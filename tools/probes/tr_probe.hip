// Probe of ds_read_b64_tr_b16 lane/element mapping on gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  const uint32_t base = (uint32_t)(uintptr_t)lds;   // LDS byte address of the array
  uint32_t addr;
  if (mode == 0) addr = l * 8;                                  // lane-linear: lane l -> elements 4l..4l+3
  else addr = ((l & 15) * 64 + (l >> 4) * 4) * 2;               // row-major [16 rows][64 cols]: lane -> row l&15, cols 4*(l>>4)..
  uint2 v;
  addr += base;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipError_t e = hipMalloc(&d, 64 * 4 * 2); printf("malloc %s\n", hipGetErrorString(e));
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    e = hipGetLastError(); printf("launch %s\n", hipGetErrorString(e));
    e = hipDeviceSynchronize(); printf("sync %s\n", hipGetErrorString(e));
    e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); printf("copy %s\n", hipGetErrorString(e));
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("L%d:%d,%d,%d,%d ", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l % 8 == 7) printf("\n"); }
  }
  return 0;
}
