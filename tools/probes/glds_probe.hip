// LDS-DMA (global_load_lds dwordx4) issue-throughput probe: how many bytes/clk/CU a 512-thread workgroup can pull from
// L2 into LDS as a function of the piece shape (rows x bytes per row), with nothing else running.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint16_t bf16_t;
template <int ROWB>   // bytes per row segment in one 1 KiB piece: 64, 128, 256, 1024
__global__ __launch_bounds__(512) void probe(const bf16_t* __restrict__ src, long ld, int iters, int wrap_elems, float* sink) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = ROWB / 16;            // lanes per row
  constexpr int RPP = 64 / LPR;             // rows per piece
  const bf16_t* p[4];
  for (int i = 0; i < 4; ++i) {
    const int row = ((wave * 4 + i) * RPP + lane / LPR) % 512;
    p[i] = src + (long)row * ld + (lane % LPR) * 8;
  }
  int koff = 0;
  for (int it = 0; it < iters; ++it) {
    uint8_t* st = smem + (it & 3) * 32768 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p[i] + koff),
                                       (__attribute__((address_space(3))) void*)(st + i * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    koff += ROWB / 2;
    if (koff >= wrap_elems) koff = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = ((float*)smem)[5];
}
template <int ROWB> void run(const bf16_t* d, float* sink, int blocks) {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)probe<ROWB>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<ROWB>, dim3(blocks), dim3(512), 131072, 0, d, 4096L, iters, 512, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * 8 * 4 * 1024 * iters;
    printf("rowbytes %4d blocks %3d: %.3f ms  %.1f GB/s  = %.1f B/clk/CU @2.1GHz  (%.0f cyc per glds per wave)\n", ROWB, blocks, ms,
           bytes / ms / 1e6, bytes / ms / 1e6 * 1e9 / 2.1e9 / blocks / 1e9 * 1.0, ms * 1e-3 * 2.1e9 / (iters * 4.0));
  }
}
int main() {
  bf16_t* d; hipMalloc(&d, 512L * 4096 * 2 + 65536); hipMemset(d, 0, 512L * 4096 * 2 + 65536);
  float* sink; hipMalloc(&sink, 64);
  for (int blocks : {256, 1}) { run<64>(d, sink, blocks); run<128>(d, sink, blocks); run<256>(d, sink, blocks); run<1024>(d, sink, blocks); }
  hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
