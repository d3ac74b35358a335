// Which bf16 MFMA shape does more flops per joule?  Pure register-operand MFMA loops (no LDS, no memory) on all 256 CUs,
// 8 waves per CU, random bf16 operands, ~3 s each, while rocm-smi is sampled from a helper thread of the driver script
// (tools/probes/mfma_power_probe.sh).  Under the 1400 W package cap the sustained TFLOP/s IS the flops-per-joule ranking.
//   mode 0: v_mfma_f32_32x32x16_bf16, 4 independent accumulator tiles (64 accumulator registers)
//   mode 1: v_mfma_f32_16x16x32_bf16, 8 independent accumulator tiles (32 accumulator registers)
//   mode 2: 32x32x16 with 8 accumulator tiles (128 registers) - the GEMM kernels' register footprint
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int MODE>
__global__ __launch_bounds__(512) void probe(const uint4* __restrict__ src, int iters, float* sink) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  bf16x8_t a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint4 u = src[(tid * 8 + i) & 0xffff], v = src[(tid * 8 + 4 + i) & 0xffff];
    a[i] = __builtin_bit_cast(bf16x8_t, u);
    b[i] = __builtin_bit_cast(bf16x8_t, v);
  }
  float out = 0.f;
  if (MODE == 1) {
    f32x4_t c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + r) & 3], b[i & 3], c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) out += c[i][0] + c[i][3];
  } else {
    constexpr int NT = MODE == 0 ? 4 : 8;
    f32x16_t c[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16 / NT; ++r)
#pragma unroll
        for (int i = 0; i < NT; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[i & 3], c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) out += c[i][0] + c[i][15];
  }
  if (out == 12345.678f) sink[tid] = out;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
  uint4* src;
  float* sink;
  hipMalloc(&src, 65536 * 16);
  hipMalloc(&sink, 4 << 20);
  uint16_t* h = (uint16_t*)malloc(65536 * 16);
  srand(1);
  for (int i = 0; i < 65536 * 8; ++i) {       // random bf16 in roughly N(0, 1): sign, exponent 120..127, random mantissa
    h[i] = (uint16_t)(((rand() & 1) << 15) | ((120 + rand() % 8) << 7) | (rand() & 127));
  }
  hipMemcpy(src, h, 65536 * 16, hipMemcpyHostToDevice);
  const int iters = 20000, grid = 256;
  // flops per launch: every wave issues iters * (32 MFMAs of 16x16x32 | 16 of 32x32x16) = iters * 16 * 32768 flops either way
  const double flops = (double)grid * 8 * iters * 16.0 * 32768.0;
  auto launch = [&]() {
    if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 0, 0, src, iters, sink);
    else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 0, 0, src, iters, sink);
    else hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(512), 0, 0, src, iters, sink);
  };
  launch();
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  int n = 0;
  double el = 0;
  while (el < seconds) {
    for (int k = 0; k < 4; ++k) launch();
    hipDeviceSynchronize();
    n += 4;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  printf("mode %d (%s): %.0f TFLOP/s sustained over %.1f s (%d launches)\n", mode,
         mode == 1 ? "16x16x32, 8 tiles" : (mode == 0 ? "32x32x16, 4 tiles" : "32x32x16, 8 tiles"), flops * n / el / 1e12, el, n);
  return 0;
}
