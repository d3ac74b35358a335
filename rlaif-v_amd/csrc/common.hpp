// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the RLAIF-V DPO hot path.
// Wave = 64 lanes everywhere; bf16 storage, fp32 accumulation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

typedef uint16_t bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (8 bf16, 4 VGPR)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;  // 16-byte staging register (first-class vector)

#define RV_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }

// fp32 -> bf16, round-to-nearest-even, through gfx950's v_cvt_pk_bf16_f32 (the compiler selects it for
// __bf16 vector conversions; a hand-rolled integer rounding costs ~6 VALU + a divergent NaN branch).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_native_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_native_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

// (Round 4 tried the logistic with v_rcp_f32 in place of the IEEE division sequence - 10 VALU instructions per element, a third of the
//  SwiGLU epilogues' VALU work: SwiGLU-backward GEMM 2.207 -> 2.167 ms, -1.5 ms per step.  Withdrawn: the 1-ulp differences re-rolled
//  the bf16 rounding noise of the full-depth config-1 step from 3.4e-4 to 1.9e-3 of the oracle's loss (the bf16-emulated oracle itself
//  sits at 1.4e-3): inside the noise, outside the 1e-3 bar the fixture was checked with.  profiles/r04_swiglu_rcp_ab.log)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `red` = >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// Bijective XCD-aware remap of a 1-D block id: block b runs on XCD b%8 (observed, speed only);
// give every XCD a contiguous chunk of tile space so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ds_read_b64_tr_b16 through inline asm.  The clang builtin makes hipcc put `s_waitcnt vmcnt(0)` in front of the
// first transposing read of a loop body (it treats it as a consumer of every LDS-DMA in flight), which drains the
// global_load_lds prefetch the ping-pong / double-buffer schedules rely on.  An asm read is invisible to that
// bookkeeping: the CALLER must `s_waitcnt lgkmcnt(..)` + __builtin_amdgcn_sched_barrier(0) before the first use.
// Two reads into adjacent register pairs form one 8 x bf16 MFMA operand without any move.
typedef __attribute__((ext_vector_type(4))) short bf16x4s_t;
__device__ __forceinline__ bf16x4s_t ds_tr16_b64_asm(uint32_t lds_addr, int imm_offset) {
  bf16x4s_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(imm_offset) : "memory");
  return v;
}
__device__ __forceinline__ bf16x8_t ds_tr16_pair_asm(uint32_t lds_addr, int off_lo, int off_hi) {
  return __builtin_shufflevector(ds_tr16_b64_asm(lds_addr, off_lo), ds_tr16_b64_asm(lds_addr, off_hi), 0, 1, 2, 3, 4,
                                 5, 6, 7);
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) { return (uint32_t)(uintptr_t)p; }

// ds_read_b128 through inline asm.  hipcc schedules a row-fragment loop "read, s_waitcnt lgkmcnt(0), MFMA" one fragment
// at a time once registers are tight (attention kernels: 16 exposed LDS latencies per tile, seen in the ISA); explicit
// reads let the kernel keep a group of fragments in flight behind the MFMAs of the previous group.  Same contract as
// ds_tr16_b64_asm: the CALLER waits (counted lgkmcnt) and fences with __builtin_amdgcn_sched_barrier(0) before the first use.
__device__ __forceinline__ bf16x8_t ds_read_b128_asm(uint32_t lds_addr, int imm_offset) {
  bf16x8_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(imm_offset) : "memory");
  return v;
}

__device__ __forceinline__ f32x4_t ds_read_f32x4_asm(uint32_t lds_addr, int imm_offset) {
  f32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(imm_offset) : "memory");
  return v;
}

// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{}): a loop whose index is a template constant
// (hard-register asm helpers are selected by it)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// Error plumbing for the C ABI (no exceptions across the boundary).
void rv_set_error(const char* msg);
#define RV_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) {                                \
      rv_set_error(hipGetErrorString(e__));                 \
      return 2;                                             \
    }                                                       \
  } while (0)
#define RV_REQUIRE(cond, msg)                               \
  do {                                                      \
    if (!(cond)) {                                          \
      rv_set_error(msg);                                    \
      return 1;                                             \
    }                                                       \
  } while (0)
