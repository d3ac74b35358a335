// Flash-style attention for gfx950, forward (causal / full, head dim 64 or 128) and backward
// (causal or full, head dim 128), built on v_mfma_f32_32x32x16_bf16.
//
// Conventions (one wave = 32 query rows or 32 key rows, 4 waves per workgroup):
//  * every MFMA is issued so that the per-lane "column" index j = lane&31 is the row the wave owns
//    (a query in fwd / dQ, a key in dK/dV).  Softmax statistics are then per-lane scalars.
//  * a B operand taken straight from 32x32 accumulators enumerates its contraction index inside each
//    group of 16 as {0-3, 8-11 | 4-7, 12-15}; the matching A operand comes from a PRE-TRANSPOSED,
//    chunk-swapped copy (rv_head_transpose) so that it is a single 16-byte LDS read.
//  * LDS tiles are XOR-swizzled per 16-byte chunk so that ds_read_b128 of "32 rows x same chunk" is
//    bank-conflict free: 256-byte rows use chunk ^ (row & 15); 128-byte rows use chunk ^ ((row >> 1) & 7).
#include "common.hpp"
#include "rlaifv_hip.h"

#include <stdlib.h>

namespace {

template <int ROWBYTES>
__device__ __forceinline__ uint32_t tile_off(int row, int c) {
  if (ROWBYTES == 256) return (uint32_t)(row * 256 + ((c ^ (row & 15)) << 4));
  return (uint32_t)(row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
}

__device__ __forceinline__ bf16x8_t pack_frag(const f32x16_t& a, int base) {
  union { uint32_t u[4]; bf16x8_t v; } r;
  r.u[0] = pack2bf(a[base + 0], a[base + 1]);
  r.u[1] = pack2bf(a[base + 2], a[base + 3]);
  r.u[2] = pack2bf(a[base + 4], a[base + 5]);
  r.u[3] = pack2bf(a[base + 6], a[base + 7]);
  return r.v;
}

__device__ __forceinline__ void zero16(f32x16_t& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// MFMA accumulating into AGPRs.  hipcc picks the VGPR form of v_mfma for builtins and, once the kernel needs more
// than 256 registers, shuttles the accumulators through v_accvgpr_read/write around EVERY MFMA (600+ copies per
// loop iteration in the dK/dV kernel).  The "a" constraint pins the long-lived accumulators in the accumulator file.
// s_nop 1 covers a VALU-written (cvt_pk) B operand; consecutive MFMAs on one accumulator need no wait states.
__device__ __forceinline__ void mfma_agpr(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// acc = 0 produced INSIDE the accumulator file (0 x 0 + 0), so the value never has a VGPR-class definition
__device__ __forceinline__ void mfma_agpr_zero(f32x16_t& acc) {
  const bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(acc) : "v"(z));
}

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// =============================================================================================
// forward
// =============================================================================================
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, long ld, int q_col0,
                                                          int k_col0, const bf16_t* __restrict__ vt,
                                                          bf16_t* __restrict__ out, long ldo,
                                                          float* __restrict__ lse, int L, int Lp, int H,
                                                          float scale) {
  constexpr int KS = HD / 16;        // k-steps of the QK^T contraction
  constexpr int ET = HD / 32;        // 32-row tiles of the output head dim
  constexpr int KROW = HD * 2;       // bytes per K row in LDS
  constexpr int KCPR = HD / 8;       // 16-byte chunks per K row
  constexpr int NCH_K = 64 * KCPR / 256;   // K chunks per thread
  constexpr int NCH_V = HD * 8 / 256;      // V^T chunks per thread
  __shared__ __attribute__((aligned(16))) uint8_t smem[64 * KROW + HD * 128];
  uint8_t* Ks = smem;
  uint8_t* Vs = smem + 64 * KROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, half = lane >> 5;
  const int nqb = (L + 127) / 128;
  const int h = blockIdx.y, s = blockIdx.z;
  const long tok0 = (long)s * L;
  // Causal work grows with the query block index and the dispatcher hands block b to CU b % 256, so a CU would
  // always draw the same index; each workgroup therefore processes the PAIR (x, nqb-1-x): equal work everywhere.
  const int npass = CAUSAL ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  const int qb = (pass == 0) ? (int)blockIdx.x : (nqb - 1 - (int)blockIdx.x);
  if (pass == 1 && qb <= (int)blockIdx.x) break;
  const int q0 = qb * 128, q0w = q0 + wave * 32;
  const int q = q0w + fr;

  // Q fragments (B operand): Q[q][16*ks + 8*half .. +7]
  bf16x8_t qf[KS];
  {
    const bf16_t* qp = qkv + (tok0 + min(q, L - 1)) * ld + q_col0 + h * HD + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8_t*)(qp + 16 * ks);
  }

  f32x16_t o[ET];
#pragma unroll
  for (int e = 0; e < ET; ++e) zero16(o[e]);
  float m_run = -INFINITY, l_run = 0.f;
  const float c = scale * LOG2E;

  const int kv_end = CAUSAL ? min(L, q0 + 128) : L;
  const int nt = (kv_end + 63) / 64;
  const bf16_t* vt_base = vt + ((long)(s * H + h) * HD) * Lp;

  u32x4_t pk[NCH_K], pv[NCH_V];
  auto prefetch = [&](int t) {
    const int k0 = t * 64;
#pragma unroll
    for (int i = 0; i < NCH_K; ++i) {
      const int ch = tid + i * 256, row = ch / KCPR, cc = ch % KCPR;
      pk[i] = *(const u32x4_t*)(qkv + (tok0 + min(k0 + row, L - 1)) * ld + k_col0 + h * HD + cc * 8);
    }
#pragma unroll
    for (int i = 0; i < NCH_V; ++i) {
      const int ch = tid + i * 256, row = ch >> 3, cc = ch & 7;
      pv[i] = *(const u32x4_t*)(vt_base + (long)row * Lp + k0 + cc * 8);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NCH_K; ++i) {
      const int ch = tid + i * 256, row = ch / KCPR, cc = ch % KCPR;
      *(u32x4_t*)(Ks + tile_off<KROW>(row, cc)) = pk[i];
    }
#pragma unroll
    for (int i = 0; i < NCH_V; ++i) {
      const int ch = tid + i * 256, row = ch >> 3, cc = ch & 7;
      *(u32x4_t*)(Vs + tile_off<128>(row, cc)) = pv[i];
    }
  };

  prefetch(0);
  for (int t = 0; t < nt; ++t) {
    const int k0 = t * 64;
    __syncthreads();
    commit();
    __syncthreads();
    if (t + 1 < nt) prefetch(t + 1);
    if (CAUSAL && k0 > q0w + 31) continue;   // wave-uniform: whole tile above this wave's diagonal

    f32x16_t sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      zero16(sacc[kt]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t kf = *(const bf16x8_t*)(Ks + tile_off<KROW>(kt * 32 + fr, 2 * ks + half));
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kt], 0, 0, 0);
      }
    }
    // scale to log2 domain, mask, tile max
    float tmax = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = sacc[kt][r] * c;
        if (key >= L || (CAUSAL && key > q)) v = -INFINITY;
        sacc[kt][r] = v;
        tmax = fmaxf(tmax, v);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = exp2f(sacc[kt][r] - m_new);
        sacc[kt][r] = p;
        psum += p;
      }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int e = 0; e < ET; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[e][r] *= alpha;
    // O^T[e][q] += V^T[e][key] * P^T[key][q]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8_t pf = pack_frag(sacc[kk >> 1], (kk & 1) * 8);
#pragma unroll
      for (int e = 0; e < ET; ++e) {
        const bf16x8_t vf = *(const bf16x8_t*)(Vs + tile_off<128>(e * 32 + fr, 2 * kk + half));
        o[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[e], 0, 0, 0);
      }
    }
  }

  if (q < L) {
    const float inv = 1.f / l_run;
    bf16_t* op = out + (tok0 + q) * ldo + h * HD;
#pragma unroll
    for (int e = 0; e < ET; ++e)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        uint2 w;
        w.x = pack2bf(o[e][rg * 4 + 0] * inv, o[e][rg * 4 + 1] * inv);
        w.y = pack2bf(o[e][rg * 4 + 2] * inv, o[e][rg * 4 + 3] * inv);
        *(uint2*)(op + e * 32 + rg * 8 + 4 * half) = w;
      }
    if (half == 0) lse[((long)s * H + h) * L + q] = (m_run + log2f(l_run)) * LN2;
  }
  __syncthreads();   // LDS tiles are reused by the second pass
  }  // pass
}

// =============================================================================================
// backward, dQ:   one workgroup = 128 queries (32 per wave), loop over 64-key tiles
//   S^T = K Q^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T
// =============================================================================================
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, long ld, int q_col0,
                                                             int k_col0, int v_col0,
                                                             const bf16_t* __restrict__ kt_,   // K^T swz [S][H][HD][Lp]
                                                             const bf16_t* __restrict__ dO, long lddo,
                                                             const float* __restrict__ lse,
                                                             const float* __restrict__ delta,
                                                             bf16_t* __restrict__ dqkv, long lddq, int L, int Lp,
                                                             int H, float scale) {
  constexpr int KS = HD / 16, ET = HD / 32, KROW = HD * 2, KCPR = HD / 8;
  constexpr int NCH = 64 * KCPR / 256;
  constexpr int NCH_T = HD * 8 / 256;
  __shared__ __attribute__((aligned(16))) uint8_t smem[2 * 64 * KROW + HD * 128];
  uint8_t* Ks = smem;
  uint8_t* Vs = smem + 64 * KROW;
  uint8_t* KTs = smem + 2 * 64 * KROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, half = lane >> 5;
  const int nqb = (L + 127) / 128;
  const int h = blockIdx.y, s = blockIdx.z;
  const long tok0 = (long)s * L;
  const int npass = CAUSAL ? 2 : 1;      // pair (x, nqb-1-x): see attn_fwd_kernel
  for (int pass = 0; pass < npass; ++pass) {
  const int qb = (pass == 0) ? (int)blockIdx.x : (nqb - 1 - (int)blockIdx.x);
  if (pass == 1 && qb <= (int)blockIdx.x) break;
  const int q0 = qb * 128, q0w = q0 + wave * 32;
  const int q = q0w + fr, qc = min(q, L - 1);

  bf16x8_t qf[KS], dof[KS];
  {
    const bf16_t* qp = qkv + (tok0 + qc) * ld + q_col0 + h * HD + 8 * half;
    const bf16_t* dp = dO + (tok0 + qc) * lddo + h * HD + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = *(const bf16x8_t*)(qp + 16 * ks);
      dof[ks] = *(const bf16x8_t*)(dp + 16 * ks);
    }
  }
  const float lse_q = lse[((long)s * H + h) * L + qc] * LOG2E;
  const float delta_q = delta[((long)s * H + h) * L + qc];
  const float c = scale * LOG2E;

  f32x16_t dq[ET];
#pragma unroll
  for (int e = 0; e < ET; ++e) zero16(dq[e]);

  const int kv_end = CAUSAL ? min(L, q0 + 128) : L;
  const int nt = (kv_end + 63) / 64;
  const bf16_t* kt_base = kt_ + ((long)(s * H + h) * HD) * Lp;

  for (int t = 0; t < nt; ++t) {
    const int k0 = t * 64;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = tid + i * 256, row = ch / KCPR, cc = ch % KCPR;
      const bf16_t* src = qkv + (tok0 + min(k0 + row, L - 1)) * ld + h * HD + cc * 8;
      *(u32x4_t*)(Ks + tile_off<KROW>(row, cc)) = *(const u32x4_t*)(src + k_col0);
      *(u32x4_t*)(Vs + tile_off<KROW>(row, cc)) = *(const u32x4_t*)(src + v_col0);
    }
#pragma unroll
    for (int i = 0; i < NCH_T; ++i) {
      const int ch = tid + i * 256, row = ch >> 3, cc = ch & 7;
      *(u32x4_t*)(KTs + tile_off<128>(row, cc)) = *(const u32x4_t*)(kt_base + (long)row * Lp + k0 + cc * 8);
    }
    __syncthreads();
    if (CAUSAL && k0 > q0w + 31) continue;

#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x16_t sacc, pacc;
      zero16(sacc);
      zero16(pacc);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t kf = *(const bf16x8_t*)(Ks + tile_off<KROW>(kt * 32 + fr, 2 * ks + half));
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc, 0, 0, 0);
        const bf16x8_t vf = *(const bf16x8_t*)(Vs + tile_off<KROW>(kt * 32 + fr, 2 * ks + half));
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], pacc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float p = exp2f(sacc[r] * c - lse_q);
        if (key >= L || (CAUSAL && key > q)) p = 0.f;
        sacc[r] = p * (pacc[r] - delta_q);   // dS^T
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int kk = kt * 2 + k2;
        const bf16x8_t df = pack_frag(sacc, k2 * 8);
#pragma unroll
        for (int e = 0; e < ET; ++e) {
          const bf16x8_t kf = *(const bf16x8_t*)(KTs + tile_off<128>(e * 32 + fr, 2 * kk + half));
          dq[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, df, dq[e], 0, 0, 0);
        }
      }
    }
  }

  if (q < L) {
    bf16_t* op = dqkv + (tok0 + q) * lddq + q_col0 + h * HD;
#pragma unroll
    for (int e = 0; e < ET; ++e)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        uint2 w;
        w.x = pack2bf(dq[e][rg * 4 + 0] * scale, dq[e][rg * 4 + 1] * scale);
        w.y = pack2bf(dq[e][rg * 4 + 2] * scale, dq[e][rg * 4 + 3] * scale);
        *(uint2*)(op + e * 32 + rg * 8 + 4 * half) = w;
      }
  }
  __syncthreads();
  }  // pass
}

// =============================================================================================
// backward, dK/dV:  one workgroup = 128 keys (32 per wave), loop over 64-query tiles
//   S = Q K^T, dP = dO V^T, P, dS;   dV^T += dO^T P,   dK^T += Q^T dS
// Register heavy (two 32x128 fp32 accumulators + K/V fragments): runs one wave per SIMD.
// =============================================================================================
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, long ld, int q_col0,
                                                              int k_col0, int v_col0,
                                                              const bf16_t* __restrict__ qt_,    // Q^T swz
                                                              const bf16_t* __restrict__ dO, long lddo,
                                                              const bf16_t* __restrict__ dot_,   // dO^T swz
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ delta,
                                                              bf16_t* __restrict__ dqkv, long lddq, int L, int Lp,
                                                              int H, float scale) {
  constexpr int KS = HD / 16, ET = HD / 32, KROW = HD * 2, KCPR = HD / 8;
  constexpr int NCH = 64 * KCPR / 256;
  constexpr int NCH_T = HD * 8 / 256;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* Qs = smem;
  uint8_t* dOs = smem + 64 * KROW;
  uint8_t* QTs = smem + 2 * 64 * KROW;
  uint8_t* dOTs = QTs + HD * 128;
  float* lse_s = (float*)(dOTs + HD * 128);
  float* delta_s = lse_s + 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, s = blockIdx.z;
  const long tok0 = (long)s * L;
  const int nkb = (L + 127) / 128;
  const int npass = CAUSAL ? 2 : 1;      // pair (x, nkb-1-x): see attn_fwd_kernel
  for (int pass = 0; pass < npass; ++pass) {
  const int kvb = (pass == 0) ? (int)blockIdx.x : (nkb - 1 - (int)blockIdx.x);
  if (pass == 1 && kvb <= (int)blockIdx.x) break;
  const int kv0 = kvb * 128, kv0w = kv0 + wave * 32;
  const int key = kv0w + fr, keyc = min(key, L - 1);

  bf16x8_t kf[KS], vf[KS];
  {
    const bf16_t* kp = qkv + (tok0 + keyc) * ld + h * HD + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = *(const bf16x8_t*)(kp + k_col0 + 16 * ks);
      vf[ks] = *(const bf16x8_t*)(kp + v_col0 + 16 * ks);
    }
  }
  const float c = scale * LOG2E;
  f32x16_t dk[ET], dv[ET];
#pragma unroll
  for (int e = 0; e < ET; ++e) { zero16(dk[e]); zero16(dv[e]); }

  const int t_begin = CAUSAL ? (kv0 / 64) : 0;
  const int nt = (L + 63) / 64;
  const bf16_t* qt_base = qt_ + ((long)(s * H + h) * HD) * Lp;
  const bf16_t* dot_base = dot_ + ((long)(s * H + h) * HD) * Lp;
  const float* lse_base = lse + ((long)s * H + h) * L;
  const float* delta_base = delta + ((long)s * H + h) * L;

  for (int t = t_begin; t < nt; ++t) {
    const int qs0 = t * 64;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = tid + i * 256, row = ch / KCPR, cc = ch % KCPR;
      const long tk = tok0 + min(qs0 + row, L - 1);
      *(u32x4_t*)(Qs + tile_off<KROW>(row, cc)) = *(const u32x4_t*)(qkv + tk * ld + q_col0 + h * HD + cc * 8);
      *(u32x4_t*)(dOs + tile_off<KROW>(row, cc)) = *(const u32x4_t*)(dO + tk * lddo + h * HD + cc * 8);
    }
#pragma unroll
    for (int i = 0; i < NCH_T; ++i) {
      const int ch = tid + i * 256, row = ch >> 3, cc = ch & 7;
      *(u32x4_t*)(QTs + tile_off<128>(row, cc)) = *(const u32x4_t*)(qt_base + (long)row * Lp + qs0 + cc * 8);
      *(u32x4_t*)(dOTs + tile_off<128>(row, cc)) = *(const u32x4_t*)(dot_base + (long)row * Lp + qs0 + cc * 8);
    }
    if (tid < 64) {
      const int qq = min(qs0 + tid, L - 1);
      lse_s[tid] = lse_base[qq] * LOG2E;
      delta_s[tid] = delta_base[qq];
    }
    __syncthreads();
    if (CAUSAL && qs0 + 63 < kv0w) continue;   // every query of the tile precedes this wave's keys

#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x16_t sacc, pacc;
      zero16(sacc);
      zero16(pacc);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t qf = *(const bf16x8_t*)(Qs + tile_off<KROW>(qt * 32 + fr, 2 * ks + half));
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[ks], sacc, 0, 0, 0);
        const bf16x8_t df = *(const bf16x8_t*)(dOs + tile_off<KROW>(qt * 32 + fr, 2 * ks + half));
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, vf[ks], pacc, 0, 0, 0);
      }
      // acc[r] <-> query ql = qt*32 + (r&3) + 8*(r>>2) + 4*half (tile-local), key = this lane's key
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int qg = qs0 + ql;
        float p = exp2f(sacc[r] * c - lse_s[ql]);
        if (qg >= L || key >= L || (CAUSAL && key > qg)) p = 0.f;
        sacc[r] = p;
        pacc[r] = p * (pacc[r] - delta_s[ql]);   // dS
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const bf16x8_t pf = pack_frag(sacc, k2 * 8);
        const bf16x8_t dsf = pack_frag(pacc, k2 * 8);
        const int cch = qt * 4 + k2 * 2 + half;
#pragma unroll
        for (int e = 0; e < ET; ++e) {
          const bf16x8_t dotf = *(const bf16x8_t*)(dOTs + tile_off<128>(e * 32 + fr, cch));
          dv[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pf, dv[e], 0, 0, 0);
          const bf16x8_t qtf = *(const bf16x8_t*)(QTs + tile_off<128>(e * 32 + fr, cch));
          dk[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dk[e], 0, 0, 0);
        }
      }
    }
  }

  if (key < L) {
    bf16_t* kp = dqkv + (tok0 + key) * lddq + h * HD;
#pragma unroll
    for (int e = 0; e < ET; ++e)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        uint2 w;
        w.x = pack2bf(dk[e][rg * 4 + 0] * scale, dk[e][rg * 4 + 1] * scale);
        w.y = pack2bf(dk[e][rg * 4 + 2] * scale, dk[e][rg * 4 + 3] * scale);
        *(uint2*)(kp + k_col0 + e * 32 + rg * 8 + 4 * half) = w;
        w.x = pack2bf(dv[e][rg * 4 + 0], dv[e][rg * 4 + 1]);
        w.y = pack2bf(dv[e][rg * 4 + 2], dv[e][rg * 4 + 3]);
        *(uint2*)(kp + v_col0 + e * 32 + rg * 8 + 4 * half) = w;
      }
  }
  __syncthreads();
  }  // pass
}

// =============================================================================================
// backward, dK/dV, version 2: no pre-transposed copies.
//   * Q / dO tiles (64 queries x 128, row-major as in HBM) are double-buffered in LDS by global_load_lds
//     (no staging registers; next tile in flight while the current one is consumed; one barrier per tile);
//   * the operands whose contraction index is the query (Q^T for dK, dO^T for dV) are read from the SAME tiles
//     with ds_read_b64_tr_b16, in the order in which P / dS leave the accumulators;
//   * chunk swizzle c ^ (((row&3)<<2) | ((row>>2)&3)): the 32-row ds_read_b128 pattern sees 16 distinct chunks and
//     the 4 rows of a transposing read fall into 4 different quarters of the 256-byte bank row;
//   * lse / delta of the tile arrive through 4-byte LDS-DMA.
// =============================================================================================
__device__ __forceinline__ uint32_t qtile_off(int row, int c) {
  return (uint32_t)(row * 256 + ((c ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4));
}

template <bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv2_kernel(const bf16_t* __restrict__ qkv, long ld, int q_col0,
                                                               int k_col0, int v_col0,
                                                               const bf16_t* __restrict__ dO, long lddo,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta,
                                                               bf16_t* __restrict__ dqkv, long lddq, int L, int H,
                                                               float scale) {
  constexpr int HD = 128, KS = 8, ET = 4;
  constexpr int STAGE = 2 * 64 * 256 + 512;          // Q tile + dO tile + lse[64] + delta[64]
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, s = blockIdx.z;
  const long tok0 = (long)s * L;
  const int nkb = (L + 127) / 128;
  const float c = scale * LOG2E;
  const float* lse_base = lse + ((long)s * H + h) * L;
  const float* delta_base = delta + ((long)s * H + h) * L;

  // LDS-DMA assignment: each wave fills 4 pieces (4 rows x 256 B) of Q and of dO per tile; wave 0 also lse/delta
  int d_row[4], d_chunk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    d_row[i] = (wave * 4 + i) * 4 + (lane >> 4);
    d_chunk[i] = (lane & 15) ^ (((d_row[i] & 3) << 2) | ((d_row[i] >> 2) & 3));
  }
  auto issue_tile = [&](int t, int buf) {
    uint8_t* st = smem + buf * STAGE;
    const int qs0 = t * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long tk = tok0 + min(qs0 + d_row[i], L - 1);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(qkv + tk * ld + q_col0 + h * HD + d_chunk[i] * 8),
          (__attribute__((address_space(3))) void*)(st + (wave * 4 + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(dO + tk * lddo + h * HD + d_chunk[i] * 8),
          (__attribute__((address_space(3))) void*)(st + 16384 + (wave * 4 + i) * 1024), 16, 0, 0);
    }
    if (wave == 0) {
      const int qq = min(qs0 + lane, L - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lse_base + qq),
                                       (__attribute__((address_space(3))) void*)(st + 32768), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(delta_base + qq),
                                       (__attribute__((address_space(3))) void*)(st + 32768 + 256), 4, 0, 0);
    }
  };

  // ---- per-lane LDS offsets, hoisted so the loops only add compile-time constants (the XOR swizzle defeats
  // the compiler's immediate-offset folding and it would otherwise keep ~100 address VGPRs alive)
  // row-operand reads (ds_read_b128): row = qt*32 + fr (+8192 per qt), chunk 2*ks + half
  uint32_t boff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) boff[ks] = qtile_off(fr, 2 * ks + half);
  // transposing reads: row = qt*32 + 16*k2 + 8u + 4*(g4>>1) + (s16>>2); column e = et*32 + 16*(g4&1) + 4*(s16&3)
  const int g4 = lane >> 4, s16 = lane & 15;
  uint32_t toff[2][ET];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int et = 0; et < ET; ++et)
      toff[u][et] = qtile_off(8 * u + 4 * (g4 >> 1) + (s16 >> 2), et * 4 + 2 * (g4 & 1) + ((s16 & 3) >> 1)) +
                    (uint32_t)((s16 & 1) * 8);
  auto tr8 = [&](const uint8_t* tile, int row0, int et) -> bf16x8_t {   // row0 = qt*32 + k2*16 (multiple of 16)
    typedef __attribute__((ext_vector_type(4))) short s4_t;
    const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s4_t*)(tile + toff[0][et] + row0 * 256));
    const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s4_t*)(tile + toff[1][et] + row0 * 256));
    bf16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return v;
  };

  const int npass = CAUSAL ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const int kvb = (pass == 0) ? (int)blockIdx.x : (nkb - 1 - (int)blockIdx.x);
    if (pass == 1 && kvb <= (int)blockIdx.x) break;
    const int kv0 = kvb * 128, kv0w = kv0 + wave * 32;
    const int key = kv0w + fr, keyc = min(key, L - 1);

    bf16x8_t kf[KS], vf[KS];
    {
      const bf16_t* kp = qkv + (tok0 + keyc) * ld + h * HD + 8 * half;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *(const bf16x8_t*)(kp + k_col0 + 16 * ks);
        vf[ks] = *(const bf16x8_t*)(kp + v_col0 + 16 * ks);
      }
    }
    f32x16_t dk[ET], dv[ET];
#pragma unroll
    for (int e = 0; e < ET; ++e) { mfma_agpr_zero(dk[e]); mfma_agpr_zero(dv[e]); }

    const int t_begin = CAUSAL ? (kv0 / 64) : 0;
    const int nt = (L + 63) / 64;
    issue_tile(t_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = t_begin; t < nt; ++t) {
      const int buf = (t - t_begin) & 1;
      const uint8_t* Qs = smem + buf * STAGE;
      const uint8_t* dOs = Qs + 16384;
      const float* lse_s = (const float*)(Qs + 32768);
      const float* delta_s = lse_s + 64;
      const int qs0 = t * 64;
      if (t + 1 < nt) issue_tile(t + 1, buf ^ 1);

      if (!(CAUSAL && qs0 + 63 < kv0w)) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          f32x16_t sacc, pacc;
          zero16(sacc);
          zero16(pacc);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t qf = *(const bf16x8_t*)(Qs + boff[ks] + qt * 8192);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[ks], sacc, 0, 0, 0);
            const bf16x8_t df = *(const bf16x8_t*)(dOs + boff[ks] + qt * 8192);
            pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, vf[ks], pacc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ql = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int qg = qs0 + ql;
            float p = exp2f(sacc[r] * c - lse_s[ql] * LOG2E);
            if (qg >= L || key >= L || (CAUSAL && key > qg)) p = 0.f;
            sacc[r] = p;
            pacc[r] = p * (pacc[r] - delta_s[ql]);
          }
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            const bf16x8_t pf = pack_frag(sacc, k2 * 8);
            const bf16x8_t dsf = pack_frag(pacc, k2 * 8);
            const int row0 = qt * 32 + k2 * 16;
#pragma unroll
            for (int e = 0; e < ET; ++e) {
              mfma_agpr(dv[e], tr8(dOs, row0, e), pf);
              mfma_agpr(dk[e], tr8(Qs, row0, e), dsf);
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // last asm MFMA -> v_accvgpr_read (hipcc pads nothing for asm)

    if (key < L) {
      bf16_t* kp = dqkv + (tok0 + key) * lddq + h * HD;
#pragma unroll
      for (int e = 0; e < ET; ++e)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          uint2 w;
          w.x = pack2bf(dk[e][rg * 4 + 0] * scale, dk[e][rg * 4 + 1] * scale);
          w.y = pack2bf(dk[e][rg * 4 + 2] * scale, dk[e][rg * 4 + 3] * scale);
          *(uint2*)(kp + k_col0 + e * 32 + rg * 8 + 4 * half) = w;
          w.x = pack2bf(dv[e][rg * 4 + 0], dv[e][rg * 4 + 1]);
          w.y = pack2bf(dv[e][rg * 4 + 2], dv[e][rg * 4 + 3]);
          *(uint2*)(kp + v_col0 + e * 32 + rg * 8 + 4 * half) = w;
        }
    }
  }  // pass
}

}  // namespace

extern "C" {

int rv_attn_fwd(const void* qkv, long ld, int q_col0, int k_col0, const void* vt, void* out, long ldo, float* lse,
                int S, int L, int H, int hd, int causal, float scale, void* stream) {
  RV_REQUIRE(hd == 64 || hd == 128, "rv_attn_fwd: head dim must be 64 or 128");
  RV_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && q_col0 % 8 == 0 && k_col0 % 8 == 0, "rv_attn_fwd: alignment");
  if (S == 0 || L == 0) return 0;
  const int Lp = rv_lp_stride(L);
  const int nb = (L + 127) / 128;
  dim3 grid(causal ? (nb + 1) / 2 : nb, H, S), block(256);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_FWD(HD_, C_)                                                                                    \
  hipLaunchKernelGGL((attn_fwd_kernel<HD_, C_>), grid, block, 0, st, (const bf16_t*)qkv, ld, q_col0, k_col0,    \
                     (const bf16_t*)vt, (bf16_t*)out, ldo, lse, L, Lp, H, scale)
  if (hd == 128) { if (causal) LAUNCH_FWD(128, true); else LAUNCH_FWD(128, false); }
  else { if (causal) LAUNCH_FWD(64, true); else LAUNCH_FWD(64, false); }
#undef LAUNCH_FWD
  RV_CHECK_LAUNCH();
  return 0;
}

int rv_attn_bwd(const void* qkv, long ld, int q_col0, int k_col0, int v_col0, const void* qt, const void* kt,
                const void* dO, long lddo, const void* dOt, const float* lse, const float* delta, void* dqkv,
                long lddq, int S, int L, int H, int hd, int causal, float scale, void* stream) {
  RV_REQUIRE(hd == 128, "rv_attn_bwd: head dim must be 128");
  RV_REQUIRE(ld % 8 == 0 && lddo % 8 == 0 && lddq % 8 == 0, "rv_attn_bwd: alignment");
  if (S == 0 || L == 0) return 0;
  const int Lp = rv_lp_stride(L);
  const int nb = (L + 127) / 128;
  dim3 grid(causal ? (nb + 1) / 2 : nb, H, S), block(256);
  hipStream_t st = (hipStream_t)stream;
  constexpr int DKV_LDS = 2 * 64 * 256 + 2 * 128 * 128 + 512;
  constexpr int DKV2_LDS = 2 * (2 * 64 * 256 + 512);
  static bool attr_done = false;
  static int dkv_version = 2;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
    hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
    hipFuncSetAttribute((const void*)attn_bwd_dkv2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV2_LDS);
    hipFuncSetAttribute((const void*)attn_bwd_dkv2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV2_LDS);
    const char* e = getenv("RV_ATTN_DKV");       // 1 = pre-transposed-copy kernel (needs qt / dOt), 2 = tr-read kernel
    if (e && e[0] == '1') dkv_version = 1;
    attr_done = true;
  }
  RV_REQUIRE(dkv_version == 2 || (qt != nullptr && dOt != nullptr), "rv_attn_bwd: qt / dOt required for RV_ATTN_DKV=1");
  if (causal) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<128, true>), grid, block, 0, st, (const bf16_t*)qkv, ld, q_col0, k_col0,
                       v_col0, (const bf16_t*)kt, (const bf16_t*)dO, lddo, lse, delta, (bf16_t*)dqkv, lddq, L, Lp, H,
                       scale);
    RV_CHECK_LAUNCH();
    if (dkv_version == 2)
      hipLaunchKernelGGL((attn_bwd_dkv2_kernel<true>), grid, block, DKV2_LDS, st, (const bf16_t*)qkv, ld, q_col0,
                         k_col0, v_col0, (const bf16_t*)dO, lddo, lse, delta, (bf16_t*)dqkv, lddq, L, H, scale);
    else
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, true>), grid, block, DKV_LDS, st, (const bf16_t*)qkv, ld, q_col0,
                         k_col0, v_col0, (const bf16_t*)qt, (const bf16_t*)dO, lddo, (const bf16_t*)dOt, lse, delta,
                         (bf16_t*)dqkv, lddq, L, Lp, H, scale);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<128, false>), grid, block, 0, st, (const bf16_t*)qkv, ld, q_col0, k_col0,
                       v_col0, (const bf16_t*)kt, (const bf16_t*)dO, lddo, lse, delta, (bf16_t*)dqkv, lddq, L, Lp, H,
                       scale);
    RV_CHECK_LAUNCH();
    if (dkv_version == 2)
      hipLaunchKernelGGL((attn_bwd_dkv2_kernel<false>), grid, block, DKV2_LDS, st, (const bf16_t*)qkv, ld, q_col0,
                         k_col0, v_col0, (const bf16_t*)dO, lddo, lse, delta, (bf16_t*)dqkv, lddq, L, H, scale);
    else
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, false>), grid, block, DKV_LDS, st, (const bf16_t*)qkv, ld, q_col0,
                         k_col0, v_col0, (const bf16_t*)qt, (const bf16_t*)dO, lddo, (const bf16_t*)dOt, lse, delta,
                         (bf16_t*)dqkv, lddq, L, Lp, H, scale);
  }
  RV_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"

extern "C" int rv_attn_lp(int L) { return rv_lp_stride(L); }
