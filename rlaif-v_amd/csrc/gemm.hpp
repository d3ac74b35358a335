// bf16 NT GEMM for gfx950:  D[m][n] = sum_k A[m][k] * B[n][k]   (both operands K-contiguous)
//
// Tile 128x128x64, 256 threads = 4 waves in 2x2, each wave 64x64 as 2x2 v_mfma_f32_32x32x16_bf16.
// The MFMA is issued with SWAPPED operands (A_op <- B rows, B_op <- A rows) so a lane ends up with
// 4 consecutive n for a fixed m: row-major bf16 stores are 8 bytes wide instead of 2.
//   acc[tm][tn][r] = D[m = m_w + tm*32 + (lane&31)][n = n_w + tn*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)]
//
// LDS image of a [128 rows][64 k] bf16 operand tile: 128-byte rows, 16-byte chunk kc of row `row`
// lives at  row*128 + ((kc ^ ((row>>1)&7)) << 4)  -> conflict-free ds_read_b128 for the 32x32x16
// fragment pattern (rows distinct mod 16 inside every b128 lane group).
// STAGE=1 fills that image with global_load_lds (LDS destination lane-linear, swizzle applied on the
// per-lane SOURCE address); STAGE=0 stages through registers (global_load_dwordx4 + ds_write_b128).
#pragma once
#include <type_traits>

#include "common.hpp"

// Wave-priority schedule of the ping-pong kernels, measured A/B on one box (tools/exp_gemm_variants.py,
// profiles/r02_gemm_variants_ab.log; 27,664-token step shapes):
//   0 = s_setprio 1 around every MFMA segment, 1 = static: the second wave group at priority 1 for the whole main loop,
//   2 = no priority changes.  TN (weight gradients): 1 beats 0 by +7.3 / +8.2 % (2: +6.3 / +6.7 %); NN: 2 beats 0 by
//   +0.6 %, 1 loses 0.5 %.  Defaults below; -DRV_GEMM_PRIO_MODE=n forces one mode on both (experiment builds).
//   RV_GEMM_DMA_SLOT = the MFMA slot (mod 4) after which a wave issues one LDS-DMA piece (0..3 within +-1 %: default 1).
#ifdef RV_GEMM_PRIO_MODE
#define RV_GEMM_PRIO_TN RV_GEMM_PRIO_MODE
#define RV_GEMM_PRIO_NN RV_GEMM_PRIO_MODE
#else
#define RV_GEMM_PRIO_TN 1
#define RV_GEMM_PRIO_NN 2
#endif
#ifndef RV_GEMM_DMA_SLOT
#define RV_GEMM_DMA_SLOT 1
#endif
#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_THREADS 256
#define GEMM_LDS_BYTES (2 * 2 * GEMM_BM * GEMM_BK * 2)  // 2 buffers x (A,B) x 16 KiB

struct GemmShape {
  const bf16_t* A; const bf16_t* B;
  int M, N, K;
  long lda, ldb;
  int group;     // row tiles per scheduling group (tile order inside an XCD chunk); 0 = kernel default
  // Optional second contraction segment (kernels instantiated with EXT = true; the fused LoRA GEMM):
  //   D[m][n] += sum_{q < K2} A2[m][c0(n) + q] * B2[n][q],   c0(n) = group_cols ? group(n) * K2 : 0,
  //   group(n) = n < group0 ? 0 : 1 + (n - group0) / group_cols   (group0 = 0 means group0 = group_cols: uniform groups;
  //   grouped-query attention: q is `hidden` wide, k and v `kv_dim` wide -> group0 = hidden, group_cols = kv_dim)
  // i.e. the K loop simply runs K2 / BK further steps whose tiles come from (A2, B2).  group_cols must be a
  // multiple of the N tile so that c0 is uniform per workgroup.
  const bf16_t* A2; const bf16_t* B2;
  long lda2, ldb2;
  int K2, group_cols;
  int group0;
  // Adapter-FIRST form (gemm_nn_a64_kernel<.., PRE = true>; the LoRA input gradient under adapter dropout):
  //   D = dropmask(A2 B2) * pre_inv_keep + A B   - the K2 segment runs first, the mask of rv_dropout for a contiguous [M][N]
  //   tensor (threshold / key as in EpiStore) is applied to the ACCUMULATORS, then the main segment accumulates on top.
  uint32_t pre_thresh16 = 0, pre_key = 0;
  float pre_inv_keep = 1.f;
};

__device__ __forceinline__ uint32_t gemm_mix32(uint32_t h) {    // same mixer as dropout_kernel (elementwise.hip)
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}

// rv_dropout's keep decision for the 4 consecutive elements (m, n .. n+3), n % 4 == 0, of a contiguous [M][N] tensor (N % 8 == 0)
__device__ __forceinline__ f32x4_t gemm_dropmask4(f32x4_t v, long m, int n, int N, uint32_t thresh16, uint32_t key, float inv_keep) {
  const long e = m * N + n;
  const uint32_t base = (uint32_t)(e >> 33) * 0x9e3779b9u + key;
  const uint32_t h0 = gemm_mix32((uint32_t)(e >> 1) ^ base), h1 = gemm_mix32(((uint32_t)(e >> 1) + 1u) ^ base);
  v.x = ((h0 & 0xffffu) >= thresh16) ? v.x * inv_keep : 0.f;
  v.y = ((h0 >> 16) >= thresh16) ? v.y * inv_keep : 0.f;
  v.z = ((h1 & 0xffffu) >= thresh16) ? v.z * inv_keep : 0.f;
  v.w = ((h1 >> 16) >= thresh16) ? v.w * inv_keep : 0.f;
  return v;
}

__device__ __forceinline__ int ext_group(int n0, int group0, int group_cols) {
  const int g0 = group0 > 0 ? group0 : group_cols;
  return n0 < g0 ? 0 : 1 + (n0 - g0) / group_cols;
}

__device__ __forceinline__ uint32_t gemm_lds_off(int row, int kc) {
  return (uint32_t)(row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
}

template <int STAGE, class Epi, bool EXT = false>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_kernel(GemmShape g, Epi epi) {
  static_assert(!EXT || STAGE == 1, "the second K segment is implemented for the LDS-DMA staging only");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_m = (g.M + GEMM_BM - 1) / GEMM_BM, tiles_n = (g.N + GEMM_BN - 1) / GEMM_BN;
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  constexpr int GROUP = 8;
  const int group_size = GROUP * tiles_n;
  const int first_m = (id / group_size) * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int tile_m = first_m + (id % group_size) % gsz;
  const int tile_n = (id % group_size) / gsz;
  const int m0 = tile_m * GEMM_BM, n0 = tile_n * GEMM_BN;

  auto As = [&](int buf) -> uint8_t* { return smem + buf * 32768; };
  auto Bs = [&](int buf) -> uint8_t* { return smem + 16384 + buf * 32768; };

  // ---- staging addresses (4 x 16-byte chunks of A and of B per thread per K tile) ----
  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
  uint32_t st_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row, kc;
    if (STAGE == 1) {
      row = wave * 32 + i * 8 + (lane >> 3);
      const int slot = lane & 7;
      kc = slot ^ ((row >> 1) & 7);
      st_off[i] = (uint32_t)((wave * 4 + i) * 1024);  // wave-uniform LDS base of this DMA piece
    } else {
      const int c = tid + i * 256;
      row = c >> 3;
      kc = c & 7;
      st_off[i] = gemm_lds_off(row, kc);
    }
    const int ra = min(m0 + row, g.M - 1), rb = min(n0 + row, g.N - 1);
    a_src[i] = g.A + (long)ra * g.lda + kc * 8;
    b_src[i] = g.B + (long)rb * g.ldb + kc * 8;
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt1 = g.K / GEMM_BK;
  const int nt = nt1 + (EXT ? g.K2 / GEMM_BK : 0);
  const int a2_col0 = (EXT && g.group_cols > 0) ? ext_group(n0, g.group0, g.group_cols) * g.K2 : 0;
  u32x4_t ra_[4], rb_[4];

  auto issue = [&](int t, int buf) {
    const long koff = (long)t * GEMM_BK;
    if (EXT && t >= nt1) {     // wave-uniform: tiles of the second segment, addresses rebuilt on the fly
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const long ko = (long)(t - nt1) * GEMM_BK + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
        const bf16_t* sa = g.A2 + (long)min(m0 + row, g.M - 1) * g.lda2 + a2_col0 + ko;
        const bf16_t* sb = g.B2 + (long)min(n0 + row, g.N - 1) * g.ldb2 + ko;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                         (__attribute__((address_space(3))) void*)(As(buf) + st_off[i]), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb,
                                         (__attribute__((address_space(3))) void*)(Bs(buf) + st_off[i]), 16, 0, 0);
      }
    } else if (STAGE == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + koff),
                                         (__attribute__((address_space(3))) void*)(As(buf) + st_off[i]), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + koff),
                                         (__attribute__((address_space(3))) void*)(Bs(buf) + st_off[i]), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra_[i] = *(const u32x4_t*)(a_src[i] + koff);
        rb_[i] = *(const u32x4_t*)(b_src[i] + koff);
      }
    }
  };
  auto commit = [&](int buf) {
    if (STAGE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *(u32x4_t*)(As(buf) + st_off[i]) = ra_[i];
        *(u32x4_t*)(Bs(buf) + st_off[i]) = rb_[i];
      }
    }
  };

  const int frow = lane & 31, fhalf = lane >> 5;
  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kc = ks * 2 + fhalf;
      bf16x8_t af[2], bfr[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        af[t] = *(const bf16x8_t*)(As(buf) + gemm_lds_off(wm * 64 + t * 32 + frow, kc));
        bfr[t] = *(const bf16x8_t*)(Bs(buf) + gemm_lds_off(wn * 64 + t * 32 + frow, kc));
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[tn], af[tm], acc[tm][tn], 0, 0, 0);
    }
  };

  issue(0, 0);
  commit(0);
  if (STAGE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    // unconditional (clamped) prefetch: a branch here makes hipcc keep the staging registers in scratch
    issue(min(t + 1, nt - 1), cur ^ 1);
    compute(cur);
    commit(cur ^ 1);
    if (STAGE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  epi.apply(acc, m0 + wm * 64, n0 + wn * 64, lane, g.M, g.N);
}

// =============================================================================================
// Skinny NT GEMM: C[M][N] = alpha * A[M][K] B[N][K]^T for N <= 256 (a multiple of 64) and many rows - the rank-r side of LoRA
// (t = dropout(x) A^T, dt = dy B: 29 k rows, N = 64 per adapter group, K = 4096 .. 22016).  These launches STREAM the activation
// operand once (2 N flop per byte: HBM-bound for N = 64) and the 128x128 kernel above - one tile of prefetch, drained completely
// every K step - left them at 1.1 - 1.8 x their streaming time (profiles/r04_lora_skinny_floor.log).  Here: 128 rows x 64 columns
// per workgroup (4 waves, a 32 x 64 strip each), a THREE-stage LDS ring (16 KB of A + 8 KB of B per stage, 72 KB: two workgroups
// per CU) filled two K tiles ahead by global_load_lds with a counted vmcnt(6), one barrier per 64-deep K tile.  Column blocks of
// the same row block sit next to each other in launch order (same XCD): the activation tile comes from HBM once.
// LDS image and swizzle as gemm_nt_kernel (128-byte rows, 16-byte chunk kc at kc ^ ((row >> 1) & 7)).
// =============================================================================================
#define GSK_BM 128
#define GSK_BN 64
#define GSK_STAGE (16384 + 8192)
#define GSK_LDS_BYTES (3 * GSK_STAGE)
__global__ __launch_bounds__(256, 2) void gemm_nt_skinny_kernel(GemmShape g, bf16_t* __restrict__ C, long ldc, float alpha) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = g.N / GSK_BN, tiles_m = (g.M + GSK_BM - 1) / GSK_BM;
  const int id = xcd_remap(blockIdx.x, tiles_m * nb);
  const int m0 = (id / nb) * GSK_BM, n0 = (id % nb) * GSK_BN;

  // DMA pieces (8 rows x 128 B = 1 KB, lane l -> row l >> 3, chunk (l & 7) ^ swizzle): wave w owns A pieces 4w .. 4w+3 (rows 32 w ..)
  // and B pieces 2w, 2w+1 (rows 16 w ..)
  const bf16_t* a_src[4];
  const bf16_t* b_src[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    a_src[i] = g.A + (long)min(m0 + row, g.M - 1) * g.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + (lane >> 3);
    b_src[i] = g.B + (long)(n0 + row) * g.ldb + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
  }
  auto issue = [&](int t, int stage) {
    uint8_t* as = smem + stage * GSK_STAGE;
    uint8_t* bs = as + 16384;
    const long koff = (long)t * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + koff),
                                       (__attribute__((address_space(3))) void*)(as + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + koff),
                                       (__attribute__((address_space(3))) void*)(bs + (wave * 2 + i) * 1024), 16, 0, 0);
  };

  f32x16_t acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int nt = g.K / 64;
  const int frow = lane & 31, fhalf = lane >> 5;
  issue(0, 0);
  issue(min(1, nt - 1), 1);
  for (int t = 0; t < nt; ++t) {
    // tile t (issued two iterations ago) has landed for this wave: everything but the newest tile's 6 pieces
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // ... and for every wave; all waves are also done reading stage (t + 2) % 3 = (t - 1) % 3
    issue(min(t + 2, nt - 1), (t + 2) % 3);                       // (clamped: redundant tail loads keep the count uniform)
    const uint8_t* as = smem + (t % 3) * GSK_STAGE;
    const uint8_t* bs = as + 16384;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kc = ks * 2 + fhalf;
      const bf16x8_t af = *(const bf16x8_t*)(as + gemm_lds_off(wave * 32 + frow, kc));
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const bf16x8_t bfr = *(const bf16x8_t*)(bs + gemm_lds_off(tn * 32 + frow, kc));
        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, af, acc[tn], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the clamped tail loads still target this workgroup's LDS
  // swapped operands: lane (frow, fhalf) holds row m = frow of the strip, columns 32 tn + 8 rg + 4 fhalf + j in register 4 rg + j
  const int m = m0 + wave * 32 + frow;
  if (m < g.M) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        uint2 o;
        o.x = pack2bf(acc[tn][rg * 4 + 0] * alpha, acc[tn][rg * 4 + 1] * alpha);
        o.y = pack2bf(acc[tn][rg * 4 + 2] * alpha, acc[tn][rg * 4 + 3] * alpha);
        *(uint2*)(C + (long)m * ldc + n0 + tn * 32 + rg * 8 + 4 * fhalf) = o;
      }
  }
}

// =============================================================================================
// 256x256x32 "ping-pong" kernel for large GEMMs.
//
// 8 waves (2 M-halves x 4 N-quarters), per-wave output 128x64 = 4x2 MFMA tiles (128 accumulator VGPRs),
// K step 32, 4-stage LDS ring (4 x (16 KiB A + 16 KiB B) = 128 KiB) filled two tiles ahead with
// global_load_lds.  Each SIMD hosts one wave of group 0 (M-half 0) and one of group 1; group 1 runs ONE
// barrier behind group 0, so at any time one wave of a SIMD is in its MFMA segment while its partner is in
// its load segment (12 ds_read_b128 + 4 LDS-DMA issues):
//     L-seg: ds_read tile p | issue DMA tile p+2 | vmcnt(4): own pieces of tile p+1 landed | lgkmcnt(0)
//     s_barrier   (X)
//     M-seg: 16 x v_mfma_f32_32x32x16_bf16 under s_setprio 1
//     s_barrier   (Y)
// Hazards (barrier instance 2p = G0.X(p) = G1.Y(p-1); 2p+1 = G0.Y(p) = G1.X(p)):
//   RAW tile p+1: every wave retires its own DMA pieces (counted vmcnt) before its X(p); G0 reads after
//       instance 2p+1, G1 after 2p+2 - both later than every wave's X(p).
//   WAR stage (p+2)%4: last read in L-seg(p-2), two phases earlier; reads are retired (lgkmcnt(0)) before X.
// LDS image of a [256 rows][32 k] tile: 64-byte rows, 16-byte chunk kc of row r at r*64 + ((kc ^ ((r>>2)&3))<<4):
// the 16 rows of every ds_read_b128 lane group land on 16 distinct 16-byte slots of the 256-byte bank row.
// =============================================================================================
#define G2_BM 256
#define G2_BN 256
#define G2_BK 32
#define G2_THREADS 512
#define G2_STAGE_BYTES 32768
#define G2_LDS_BYTES (4 * G2_STAGE_BYTES)

// DMA_IN_MSEG = true issues the LDS-DMA pieces of tile p+3 between the MFMAs of M-seg(p) (free issue slots of the
// matrix pipe) instead of tile p+2 at the end of L-seg(p): the load segment shrinks to the 12 ds_reads.
// WAR for stage (p+3)%4 = (p-1)%4: G0 issues after instance 2p = G1.Y(p-1), G1 after 2p+1 = G0.Y(p); both groups
// retired their L-seg(p-1) reads before X(p-1).  RAW and the counted vmcnt(4) are unchanged.
// ABLATE (debug/profiling only, results are wrong): 1 = no LDS-DMA in the main loop, 2 = no ds_reads in the loop,
// 3 = DMA always re-reads K tile 0 (same instruction stream, every load an L2 hit).
// DIST = prefetch distance in K tiles when DMA_IN_MSEG (ring of DIST+1 stages: 3 -> 128 KiB, 4 -> 160 KiB).
// SPLIT = 1 (with DMA_IN_MSEG, DIST 3): pieces 0,1 of tile p+3 are issued at the end of L-seg(p), pieces 2,3 inside
// M-seg(p) - a VMEM issue stalls the in-order wave for ~60 cycles and starves the matrix pipe, so half of them move
// to the segment that is not using it.
template <class Epi, bool DMA_IN_MSEG, int ABLATE = 0, int DIST = 3, int SPLIT = 0, bool EXT = false>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm_nt_256_kernel(GemmShape g, Epi epi) {
  constexpr int NST = DMA_IN_MSEG ? DIST + 1 : 4;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const int tiles_m = (g.M + G2_BM - 1) / G2_BM, tiles_n = (g.N + G2_BN - 1) / G2_BN;
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int GROUP = g.group > 0 ? g.group : 4;   // measured at M = 27.6 k: 4 beats 8 by 0-3.5 %, 16/32 lose 4-10 %
  const int group_size = GROUP * tiles_n;
  const int first_m = (id / group_size) * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int tile_m = first_m + (id % group_size) % gsz;
  const int tile_n = (id % group_size) / gsz;
  const int m0 = tile_m * G2_BM, n0 = tile_n * G2_BN;

  // ---- LDS-DMA sources: 2 pieces (1 KiB = 16 rows x 64 B) of A and of B per wave per K tile ----
  const bf16_t* a_src[2];
  const bf16_t* b_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 16 + (lane >> 2);
    const int kc = (lane & 3) ^ ((row >> 2) & 3);
    a_src[i] = g.A + (long)min(m0 + row, g.M - 1) * g.lda + kc * 8;
    b_src[i] = g.B + (long)min(n0 + row, g.N - 1) * g.ldb + kc * 8;
  }
  const uint32_t piece0 = (uint32_t)(wave * 2) * 1024u;
  const int nt1 = g.K / G2_BK;
  const int nt = nt1 + (EXT ? g.K2 / G2_BK : 0);
  const int a2_col0 = (EXT && g.group_cols > 0) ? ext_group(n0, g.group0, g.group_cols) * g.K2 : 0;

  auto issue_piece = [&](int t, int j) {   // j: 0 = A piece 0, 1 = B piece 0, 2 = A piece 1, 3 = B piece 1
    uint8_t* st = smem + (t % NST) * G2_STAGE_BYTES;
    const long koff = (ABLATE == 3) ? 0 : (long)t * G2_BK;
    const int i = j >> 1;
    if (EXT && t >= nt1) {     // wave-uniform: second K segment, addresses rebuilt on the fly (2 of ~130 steps)
      const int row = (wave * 2 + i) * 16 + (lane >> 2);
      const long ko = (long)(t - nt1) * G2_BK + (((lane & 3) ^ ((row >> 2) & 3)) << 3);
      const bf16_t* src = (j & 1) == 0 ? g.A2 + (long)min(m0 + row, g.M - 1) * g.lda2 + a2_col0 + ko
                                       : g.B2 + (long)min(n0 + row, g.N - 1) * g.ldb2 + ko;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + ((j & 1) ? 16384 : 0) + piece0 + i * 1024),
                                       16, 0, 0);
    } else if ((j & 1) == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + koff),
                                       (__attribute__((address_space(3))) void*)(st + piece0 + i * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + koff),
                                       (__attribute__((address_space(3))) void*)(st + 16384 + piece0 + i * 1024), 16, 0, 0);
  };
  auto issue = [&](int t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_piece(t, j);
  };
  // steady-state form: per-lane source pointers that already point at the tile to fetch (advanced once per K tile),
  // so a DMA issue between two MFMAs is "write M0, global_load_lds" and nothing else
  const bf16_t* a_run[2] = {a_src[0] + DIST * G2_BK, a_src[1] + DIST * G2_BK};
  const bf16_t* b_run[2] = {b_src[0] + DIST * G2_BK, b_src[1] + DIST * G2_BK};
  auto issue_piece_run = [&](int t, int j) {
    uint8_t* st = smem + (t % NST) * G2_STAGE_BYTES;
    const int i = j >> 1;
    if ((j & 1) == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_run[i],
                                       (__attribute__((address_space(3))) void*)(st + piece0 + i * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_run[i],
                                       (__attribute__((address_space(3))) void*)(st + 16384 + piece0 + i * 1024), 16, 0, 0);
  };

  // ---- fragment read offsets: row = base + t*32 + fr with base a multiple of 32 => swizzle depends on fr only
  const int fr = lane & 31, half = lane >> 5;
  uint32_t a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const uint32_t sw = (uint32_t)(((ks * 2 + half) ^ ((fr >> 2) & 3)) << 4);
    a_off[ks] = (uint32_t)((wm * 128 + fr) * 64) + sw;
    b_off[ks] = (uint32_t)((wn * 64 + fr) * 64) + sw + 16384u;
  }

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue(0);
  if (nt > 1) issue(1);
  if (DMA_IN_MSEG && nt > 2) issue(2);
  if (DMA_IN_MSEG && DIST > 3 && nt > 3) issue(3);
  {
    const int issued = DMA_IN_MSEG ? min(nt, DIST) : min(nt, 2);
    if (issued >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (issued == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (issued == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0

  // One K tile.  STEADY (compile time) = this is not one of the last DIST tiles: the DMA of tile p+DIST is issued
  // unconditionally and the counted wait is a constant - no scalar branch or select ends up between the MFMAs
  // (the rolled-up form had one per DMA piece; hipcc materialised each `if (dma)` as s_cbranch + v_cndmask chains).
  auto tile = [&](const int p, auto steady_c) {
    constexpr bool STEADY = decltype(steady_c)::value;
    const uint8_t* st = smem + ((ABLATE == 2 ? 0 : p) % NST) * G2_STAGE_BYTES;
    bf16x8_t af[2][4], bfr[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int t = 0; t < 2; ++t) bfr[ks][t] = *(const bf16x8_t*)(st + b_off[ks] + t * 2048);
#pragma unroll
      for (int t = 0; t < 4; ++t) af[ks][t] = *(const bf16x8_t*)(st + a_off[ks] + t * 2048);
    }
    if (!DMA_IN_MSEG && p + 2 < nt && ABLATE != 1) issue(p + 2);
    if (SPLIT == 1) {
      if (p + DIST < nt) {
        issue_piece(p + DIST, 0);
        issue_piece(p + DIST, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // tile p+2 (4) + the two pieces just issued
      } else if (p + 2 < nt) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      // tiles newer than p+1 that are already in flight HERE (each = 4 DMA pieces of this wave): with DMA_IN_MSEG
      // tile p+DIST is only issued in the M segment that follows, so they are p+2 .. p+DIST-1; otherwise p+2.
      // (An earlier revision counted DIST-1 and thereby waited for tile p only - results then depended on the DMA
      // beating the consumer by three iterations, which a memory-bound launch mix can break.)
      const int newer = STEADY ? DIST - 2 : ((ABLATE == 1) ? 0 : min(DMA_IN_MSEG ? DIST - 2 : 1, nt - 2 - p));
      if (newer >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (newer == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    const bool dma = STEADY || (DMA_IN_MSEG && (p + DIST < nt) && ABLATE != 1);
    // SLOT = the MFMA slot (mod 4) after which this wave issues one LDS-DMA piece (compile-time: a scalar branch
    // between MFMAs costs ~9 %, and four slot-specialised copies of the segment made hipcc spill the accumulators).
    auto mseg = [&](auto slot_c) {
      constexpr int SLOT = decltype(slot_c)::value;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][tn], af[ks][tm], acc[tm][tn], 0, 0, 0);
            const int k = (ks * 4 + tm) * 2 + tn;
            if (DMA_IN_MSEG && SPLIT != 1 && (k & 3) == SLOT) {
              __builtin_amdgcn_sched_barrier(0);
              if (STEADY && !EXT) issue_piece_run(p + DIST, k >> 2);
              else if (dma) issue_piece(p + DIST, k >> 2);
              __builtin_amdgcn_sched_barrier(0);
            }
            if (DMA_IN_MSEG && SPLIT == 1 && (k & 7) == 3) {
              __builtin_amdgcn_sched_barrier(0);
              if (dma) issue_piece(p + DIST, 2 + (k >> 3));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
    };
    // (measured and dropped: staggering the four SIMDs' DMA slots - by a scalar branch per slot -9 %, by slot-specialised
    //  copies of the segment -> accumulator spills, by a 16-clk time skew per SIMD -11 %: the closing barrier pays for it)
    mseg(std::integral_constant<int, 1>{});
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    int p = 0;
    if (DMA_IN_MSEG && SPLIT == 0 && ABLATE == 0)
      for (; p + DIST < nt; ++p) {
        tile(p, std::true_type{});
        a_run[0] += G2_BK; a_run[1] += G2_BK; b_run[0] += G2_BK; b_run[1] += G2_BK;
      }
    for (; p < nt; ++p) tile(p, std::false_type{});
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();   // re-balance the barrier count

  epi.apply(*reinterpret_cast<f32x16_t(*)[2][2]>(&acc[0]), m0 + wm * 128, n0 + wn * 64, lane, g.M, g.N);
  epi.apply(*reinterpret_cast<f32x16_t(*)[2][2]>(&acc[2]), m0 + wm * 128 + 64, n0 + wn * 64, lane, g.M, g.N);
}

// =============================================================================================
// 256x256x64 ping-pong kernel: same two-group schedule, but a K tile is 64 deep so every LDS-DMA lane group
// reads FULL 128-byte lines from L2 (the 32-deep tile reads half lines twice).  Two 64 KiB stages; a tile is
// consumed in two phases (k halves); all 8 DMA pieces of tile t+1 are issued between the MFMAs of phase 2t.
//   WAR stage (t+1)&1 (last read in phase 2t-1): G0 issues after instance 4t = G1.Y(2t-1), G1 after 4t+1.
//   RAW tile t+1 (first read in L-seg(2t+2)): every wave drains its DMA (vmcnt(0)) before its X(2t+1).
// =============================================================================================
#define G3_STAGE_BYTES 65536
#define G3_LDS_BYTES (2 * G3_STAGE_BYTES)

template <class Epi>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm_nt_256x64_kernel(GemmShape g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const int tiles_m = (g.M + G2_BM - 1) / G2_BM, tiles_n = (g.N + G2_BN - 1) / G2_BN;
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  constexpr int GROUP = 8;
  const int group_size = GROUP * tiles_n;
  const int first_m = (id / group_size) * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int tile_m = first_m + (id % group_size) % gsz;
  const int tile_n = (id % group_size) / gsz;
  const int m0 = tile_m * G2_BM, n0 = tile_n * G2_BN;

  // 4 pieces (1 KiB = 8 rows x 128 B) of A and of B per wave per K tile
  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((row >> 1) & 7);
    a_src[i] = g.A + (long)min(m0 + row, g.M - 1) * g.lda + kc * 8;
    b_src[i] = g.B + (long)min(n0 + row, g.N - 1) * g.ldb + kc * 8;
  }
  const uint32_t piece0 = (uint32_t)(wave * 4) * 1024u;
  auto issue_piece = [&](int t, int j) {   // j = 0..7: A piece j>>1 for even j, B piece j>>1 for odd j
    uint8_t* st = smem + (t & 1) * G3_STAGE_BYTES;
    const long koff = (long)t * 64;
    const int i = j >> 1;
    if ((j & 1) == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + koff),
                                       (__attribute__((address_space(3))) void*)(st + piece0 + i * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + koff),
                                       (__attribute__((address_space(3))) void*)(st + 32768 + piece0 + i * 1024), 16, 0, 0);
  };

  const int fr = lane & 31, half = lane >> 5;
  uint32_t a_off[2], b_off[2];   // k-substep ks of the FIRST half tile; the second half is the same offset ^ 64
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const uint32_t sw = (uint32_t)(((ks * 2 + half) ^ ((fr >> 1) & 7)) << 4);
    a_off[ks] = (uint32_t)((wm * 128 + fr) * 128) + sw;
    b_off[ks] = (uint32_t)((wn * 64 + fr) * 128) + sw + 32768u;
  }

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = g.K / 64;
#pragma unroll
  for (int j = 0; j < 8; ++j) issue_piece(0, j);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();

  for (int t = 0; t < nt; ++t) {
    const uint8_t* st = smem + (t & 1) * G3_STAGE_BYTES;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8_t af[2][4], bfr[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int u = 0; u < 2; ++u) bfr[ks][u] = *(const bf16x8_t*)(st + ((b_off[ks] ^ (h * 64)) + u * 4096));
#pragma unroll
        for (int u = 0; u < 4; ++u) af[ks][u] = *(const bf16x8_t*)(st + ((a_off[ks] ^ (h * 64)) + u * 4096));
      }
      if (h == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      const bool dma = (h == 0) && (t + 1 < nt);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][tn], af[ks][tm], acc[tm][tn], 0, 0, 0);
            const int k = (ks * 4 + tm) * 2 + tn;
            if (h == 0 && (k & 1) == 1) {
              __builtin_amdgcn_sched_barrier(0);
              if (dma) issue_piece(t + 1, k >> 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();

  epi.apply(*reinterpret_cast<f32x16_t(*)[2][2]>(&acc[0]), m0 + wm * 128, n0 + wn * 64, lane, g.M, g.N);
  epi.apply(*reinterpret_cast<f32x16_t(*)[2][2]>(&acc[2]), m0 + wm * 128 + 64, n0 + wn * 64, lane, g.M, g.N);
}

// =============================================================================================
// TN GEMM (weight gradients):  out[i][j] = sum_r P[r][i] * Q[r][j]   (P [R][I], Q [R][J] row-major, the
// contraction index r is the ROW of both operands - e.g. dW = dY^T X with r = token).
// Same 256x256 ping-pong schedule as gemm_nt_256_kernel; the operand tiles are staged exactly as they lie
// in memory ([32 r][256 cols], 512-byte rows, coalesced 512-byte global segments) and the MFMA fragments
// ("8 consecutive r for one column") come from ds_read_b64_tr_b16, gfx950's transposing LDS read:
// inside a 16-lane group lane t receives element (t&3) of the 8-byte chunk addressed by lane 4j+(t>>2),
// j = 0..3 - so when lane s addresses row (s>>2), columns 4(s&3).., lane t ends up with 4 consecutive rows of
// column t.  No transposed copies of dY / X are ever written to HBM.
// LDS swizzle: the 64-byte block b of row r lives at block b ^ (r & 3) (4 consecutive rows of a transposing
// read hit 4 different quarters of the 256-byte bank row); applied on the DMA source address.
// Rows r >= R of the last tile are redirected to a zero row (zero_row: >= 512 zero bytes in HBM).
// =============================================================================================
// ---------------------------------------------------------------------------------------------
// 16x16x32 MFMA variant ("MI16").  profiles/r02_mfma_shape_power_probe.log: under the package power cap a pure
// v_mfma_f32_16x16x32_bf16 loop sustains 1953 TFLOP/s at 2.03 GHz, v_mfma_f32_32x32x16_bf16 1750 at 1.85 GHz: the small
// shape touches half as many accumulator registers per flop, and these GEMMs are power-bound.
// The main loop is the same schedule with 32 MFMAs of 4 passes per phase instead of 16 of 8; the epilogues still see the
// 32x32 accumulator layout - acc16_block_to_acc32() moves a 64 x 64 block of the wave's tile from
//     16x16 tiles:  lane l, register j of tile (t_m, t_n)  <->  m = 16 t_m + (l & 15),  n = 16 t_n + 4 (l >> 4) + j
// to  32x32 tiles:  lane L, register r of tile (T_m, T_n)  <->  m = 32 T_m + (L & 31),  n = 32 T_n + (r & 3) + 8 (r >> 2) + 4 (L >> 5)
// i.e. target (L, r) = source tile (2 T_m + ((L >> 4) & 1), 2 T_n + (r >> 3)), lane (L & 15) | ((L >> 5) << 4) | (((r >> 2) & 1) << 5),
// register r & 3: two ds_bpermute + one select per register, 256 per wave per output tile (~1 % of a K = 4096 tile).
// ---------------------------------------------------------------------------------------------
// (element access through a helper: __builtin_bit_cast(int, vec[j]) on a vector-element lvalue reads element 0 for every j
//  with this hipcc - seen as 64 instead of 256 ds_bpermute in the ISA and as "every 4 columns equal" on the GPU)
__device__ __forceinline__ int acc16_elem_bits(const f32x4_t& v, int j) {
  const float f = j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
  return __builtin_bit_cast(int, f);
}
// Two lane-swap instructions per register pair instead: with a = tile row bit, p / q = source lane bits 4 / 5 the move is the
// rotation (a -> lane bit 4, p -> lane bit 5, q -> register bit 2):
//   v_permlane16_swap(S[a=0], S[a=1])  exchanges the register bit a with lane bit 4  (odd 16-lane rows of the first operand <->
//                                      even rows of the second): Y[p] holds a in lane bit 4;
//   v_permlane32_swap(Y[0], Y[1])      exchanges the register bit p with lane bit 5  (upper half of the first <-> lower half of
//                                      the second): Z[q] holds p in lane bit 5;   target register r = 8 b + 4 q + j.
// 128 VALU swaps per wave per output tile (RV_GEMM_MI16_BPERMUTE: the ds_bpermute form above, 256 + 128 selects).
__device__ __forceinline__ void acc16_block_to_acc32(const f32x4_t (&a)[4][4], f32x16_t (&out)[2][2], int lane) {
#ifdef RV_GEMM_MI16_BPERMUTE
  const int base = (lane & 15) | ((lane >> 5) << 4);
  const bool odd = (lane >> 4) & 1;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int src_lane = base | (((r >> 2) & 1) << 5);
        const int v0 = __builtin_amdgcn_ds_bpermute(src_lane << 2, acc16_elem_bits(a[2 * tm][2 * tn + (r >> 3)], r & 3));
        const int v1 = __builtin_amdgcn_ds_bpermute(src_lane << 2, acc16_elem_bits(a[2 * tm + 1][2 * tn + (r >> 3)], r & 3));
        out[tm][tn][r] = __builtin_bit_cast(float, odd ? v1 : v0);
      }
#else
  (void)lane;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned s0 = (unsigned)acc16_elem_bits(a[2 * tm][2 * tn + b], j);
          const unsigned s1 = (unsigned)acc16_elem_bits(a[2 * tm + 1][2 * tn + b], j);
          const u32x2_t y = __builtin_amdgcn_permlane16_swap(s0, s1, false, false);
          const u32x2_t z = __builtin_amdgcn_permlane32_swap(y[0], y[1], false, false);
          out[tm][tn][8 * b + j] = __builtin_bit_cast(float, (unsigned)z[0]);
          out[tm][tn][8 * b + 4 + j] = __builtin_bit_cast(float, (unsigned)z[1]);
        }
#endif
}

__device__ bf16_t g_zero_row[256];   // 512 zero bytes: DMA source for contraction rows beyond R

// Output tile of launch-order position `bid` (XCD-aware remap + grouped order; sweep at 27.6 k tokens: GROUP 4 beats 8 by
// 1.6-3.5 %, 16 loses another 2 %).  Shared with the tail-split reduce kernel, which must walk the same tiles.
__device__ __forceinline__ void tn_tile_origin(int bid, int I, int J, int group, int& i0, int& j0) {
  const int tiles_i = (I + 255) / 256, tiles_j = (J + 255) / 256;
  const int nwg = tiles_i * tiles_j;
  const int id = xcd_remap(bid, nwg);
  const int GROUP = group > 0 ? group : 4;
  const int group_size = GROUP * tiles_j;
  const int first_i = (id / group_size) * GROUP;
  const int gsz = min(tiles_i - first_i, GROUP);
  i0 = (first_i + (id % group_size) % gsz) * 256;
  j0 = ((id % group_size) / gsz) * 256;
}

struct EpiStoreF32;
// An epilogue that wants a wave's 64 output columns as TWO 32-column blocks 64 columns apart (the half-split RoPE pairs (i, i + 64) of
// a 128-wide head in the same lane and register: EpiStoreRope) declares `static constexpr bool kRopeCols = true`.
template <class E, class = void> struct EpiRopeCols : std::false_type {};
template <class E> struct EpiRopeCols<E, std::void_t<decltype(E::kRopeCols)>> : std::bool_constant<E::kRopeCols> {};

template <class Epi, int DIST = 3, bool MI16 = false>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm_tn_256_kernel(const bf16_t* __restrict__ P, long ldp,
                                                                    const bf16_t* __restrict__ Q, long ldq, int R,
                                                                    int I, int J, Epi epi, int r_chunk,
                                                                    long split_stride, int group, int bid0) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int NST = DIST + 1;
  int i0, j0;
  tn_tile_origin((int)blockIdx.x + bid0, I, J, group, i0, j0);
  if (gridDim.y > 1) {   // split-K: workgroup row y reduces contraction rows [y*r_chunk, (y+1)*r_chunk) into its own slab
    const long r0 = (long)blockIdx.y * r_chunk;
    P += r0 * ldp;
    Q += r0 * ldq;
    R = (int)min((long)r_chunk, (long)R - r0);
    if constexpr (std::is_same<Epi, EpiStoreF32>::value) {
      if (split_stride < 0) {   // tail split (rv_gemm_tn_bf16_ws): tile-dense slabs [split][tail tile][256][256]
        epi.C += ((long)blockIdx.y * gridDim.x + blockIdx.x) * 65536 - ((long)i0 * 256 + j0);
        epi.ldc = 256;
      } else {
        epi.C += (long)blockIdx.y * split_stride;
      }
    } else {
      epi.C += (long)blockIdx.y * split_stride;
    }
  }
  const bf16_t* zero_row = g_zero_row;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 2, wj = wave & 3;

  // ---- LDS-DMA: a stage = P tile (16 KiB) + Q tile (16 KiB); one piece = 2 rows x 512 B; 2 pieces each per wave
  int p_row[2];   // tile row filled by this lane in piece i
#pragma unroll
  for (int i = 0; i < 2; ++i) p_row[i] = (wave * 2 + i) * 2 + (lane >> 5);
  // source column (elements) of LDS chunk c = lane&31 of row r: 64-byte block (c>>2) ^ (r&3), 16-byte slot c&3
  auto src_col = [&](int r, int base, int limit) {
    const int c = lane & 31;
    // MI16: rows 8..15 / 24..31 swap the 32-byte halves of every 64-byte block (see gemm_nn_a64_kernel)
    const int slot = MI16 ? ((c & 3) ^ (((r >> 3) & 1) << 1)) : (c & 3);
    const int col = (((c >> 2) ^ (r & 3)) << 5) + (slot << 3);
    return min(base + col, limit - 8);
  };
  const uint32_t piece0 = (uint32_t)(wave * 2) * 1024u;
  auto issue_piece = [&](int t, int jj) {   // jj: 0 = P piece 0, 1 = Q piece 0, 2 = P piece 1, 3 = Q piece 1
    uint8_t* st = smem + (t % NST) * G2_STAGE_BYTES;
    const int i = jj >> 1;
    const int r = p_row[i];
    const long rg = (long)t * 32 + r;
    const bool valid = rg < R;
    if ((jj & 1) == 0) {
      const bf16_t* src = valid ? (P + rg * ldp + src_col(r, i0, I)) : (zero_row + ((lane & 31) << 3));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + piece0 + i * 1024), 16, 0, 0);
    } else {
      const bf16_t* src = valid ? (Q + rg * ldq + src_col(r, j0, J)) : (zero_row + ((lane & 31) << 3));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + 16384 + piece0 + i * 1024), 16, 0, 0);
    }
  };
  auto issue = [&](int t) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) issue_piece(t, jj);
  };

  // ---- transposing fragment reads
  const int g4 = lane >> 4, s16 = lane & 15;
  const uint32_t lane_part = (uint32_t)((8 * (g4 >> 1) + (s16 >> 2)) * 512 + 32 * (g4 & 1) + 8 * (s16 & 3));
  uint32_t q_blk[2], p_blk[4];
#pragma unroll
  for (int t = 0; t < 2; ++t) q_blk[t] = lane_part + 16384u + (uint32_t)((((wj * 2 + t) ^ (s16 >> 2))) << 6);
#pragma unroll
  for (int t = 0; t < 4; ++t) p_blk[t] = lane_part + (uint32_t)((((wi * 4 + t) ^ (s16 >> 2))) << 6);

  // MI16 fragments (16x16x32 MFMAs): contraction rows 8 g4 + (s16 >> 2) (+4), 16 columns = half (t & 1) of 64-byte block t >> 1
  const uint32_t lane16 = (uint32_t)((8 * g4 + (s16 >> 2)) * 512 + 8 * (s16 & 3));
  uint32_t q16[4], p16[8];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    q16[t] = lane16 + 16384u + (uint32_t)(((wj * 2 + (t >> 1)) ^ (s16 >> 2)) << 6) + (uint32_t)(((t & 1) ^ (g4 & 1)) << 5);
#pragma unroll
  for (int t = 0; t < 8; ++t)
    p16[t] = lane16 + (uint32_t)(((wi * 4 + (t >> 1)) ^ (s16 >> 2)) << 6) + (uint32_t)(((t & 1) ^ (g4 & 1)) << 5);

  f32x16_t acc[MI16 ? 1 : 4][2];
  f32x4_t acc16[MI16 ? 8 : 1][4];
  if (MI16) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc16[MI16 ? i : 0][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[MI16 ? 0 : i][j][r] = 0.f;
  }

  const int nt = (R + 31) / 32;
  issue(0);
  if (nt > 1) issue(1);
  if (nt > 2) issue(2);
  if (DIST > 3 && nt > 3) issue(3);
  {
    const int issued = min(nt, DIST);
    if (issued >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (issued == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (issued == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wi == 1) __builtin_amdgcn_s_barrier();
  if (RV_GEMM_PRIO_TN == 1 && wi == 1) __builtin_amdgcn_s_setprio(1);     // wi is wave-uniform (readfirstlane)

  // STEADY tiles (all but the last DIST+1): the fetched tile p+DIST is neither past the end nor ragged, so its DMA
  // needs no validity select and no branch, and uses pointers advanced once per tile (see gemm_nt_256_kernel)
  const bf16_t* p_run[2];
  const bf16_t* q_run[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    p_run[i] = P + ((long)DIST * 32 + p_row[i]) * ldp + src_col(p_row[i], i0, I);
    q_run[i] = Q + ((long)DIST * 32 + p_row[i]) * ldq + src_col(p_row[i], j0, J);
  }
  auto issue_piece_run = [&](int t, int jj) {
    uint8_t* st = smem + (t % NST) * G2_STAGE_BYTES;
    const int i = jj >> 1;
    if ((jj & 1) == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p_run[i],
                                       (__attribute__((address_space(3))) void*)(st + piece0 + i * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)q_run[i],
                                       (__attribute__((address_space(3))) void*)(st + 16384 + piece0 + i * 1024), 16, 0, 0);
  };
  auto tile = [&](const int p, auto steady_c) {
    constexpr bool STEADY = decltype(steady_c)::value;
    const uint8_t* st = smem + (p % NST) * G2_STAGE_BYTES;
    bf16x8_t qf[2][2], pf[2][4];            // MI16 views: qf16[t] = qf[t >> 1][t & 1], pf16[t] = pf[t >> 2][t & 3]
    const uint32_t sta = lds_addr_of(st);
    if (MI16) {
#pragma unroll
      for (int t = 0; t < 4; ++t) qf[t >> 1][t & 1] = ds_tr16_pair_asm(sta + q16[t], 0, 2048);
#pragma unroll
      for (int t = 0; t < 8; ++t) pf[t >> 2][t & 3] = ds_tr16_pair_asm(sta + p16[t], 0, 2048);
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t a0 = sta + q_blk[t];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks][t] = ds_tr16_pair_asm(a0, ks * 8192, ks * 8192 + 2048);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t a0 = sta + p_blk[t];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) pf[ks][t] = ds_tr16_pair_asm(a0, ks * 8192, ks * 8192 + 2048);
      }
    }
    {
      const int newer = STEADY ? DIST - 2 : min(DIST - 2, nt - 2 - p);   // tiles p+2 .. p+DIST-1 (p+DIST comes after this wait)
      if (newer >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (RV_GEMM_PRIO_TN == 0) __builtin_amdgcn_s_setprio(1);
    const bool dma = (p + DIST < nt);
#pragma unroll
    for (int kk = 0; kk < (MI16 ? 32 : 16); ++kk) {
      if (MI16) {
        const int ti = kk >> 2, tj = kk & 3;
        acc16[MI16 ? ti : 0][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[tj >> 1][tj & 1], pf[ti >> 2][ti & 3],
                                                                        acc16[MI16 ? ti : 0][tj], 0, 0, 0);
      } else {
        const int ks = kk >> 3, ti = (kk >> 1) & 3, tj = kk & 1;
        acc[MI16 ? 0 : ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks][tj], pf[ks][ti], acc[MI16 ? 0 : ti][tj], 0, 0, 0);
      }
      const int k = MI16 ? (kk >> 1) : kk;                       // DMA slots: every 4th (MI16: 8th) MFMA
      if ((k & 3) == RV_GEMM_DMA_SLOT && (!MI16 || (kk & 1) == 1)) {
        __builtin_amdgcn_sched_barrier(0);
        if (STEADY) issue_piece_run(p + DIST, k >> 2);
        else if (dma) issue_piece(p + DIST, k >> 2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (RV_GEMM_PRIO_TN == 0) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    int p = 0;
    for (; p + DIST < nt - 1; ++p) {
      tile(p, std::true_type{});
      p_run[0] += 32 * ldp; p_run[1] += 32 * ldp; q_run[0] += 32 * ldq; q_run[1] += 32 * ldq;
    }
    for (; p < nt; ++p) tile(p, std::false_type{});
  }
  if (wi == 0) __builtin_amdgcn_s_barrier();

  if (MI16) {
#pragma unroll
    for (int hm = 0; hm < 2; ++hm) {
      f32x16_t blk[2][2];
      acc16_block_to_acc32(*reinterpret_cast<f32x4_t(*)[4][4]>(&acc16[MI16 ? 4 * hm : 0]), blk, lane);
      epi.apply(blk, i0 + wi * 128 + hm * 64, j0 + wj * 64, lane, I, J);
    }
  } else {
    epi.apply(*reinterpret_cast<f32x16_t(*)[2][2]>(&acc[0]), i0 + wi * 128, j0 + wj * 64, lane, I, J);
    epi.apply(*reinterpret_cast<f32x16_t(*)[2][2]>(&acc[MI16 ? 0 : 2]), i0 + wi * 128 + 64, j0 + wj * 64, lane, I, J);
  }
}

// =============================================================================================
// NN GEMM:  D[m][n] = sum_k A[m][k] * B[k][n]   (A [M][K] K-contiguous, B [K][N] N-contiguous, both row-major).
// The A side is gemm_nt_256_kernel's (64-byte row slices by LDS-DMA, ds_read_b128 fragments), the B side is
// gemm_tn_256_kernel's Q operand: the [32 k][256 n] tile is staged as it lies in memory - FULL 512-byte row segments per
// DMA piece - and the "8 consecutive k of one column" fragments come from ds_read_b64_tr_b16.  Full-line requests are
// what the L2->LDS fabric likes (profiles/r01_glds_probe.log: 34 vs 20 TB/s chip-wide), which is why the TN kernel
// outruns the NT kernel; this kernel gives the weight operand of every forward / input-gradient GEMM the same
// treatment: forward runs on the W^T copy ([in][out]), the input gradient on W itself ([out][in]).
// Same ping-pong schedule, 4-stage ring, counted vmcnt and hazards as gemm_nt_256_kernel.  K % 32 == 0.
// =============================================================================================
// EXT: second contraction segment (GemmShape::A2 [M][..] K2-contiguous slices, B2 [K2][N] row-major) - the fused LoRA form.
template <class Epi, bool EXT = false>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm_nn_256_kernel(GemmShape g, Epi epi) {
  constexpr int NST = 4, DIST = 3;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const int tiles_m = (g.M + G2_BM - 1) / G2_BM, tiles_n = (g.N + G2_BN - 1) / G2_BN;
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int GROUP = g.group > 0 ? g.group : 4;
  const int group_size = GROUP * tiles_n;
  const int first_m = (id / group_size) * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int tile_m = first_m + (id % group_size) % gsz;
  const int tile_n = (id % group_size) / gsz;
  const int m0 = tile_m * G2_BM, n0 = tile_n * G2_BN;
  const long ldb = g.ldb;

  // ---- A pieces: 16 rows x 64 B (as gemm_nt_256_kernel); B pieces: 2 k-rows x 512 B (as gemm_tn_256_kernel's Q)
  const bf16_t* a_src[2];
  const bf16_t* b_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 16 + (lane >> 2);
    const int kc = (lane & 3) ^ ((row >> 2) & 3);
    a_src[i] = g.A + (long)min(m0 + row, g.M - 1) * g.lda + kc * 8;
    const int r = (wave * 2 + i) * 2 + (lane >> 5);              // k row of the tile filled by this lane
    const int c = lane & 31;
    const int col = (((c >> 2) ^ (r & 3)) << 5) + ((c & 3) << 3);   // 64-byte block (c>>2) ^ (r&3), 16-byte slot c&3
    b_src[i] = g.B + (long)r * ldb + min(n0 + col, g.N - 8);
  }
  const uint32_t piece0 = (uint32_t)(wave * 2) * 1024u;
  const int nt1 = g.K / G2_BK;
  const int nt = nt1 + (EXT ? g.K2 / G2_BK : 0);
  const int a2_col0 = (EXT && g.group_cols > 0) ? ext_group(n0, g.group0, g.group_cols) * g.K2 : 0;
  auto issue_piece = [&](int t, int j) {   // j: 0 = A piece 0, 1 = B piece 0, 2 = A piece 1, 3 = B piece 1
    uint8_t* st = smem + (t % NST) * G2_STAGE_BYTES;
    const int i = j >> 1;
    if (EXT && t >= nt1) {     // wave-uniform: tiles of the second segment, addresses rebuilt on the fly
      const bf16_t* src;
      if ((j & 1) == 0) {
        const int row = (wave * 2 + i) * 16 + (lane >> 2);
        src = g.A2 + (long)min(m0 + row, g.M - 1) * g.lda2 + a2_col0 + (long)(t - nt1) * G2_BK +
              (((lane & 3) ^ ((row >> 2) & 3)) << 3);
      } else {
        const int r = (wave * 2 + i) * 2 + (lane >> 5), c = lane & 31;
        src = g.B2 + ((long)(t - nt1) * G2_BK + r) * g.ldb2 + min(n0 + (((c >> 2) ^ (r & 3)) << 5) + ((c & 3) << 3), g.N - 8);
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + ((j & 1) ? 16384 : 0) + piece0 + i * 1024),
                                       16, 0, 0);
    } else if ((j & 1) == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + (long)t * G2_BK),
                                       (__attribute__((address_space(3))) void*)(st + piece0 + i * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + (long)t * G2_BK * ldb),
                                       (__attribute__((address_space(3))) void*)(st + 16384 + piece0 + i * 1024), 16, 0, 0);
  };
  auto issue = [&](int t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_piece(t, j);
  };
  const bf16_t* a_run[2] = {a_src[0] + DIST * G2_BK, a_src[1] + DIST * G2_BK};
  const bf16_t* b_run[2] = {b_src[0] + (long)DIST * G2_BK * ldb, b_src[1] + (long)DIST * G2_BK * ldb};
  auto issue_piece_run = [&](int t, int j) {
    uint8_t* st = smem + (t % NST) * G2_STAGE_BYTES;
    const int i = j >> 1;
    if ((j & 1) == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_run[i],
                                       (__attribute__((address_space(3))) void*)(st + piece0 + i * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_run[i],
                                       (__attribute__((address_space(3))) void*)(st + 16384 + piece0 + i * 1024), 16, 0, 0);
  };

  // ---- A fragments (ds_read_b128) and B fragments (transposing reads)
  const int fr = lane & 31, half = lane >> 5;
  uint32_t a_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    a_off[ks] = (uint32_t)((wm * 128 + fr) * 64) + (uint32_t)(((ks * 2 + half) ^ ((fr >> 2) & 3)) << 4);
  const int g4 = lane >> 4, s16 = lane & 15;
  const uint32_t lane_part = (uint32_t)((8 * (g4 >> 1) + (s16 >> 2)) * 512 + 32 * (g4 & 1) + 8 * (s16 & 3));
  uint32_t q_blk[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) q_blk[t] = lane_part + 16384u + (uint32_t)((((wn * 2 + t) ^ (s16 >> 2))) << 6);

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue(0);
  if (nt > 1) issue(1);
  if (nt > 2) issue(2);
  {
    const int issued = min(nt, DIST);
    if (issued == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (issued == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0

  auto tile = [&](const int p, auto steady_c) {
    constexpr bool STEADY = decltype(steady_c)::value;
    const uint8_t* st = smem + (p % NST) * G2_STAGE_BYTES;
    bf16x8_t af[2][4], bfr[2][2];
    const uint32_t sta = lds_addr_of(st);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint32_t a0 = sta + q_blk[t];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bfr[ks][t] = ds_tr16_pair_asm(a0, ks * 8192, ks * 8192 + 2048);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) af[ks][t] = *(const bf16x8_t*)(st + a_off[ks] + t * 2048);
    {
      const int newer = STEADY ? DIST - 2 : min(DIST - 2, nt - 2 - p);
      if (newer >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    const bool dma = STEADY || (p + DIST < nt);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][tn], af[ks][tm], acc[tm][tn], 0, 0, 0);
          const int k = (ks * 4 + tm) * 2 + tn;
          if ((k & 3) == 1) {
            __builtin_amdgcn_sched_barrier(0);
            if (STEADY) issue_piece_run(p + DIST, k >> 2);
            else if (dma) issue_piece(p + DIST, k >> 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    int p = 0;
    for (; p + DIST < nt1; ++p) {      // steady: the fetched tile p+DIST lies in the main segment
      tile(p, std::true_type{});
      a_run[0] += G2_BK; a_run[1] += G2_BK; b_run[0] += G2_BK * ldb; b_run[1] += G2_BK * ldb;
    }
    for (; p < nt; ++p) tile(p, std::false_type{});
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();   // re-balance the barrier count

  epi.apply(*reinterpret_cast<f32x16_t(*)[2][2]>(&acc[0]), m0 + wm * 128, n0 + wn * 64, lane, g.M, g.N);
  epi.apply(*reinterpret_cast<f32x16_t(*)[2][2]>(&acc[2]), m0 + wm * 128 + 64, n0 + wn * 64, lane, g.M, g.N);
}


// =============================================================================================
// NN GEMM with FULL-LINE fetches on BOTH operands ("A64"): as gemm_nn_256_kernel, but the A operand is staged in
// 64-deep tiles - one LDS-DMA piece = 8 rows x 128 B, a whole cache line per row - held in a 3-stage ring of its own
// (3 x 32 KiB) next to the 4-stage ring of 32-deep B tiles (4 x 16 KiB): 160 KiB of LDS, the CU's whole allocation.
// Phase p (32 of K) computes on A tile p>>1 (k half p&1) and B tile p.  Its M segment issues, interleaved A,B,A,B:
//     A pieces 2(p&1), 2(p&1)+1 of A tile (p>>1)+2      and      B pieces 0, 1 of B tile p+3.
// Hazards (barrier instances as in gemm_nt_256_kernel; loads retire in order, every M segment issues exactly 4):
//   RAW: the wait of L-seg(p) is vmcnt(4) = "everything issued up to M-seg(p-2) has landed" - that covers B tile p+1
//        (issued in M-seg(p-2)) and A tile (p+1)>>1 (issued in M-segs 2((p+1)>>1)-4, -3 <= p-2), one barrier or more
//        before any group reads them.  Prologue order A0 B0 | A1 B1 B2: vmcnt(8) before the first barrier, vmcnt(2) in
//        L-seg(0).
//   WAR: A stage ((p>>1)+2)%3 = ((p>>1)-1)%3 was last read in L-seg(2(p>>1)-1) <= L-seg(p-1), retired by both groups
//        before G1.X(p-1) = barrier instance 2p-1, which precedes every M-seg(p).  B ring as in the NT kernel.
// Requires K % 64 == 0.
// =============================================================================================
#define G4_A_STAGE 32768
#define G4_B_STAGE 16384
#define G4_LDS_BYTES (3 * G4_A_STAGE + 4 * G4_B_STAGE)

// EXT: second contraction segment (A2 slices / B2 [K2][N], K2 % 64 == 0: the fused LoRA form).  The steady loop only
// covers phases whose fetches lie in the main segment; the few phases around the seam and the adapter's own K2/32
// phases run in the generic form (addresses rebuilt on the fly, full drain per phase).
// PRE: adapter-FIRST form (GemmShape): the K2 segment (A2 [M][K2], B2 [K2][N]) runs before the main loop, 64 of K2 per step
// through A stage 2 / B stages 2, 3 with all 8 waves in step (the main prologue - A stages 0, 1, B stages 0, 1 - is issued with
// the last step and lands under it), the dropout mask is applied to the accumulators while the main prologue is in flight,
// then the unchanged main loop accumulates on top.  Replaces a second pass over the [M][N] output (rv_gemm_nt_dropout_bf16 with
// the output as its own residual: 2 x M x N x 2 B of HBM traffic per projection).
template <class Epi, bool EXT = false, bool MI16 = false, bool PRE = false>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm_nn_a64_kernel(GemmShape g, Epi epi) {
  static_assert(!(EXT && PRE), "the adapter segment runs either first (PRE) or last (EXT)");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const smA = smem;
  uint8_t* const smB = smem + 3 * G4_A_STAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const int tiles_m = (g.M + G2_BM - 1) / G2_BM, tiles_n = (g.N + G2_BN - 1) / G2_BN;
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int GROUP = g.group > 0 ? g.group : 4;
  const int group_size = GROUP * tiles_n;
  const int first_m = (id / group_size) * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int tile_m = first_m + (id % group_size) % gsz;
  const int tile_n = (id % group_size) / gsz;
  const int m0 = tile_m * G2_BM, n0 = tile_n * G2_BN;
  const long ldb = g.ldb;

  // ---- A pieces: wave w owns pieces 4w .. 4w+3 of a 64-deep tile; piece = 8 rows x 128 B, LDS image row*128 +
  //      ((kc ^ ((row>>1)&7))<<4) (the 128x128 kernel's conflict-free layout), swizzle applied on the source address.
  //      B pieces: 2 k-rows x 512 B, as gemm_nn_256_kernel.
  const bf16_t* a_src[4];
  const bf16_t* b_src[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((row >> 1) & 7);
    a_src[i] = g.A + (long)min(m0 + row, g.M - 1) * g.lda + kc * 8;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 2 + (lane >> 5);
    const int c = lane & 31;
    // MI16: k rows 8..15 and 24..31 additionally swap the two 32-byte halves of every 64-byte block, so that the two
    // 16-lane groups of a transposing read (k rows 8 apart, same 16 columns) hit disjoint banks
    const int slot = MI16 ? ((c & 3) ^ (((r >> 3) & 1) << 1)) : (c & 3);
    const int col = (((c >> 2) ^ (r & 3)) << 5) + (slot << 3);
    b_src[i] = g.B + (long)r * ldb + min(n0 + col, g.N - 8);
  }
  const uint32_t a_piece0 = (uint32_t)(wave * 4) * 1024u, b_piece0 = (uint32_t)(wave * 2) * 1024u;
  auto issue_a = [&](int T, int i, const bf16_t* src) {      // piece i (0..3) of A tile T
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smA + (T % 3) * G4_A_STAGE + a_piece0 + i * 1024),
                                     16, 0, 0);
  };
  auto issue_b = [&](int t, int i, const bf16_t* src) {      // piece i (0..1) of B tile t
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smB + (t % 4) * G4_B_STAGE + b_piece0 + i * 1024),
                                     16, 0, 0);
  };

  // ---- fragments
  const int fr = lane & 31, half = lane >> 5;
  const uint32_t a_row_off = (uint32_t)((wm * 128 + fr) * 128);
  const int a_sw = (fr >> 1) & 7;                               // (row >> 1) & 7 with row = wm*128 + t*32 + fr
  const int g4 = lane >> 4, s16 = lane & 15;
  const uint32_t lane_part = (uint32_t)((8 * (g4 >> 1) + (s16 >> 2)) * 512 + 32 * (g4 & 1) + 8 * (s16 & 3));
  // the wave's two 32-column blocks of the 256-column tile: 2 wn, 2 wn + 1 - or, for the RoPE epilogue, blocks b and b + 2 of head
  // wn >> 1 (columns c .. c + 31 and c + 64 .. c + 95: rotation partners meet in one lane; only the LDS column offset of the B
  // fragments and the epilogue's column arithmetic change, the tile in HBM keeps its layout)
  constexpr bool RC = EpiRopeCols<Epi>::value;
  const int wblk = RC ? ((wn >> 1) * 4 + (wn & 1)) : wn * 2;
  constexpr int wstep = RC ? 2 : 1;
  uint32_t q_blk[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) q_blk[t] = lane_part + (uint32_t)((((wblk + t * wstep) ^ (s16 >> 2))) << 6);

  // MI16 fragments: A tile t_m (16 rows): row wm*128 + 16 t_m + s16, 16-byte chunk 4h + g4 of the 64-deep row;
  // B tile t_n (16 columns): k rows 8 g4 + (s16 >> 2) (+4 for the second read of the pair), 64-byte block wn*2 + (t_n >> 1),
  // half (t_n & 1) ^ (g4 & 1), 8-byte chunk s16 & 3: the transposing read hands lane s16 of the group column s16, 4 k values
  const uint32_t a16_pre = (uint32_t)((wm * 128 + s16) * 128) + (uint32_t)((g4 ^ (s16 >> 1)) << 4);
  uint32_t q16[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    q16[t] = (uint32_t)((8 * g4 + (s16 >> 2)) * 512) + (uint32_t)(((wblk + (t >> 1) * wstep) ^ (s16 >> 2)) << 6) +
             (uint32_t)((((t & 1) ^ (g4 & 1)) << 5) + 8 * (s16 & 3));

  f32x16_t acc[MI16 ? 1 : 4][2];
  f32x4_t acc16[MI16 ? 8 : 1][4];
  if (MI16) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc16[MI16 ? i : 0][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[MI16 ? 0 : i][j][r] = 0.f;
  }

  const int nt1 = g.K / G2_BK;                          // 32-deep phases of the main segment (even)
  const int nt = nt1 + (EXT ? g.K2 / G2_BK : 0);        // + the second segment
  const int ntA1 = nt1 >> 1, ntA = nt >> 1;             // 64-deep A tiles
  const int a2_col0 = (EXT && g.group_cols > 0) ? ext_group(n0, g.group0, g.group_cols) * g.K2 : 0;
  // generic sources (any tile of either segment), rebuilt per piece
  auto a_tile_src = [&](int T, int i) -> const bf16_t* {          // piece i (0..3) of A tile T
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((row >> 1) & 7);
    const long mrow = min(m0 + row, g.M - 1);
    if (EXT && T >= ntA1) return g.A2 + mrow * g.lda2 + a2_col0 + (long)(T - ntA1) * 64 + kc * 8;
    return g.A + mrow * g.lda + (long)T * 64 + kc * 8;
  };
  auto b_tile_src = [&](int t, int i) -> const bf16_t* {          // piece i (0..1) of B tile t
    const int r = (wave * 2 + i) * 2 + (lane >> 5), c = lane & 31;
    const int slot = MI16 ? ((c & 3) ^ (((r >> 3) & 1) << 1)) : (c & 3);
    const int col = min(n0 + (((c >> 2) ^ (r & 3)) << 5) + (slot << 3), g.N - 8);
    if (EXT && t >= nt1) return g.B2 + ((long)(t - nt1) * G2_BK + r) * g.ldb2 + col;
    return g.B + ((long)t * G2_BK + r) * ldb + col;
  };

  // running sources of the pieces issued in M-seg(p): A tile (p>>1)+2, B tile p+3 (clamped to the last tile at the end:
  // the redundant loads land in stages nobody reads any more)
  const bf16_t* a_run[4];
  const bf16_t* b_run[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_run[i] = a_src[i] + 2 * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) b_run[i] = b_src[i] + 3L * G2_BK * ldb;

  // A fragment address of k half h: chunk (4h + 2ks + half) ^ sw = the h = 0 chunk with bit 2 flipped -> offset ^ (h << 6)
  uint32_t a_pre[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) a_pre[ks] = a_row_off + (uint32_t)((((ks * 2 + half) ^ a_sw)) << 4);

  // One phase.  MODE (compile time): 2 = adapter-first step (PRE: computes only, nothing is issued), 1 = steady (issues 2 A + 2 B pieces from the running pointers, counted wait),
  // 0 = generic (issues whatever tiles still exist - B tile p+3, with EXT also A tile (p>>1)+2 - and drains completely).  Only three instantiations exist (steady even, steady odd, tail): more copies of
  // the segment made hipcc spill the accumulators.
  auto phase = [&](const int p, const int h, auto hc, auto mode_c) {
    constexpr int H = decltype(hc)::value;              // = h in steady phases (selects the A pieces to issue)
    constexpr int MODE = decltype(mode_c)::value;
    const uint8_t* stA = smA + ((p >> 1) % 3) * G4_A_STAGE;
    const uint32_t stB = lds_addr_of(smB + (p % 4) * G4_B_STAGE);
    const uint32_t hx = (uint32_t)h << 6;
    bf16x8_t af[2][4], bfr[2][2];           // MI16 views: af16[t_m] = af[t_m >> 2][t_m & 3], bfr16[t_n] = bfr[t_n >> 1][t_n & 1]
    if (MI16) {
#pragma unroll
      for (int t = 0; t < 4; ++t) bfr[t >> 1][t & 1] = ds_tr16_pair_asm(stB + q16[t], 0, 2048);
#pragma unroll
      for (int t = 0; t < 8; ++t) af[t >> 2][t & 3] = *(const bf16x8_t*)(stA + (a16_pre ^ hx) + t * 2048);
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t a0 = stB + q_blk[t];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bfr[ks][t] = ds_tr16_pair_asm(a0, ks * 8192, ks * 8192 + 2048);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) af[ks][t] = *(const bf16x8_t*)(stA + (a_pre[ks] ^ hx) + t * 4096);
    }
    if (MODE == 1) {
      if (p == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // prologue: only B tile 2 may still be in flight
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (MODE == 2) {
      // adapter-first step: its tiles landed (and were published by a barrier) before the fragment reads above; nothing to wait for
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (RV_GEMM_PRIO_NN == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < (MI16 ? 32 : 16); ++kk) {
          if (MI16) {
            const int tm = kk >> 2, tn = kk & 3;
            acc16[MI16 ? tm : 0][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tn >> 1][tn & 1], af[tm >> 2][tm & 3],
                                                                            acc16[MI16 ? tm : 0][tn], 0, 0, 0);
          } else {
            const int ks = kk >> 3, tm = (kk >> 1) & 3, tn = kk & 1;
            acc[MI16 ? 0 : tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][tn], af[ks][tm], acc[MI16 ? 0 : tm][tn], 0, 0, 0);
          }
          const int k = MI16 ? (kk >> 1) : kk;                       // DMA slots: every 4th (MI16: 8th) MFMA
          if (MODE != 2 && (k & 3) == RV_GEMM_DMA_SLOT && (!MI16 || (kk & 1) == 1)) {
            const int j = k >> 2;                                  // 0: A, 1: B, 2: A, 3: B
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 1) {
              if ((j & 1) == 0) issue_a((p >> 1) + 2, 2 * H + (j >> 1), a_run[2 * H + (j >> 1)]);
              else issue_b(p + 3, j >> 1, b_run[j >> 1]);
            } else if ((j & 1) == 1) {
              if (p + 3 < nt) issue_b(p + 3, j >> 1, b_tile_src(p + 3, j >> 1));
            } else if (EXT) {
              if ((p >> 1) + 2 < ntA) issue_a((p >> 1) + 2, 2 * h + (j >> 1), a_tile_src((p >> 1) + 2, 2 * h + (j >> 1)));
            }
            __builtin_amdgcn_sched_barrier(0);
          }
    }
    if (RV_GEMM_PRIO_NN == 0) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  if constexpr (PRE) {
    // ---- adapter-first segment (all 8 waves in step: the group skew starts after it)
    const int nA2 = g.K2 >> 6;
    const int pre_col0 = g.group_cols > 0 ? ext_group(n0, g.group0, g.group_cols) * g.K2 : 0;     // adapter group of this column tile
    for (int st2 = 0; st2 < nA2; ++st2) {
      const bool last = st2 == nA2 - 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int kc = (lane & 7) ^ ((row >> 1) & 7);
        issue_a(2, i, g.A2 + (long)min(m0 + row, g.M - 1) * g.lda2 + pre_col0 + (long)st2 * 64 + kc * 8);
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = (wave * 2 + i) * 2 + (lane >> 5), c = lane & 31;
          const int slot = MI16 ? ((c & 3) ^ (((r >> 3) & 1) << 1)) : (c & 3);
          const int col = min(n0 + (((c >> 2) ^ (r & 3)) << 5) + (slot << 3), g.N - 8);
          issue_b(2 + hh, i, g.B2 + ((long)st2 * 64 + hh * 32 + r) * g.ldb2 + col);
        }
      if (last) {                                   // main prologue, first part: A0 B0 | A1 B1
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_a(0, i, a_src[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) issue_b(0, i, b_src[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_a(1, i, a_src[i] + (ntA1 > 1 ? 64 : 0));
#pragma unroll
        for (int i = 0; i < 2; ++i) issue_b(1, i, b_src[i] + (long)G2_BK * ldb);
      }
      // the step's 8 pieces per wave have landed (in the last step the 12 pieces of the main prologue stay in flight) and are
      // visible to every wave BEFORE the phases read their fragments (a phase waits for the data of the NEXT one, not its own)
      if (last) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      phase(10, 0, I0{}, I2{});                     // A stage (10 >> 1) % 3 = 2, B stage 10 % 4 = 2
      phase(11, 1, I0{}, I2{});                     // A stage 2 (k half 1), B stage 3
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) issue_b(2, i, b_src[i] + (long)(nt1 > 2 ? 2 : 1) * G2_BK * ldb);    // main prologue: B2
    // dropout mask on the adapter term (the main prologue is in flight)
    if (g.pre_thresh16) {
      if (MI16) {
#pragma unroll
        for (int tm = 0; tm < 8; ++tm)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) {
            acc16[MI16 ? tm : 0][tn] = gemm_dropmask4(acc16[MI16 ? tm : 0][tn], (long)m0 + wm * 128 + 16 * tm + (lane & 15),
                                                      n0 + wn * 64 + 16 * tn + 4 * (lane >> 4), g.N, g.pre_thresh16, g.pre_key,
                                                      g.pre_inv_keep);
          }
      } else {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              f32x16_t& v = acc[MI16 ? 0 : tm][tn];
              const f32x4_t o = gemm_dropmask4(f32x4_t{v[4 * rg], v[4 * rg + 1], v[4 * rg + 2], v[4 * rg + 3]},
                                               (long)m0 + wm * 128 + 32 * tm + (lane & 31), n0 + wn * 64 + 32 * tn + 8 * rg + 4 * (lane >> 5),
                                               g.N, g.pre_thresh16, g.pre_key, g.pre_inv_keep);
              v[4 * rg] = o.x; v[4 * rg + 1] = o.y; v[4 * rg + 2] = o.z; v[4 * rg + 3] = o.w;
            }
      }
    }
  } else {
    // prologue: A0 B0 | A1 B1 B2  (K >= 256 is required by the launcher, so these all lie in the main segment)
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_a(0, i, a_src[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) issue_b(0, i, b_src[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_a(1, i, a_src[i] + (ntA1 > 1 ? 64 : 0));
#pragma unroll
    for (int i = 0; i < 2; ++i) issue_b(1, i, b_src[i] + (long)G2_BK * ldb);
#pragma unroll
    for (int i = 0; i < 2; ++i) issue_b(2, i, b_src[i] + (long)(nt1 > 2 ? 2 : 1) * G2_BK * ldb);
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0
  if (RV_GEMM_PRIO_NN == 1 && wm == 1) __builtin_amdgcn_s_setprio(1);     // wm is wave-uniform (readfirstlane)

  // steady pairs: p + 4 < nt1 (A tile (p>>1)+2 and B tiles p+3, p+4 exist and lie in the main segment)
  int p = 0;
  for (; p + 4 < nt1; p += 2) {
    phase(p, 0, I0{}, I1{});
    b_run[0] += G2_BK * ldb; b_run[1] += G2_BK * ldb;
    phase(p + 1, 1, I1{}, I1{});
    b_run[0] += G2_BK * ldb; b_run[1] += G2_BK * ldb;
    a_run[0] += 64; a_run[1] += 64; a_run[2] += 64; a_run[3] += 64;
  }
  // tail / seam: generic phases (without EXT only phase nt-4 still has something to fetch: the last B tile)
  for (; p < nt; ++p) phase(p, p & 1, I0{}, I0{});
  if (wm == 0) __builtin_amdgcn_s_barrier();   // re-balance the barrier count

  if (MI16) {
#pragma unroll
    for (int hm = 0; hm < 2; ++hm) {
      f32x16_t blk[2][2];
      acc16_block_to_acc32(*reinterpret_cast<f32x4_t(*)[4][4]>(&acc16[MI16 ? 4 * hm : 0]), blk, lane);
      epi.apply(blk, m0 + wm * 128 + hm * 64, n0 + wblk * 32, lane, g.M, g.N);
    }
  } else {
    epi.apply(*reinterpret_cast<f32x16_t(*)[2][2]>(&acc[0]), m0 + wm * 128, n0 + wblk * 32, lane, g.M, g.N);
    epi.apply(*reinterpret_cast<f32x16_t(*)[2][2]>(&acc[MI16 ? 0 : 2]), m0 + wm * 128 + 64, n0 + wblk * 32, lane, g.M, g.N);
  }
}

#ifdef RV_GEMM_H128
// =============================================================================================
// EXPERIMENT (-DRV_GEMM_H128 builds only): the NN GEMM as TWO 4-wave workgroups per CU on 128 x 256 tiles, so
// that one workgroup's epilogue (the gate|up load burst + stores of the fused SwiGLU forms) can run under the other's main loop.
// Per workgroup 80 KiB of LDS: A 2 stages x 16 KiB (128 rows x 64 deep, the A64 image), B 3 stages x 16 KiB (32 k rows x 256).
// A wave owns 128 x 64 of the tile (the A64 kernel's strip of wave group 0) and runs ONE barrier per 32-deep phase:
//     wait (tile data of THIS phase: counted)  ->  barrier  ->  fragment reads  ->  32 MFMAs with the DMA issues between them
// M-seg(p) issues B tile p+2 (4 pieces per wave) and, in even phases, A tile (p>>1)+1 (4 pieces).  Hazards:
//   RAW: phase p needs B tile p (issued in M-seg(p-2)) and A tile p>>1 (issued in M-seg(p-2) for even p, M-seg(p-3) for odd p): all
//        but the pieces of M-seg(p-1) have landed -> vmcnt(4) in even phases (M-seg(p-1) was odd: 4 pieces), vmcnt(8) in odd
//        ones; prologue order A0 B0 B1: vmcnt(4) in phase 0 leaves B1 in flight.
//   WAR: B stage (p+2)%3 = (p-1)%3 and A stage ((p>>1)+1)&1 = ((p>>1)-1)&1 were last read in L-seg(p-1), which every wave has
//        behind it when it passes barrier(p).
// The ping-pong of the 8-wave kernels (one wave group in its MFMA segment while the other reads fragments) is left to the
// hardware here: the two workgroups of a CU are not synchronised.  `stagger`: workgroups of the first dispatch round whose LDS
// allocation does not start at 0 (the second one on its CU) first sleep that many s_memtime ticks, so the pair starts half a
// tile apart.
// =============================================================================================
#define H128_A_STAGE 16384
#define H128_B_STAGE 16384
#define H128_LDS_BYTES (2 * H128_A_STAGE + 3 * H128_B_STAGE)
template <class Epi>
__global__ __launch_bounds__(256, 2) void gemm_nn_h128_kernel(GemmShape g, Epi epi, int stagger, unsigned* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const smA = smem;
  uint8_t* const smB = smem + 2 * H128_A_STAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);

  const unsigned lds_alloc = __builtin_amdgcn_s_getreg(6 | (31 << 11));        // HW_REG_LDS_ALLOC, all 32 bits
  if (dbg && tid == 0) dbg[blockIdx.x] = lds_alloc;
  if (stagger > 0 && (int)blockIdx.x < 2 * 256 && (lds_alloc & 0xfffu) != 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)stagger) __builtin_amdgcn_s_sleep(32);
  }

  const int tiles_m = (g.M + 127) / 128, tiles_n = (g.N + G2_BN - 1) / G2_BN;
  const int nwg = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int GROUP = g.group > 0 ? 2 * g.group : 8;            // the same 1024-row clusters as the 256-row kernel's GROUP 4
  const int group_size = GROUP * tiles_n;
  const int first_m = (id / group_size) * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int tile_m = first_m + (id % group_size) % gsz;
  const int tile_n = (id % group_size) / gsz;
  const int m0 = tile_m * 128, n0 = tile_n * G2_BN;
  const long ldb = g.ldb;

  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wn * 4 + i) * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((row >> 1) & 7);
    a_src[i] = g.A + (long)min(m0 + row, g.M - 1) * g.lda + kc * 8;
    const int r = (wn * 4 + i) * 2 + (lane >> 5);
    const int c = lane & 31;
    const int slot = (c & 3) ^ (((r >> 3) & 1) << 1);
    const int col = (((c >> 2) ^ (r & 3)) << 5) + (slot << 3);
    b_src[i] = g.B + (long)r * ldb + min(n0 + col, g.N - 8);
  }
  const uint32_t piece0 = (uint32_t)(wn * 4) * 1024u;
  auto issue_a = [&](int T, int i, const bf16_t* src) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smA + (T & 1) * H128_A_STAGE + piece0 + i * 1024), 16, 0, 0);
  };
  auto issue_b = [&](int t, int i, const bf16_t* src) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smB + (t % 3) * H128_B_STAGE + piece0 + i * 1024), 16, 0, 0);
  };

  const int g4 = lane >> 4, s16 = lane & 15;
  const uint32_t a16_pre = (uint32_t)(s16 * 128) + (uint32_t)((g4 ^ (s16 >> 1)) << 4);
  uint32_t q16[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    q16[t] = (uint32_t)((8 * g4 + (s16 >> 2)) * 512) + (uint32_t)(((wn * 2 + (t >> 1)) ^ (s16 >> 2)) << 6) +
             (uint32_t)((((t & 1) ^ (g4 & 1)) << 5) + 8 * (s16 & 3));

  f32x4_t acc16[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nt = g.K / G2_BK;                    // 32-deep phases (even: K % 64 == 0)
  const int ntA = nt >> 1;
  // prologue: A0 B0 B1
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_a(0, i, a_src[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_b(0, i, b_src[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_b(1, i, b_src[i] + (long)G2_BK * ldb);

  const bf16_t* a_run[4];
  const bf16_t* b_run[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_run[i] = a_src[i] + (ntA > 1 ? 64 : 0);                       // A tile 1
    b_run[i] = b_src[i] + (long)(nt > 2 ? 2 : 1) * G2_BK * ldb;      // B tile 2
  }
  auto phase = [&](const int p, auto hc) {
    constexpr int H = decltype(hc)::value;                // = p & 1
    if (H == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const uint8_t* stA = smA + ((p >> 1) & 1) * H128_A_STAGE;
    const uint32_t stB = lds_addr_of(smB + (p % 3) * H128_B_STAGE);
    const uint32_t hx = (uint32_t)H << 6;
    bf16x8_t af[8], bfr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bfr[t] = ds_tr16_pair_asm(stB + q16[t], 0, 2048);
#pragma unroll
    for (int t = 0; t < 8; ++t) af[t] = *(const bf16x8_t*)(stA + (a16_pre ^ hx) + t * 2048);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      const int tm = kk >> 2, tn = kk & 3;
      acc16[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tn], af[tm], acc16[tm][tn], 0, 0, 0);
      if ((kk & 3) == 3) {                                // 8 DMA slots per phase: B pieces in slots 0..3, A pieces (even phases) in 4..7
        const int j = kk >> 2;
        __builtin_amdgcn_sched_barrier(0);
        if (j < 4) issue_b((p + 2), j, b_run[j]);
        else if (H == 0) issue_a((p >> 1) + 1, j - 4, a_run[j - 4]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  for (int p = 0; p < nt; p += 2) {
    // running sources of M-seg(p): B tile p+2, A tile (p>>1)+1; they stop at the last tile (redundant tail loads land in stages
    // nobody reads any more)
    phase(p, I0{});
    if (p + 3 < nt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) b_run[i] += (long)G2_BK * ldb;
    }
    if ((p >> 1) + 2 < ntA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a_run[i] += 64;
    }
    phase(p + 1, I1{});
    if (p + 4 < nt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) b_run[i] += (long)G2_BK * ldb;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the tail loads still target this workgroup's LDS

#pragma unroll
  for (int hm = 0; hm < 2; ++hm) {
    f32x16_t blk[2][2];
    acc16_block_to_acc32(*reinterpret_cast<f32x4_t(*)[4][4]>(&acc16[4 * hm]), blk, lane);
    epi.apply(blk, m0 + hm * 64, n0 + wn * 64, lane, g.M, g.N);
  }
}
#endif   // RV_GEMM_H128

// ---------------------------------------------------------------------------------------------
// Epilogues.  apply() receives the wave's 64x64 accumulators and its tile origin.
// ---------------------------------------------------------------------------------------------

enum { RV_ACT_NONE = 0, RV_ACT_QUICK_GELU = 1, RV_ACT_GELU = 2 };

// C[m][n] = act(acc + bias[n]) + R[m][n]   (bf16 out; bias/R optional)
__device__ __forceinline__ void epi_unpack8(const uint4& v, float (&f)[8]) {
  f[0] = bf2f((bf16_t)(v.x & 0xffff)); f[1] = bf2f((bf16_t)(v.x >> 16));
  f[2] = bf2f((bf16_t)(v.y & 0xffff)); f[3] = bf2f((bf16_t)(v.y >> 16));
  f[4] = bf2f((bf16_t)(v.z & 0xffff)); f[5] = bf2f((bf16_t)(v.z >> 16));
  f[6] = bf2f((bf16_t)(v.w & 0xffff)); f[7] = bf2f((bf16_t)(v.w >> 16));
}
__device__ __forceinline__ uint4 epi_pack8(const float (&f)[8]) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}
// Lanes l and l + 32 hold the two 4-column halves of the same row's 8-column groups (a: group 2g, b: group 2g + 1).  One
// exchange gives lane l {own a | partner's a} = columns 0-7 and lane l + 32 {partner's b | own b} = columns 8-15.  Round 3:
// v_permlane32_swap (upper half of the first operand <-> lower half of the second: exactly this exchange, one VALU op for
// both outputs) instead of __shfl_xor, which hipcc lowers to ds_bpermute_b32 + two v_cndmask + an lgkmcnt wait - 64 LDS round
// trips per wave per output tile in an epilogue nothing overlaps.  Same bits.
#ifndef RV_EPI_PERMLANE
#define RV_EPI_PERMLANE 1
#endif
__device__ __forceinline__ void epi_xhalf(float a, float b, int half, float& lo, float& hi) {
#if RV_EPI_PERMLANE
  (void)half;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  const u32x2_t r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  lo = __builtin_bit_cast(float, (unsigned)r[0]);
  hi = __builtin_bit_cast(float, (unsigned)r[1]);
#else
  const float recv = __shfl_xor(half ? a : b, 32);
  lo = half ? recv : a;
  hi = half ? b : recv;
#endif
}

struct EpiStore {
  bf16_t* C; long ldc;
  const bf16_t* bias;
  const bf16_t* R; long ldr;
  int act;
  float alpha;
  // optional dropout of the GEMM result BEFORE the residual is added (rv_gemm_nt_dropout_bf16): the mask of
  // rv_dropout for a contiguous [M][N] tensor with the same (p, seed); drop_thresh16 = 0 disables it
  uint32_t drop_thresh16 = 0, drop_key = 0;
  float drop_inv_keep = 1.f;
  int narrow = 0;            // 1: force the 8-byte store path (A/B knob RV_EPI_WIDE=0)
  // 16-byte stores: lanes l and l+32 hold the two 4-column halves of the same row's 8-column groups; one exchange per
  // value turns {cols 0-3 | 8-11} + {cols 4-7 | 12-15} into {0-7} + {8-15}, halving the number of store (and residual
  // load) instructions - the epilogue of a 1-workgroup-per-CU kernel is store-ISSUE bound and nothing overlaps it.
  __device__ __forceinline__ void apply_wide(f32x16_t (&acc)[2][2], int mw, int nw, int lane, int M, int N) const {
    const int half = lane >> 5;
    // The block's 8 residual loads are issued BEFORE the first one is used (addresses clamped into the tensor; rows / columns
    // beyond the edge read valid memory and are never stored).  Inside the loop they sat behind the `m >= M` / `n >= N`
    // `continue`s - control flow hipcc does not move loads across - so every iteration paid its own load -> wait -> add ->
    // store chain while all of the XCD's CUs were in their epilogues (round 3: +1 % on the o / down projections, bit-identical).
    uint4 rpre[2][4];
    if (R) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const long mc = min(mw + tm * 32 + (lane & 31), M - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          rpre[tm][i] = *(const uint4*)(R + mc * ldr + min(nw + (i >> 1) * 32 + (i & 1) * 16 + 8 * half, N - 8));
      }
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int m = mw + tm * 32 + (lane & 31);
      if (m >= M) continue;                       // lanes l and l+32 share m: partners skip together
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
        for (int rgp = 0; rgp < 2; ++rgp) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = acc[tm][tn][(2 * rgp) * 4 + j] * alpha, b = acc[tm][tn][(2 * rgp + 1) * 4 + j] * alpha;
            epi_xhalf(a, b, half, v[j], v[4 + j]);
          }
          const int n = nw + tn * 32 + rgp * 16 + 8 * half;
          if (n >= N) continue;
          if (bias) {
            float bb[8];
            epi_unpack8(*(const uint4*)(bias + n), bb);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += bb[j];
          }
          if (act == RV_ACT_QUICK_GELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] / (1.f + __expf(-1.702f * v[j]));
          } else if (act == RV_ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.5f * v[j] * (1.f + erff(v[j] * 0.70710678118654752f));
          }
          if (drop_thresh16) {
            const long e = (long)m * N + n;
            const uint32_t base = (uint32_t)(e >> 33) * 0x9e3779b9u + drop_key;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t h = gemm_mix32(((uint32_t)(e >> 1) + (uint32_t)q) ^ base);
              v[2 * q] = ((h & 0xffffu) >= drop_thresh16) ? v[2 * q] * drop_inv_keep : 0.f;
              v[2 * q + 1] = ((h >> 16) >= drop_thresh16) ? v[2 * q + 1] * drop_inv_keep : 0.f;
            }
          }
          if (R) {
            float rr[8];
            epi_unpack8(rpre[tm][tn * 2 + rgp], rr);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rr[j];
          }
          *(uint4*)(C + (long)m * ldc + n) = epi_pack8(v);
        }
      }
    }
  }
  __device__ __forceinline__ void apply(f32x16_t (&acc)[2][2], int mw, int nw, int lane, int M, int N) const {
    if (!narrow && ((ldc | N) & 7) == 0 && (((uintptr_t)C | (uintptr_t)bias) & 15) == 0 &&
        (R == nullptr || ((ldr & 7) == 0 && ((uintptr_t)R & 15) == 0))) {
      apply_wide(acc, mw, nw, lane, M, N);
      return;
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int m = mw + tm * 32 + (lane & 31);
      if (m >= M) continue;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int n = nw + tn * 32 + rg * 8 + 4 * (lane >> 5);
          if (n >= N) continue;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[tm][tn][rg * 4 + j] * alpha;
          if (bias) {
            const uint2 bb = *(const uint2*)(bias + n);
            v[0] += bf2f((bf16_t)(bb.x & 0xffff)); v[1] += bf2f((bf16_t)(bb.x >> 16));
            v[2] += bf2f((bf16_t)(bb.y & 0xffff)); v[3] += bf2f((bf16_t)(bb.y >> 16));
          }
          if (act == RV_ACT_QUICK_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] / (1.f + __expf(-1.702f * v[j]));
          } else if (act == RV_ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = 0.5f * v[j] * (1.f + erff(v[j] * 0.70710678118654752f));
          }
          if (drop_thresh16) {
            const long e = (long)m * N + n;                               // element index in the logical [M][N] tensor
            const uint32_t base = (uint32_t)(e >> 33) * 0x9e3779b9u + drop_key;
            const uint32_t h0 = gemm_mix32((uint32_t)(e >> 1) ^ base), h1 = gemm_mix32(((uint32_t)(e >> 1) + 1u) ^ base);
            v[0] = ((h0 & 0xffffu) >= drop_thresh16) ? v[0] * drop_inv_keep : 0.f;
            v[1] = ((h0 >> 16) >= drop_thresh16) ? v[1] * drop_inv_keep : 0.f;
            v[2] = ((h1 & 0xffffu) >= drop_thresh16) ? v[2] * drop_inv_keep : 0.f;
            v[3] = ((h1 >> 16) >= drop_thresh16) ? v[3] * drop_inv_keep : 0.f;
          }
          if (R) {
            const uint2 rr = *(const uint2*)(R + (long)m * ldr + n);
            v[0] += bf2f((bf16_t)(rr.x & 0xffff)); v[1] += bf2f((bf16_t)(rr.x >> 16));
            v[2] += bf2f((bf16_t)(rr.y & 0xffff)); v[3] += bf2f((bf16_t)(rr.y >> 16));
          }
          uint2 o;
          o.x = pack2bf(v[0], v[1]);
          o.y = pack2bf(v[2], v[3]);
          *(uint2*)(C + (long)m * ldc + n) = o;
        }
      }
    }
  }
};

// RoPE fused into the q|k|v projection (round 6; HF apply_rotary_pos_emb / rotate_half behind
// llava_llama.py:91-102): the epilogue rotates the q and k heads from the fp32 ACCUMULATORS - one rounding instead of the two of
// GEMM -> bf16 -> rope_kernel (7 % of the per-token error sat in that second rounding: profiles/r05_rounding_attribution.json) and
// one pass over [tokens, 2 d] less per layer.  kRopeCols: the kernel hands this wave columns nw .. nw + 31 (tn = 0) and
// nw + 64 .. nw + 95 (tn = 1) of one 128-wide head, so x1 = acc[tm][0][r] and x2 = acc[tm][1][r] are a rotation pair:
//     y1 = x1 cos - x2 sin,   y2 = x2 cos + x1 sin        (tables [position][64] fp32, position = pos[m] or m % L)
// Columns >= rope_cols (the v heads) are stored as they are.  Head dim 128 only; N, rope_cols multiples of 256.
struct EpiStoreRope {
  static constexpr bool kRopeCols = true;
  bf16_t* C; long ldc;
  const float* cos_tab; const float* sin_tab;
  const int* pos; int L;
  int rope_cols;
  __device__ __forceinline__ void apply(f32x16_t (&acc)[2][2], int mw, int nw, int lane, int M, int N) const {
    const int half = lane >> 5;
    const bool rot = nw < rope_cols;                 // wave-uniform (a 256-column tile holds q / k heads or v heads, never both)
    const int i0 = (nw & 127) + 8 * half;            // head-local index of the lane's first column of block tn = 0 (< 64)
    f32x4_t cs[2][2][2], sn[2][2][2];                // [tm][rgp][first / second four columns]
    if (rot) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int mc = min(mw + tm * 32 + (lane & 31), M - 1);
        const long p = pos ? pos[mc] : (mc % L);
#pragma unroll
        for (int rgp = 0; rgp < 2; ++rgp) {
          const float* c = cos_tab + p * 64 + i0 + rgp * 16;
          const float* s_ = sin_tab + p * 64 + i0 + rgp * 16;
          cs[tm][rgp][0] = *(const f32x4_t*)c;
          cs[tm][rgp][1] = *(const f32x4_t*)(c + 4);
          sn[tm][rgp][0] = *(const f32x4_t*)s_;
          sn[tm][rgp][1] = *(const f32x4_t*)(s_ + 4);
        }
      }
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int m = mw + tm * 32 + (lane & 31);
      if (m >= M) continue;                       // lanes l and l+32 share m: partners skip together
#pragma unroll
      for (int rgp = 0; rgp < 2; ++rgp) {
        float x1[8], x2[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          epi_xhalf(acc[tm][0][(2 * rgp) * 4 + j], acc[tm][0][(2 * rgp + 1) * 4 + j], half, x1[j], x1[4 + j]);
          epi_xhalf(acc[tm][1][(2 * rgp) * 4 + j], acc[tm][1][(2 * rgp + 1) * 4 + j], half, x2[j], x2[4 + j]);
        }
        const int n1 = nw + rgp * 16 + 8 * half;
        if (n1 >= N) continue;
        float y1[8], y2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float c = rot ? cs[tm][rgp][j >> 2][j & 3] : 1.f, s_ = rot ? sn[tm][rgp][j >> 2][j & 3] : 0.f;
          y1[j] = x1[j] * c - x2[j] * s_;
          y2[j] = x2[j] * c + x1[j] * s_;
        }
        *(uint4*)(C + (long)m * ldc + n1) = epi_pack8(y1);
        if (n1 + 64 < N) *(uint4*)(C + (long)m * ldc + n1 + 64) = epi_pack8(y2);
      }
    }
  }
};

// SwiGLU fused into the gate|up projection (HF LlamaMLP: down(silu(gate(x)) * up(x)), llava_llama.py:91-102 -> modeling_llama
// LlamaMLP.forward).  The fused weight stores gate and up INTERLEAVED (row 2j = gate_j, row 2j+1 = up_j), so a lane that holds 8
// consecutive output columns holds 4 complete (gate, up) pairs: it writes the gate|up tile (kept for backward) AND the
// activation tile act[m][j] = silu(g) * u - the separate swiglu pass (1.2 GB read + 0.6 GB write per layer) disappears.
// Round 6: the activation is computed from the fp32 ACCUMULATORS of the GEMM, not from the bf16-rounded g, u the
// unfused kernel would read back - one rounding (of the product) instead of three; the rounding-point study attributed 19 % of the
// per-token log-prob error to the gate|up term (profiles/r05_rounding_attribution.json).  Free in time; the price is that the
// fused forward is no longer bit-identical to GEMM -> bf16 -> swiglu kernel (it is CLOSER to the fp32 product: asserted in
// tests/test_kernels_gpu.py::test_swiglu_fused_gemm_epilogues).  The kept gate|up tile is still the rounded one, and the backward
// (EpiSwiGLUBwd, swiglu_bwd_kernel) differentiates at it, as does RV_KEEP_RECOMPUTABLE=0's recomputed activation.
struct EpiSwiGLU {
  bf16_t* C; long ldc;        // gate|up, interleaved columns [M][N]
  bf16_t* ACT; long lda;      // activation [M][N/2]
  // EXPERIMENT (round 6; RV_GU_TILE_MAJOR=1): the kept gate|up tensor TILE-MAJOR - 256 x 256 tiles, each 128 KB
  // contiguous, tile (tm, tn) at ((tm * tiles_n + tn) << 16) - so that the backward epilogue reads two contiguous 128 KB regions
  // per output tile instead of 256 row segments 44 KB apart.  Requires a buffer of ceil(M / 256) * 256 rows.
  int tile_major = 0, tiles_n = 0;
  // optional (round 6, fused-LoRA form rv_gemm_nn_lora_swiglu_bf16): ACTD = rv_dropout(ACT) for the contiguous [M][N/2] activation
  // with the same (p, seed) - the dropped adapter input of the down projection, written by its producer (what
  // swiglu_fwd_kernel<0, true> writes); drop_thresh16 = 0 / ACTD = NULL disables it
  bf16_t* ACTD = nullptr;
  uint32_t drop_thresh16 = 0, drop_key = 0;
  float drop_inv_keep = 1.f;
  __device__ __forceinline__ long gu_index(int m, int n) const {
    return tile_major ? ((((long)(m >> 8) * tiles_n + (n >> 8)) << 16) + ((m & 255) << 8) + (n & 255)) : ((long)m * ldc + n);
  }
  __device__ __forceinline__ void apply(f32x16_t (&acc)[2][2], int mw, int nw, int lane, int M, int N) const {
    const int half = lane >> 5;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int m = mw + tm * 32 + (lane & 31);
      if (m >= M) continue;                       // lanes l and l+32 share m: partners skip together
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
        for (int rgp = 0; rgp < 2; ++rgp) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = acc[tm][tn][(2 * rgp) * 4 + j], b = acc[tm][tn][(2 * rgp + 1) * 4 + j];
            epi_xhalf(a, b, half, v[j], v[4 + j]);
          }
          const int n = nw + tn * 32 + rgp * 16 + 8 * half;
          if (n >= N) continue;
          *(uint4*)(C + gu_index(m, n)) = epi_pack8(v);
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = v[2 * j] / (1.f + __expf(-v[2 * j])) * v[2 * j + 1];
          uint2 w;
          w.x = pack2bf(o[0], o[1]);
          w.y = pack2bf(o[2], o[3]);
          *(uint2*)(ACT + (long)m * lda + (n >> 1)) = w;
          if (ACTD) {                   // rv_dropout's arithmetic on the ROUNDED activation (mask index = element index in [M][N/2])
            const long e = (long)m * (N >> 1) + (n >> 1);
            const uint32_t base = (uint32_t)(e >> 33) * 0x9e3779b9u + drop_key;
            const uint32_t h0 = gemm_mix32((uint32_t)(e >> 1) ^ base), h1 = gemm_mix32(((uint32_t)(e >> 1) + 1u) ^ base);
            const float a0 = bf2f((bf16_t)(w.x & 0xffff)), a1 = bf2f((bf16_t)(w.x >> 16));
            const float a2 = bf2f((bf16_t)(w.y & 0xffff)), a3 = bf2f((bf16_t)(w.y >> 16));
            uint2 wd;
            wd.x = pack2bf(((h0 & 0xffffu) >= drop_thresh16) ? a0 * drop_inv_keep : 0.f, ((h0 >> 16) >= drop_thresh16) ? a1 * drop_inv_keep : 0.f);
            wd.y = pack2bf(((h1 & 0xffffu) >= drop_thresh16) ? a2 * drop_inv_keep : 0.f, ((h1 >> 16) >= drop_thresh16) ? a3 * drop_inv_keep : 0.f);
            *(uint2*)(ACTD + (long)m * (N >> 1) + (n >> 1)) = wd;
          }
        }
      }
    }
  }
};

// SwiGLU backward fused into the input-gradient GEMM of the down projection: acc = d act [M][f]; with the kept gate|up tile
// (interleaved) it emits d(gate|up) [M][2f] directly - d act never reaches HBM.  d act is rounded to bf16 first (what the
// unfused path stores), then the arithmetic of swiglu_bwd_kernel.
struct EpiSwiGLUBwd {
  const bf16_t* GU; long ldgu;
  bf16_t* DGU; long lddgu;
  int tile_major = 0, tiles_n = 0;      // layout of GU (see EpiSwiGLU); tiles_n counts 256-column tiles of the 2f-wide tensor
  // All 16 gate|up loads of the wave's 64 x 64 block are issued BEFORE the first one is used (addresses clamped into the
  // tensor; out-of-range rows / columns are loaded from valid memory and never stored): one exposed memory latency per block
  // instead of eight.  The edge tests used to be `continue`s in front of the loads - control flow hipcc does not move loads
  // across - so every iteration paid load -> wait -> exp -> store in sequence (ISA before: LLwSS x 8; after: L x 16, w, SS x 8).
  // Round 3: 1071 -> 1108 TF/s on the step's shape, bit-identical results (profiles/r03_gemm_lib_ab_*.log).
  __device__ __forceinline__ void apply(f32x16_t (&acc)[2][2], int mw, int nw, int lane, int M, int N) const {
    const int half = lane >> 5;
    uint4 ga[2][4], gb[2][4];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const long mc = min(mw + tm * 32 + (lane & 31), M - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int nc = min(nw + (i >> 1) * 32 + (i & 1) * 16 + 8 * half, N - 8);
        const int c2 = 2 * nc;
        const bf16_t* gp = tile_major ? GU + ((((long)(mc >> 8) * tiles_n + (c2 >> 8)) << 16) + ((mc & 255) << 8) + (c2 & 255))
                                      : GU + mc * ldgu + c2;
        ga[tm][i] = *(const uint4*)gp;
        gb[tm][i] = *(const uint4*)(gp + 8);
      }
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int m = mw + tm * 32 + (lane & 31);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
        for (int rgp = 0; rgp < 2; ++rgp) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = acc[tm][tn][(2 * rgp) * 4 + j], b = acc[tm][tn][(2 * rgp + 1) * 4 + j];
            epi_xhalf(a, b, half, v[j], v[4 + j]);
          }
          const int n = nw + tn * 32 + rgp * 16 + 8 * half;
          float da[8], gu0[8], gu1[8], o0[8], o1[8];
          epi_unpack8(epi_pack8(v), da);
          epi_unpack8(ga[tm][tn * 2 + rgp], gu0);
          epi_unpack8(gb[tm][tn * 2 + rgp], gu1);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float gg = (j < 4) ? gu0[2 * j] : gu1[2 * (j - 4)], uu = (j < 4) ? gu0[2 * j + 1] : gu1[2 * (j - 4) + 1];
            const float sg = 1.f / (1.f + __expf(-gg));
            const float dg = da[j] * uu * sg * (1.f + gg * (1.f - sg)), du = da[j] * (gg * sg);
            if (j < 4) { o0[2 * j] = dg; o0[2 * j + 1] = du; } else { o1[2 * (j - 4)] = dg; o1[2 * (j - 4) + 1] = du; }
          }
          if (m < M && n < N) {
            bf16_t* dp = DGU + (long)m * lddgu + 2 * n;
            *(uint4*)dp = epi_pack8(o0);
            *(uint4*)(dp + 8) = epi_pack8(o1);
          }
        }
      }
    }
  }
};

// fp32 output (used where a downstream reduction wants full precision).
struct EpiStoreF32 {
  float* C; long ldc;
  // optional (rv_gemm_nt_bf16_f32res): C = acc + bias[n] + R[m][n] with an fp32 residual - the fp32 residual stream of the frozen
  // CLIP tower (RV_CLIP_FP32_RESID).  R may alias C (every element is read and written by the same lane).
  const bf16_t* bias = nullptr;
  const float* R = nullptr; long ldr = 0;
  __device__ __forceinline__ void apply(f32x16_t (&acc)[2][2], int mw, int nw, int lane, int M, int N) const {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int m = mw + tm * 32 + (lane & 31);
      if (m >= M) continue;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int n = nw + tn * 32 + rg * 8 + 4 * (lane >> 5);
          if (n >= N) continue;
          float4 o = make_float4(acc[tm][tn][rg * 4], acc[tm][tn][rg * 4 + 1], acc[tm][tn][rg * 4 + 2],
                                 acc[tm][tn][rg * 4 + 3]);
          if (bias) {
            const uint2 bb = *(const uint2*)(bias + n);
            o.x += bf2f((bf16_t)(bb.x & 0xffff)); o.y += bf2f((bf16_t)(bb.x >> 16));
            o.z += bf2f((bf16_t)(bb.y & 0xffff)); o.w += bf2f((bf16_t)(bb.y >> 16));
          }
          if (R) {
            const float4 rr = *(const float4*)(R + (long)m * ldr + n);
            o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
          }
          *(float4*)(C + (long)m * ldc + n) = o;
        }
    }
  }
};

// Fused LM head, forward: logits never reach HBM.  Per (row, 64-column block) write the running
// (max, sum exp(x - max)) pair and, in the one block whose columns contain the target id, the
// target logit.  Requires N % 64 == 0.
struct EpiLogpFwd {
  const int* tgt;      // [M] target vocabulary id per selected row
  float* pmax;         // [M][N/64]
  float* psum;         // [M][N/64]
  float* tgt_logit;    // [M]
  int n_valid;         // columns [n_valid, N) are vocabulary padding (N % 64 == 0, N - 64 < n_valid <= N): not in the softmax
  __device__ __forceinline__ void apply(f32x16_t (&acc)[2][2], int mw, int nw, int lane, int M, int N) const {
    const int nblk = N >> 6;
    if (nw >= N) return;
    if (nw + 64 > n_valid) {           // the last 64-column block of a padded vocabulary (wave-uniform)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (nw + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) >= n_valid) acc[tm][tn][r] = -INFINITY;
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int m = mw + tm * 32 + (lane & 31);
      const bool ok = m < M;
      const int t = ok ? tgt[m] : -1;
      float mx = -INFINITY;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[tm][tn][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float s = 0.f;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s += __expf(acc[tm][tn][r] - mx);
          const int n = nw + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (ok && n == t) tgt_logit[m] = acc[tm][tn][r];
        }
      s += __shfl_xor(s, 32, 64);
      if (ok && (lane >> 5) == 0) {
        pmax[(long)m * nblk + (nw >> 6)] = mx;
        psum[(long)m * nblk + (nw >> 6)] = s;
      }
    }
  }
};

// Fused LM head, backward: recompute the logits tile and emit
//   dlogits[m][n] = coef[m] * ((n == tgt[m]) - exp(logit - lse[m]))      (bf16)
struct EpiLogpBwd {
  const int* tgt; const float* lse; const float* coef;
  bf16_t* dlogits; long ldd;
  int n_valid;         // columns [n_valid, N): vocabulary padding, written as zeros
  __device__ __forceinline__ void apply(f32x16_t (&acc)[2][2], int mw, int nw, int lane, int M, int N) const {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int m = mw + tm * 32 + (lane & 31);
      if (m >= M) continue;
      const int t = tgt[m];
      const float l = lse[m], c = coef[m];
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int n = nw + tn * 32 + rg * 8 + 4 * (lane >> 5);
          if (n >= N) continue;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float p = __expf(acc[tm][tn][rg * 4 + j] - l);
            v[j] = (n + j < n_valid) ? c * (((n + j) == t ? 1.f : 0.f) - p) : 0.f;
          }
          uint2 o;
          o.x = pack2bf(v[0], v[1]);
          o.y = pack2bf(v[2], v[3]);
          *(uint2*)(dlogits + (long)m * ldd + n) = o;
        }
    }
  }
};
