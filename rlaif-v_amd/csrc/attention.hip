// Flash-style attention for gfx950, forward (causal / full, head dim 64 or 128) and backward
// (causal or full, head dim 128), built on v_mfma_f32_32x32x16_bf16.
//
// Conventions (one wave = 32 query rows or 32 key rows, 4 waves per workgroup):
//  * every MFMA is issued so that the per-lane "column" index j = lane&31 is the row the wave owns
//    (a query in fwd / dQ, a key in dK/dV).  Softmax statistics are then per-lane scalars.
//  * a B operand taken straight from 32x32 accumulators enumerates its contraction index inside each
//    group of 16 as {0-3, 8-11 | 4-7, 12-15}; the matching A operand (V^T, K^T, Q^T, dO^T) is read from the
//    row-major tile with ds_read_b64_tr_b16 in exactly that order - no transposed copies exist in HBM.
//  * K/V (or Q/dO) tiles are double-buffered in LDS by global_load_lds; XOR chunk swizzles keep both the
//    32-row ds_read_b128 pattern and the 4-row transposing reads bank-conflict free.
#include "common.hpp"
#include "rlaifv_hip.h"

#include <stdlib.h>
#include <type_traits>

namespace {

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

// Combine a per-lane value with the one of lane ^ 32 (the two lanes that share a query / key row of a 32x32 MFMA tile) with
// v_permlane32_swap instead of __shfl_xor: hipcc lowers the shuffle to ds_bpermute_b32 + s_waitcnt lgkmcnt(0), and that wait
// also drains every fragment read issued ahead of it (the V^T prefetch of the forward's softmax section).  With both operands
// = x the swap leaves {own, partner} in lanes < 32 and {partner, own} in lanes >= 32: the combination is symmetric.
#ifndef RV_ATTN_PERMLANE
#define RV_ATTN_PERMLANE 1
#endif
#ifndef RV_ATTN_FWD_NW_DEFAULT
#define RV_ATTN_FWD_NW_DEFAULT 4
#endif
#ifndef RV_ATTN_FWD_DEFAULT
#define RV_ATTN_FWD_DEFAULT 2      // forward kernel at head dim 128: 2 = attn_fwd2_kernel, 3 = attn_fwd3_kernel (attn_fwd3.inc)
#endif
__device__ __forceinline__ void xhalf_pair(float x, float& a, float& b) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const u32x2_t r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  a = __builtin_bit_cast(float, (unsigned)r[0]);
  b = __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float xhalf_max(float x) {
#if RV_ATTN_PERMLANE
  float a, b;
  xhalf_pair(x, a, b);
  return fmaxf(a, b);
#else
  return fmaxf(x, __shfl_xor(x, 32, 64));
#endif
}
__device__ __forceinline__ float xhalf_sum(float x) {
#if RV_ATTN_PERMLANE
  float a, b;
  xhalf_pair(x, a, b);
  return a + b;
#else
  return x + __shfl_xor(x, 32, 64);
#endif
}

// RV_ATTN_PROF (experiment builds): s_memtime stamps in the tile loops of the forward (slots 0..7) and dQ (slots 8..15) kernels,
// accumulated per wave 0 of every workgroup into rv_attn_prof[]: +0 DMA issue, +1 first MFMA phase (S^T [and dP^T]),
// +2 softmax / dS + second MFMA phase, +3 wait for the next tile's DMA, +4 barrier, +5 per-pass prologue / epilogue, +6 whole
// kernel, +7 workgroups.  tools/exp_attn_prof.py reads them.  Each stamp drains lgkmcnt (s_memtime is an SMEM read).
#ifdef RV_ATTN_PROF
__device__ unsigned long long rv_attn_prof[16];
#define APROF_DECL unsigned apr[6] = {0, 0, 0, 0, 0, 0}; const unsigned long long apr_t0 = __builtin_amdgcn_s_memtime(); unsigned long long apr_last = apr_t0;
#define APROF(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); apr[i] += (unsigned)(t_ - apr_last); apr_last = t_; }
#define APROF_END(base) if (threadIdx.x == 0) { for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&rv_attn_prof[(base) + i_], (unsigned long long)apr[i_]); \
    atomicAdd(&rv_attn_prof[(base) + 6], __builtin_amdgcn_s_memtime() - apr_t0); atomicAdd(&rv_attn_prof[(base) + 7], 1ull); }
#else
#define APROF_DECL
#define APROF(i)
#define APROF_END(base)
#endif
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// =============================================================================================
// Version-2 forward and dQ kernels: K / V tiles (64 keys, row-major as in HBM) double-buffered in LDS by
// global_load_lds, and the key-contracted operand (V^T in forward, K^T in dQ) read from the same tiles by
// ds_read_b64_tr_b16 - no rv_head_transpose copies are needed any more.
// Row-operand reads are ds_read_b128 (32 rows x one chunk); chunk swizzles (see qtile_off for 256-byte rows):
//   128-byte rows (head dim 64): chunk ^ f(row), f = ((x&1)<<2)|(x>>1) with x = (row>>1)&7.
// =============================================================================================
// Block -> (x, head, sequence).  Workgroups of one (sequence, head) share its K/V (1.8 MB at L = 3.5 k); the
// dispatcher puts block b on XCD b % 8, so the plain (x, h, s) grid spreads every (s, h) over all 8 private L2s
// and each K/V tile is re-fetched from HBM by every query block (PMC: 6.8 GB per forward launch = 6 TB/s, the
// kernels were HBM bound).  Here the nx blocks of an (s, h) occupy consecutive slots of ONE XCD.
__device__ __forceinline__ void attn_block_coords(int nx, int H, int S, int& x, int& h, int& s) {
  const int id = blockIdx.x, HS = H * S;
  const int mode = (nx >> 16) & 15;      // experiment switch packed into the high bits by the launcher (bit 20: pair_blocks)
  nx &= 0xffff;
  int hs;
  if (mode == 1 && (HS & 7) == 0) {        // the nx blocks of an (s,h) on consecutive slots of one XCD
    const int xcd = id & 7, j = id >> 3;
    hs = (j / nx) * 8 + xcd;
    x = j % nx;
  } else if (mode == 2) {                  // (s,h) fastest: an (s,h) always lands on the same XCD, x spread in time
    hs = id % HS;
    x = id / HS;
  } else {                                 // x fastest (plain)
    hs = id / nx;
    x = id % nx;
  }
  h = hs % H;
  s = hs / H;
}

// Which two 128-row blocks a causal workgroup processes.  The plain pairing (x, n-1-x) is balanced for a plain causal row
// only: in a packed pair row [shared | chosen | rejected] a rejected-branch query block skips the chosen-branch key tiles
// and a chosen-branch key block is never visited by the rejected-branch queries, so the (x, n-1-x) sums range over 36..58
// tiles (forward / dQ) and 10..56 tiles (dK / dV) at the bench shape.  Here every lane computes the tile count of block
// `lane` with the SAME bounds the kernels use, the blocks are ranked by descending work, and workgroup bx takes the blocks
// of rank bx (heavy) and n-1-bx (light): the sums are equal where pairs can make them equal (forward / dQ: 36..40) and
// otherwise DEscend with bx, i.e. the dispatcher - which issues workgroups in id order - sees the long ones first (LPT).
// KEYS = false: query blocks (forward, dQ); true: key blocks (dK / dV).  n > 64 or bit 20 of nx clear: plain pairing.
// single = true (bit 21 of nx; the launcher then starts ONE workgroup per block): workgroup bx takes the block of rank bx
// alone - twice as many, half as long workgroups in strict longest-first order, so the tail of the launch is one LIGHT block.
template <bool KEYS, int BS = 128>
__device__ __forceinline__ void pair_blocks(int bx, int n, int L, int sh, int e1, int lane, bool balanced, bool single,
                                            int& first, int& second) {
  if (!balanced || n > 64) {
    first = bx;
    second = (!single && n - 1 - bx > bx) ? n - 1 - bx : -1;
    return;
  }
  const int x = lane, b0 = x * BS;
  int w;
  if (!KEYS) {
    const int nt = (min(L, b0 + BS) + 63) >> 6;
    const int skip = (b0 >= e1 && e1 > sh) ? max((e1 >> 6) - ((sh + 63) >> 6), 0) : 0;
    w = nt - skip;
  } else {
    const int nt = (b0 >= sh && b0 + BS - 1 < e1) ? min((L + 63) >> 6, (e1 + 63) >> 6) : (L + 63) >> 6;
    w = nt - (b0 >> 6);
  }
  if (x >= n) w = -1;
  int rank = 0;
  for (int y = 0; y < n; ++y) {
    const int wy = __builtin_amdgcn_readlane(w, y);
    rank += (wy > w || (wy == w && y < x)) ? 1 : 0;
  }
  const unsigned long long m1 = __ballot(x < n && rank == bx);
  const unsigned long long m2 = __ballot(x < n && rank == n - 1 - bx);
  first = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)m1) - 1);
  second = (!single && n - 1 - bx > bx) ? __builtin_amdgcn_readfirstlane((int)__ffsll((long long)m2) - 1) : -1;
}

// Pad-free rows: the grid has room for ceil(Lmax / 128) blocks per row; a workgroup whose index lies beyond THIS row's blocks
// (paired: beyond its ceil(n / 2) block pairs) has nothing to do.  Uniform over the workgroup, taken before any barrier.
__device__ __forceinline__ bool varlen_done(int bx, int n, bool paired) { return paired ? 2 * bx >= n + (n & 1) : bx >= n; }

template <int HD>
__device__ __forceinline__ uint32_t kvtile_off(int row, int c) {
  if (HD == 128) return (uint32_t)(row * 256 + ((c ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4));
  const int x = (row >> 1) & 7;
  return (uint32_t)(row * 128 + ((c ^ (((x & 1) << 2) | (x >> 1))) << 4));
}

// Element masks of a 32 x 32 score block (accumulator register r of a lane <-> index (r&3) + 8*(r>>2) + 4*half along the
// MFMA's row axis).  The predicate of the whole path is
//     hidden(q, key)  <=>  key >= L  ||  (CAUSAL && key > q)  ||  (q >= e1 && sh <= key < e1)
// (causal + the packed pair's "rejected branch does not see the chosen branch" window).  Written out per element it compiles
// to 5 compares and 6 scalar mask operations (17 instructions per element, 544 per forward tile - longer than the rest of
// the tile); with the lane-constant part folded into interval bounds it is one or two unsigned compares per element.
template <bool CAUSAL>
struct QueryLaneMask {            // this lane owns ONE query, the 16 registers run over keys (forward, dQ)
  int hi, xs;                     // last visible key; start of the hidden window
  unsigned xw;                    // width of the hidden window (0: none)
  __device__ __forceinline__ void init(int q, int L, int sh, int e1) {
    hi = CAUSAL ? min(q, L - 1) : L - 1;
    xs = sh;
    xw = (q >= e1 && e1 > sh) ? (unsigned)(e1 - sh) : 0u;
  }
  // kb = key of register 0 of this lane (block start + 4 * half)
  __device__ __forceinline__ void apply(f32x16_t& sacc, int kb) const {
    const int a = hi - kb, b = xs - kb;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cr = (r & 3) + 8 * (r >> 2);
      if (cr > a || (unsigned)(cr - b) < xw) sacc[r] = -INFINITY;
    }
  }
};
template <bool CAUSAL>
struct KeyLaneMask {              // this lane owns ONE key, the 16 registers run over queries (dK / dV)
  int lo;                         // first query that sees the key
  unsigned wd;                    // number of queries that see it: [lo, lo + wd)
  __device__ __forceinline__ void init(int key, int L, int sh, int e1) {
    const bool in_window = key >= sh && key < e1;
    lo = CAUSAL ? key : 0;
    const int up = in_window ? e1 : L;        // e1 <= L
    wd = (key < L && up > lo) ? (unsigned)(up - lo) : 0u;
  }
  // qb = query of register 0 of this lane (sub-tile start + 4 * half)
  __device__ __forceinline__ void apply(f32x16_t& sacc, int qb) const {
    const int a = lo - qb;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cr = (r & 3) + 8 * (r >> 2);
      if ((unsigned)(cr - a) >= wd) sacc[r] = -INFINITY;
    }
  }
};

// lane constants of the transposing reads over a [64 rows][HD] tile (rows = contraction index)
template <int HD>
struct TrOffsets {
  uint32_t off[2][HD / 32];
  __device__ __forceinline__ void init(int lane) {
    const int g4 = lane >> 4, s16 = lane & 15;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int et = 0; et < HD / 32; ++et)
        off[u][et] = kvtile_off<HD>(8 * u + 4 * (g4 >> 1) + (s16 >> 2), et * 4 + 2 * (g4 & 1) + ((s16 & 3) >> 1)) +
                     (uint32_t)((s16 & 1) * 8);
  }
  // 8 contraction rows row0 + {4*half + 0..3, 8 + 4*half + 0..3} of column et*32 + (lane&31); row0 % 16 == 0.
  // Inline-asm reads (common.hpp ds_tr16_b64_asm): the CALLER waits lgkmcnt and fences before the first use.
  __device__ __forceinline__ bf16x8_t read(uint32_t tile_addr, int row0, int et) const {
    return __builtin_shufflevector(ds_tr16_b64_asm(tile_addr + off[0][et], row0 * (HD * 2)),
                                   ds_tr16_b64_asm(tile_addr + off[1][et], row0 * (HD * 2)), 0, 1, 2, 3, 4, 5, 6, 7);
  }
};

// LDS-DMA of a [64 rows][HD] tile: 64*HD*2/1024 pieces of 1 KiB, split over the 4 waves.
// Addressing: a lane's source = (sequence base + tile row offset) [wave-uniform, SGPR pair] + voff[i] [32-bit VGPR, constant
// for the whole kernel] - the `saddr` form of global_load_lds, no per-piece VALU.  (The first version rebuilt a 64-bit address
// per piece - ~50 VALU per tile per wave; the forward ablation run prices the DMA issue at 20 % of the kernel,
// profiles/r02_attn_fwd_ablation.log.)  Only a tile that crosses the sequence end takes the per-lane clamped path.
template <int HD, int NW = 4>
struct TileDma {
  static constexpr int NP = 64 * HD * 2 / 1024 / NW;      // pieces per wave (4 waves: 4 for HD 128, 2 for HD 64; 8 waves: 2)
  static constexpr int RPP = 1024 / (HD * 2);             // rows per piece (4 / 8)
  static constexpr int CPR = HD / 8;                      // 16-byte chunks per row
  uint32_t voff[NP];                                      // row * (row stride in bytes) + source chunk * 16
  uint32_t ldb;
  __device__ __forceinline__ static int piece_row(int wave, int lane, int i) { return (wave * NP + i) * RPP + lane / CPR; }
  __device__ __forceinline__ static int piece_col(int row, int lane) {
    const int cp = lane % CPR;                             // chunk position in the LDS row
    // inverse of the read swizzle (an involution): source chunk = cp ^ f(row)
    return (int)((kvtile_off<HD>(row, cp) - (uint32_t)(row * HD * 2)) >> 4) * 8;
  }
  __device__ __forceinline__ void init(int wave, int lane, long ld) {
    ldb = (uint32_t)(ld * 2);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = piece_row(wave, lane, i);
      voff[i] = (uint32_t)row * ldb + (uint32_t)piece_col(row, lane) * 2u;
    }
  }
  // piece I of the NP pieces of this wave (the forward kernel of round 6 spreads a tile's pieces over its MFMA gaps)
  template <int I>
  __device__ __forceinline__ void issue_piece(const bf16_t* base, long ld, long tok0, int r0, int L, uint8_t* dst, int wave) const {
    const char* seq = (const char*)(base + tok0 * ld);     // wave-uniform
    if (r0 + 64 <= L) {
      const char* b = seq + (size_t)(uint32_t)r0 * ldb;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + voff[I]),
                                       (__attribute__((address_space(3))) void*)(dst + (wave * NP + I) * 1024), 16, 0, 0);
    } else {
      const int lane = threadIdx.x & 63;
      const int row = piece_row(wave, lane, I);
      const uint32_t r = (uint32_t)min(r0 + row, L - 1);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(seq + (r * ldb + (uint32_t)piece_col(row, lane) * 2u)),
          (__attribute__((address_space(3))) void*)(dst + (wave * NP + I) * 1024), 16, 0, 0);
    }
  }
  __device__ __forceinline__ void issue(const bf16_t* base, long ld, long tok0, int r0, int L, uint8_t* dst, int wave) const {
    const char* seq = (const char*)(base + tok0 * ld);     // wave-uniform
    if (r0 + 64 <= L) {
      const char* b = seq + (size_t)(uint32_t)r0 * ldb;
#pragma unroll
      for (int i = 0; i < NP; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + voff[i]),
                                         (__attribute__((address_space(3))) void*)(dst + (wave * NP + i) * 1024), 16, 0, 0);
    } else {
      const int lane = threadIdx.x & 63;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int row = piece_row(wave, lane, i);
        const uint32_t r = (uint32_t)min(r0 + row, L - 1);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(seq + (r * ldb + (uint32_t)piece_col(row, lane) * 2u)),
            (__attribute__((address_space(3))) void*)(dst + (wave * NP + i) * 1024), 16, 0, 0);
      }
    }
  }
};

// ABL (experiment builds, -DRV_ATTN_EXPERIMENTS; results WRONG by construction): 1 = no softmax arithmetic, 2 = no PV MFMAs,
// 3 = no QK^T MFMAs, 6 = no LDS-DMA of the following tiles.
#ifndef RV_DKV_S2
#define RV_DKV_S2 1            // dK/dV kernel: S^T accumulated as two independent partial sums (0: one 8-deep dependent chain)
#endif
#ifndef RV_ATTN_SMSPLIT
#define RV_ATTN_SMSPLIT 1      // softmax / dS slices computed under the MFMAs of the previous slice (round 3: forward -2.7 %)
#endif
#ifndef RV_ATTN_FWD_PRIO
#ifndef RV_ATTN_DQ_PRIO
#define RV_ATTN_DQ_PRIO 0      // 1 = s_setprio 1 in the dQ kernel's MFMA phases (experiment, round 4)
#endif
#define RV_ATTN_FWD_PRIO 1     // 1 = s_setprio 1 in the QK^T / PV MFMA phases (measured -0.5..-3 % vs 0, profiles/r02_attn_fwd_prio.log); 2 = in the softmax section (+1..2 %)
#endif
// NW = waves per workgroup = 32-query slices that share one K / V ring (round 5).  NW = 4: 128 queries, two workgroups per CU
// (rounds 1-4).  NW = 8: 256 queries, ONE workgroup per CU - every K / V tile is staged once per 256 queries, i.e. HALF the LDS-DMA
// instructions per CU and key tile (each wave issues 4 instead of 8): the phase stamps of round 4 put 18 % of a tile in those issues,
// which the CU's address path serves one wave-instruction at a time (profiles/r04_attn_fwd_dq_phase_profile.log).
template <int HD, bool CAUSAL, int ABL = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_fwd2_kernel(const bf16_t* __restrict__ qkv, long ld, int q_col0,
                                                           int k_col0, int v_col0, bf16_t* __restrict__ out, long ldo,
                                                           float* __restrict__ lse, int Lmax, int H, int nx, float scale,
                                                           const int* __restrict__ seg_sh,
                                                           const int* __restrict__ seg_e1, int kv_group,
                                                           const int* __restrict__ row_off,
                                                           const int* __restrict__ row_len) {
  constexpr int KS = HD / 16, ET = HD / 32, TILE = 64 * HD * 2, STAGE = 2 * TILE;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, half = lane >> 5;
  int bx, h, s;
  attn_block_coords(nx, H, (int)gridDim.x / ((nx & 0xffff) * H), bx, h, s);
  // PAD-FREE rows (row_off / row_len, both NULL = S rectangular rows of Lmax tokens): sequence s occupies token rows
  // [row_off[s], row_off[s] + row_len[s]) of the token-major buffers; lse (and the backward's delta planes) keep the [S][H][Lmax]
  // stride.  The grid is sized for Lmax: workgroups beyond this row's last block (pair) leave at once (varlen_done).
  const int L = row_len ? row_len[s] : Lmax;
  const long tok0 = row_off ? (long)row_off[s] : (long)s * Lmax;
  constexpr int BQ = 32 * NW;                     // queries per workgroup block
  const int nqb = (L + BQ - 1) / BQ;
  if (varlen_done(bx, nqb, CAUSAL && !((nx >> 21) & 1))) return;
  // packed (chosen | rejected) rows: queries at index >= e1 (the rejected branch) do not see keys in [sh, e1)
  const int sh = seg_sh ? seg_sh[s] : 0, e1 = seg_e1 ? seg_e1[s] : 0;
  const float c = scale * LOG2E;

  TileDma<HD, NW> dma;
  dma.init(wave, lane, ld);
  TrOffsets<HD> tro;
  tro.init(lane);
  uint32_t boff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) boff[ks] = kvtile_off<HD>(fr, 2 * ks + half);
  const bf16_t* kbase = qkv + k_col0 + (h / kv_group) * HD;     // grouped-query attention: query head h -> kv head h / G
  const bf16_t* vbase = qkv + v_col0 + (h / kv_group) * HD;

  int blk_first, blk_second;
  pair_blocks<false, BQ>(bx, nqb, L, sh, e1, lane, CAUSAL && ((nx >> 20) & 1), CAUSAL && ((nx >> 21) & 1), blk_first, blk_second);
  APROF_DECL
  const int npass = CAUSAL ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const int qb = (pass == 0) ? blk_first : blk_second;
    if (qb < 0) break;
    const int q0 = qb * BQ, q0w = q0 + wave * 32;
    const int q = q0w + fr;
    QueryLaneMask<CAUSAL> qmask;
    qmask.init(q, L, sh, e1);

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

    const int kv_end = CAUSAL ? min(L, q0 + BQ) : L;
    const int nt = (kv_end + 63) / 64;
    // A query block that lies entirely in the rejected branch never sees the key tiles that lie entirely in the chosen
    // branch [sh, e1): those tiles are not fetched at all (block-uniform: DMA and barriers are workgroup-wide).  The loop
    // runs over the nte remaining tiles; the LDS stage alternates with the loop index, not the tile index.
    int skip_lo = nt, skip_n = 0;
    if (q0 >= e1 && e1 > sh) {
      const int lo = (sh + 63) >> 6, hi = e1 >> 6;
      if (hi > lo) { skip_lo = lo; skip_n = hi - lo; }
    }
    const int nte = nt - skip_n;
    dma.issue(kbase, ld, tok0, (skip_lo == 0 ? skip_n : 0) * 64, L, smem, wave);
    dma.issue(vbase, ld, tok0, (skip_lo == 0 ? skip_n : 0) * 64, L, smem + TILE, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    APROF(5);
    for (int i = 0; i < nte; ++i) {
      const int t = i < skip_lo ? i : i + skip_n;
      const int k0 = t * 64;
      const uint8_t* Ks = smem + (i & 1) * STAGE;
      const uint8_t* Vs = Ks + TILE;
      if (i + 1 < nte && ABL != 6) {
        const int kn = ((i + 1 < skip_lo) ? i + 1 : i + 1 + skip_n) * 64;
        dma.issue(kbase, ld, tok0, kn, L, smem + ((i + 1) & 1) * STAGE, wave);
        dma.issue(vbase, ld, tok0, kn, L, smem + ((i + 1) & 1) * STAGE + TILE, wave);
      }
      APROF(0);
      if (!(CAUSAL && k0 > q0w + 31) && !(q0w >= e1 && k0 >= sh && k0 + 63 < e1)) {
        // S^T = K Q^T.  The K fragments are fetched in groups of four explicit ds_read_b128, group g+1 in flight while
        // group g multiplies (the compiler's own schedule was read / wait / MFMA sixteen times per tile).  Fragment ks of
        // a row sits at chunk (2 ks + half) ^ swz(row): address = address(ks = 0) ^ (ks << 5).
        constexpr int NG = KS / 2;                  // a group = fragments (ks, ks + 1) of BOTH 32-key halves: 4 reads, 4 MFMAs
        const uint32_t ka0 = lds_addr_of(Ks) + boff[0];
        f32x16_t sacc[2];
        zero16(sacc[0]);
        zero16(sacc[1]);
        bf16x8_t kA[4], kB[4];                       // element j of a group: half kt = j & 1, k step 2 g + (j >> 1)
        if (RV_ATTN_FWD_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j) kA[j] = ds_read_b128_asm(ka0 ^ (uint32_t)((j >> 1) << 5), (j & 1) * 32 * HD * 2);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          bf16x8_t (&kcur)[4] = (g & 1) ? kB : kA;
          bf16x8_t (&knxt)[4] = (g & 1) ? kA : kB;
          if (g + 1 < NG) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              knxt[j] = ds_read_b128_asm(ka0 ^ (uint32_t)((2 * (g + 1) + (j >> 1)) << 5), (j & 1) * 32 * HD * 2);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j) {   // the two halves alternate: consecutive MFMAs never share an accumulator
            if (ABL == 3) { asm volatile("" ::"v"(kcur[j])); continue; }
            sacc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kcur[j], qf[2 * g + (j >> 1)], sacc[j & 1], 0, 0, 0);
          }
        }
        // The tile needs element masks only when it touches the sequence end, the causal diagonal of this wave
        // or the chosen-branch window of a packed pair; interior tiles (the vast majority) skip ~200 VALU ops.
        APROF(1);
        const bool need_mask = (k0 + 63 >= L) || (CAUSAL && k0 + 63 > q0w) ||
                               (q0w + 31 >= e1 && k0 + 63 >= sh && k0 < e1);
        if (need_mask) {
          qmask.apply(sacc[0], k0 + 4 * half);
          qmask.apply(sacc[1], k0 + 32 + 4 * half);
        }
        // V^T fragments of the first 16 keys are requested NOW: they do not depend on P, so their LDS latency hides under
        // the softmax VALU work; afterwards the fragments of 16-key group kk+1 are in flight while group kk multiplies.
        const uint32_t vs_addr = lds_addr_of(Vs);
        bf16x8_t vA[ET], vB[ET];
        if (RV_ATTN_FWD_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (RV_ATTN_FWD_PRIO == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int e = 0; e < ET; ++e) vA[e] = tro.read(vs_addr, 0, e);
        // (Round 3 tried the same arithmetic with the instruction-level parallelism spelled out - four independent maxima, all 32
        // exponent arguments before the first v_exp, four partial sums instead of one 16-deep v_max3 chain, 32 dependent
        // v_fma -> v_exp pairs and one 32-deep v_add chain: no measurable change, profiles/r03_attn_rounds_softmax_ilp.log - the
        // second wave of the SIMD already fills those bubbles; removed.)
        float tmax = -INFINITY;                      // raw-score maximum (c > 0 keeps the order)
        if (ABL != 1)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[kt][r]);
        tmax = xhalf_max(tmax);
        const float m_new = fmaxf(m_run, tmax * c);
        const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);   // row still empty
        const float neg_m = (m_new == -INFINITY) ? 0.f : -m_new;     // a fully masked row keeps p = exp2(-inf) = 0
#if RV_ATTN_SMSPLIT
        // Exponentials in four slices of 16 keys, each computed UNDER the PV MFMAs of the previous slice (the matrix pipe runs
        // asynchronously: between two MFMA issues of a wave ~5 VALU slots are free): slice kk+1's 8 v_fma / v_exp / v_add go two
        // per MFMA of slice kk.  Only the maximum has to be complete before the first exponential.
        float psum = 0.f;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {                // the running maximum rarely moves after the first tiles
#pragma unroll
          for (int e = 0; e < ET; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[e][r] *= alpha;
        }
        auto exp_slice = [&](int kk, int j0, int j1) {      // elements j0..j1-1 (of 8) of slice kk
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < j0 || j >= j1 || ABL == 1) continue;
            const int r = (kk & 1) * 8 + j;
            const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kk >> 1][r], c, neg_m));
            sacc[kk >> 1][r] = p;
            psum += p;
          }
        };
        exp_slice(0, 0, 8);
        if (RV_ATTN_FWD_PRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (RV_ATTN_FWD_PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const bf16x8_t pf = pack_frag(sacc[kk >> 1], (kk & 1) * 8);
          bf16x8_t (&vcur)[ET] = (kk & 1) ? vB : vA;
          bf16x8_t (&vnxt)[ET] = (kk & 1) ? vA : vB;
          if (kk < 3) {
#pragma unroll
            for (int e = 0; e < ET; ++e) vnxt[e] = tro.read(vs_addr, (kk + 1) * 16, e);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * ET) : "memory");   // everything older than the 2*ET reads just issued
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < ET; ++e) {
            if (ABL != 2) o[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vcur[e], pf, o[e], 0, 0, 0);
            if (kk < 3) {
              exp_slice(kk + 1, (8 / ET) * e, (8 / ET) * (e + 1));
              __builtin_amdgcn_sched_barrier(0);              // keep the slice pieces between the MFMAs
            }
          }
        }
        psum = xhalf_sum(psum);
        l_run = l_run * alpha + psum;
      }
#else
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (ABL == 1) continue;
            const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], c, neg_m));   // one v_fma + one v_exp
            sacc[kt][r] = p;
            psum += p;
          }
        psum = xhalf_sum(psum);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {                // the running maximum rarely moves after the first tiles
#pragma unroll
          for (int e = 0; e < ET; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[e][r] *= alpha;
        }
        if (RV_ATTN_FWD_PRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (RV_ATTN_FWD_PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const bf16x8_t pf = pack_frag(sacc[kk >> 1], (kk & 1) * 8);
          bf16x8_t (&vcur)[ET] = (kk & 1) ? vB : vA;
          bf16x8_t (&vnxt)[ET] = (kk & 1) ? vA : vB;
          if (kk < 3) {
#pragma unroll
            for (int e = 0; e < ET; ++e) vnxt[e] = tro.read(vs_addr, (kk + 1) * 16, e);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * ET) : "memory");   // everything older than the 2*ET reads just issued
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < ET; ++e) {
            if (ABL == 2) { asm volatile("" ::"v"(vcur[e]), "v"(pf)); continue; }
            o[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vcur[e], pf, o[e], 0, 0, 0);
          }
        }
      }
#endif
      if (RV_ATTN_FWD_PRIO == 1) __builtin_amdgcn_s_setprio(0);
      APROF(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      APROF(3);
      __syncthreads();
      APROF(4);
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
      if (half == 0) lse[((long)s * H + h) * Lmax + q] = (m_run + log2f(l_run)) * LN2;
    }
  }  // pass
  APROF(5);
  APROF_END(0)
}

// Inverse RoPE fused into the stores of dQ / dK (apply_rotary_pos_emb's autograd, HF modeling_llama: the half-split pairs
// (i, i + 64) of a 128-wide head sit in accumulator tiles e and e + 2 at the same lane and register).  The values are first
// rounded to bf16 - what the separate rv_rope_inplace(backward) pass used to read back - then rotated in fp32:
//   dx1 = dy1 cos + dy2 sin,   dx2 = dy2 cos - dy1 sin.
__device__ __forceinline__ void store_rope_bwd_pair(const f32x16_t& lo, const f32x16_t& hi, float scale, const float* cr,
                                                    const float* sr, int e01, int half, bf16_t* head_base) {
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int i0 = e01 * 32 + rg * 8 + 4 * half;
    const f32x4_t cc = *(const f32x4_t*)(cr + i0), ss = *(const f32x4_t*)(sr + i0);
    float o1[4], o2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = bf2f(f2bf(lo[rg * 4 + j] * scale)), b = bf2f(f2bf(hi[rg * 4 + j] * scale));
      o1[j] = a * cc[j] + b * ss[j];
      o2[j] = b * cc[j] - a * ss[j];
    }
    uint2 w;
    w.x = pack2bf(o1[0], o1[1]);
    w.y = pack2bf(o1[2], o1[3]);
    *(uint2*)(head_base + i0) = w;
    w.x = pack2bf(o2[0], o2[1]);
    w.y = pack2bf(o2[2], o2[3]);
    *(uint2*)(head_base + 64 + i0) = w;
  }
}

template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq2_kernel(const bf16_t* __restrict__ qkv, long ld, int q_col0,
                                                              int k_col0, int v_col0, const bf16_t* __restrict__ dO,
                                                              long lddo, const bf16_t* __restrict__ O, long ldo,
                                                              const float* __restrict__ lse,
                                                              float* __restrict__ delta,
                                                              bf16_t* __restrict__ dqkv, long lddq, int Lmax, int H,
                                                              int nx, float scale, const int* __restrict__ seg_sh,
                                                              const int* __restrict__ seg_e1, int kv_group,
        const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, const int* __restrict__ rope_pos,
        const int* __restrict__ row_off, const int* __restrict__ row_len) {
  constexpr int HD = 128, KS = 8, ET = 4, TILE = 64 * HD * 2, STAGE = 2 * TILE;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, half = lane >> 5;
  int bx, h, s;
  attn_block_coords(nx, H, (int)gridDim.x / ((nx & 0xffff) * H), bx, h, s);
  // PAD-FREE rows (row_off / row_len, both NULL = S rectangular rows of Lmax tokens): sequence s occupies token rows
  // [row_off[s], row_off[s] + row_len[s]) of the token-major buffers; lse (and the backward's delta planes) keep the [S][H][Lmax]
  // stride.  The grid is sized for Lmax: workgroups beyond this row's last block (pair) leave at once (varlen_done).
  const int L = row_len ? row_len[s] : Lmax;
  const long tok0 = row_off ? (long)row_off[s] : (long)s * Lmax;
  const int nqb = (L + 127) / 128;
  if (varlen_done(bx, nqb, CAUSAL && !((nx >> 21) & 1))) return;
  // packed (chosen | rejected) rows: queries at index >= e1 (the rejected branch) do not see keys in [sh, e1)
  const int sh = seg_sh ? seg_sh[s] : 0, e1 = seg_e1 ? seg_e1[s] : 0;
  const float c = scale * LOG2E;

  TileDma<HD> dma;
  dma.init(wave, lane, ld);
  TrOffsets<HD> tro;
  tro.init(lane);
  uint32_t boff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) boff[ks] = kvtile_off<HD>(fr, 2 * ks + half);
  const bf16_t* kbase = qkv + k_col0 + (h / kv_group) * HD;
  const bf16_t* vbase = qkv + v_col0 + (h / kv_group) * HD;

  int blk_first, blk_second;
  pair_blocks<false>(bx, nqb, L, sh, e1, lane, CAUSAL && ((nx >> 20) & 1), CAUSAL && ((nx >> 21) & 1), blk_first, blk_second);
  APROF_DECL
  const int npass = CAUSAL ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const int qb = (pass == 0) ? blk_first : blk_second;
    if (qb < 0) break;
    const int q0 = qb * 128, q0w = q0 + wave * 32;
    const int q = q0w + fr, qc = min(q, L - 1);
    QueryLaneMask<CAUSAL> qmask;
    qmask.init(q, L, sh, e1);

    const int kv_end = CAUSAL ? min(L, q0 + 128) : L;
    const int nt = (kv_end + 63) / 64;
    int skip_lo = nt, skip_n = 0;           // chosen-branch key tiles a rejected-branch query block never sees (see forward)
    if (q0 >= e1 && e1 > sh) {
      const int lo = (sh + 63) >> 6, hi = e1 >> 6;
      if (hi > lo) { skip_lo = lo; skip_n = hi - lo; }
    }
    const int nte = nt - skip_n;
    // the first K / V tile is requested BEFORE the row operands: the delta computation below waits for Q / dO / O, and the tile's
    // latency used to start only after it (the ring is free: the previous pass ended on a barrier and its read-out does not use LDS)
    dma.issue(kbase, ld, tok0, (skip_lo == 0 ? skip_n : 0) * 64, L, smem, wave);
    dma.issue(vbase, ld, tok0, (skip_lo == 0 ? skip_n : 0) * 64, L, smem + TILE, wave);

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
    const float lse_q = lse[((long)s * H + h) * Lmax + qc] * LOG2E;
    // delta[q] = sum_e dO[q][e] * O[q][e] (the softmax-backward row term) is computed HERE from the dO fragments the
    // kernel holds anyway (a lane owns 64 of the row's 128 columns, its partner lane^32 the rest) and published for the
    // dK/dV kernel, which runs after this one on the same stream - no separate attn_delta pass over dO and O.
    float delta_q = 0.f;
    {
      const bf16_t* op = O + (tok0 + qc) * ldo + h * HD + 8 * half;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t of = *(const bf16x8_t*)(op + 16 * ks);
#pragma unroll
        for (int j = 0; j < 8; ++j) delta_q = fmaf(bf2f((bf16_t)dof[ks][j]), bf2f((bf16_t)of[j]), delta_q);
      }
      delta_q = xhalf_sum(delta_q);
      if (q < L && half == 0) {
        // workspace planes [3][S, H, L]: delta (versions 2-4 of the dK/dV kernel), -delta and -lse / scale (version 5 reads them
        // straight into its MFMA accumulator inputs)
        const long idx = ((long)s * H + h) * Lmax + q, plane = (long)((int)gridDim.x / ((nx & 0xffff) * H)) * H * Lmax;
        delta[idx] = delta_q;
        delta[plane + idx] = -delta_q;
        delta[2 * plane + idx] = -lse[idx] / scale;
      }
    }
    f32x16_t dq[ET];
#pragma unroll
    for (int e = 0; e < ET; ++e) zero16(dq[e]);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    APROF(5);
    for (int i = 0; i < nte; ++i) {
      const int t = i < skip_lo ? i : i + skip_n;
      const int k0 = t * 64;
      const uint8_t* Ks = smem + (i & 1) * STAGE;
      const uint8_t* Vs = Ks + TILE;
      if (i + 1 < nte) {
        const int kn = ((i + 1 < skip_lo) ? i + 1 : i + 1 + skip_n) * 64;
        dma.issue(kbase, ld, tok0, kn, L, smem + ((i + 1) & 1) * STAGE, wave);
        dma.issue(vbase, ld, tok0, kn, L, smem + ((i + 1) & 1) * STAGE + TILE, wave);
      }
      APROF(0);
      if (!(CAUSAL && k0 > q0w + 31) && !(q0w >= e1 && k0 >= sh && k0 + 63 < e1)) {
        const uint32_t ka0 = lds_addr_of(Ks) + boff[0];       // fragment ks: ka0 ^ (ks << 5); V tile = K tile + TILE
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          // S^T = K Q^T and dP^T = V dO^T.  Explicit reads, one pair {K[ks], V[ks]} per step in a 3-slot ring: pairs ks+1
          // and ks+2 are in flight while pair ks multiplies (the compiler's schedule exposed one LDS latency per pair).
          f32x16_t sacc, pacc;
          zero16(sacc);
          zero16(pacc);
          bf16x8_t fr3[3][2];
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              fr3[p][j] = ds_read_b128_asm(ka0 ^ (uint32_t)(p << 5), j * TILE + kt * 32 * HD * 2);
          if (RV_ATTN_DQ_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            if (ks + 2 < KS) {
#pragma unroll
              for (int j = 0; j < 2; ++j)
                fr3[(ks + 2) % 3][j] = ds_read_b128_asm(ka0 ^ (uint32_t)((ks + 2) << 5), j * TILE + kt * 32 * HD * 2);
              asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            } else if (ks + 1 < KS) {
              asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            } else {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr3[ks % 3][0], qf[ks], sacc, 0, 0, 0);
            pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr3[ks % 3][1], dof[ks], pacc, 0, 0, 0);
          }
          if (RV_ATTN_DQ_PRIO == 1) __builtin_amdgcn_s_setprio(0);
          // K^T fragments of the first 16 keys: independent of dS, requested before the exp section (latency hidden)
          const uint32_t ks_addr = lds_addr_of(Ks);
          bf16x8_t kfr0[ET], kfr1[ET];
#pragma unroll
          for (int e = 0; e < ET; ++e) kfr0[e] = tro.read(ks_addr, kt * 32, e);
          const bool need_mask = (k0 + 63 >= L) || (CAUSAL && k0 + 63 > q0w) ||
                                 (q0w + 31 >= e1 && k0 + 63 >= sh && k0 < e1);
          if (need_mask) qmask.apply(sacc, k0 + kt * 32 + 4 * half);
          // (Round 3 tried computing dS of the second 16 keys under the dQ MFMAs of the first 16, as the forward does with its
          // softmax slices: neutral to +1 % slower here - 4 VALU per element and 246 live registers leave the gaps no room;
          // profiles/r03_attn_smsplit_fwd_dq.log.)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sacc[r], c, -lse_q));   // masked: exp2(-inf) = 0
            sacc[r] = p * (pacc[r] - delta_q);
          }
          {
            const bf16x8_t df = pack_frag(sacc, 0);
#pragma unroll
            for (int e = 0; e < ET; ++e) kfr1[e] = tro.read(ks_addr, kt * 32 + 16, e);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * ET) : "memory");     // kfr0 landed; kfr1 may still be in flight
            if (RV_ATTN_DQ_PRIO == 1) __builtin_amdgcn_s_setprio(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < ET; ++e) dq[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr0[e], df, dq[e], 0, 0, 0);
          }
          {
            const bf16x8_t df = pack_frag(sacc, 8);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < ET; ++e) dq[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr1[e], df, dq[e], 0, 0, 0);
            if (RV_ATTN_DQ_PRIO == 1) __builtin_amdgcn_s_setprio(0);
          }
        }
      }
      APROF(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      APROF(3);
      __syncthreads();
      APROF(4);
    }

    if (q < L) {
      bf16_t* op = dqkv + (tok0 + q) * lddq + q_col0 + h * HD;
      if (rope_cos) {
        const long pos = rope_pos ? rope_pos[tok0 + q] : q;
        store_rope_bwd_pair(dq[0], dq[2], scale, rope_cos + pos * 64, rope_sin + pos * 64, 0, half, op);
        store_rope_bwd_pair(dq[1], dq[3], scale, rope_cos + pos * 64, rope_sin + pos * 64, 1, half, op);
      } else {
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
    }
  }  // pass
  APROF(5);
  APROF_END(8)
}

// =============================================================================================
// backward, dK/dV: no pre-transposed copies.
//   * Q / dO tiles (64 queries x 128, row-major as in HBM) are ring-buffered in LDS by global_load_lds;
//   * the operands whose contraction index is the query (Q^T for dK, dO^T for dV) are read from the SAME tiles
//     with ds_read_b64_tr_b16, in the order in which P / dS leave the accumulators;
//   * chunk swizzle c ^ (((row&3)<<2) | ((row>>2)&3)): the 32-row ds_read_b128 pattern sees 16 distinct chunks and
//     the 4 rows of a transposing read fall into 4 different quarters of the 256-byte bank row;
//   * lse / delta of the tile arrive through 4-byte LDS-DMA.
// (Version 2 of the kernel - the compiler-scheduled one - lives in history: commit 5e465f4 and before.)
// =============================================================================================
__device__ __forceinline__ uint32_t qtile_off(int row, int c) {
  return (uint32_t)(row * 256 + ((c ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4));
}

// =============================================================================================
// backward, dK/dV, version 3.  Same decomposition, tiles, LDS image and arithmetic as version 2 (bit-identical results);
// what changes is who schedules the inner loop.  rocprofv3 PMC on version 2 (profiles/r02_pmc_hot_kernels_before.txt):
// MFMA busy 23 % of cycles, 40 % of the wave's time parked in s_waitcnt, 450 VALU instructions per 64-query tile.  Its ISA
// shows why: with all 256 architectural VGPRs in use hipcc emits "ds_read_b128 x2, s_waitcnt lgkmcnt(0), MFMA x2" per k
// step (sixteen exposed LDS latencies per tile: one wave per SIMD, nothing else to run), a v_add per LDS address and
// 32 v_mov per sub-tile to clear accumulators.  Here:
//   * K / V fragments live in AGPRs (MFMA reads its B operand from the accumulator file) - 64 VGPRs back;
//   * every LDS read is an explicit asm read off ONE per-lane base register per fragment column with the buffer / tile /
//     sub-tile selected by the 16-bit immediate (the tile loop is unrolled by two so the buffer index is static);
//   * the Q / dO row fragments of a 32-query sub-tile are fetched in two batches of 8 reads: batch 1 is in flight while
//     batch 0 multiplies, and both batches of the second sub-tile are requested while the first one's dV / dK MFMAs run;
//   * S / dP accumulate in VGPRs through asm MFMAs whose first instruction takes the inline constant 0 as C.
// =============================================================================================
#include "attn_agpr.inc"

// ABL (experiment builds only, -DRV_ATTN_EXPERIMENTS; results are WRONG by construction): 1 = no exp / dS arithmetic,
// 2 = no dV / dK MFMAs, 3 = no S / dP MFMAs, 4 = no transposing LDS reads, 5 = no row-fragment LDS reads, 6 = no DMA of the
// next tile.  Timing deltas against ABL 0 price the stages (cdna_hip_programming.md 5.4 rule 17: values stay live).
template <bool CAUSAL, int ABL = 0>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv3_kernel(const bf16_t* __restrict__ qkv, long ld, int q_col0,
                                                               int k_col0, int v_col0,
                                                               const bf16_t* __restrict__ dO, long lddo,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta,
                                                               bf16_t* __restrict__ dqkv, long lddq, int Lmax, int H,
                                                               int nx, float scale, const int* __restrict__ seg_sh,
                                                               const int* __restrict__ seg_e1, int kv_group,
        const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, const int* __restrict__ rope_pos,
        const int* __restrict__ row_off, const int* __restrict__ row_len) {
  constexpr int HD = 128, KS = 8, ET = 4;
  constexpr int STAGE = 2 * 64 * 256 + 512;          // Q tile + dO tile + lse[64] + delta[64] = 0x8200
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, half = lane >> 5;
  int bx, h, s;
  attn_block_coords(nx, H, (int)gridDim.x / ((nx & 0xffff) * H), bx, h, s);
  const int L = row_len ? row_len[s] : Lmax;                   // pad-free rows: see attn_fwd2_kernel
  const long tok0 = row_off ? (long)row_off[s] : (long)s * Lmax;
  const int sh = seg_sh ? seg_sh[s] : 0, e1 = seg_e1 ? seg_e1[s] : 0;
  const int nkb = (L + 127) / 128;
  if (varlen_done(bx, nkb, CAUSAL && !((nx >> 21) & 1))) return;
  const float c = scale * LOG2E;
  const int HQ = H * kv_group;

  int d_row[4], d_chunk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    d_row[i] = (wave * 4 + i) * 4 + (lane >> 4);
    d_chunk[i] = (lane & 15) ^ (((d_row[i] & 3) << 2) | ((d_row[i] >> 2) & 3));
  }
  const uint32_t ldqb = (uint32_t)(ld * 2), lddob = (uint32_t)(lddo * 2);       // row strides in bytes (L * stride < 2^32: checked by the launcher)
  // One LDS-DMA piece of tile t into buffer buf: j = 0..7 -> Q piece j>>1 (even j) / dO piece j>>1 (odd j); j = 8: lse and
  // delta of the tile (wave 0).  An LDS-DMA issue occupies the wave for 60-180 cycles (address path); with one wave per
  // SIMD nothing else would run meanwhile, so the pieces are issued one by one BETWEEN the MFMAs of the dV / dK segment,
  // whose matrix work keeps executing underneath.
  auto issue_piece = [&](int hq, int t, int buf, int j) {
    if (ABL == 6 && t > 0) return;
    uint8_t* st = smem + buf * STAGE;
    // the tile index goes through an opaque scalar: the address arithmetic below is then computed HERE (a handful of
    // 32-bit VALU in the shadow of the surrounding MFMAs) instead of being hoisted to the top of the tile, where nine
    // 64-bit addresses stay live across the whole body (the causal instantiation spilled)
    int qs0 = t * 64;
    asm volatile("" : "+s"(qs0));
    if (j < 8) {
      const int i = j >> 1;
      const uint32_t r = (uint32_t)min(qs0 + d_row[i], L - 1);           // row inside sequence s (< 2^16)
      if ((j & 1) == 0) {
        const char* base = (const char*)(qkv + tok0 * ld + q_col0 + hq * HD);      // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (r * ldqb + d_chunk[i] * 16u)),
                                         (__attribute__((address_space(3))) void*)(st + (wave * 4 + i) * 1024), 16, 0, 0);
      } else {
        const char* base = (const char*)(dO + tok0 * lddo + hq * HD);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (r * lddob + d_chunk[i] * 16u)),
                                         (__attribute__((address_space(3))) void*)(st + 16384 + (wave * 4 + i) * 1024), 16, 0, 0);
      }
    } else if (wave == 0) {
      const int qq = min(qs0 + lane, L - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lse + ((long)s * HQ + hq) * Lmax + qq),
                                       (__attribute__((address_space(3))) void*)(st + 32768), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(delta + ((long)s * HQ + hq) * Lmax + qq),
                                       (__attribute__((address_space(3))) void*)(st + 32768 + 256), 4, 0, 0);
    }
  };
  auto issue_tile = [&](int hq, int t, int buf) {
#pragma unroll
    for (int j = 0; j < 9; ++j) issue_piece(hq, t, buf, j);
  };

  // per-lane LDS bases (buffer 0, Q tile, sub-tile 0); everything else is an immediate:
  //   row fragment ks of row fr:  rb[ks] + BUF*STAGE + isdO*16384 + qt*8192
  //   transposing read (u, et):    tb[u][et] + BUF*STAGE + isdO*16384 + (qt*32 + k2*16)*256     (all < 65536)
  const uint32_t lds0 = lds_addr_of(smem);
  uint32_t rb[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) rb[ks] = lds0 + qtile_off(fr, 2 * ks + half);
  uint32_t lh = lds0 + 32768u + (uint32_t)(half * 16);   // lse / delta: floats 4*half .. 4*half+3 of each group of 8 queries
  const int g4 = lane >> 4, s16 = lane & 15;
  uint32_t tb[2][ET];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int et = 0; et < ET; ++et)
      tb[u][et] = lds0 + qtile_off(8 * u + 4 * (g4 >> 1) + (s16 >> 2), et * 4 + 2 * (g4 & 1) + ((s16 & 3) >> 1)) +
                  (uint32_t)((s16 & 1) * 8);

  int blk_first, blk_second;
  pair_blocks<true>(bx, nkb, L, sh, e1, lane, CAUSAL && ((nx >> 20) & 1), CAUSAL && ((nx >> 21) & 1), blk_first, blk_second);
  const int npass = CAUSAL ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const int kvb = (pass == 0) ? blk_first : blk_second;
    if (kvb < 0) break;
    const int kv0 = kvb * 128, kv0w = kv0 + wave * 32;
    const int key = kv0w + fr, keyc = min(key, L - 1);
    KeyLaneMask<CAUSAL> kmask;
    kmask.init(key, L, sh, e1);

    // B operands of S^T / dP^T: K / V fragments of this wave's 32 keys, loaded STRAIGHT INTO fixed AGPRs (gfx90a+ vector
    // memory can target the accumulator file); dK / dV accumulate in fixed AGPRs as well (map in attn_agpr.inc).  With
    // compiler-allocated "a" operands the causal instantiation copied 16-register tiles around every MFMA.
    {
      const bf16_t* kp = qkv + (tok0 + keyc) * ld + h * HD + 8 * half;
      static_for<KS>([&](auto ic) {
        constexpr int ks = decltype(ic)::value;
        frag_load<ks>(kp + k_col0 + 16 * ks);
        frag_load<8 + ks>(kp + v_col0 + 16 * ks);
      });
      static_for<8>([&](auto ic) { acc_zero<decltype(ic)::value>(); });
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // asm loads are invisible to hipcc's own wait bookkeeping
    }

    const int t_begin = CAUSAL ? (kv0 / 64) : 0;
    const int nt = (kv0 >= sh && kv0 + 127 < e1) ? min((L + 63) / 64, (e1 + 63) / 64) : (L + 63) / 64;

    // One 64-query tile out of LDS buffer `buf`.  The per-lane read bases (rb, tb, lh) already point INTO that buffer: they
    // are flipped by +-STAGE at the end of every tile (17 VALU), so one copy of the body serves both buffers and the
    // immediates stay compile-time (two unrolled copies made hipcc shuffle the 128 accumulator registers between them).
    auto tile = [&](const int hq, const int t, const int buf) {
      constexpr int SB = 0;
      const int BUF = buf;
      const int qs0 = t * 64;
      // the tile fetched meanwhile: t + 1, or once more the last one (into the buffer nobody reads any more) - the DMA
      // issues inside the MFMA stream stay UNCONDITIONAL: a branch there splits the segment into basic blocks and hipcc
      // then copies a 16-register accumulator between them at every seam
      const int tn = min(t + 1, nt - 1);
      // this wave's 32 keys see nothing of the tile (above the causal diagonal, or rejected-branch queries x chosen-branch
      // keys): it only issues its share of the next tile's DMA
      if ((CAUSAL && qs0 + 63 < kv0w) || (qs0 >= e1 && kv0w >= sh && kv0w + 31 < e1)) {
        issue_tile(hq, tn, BUF ^ 1);
      } else {
        bf16x8_t fq[KS], fo[KS];                   // row fragments of the current sub-tile: Q[ks], dO[ks]
        // Row fragments: batch 0 = the 8 Q fragments (all k steps), batch 1 = the 8 dO fragments.  The S^T chain runs on batch 0
        // while batch 1 lands; the dP^T chain then runs UNDER the exp section, which needs S^T only (the first version
        // interleaved the two chains step by step and started the exponentials after both: 8 MFMAs of exposed VALU wait).
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fq[ks] = ds_read_b128_asm(rb[ks], SB);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const int QO = SB + qt * 8192;           // immediate part of this sub-tile's row-fragment reads
          f32x16_t sacc, pacc;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) fo[ks] = ds_read_b128_asm(rb[ks], QO + 16384);
          asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // the Q fragments landed (the 8 dO reads may be in flight)
          __builtin_amdgcn_sched_barrier(0);
          // ---- S^T = Q K^T as TWO partial sums (even / odd k steps): an MFMA that accumulates on the previous one's result
          // cannot issue before that result exists (16 passes) while independent ones issue every 8 - a single 8-deep chain
          // runs at half rate.  The halves are added inside the exponent's argument.
          f32x16_t sacc2;
#if RV_DKV_S2
          smfma_first<0>(sacc, fq[0]);
          smfma_first<1>(sacc2, fq[1]);
          static_for<3>([&](auto ic) {
            constexpr int ks = 2 * (decltype(ic)::value + 1);
            smfma<ks>(sacc, fq[ks]);
            smfma<ks + 1>(sacc2, fq[ks + 1]);
          });
#else
          smfma_first<0>(sacc, fq[0]);
          static_for<KS - 1>([&](auto ic) { smfma<decltype(ic)::value + 1>(sacc, fq[decltype(ic)::value + 1]); });
          zero16(sacc2);
#endif
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the dO fragments landed under the chain above
          __builtin_amdgcn_sched_barrier(0);
          // lse and delta of this lane's 16 queries (4 consecutive floats per accumulator row group), then the transposed
          // fragments of the first 16 queries (independent of P / dS): all requested before the dP^T chain starts
          f32x4_t lsev[4], delv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lsev[j] = ds_read_f32x4_asm(lh, SB + qt * 128 + j * 32);
            delv[j] = ds_read_f32x4_asm(lh, SB + 256 + qt * 128 + j * 32);
          }
          bf16x8_t dotf0[ET], qtf0[ET], dotf1[ET], qtf1[ET];
          const int TO = SB + qt * 32 * 256;
#pragma unroll
          for (int e = 0; e < ET; ++e) {
            dotf0[e] = __builtin_shufflevector(ds_tr16_b64_asm(tb[0][e], TO + 16384), ds_tr16_b64_asm(tb[1][e], TO + 16384),
                                               0, 1, 2, 3, 4, 5, 6, 7);
            qtf0[e] = __builtin_shufflevector(ds_tr16_b64_asm(tb[0][e], TO), ds_tr16_b64_asm(tb[1][e], TO), 0, 1, 2, 3, 4, 5,
                                              6, 7);
          }
          // ---- dP^T = dO V^T: the first three MFMAs also serve as the XDL-write -> VALU wait states of the S^T chain
          // (>= 64 cycles pass before the third can issue); nothing that reads sacc may be scheduled above this fence
          smfma_first<8>(pacc, fo[0]);
          smfma<9>(pacc, fo[1]);
          smfma<10>(pacc, fo[2]);
          asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");   // lse / delta (the 8 oldest of 24 reads) landed
          __builtin_amdgcn_sched_barrier(0);
          const int qsub = qs0 + qt * 32;                      // first query of this 32-row sub-tile
          const bool need_mask = (qsub + 31 >= L) || (kv0w + 31 >= L) || (CAUSAL && kv0w + 31 > qsub) ||
                                 (qsub + 31 >= e1 && kv0w + 31 >= sh && kv0w < e1);
          if (need_mask) kmask.apply(sacc, qsub + 4 * half);
          // exponentials (S^T only), three per remaining dP^T MFMA; the fences pin the interleave
          auto exps = [&](int r0, int r1) {
#pragma unroll
            for (int r = r0; r < r1; ++r)
              sacc[r] = __builtin_amdgcn_exp2f(fmaf(sacc[r] + sacc2[r], c, -LOG2E * lsev[r >> 2][r & 3]));
          };
          exps(0, 3);
          __builtin_amdgcn_sched_barrier(0);
          smfma<11>(pacc, fo[3]);
          __builtin_amdgcn_sched_barrier(0);
          exps(3, 6);
          __builtin_amdgcn_sched_barrier(0);
          smfma<12>(pacc, fo[4]);
          __builtin_amdgcn_sched_barrier(0);
          exps(6, 9);
          __builtin_amdgcn_sched_barrier(0);
          smfma<13>(pacc, fo[5]);
          __builtin_amdgcn_sched_barrier(0);
          exps(9, 12);
          __builtin_amdgcn_sched_barrier(0);
          smfma<14>(pacc, fo[6]);
          __builtin_amdgcn_sched_barrier(0);
          exps(12, 14);
          __builtin_amdgcn_sched_barrier(0);
          smfma<15>(pacc, fo[7]);
          __builtin_amdgcn_sched_barrier(0);
          exps(14, 16);
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 15\n\ts_nop 3" : "+v"(pacc));      // XDL write -> VALU read wait states of the dP^T chain
#pragma unroll
          for (int r = 0; r < 16; ++r) pacc[r] = sacc[r] * (pacc[r] - delv[r >> 2][r & 3]);
          const bf16x8_t pf0 = pack_frag(sacc, 0), dsf0 = pack_frag(pacc, 0);
          const bf16x8_t pf1 = pack_frag(sacc, 8), dsf1 = pack_frag(pacc, 8);
#pragma unroll
          for (int e = 0; e < ET; ++e)
            dotf1[e] = __builtin_shufflevector(ds_tr16_b64_asm(tb[0][e], TO + 16384 + 16 * 256),
                                               ds_tr16_b64_asm(tb[1][e], TO + 16384 + 16 * 256), 0, 1, 2, 3, 4, 5, 6, 7);
          asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // everything older than the 8 reads just issued
          __builtin_amdgcn_sched_barrier(0);
          static_for<ET>([&](auto ic) {
            constexpr int e = decltype(ic)::value;
            if (ABL == 2) asm volatile("" ::"v"(dotf0[e]), "v"(pf0), "v"(qtf0[e]), "v"(dsf0));
            if (ABL != 2) acc_mfma<4 + e>(dotf0[e], pf0);
            if (qt == 0) { __builtin_amdgcn_sched_barrier(0); issue_piece(hq, tn, BUF ^ 1, 2 * e); __builtin_amdgcn_sched_barrier(0); }
            if (ABL != 2) acc_mfma<e>(qtf0[e], dsf0);
            if (qt == 0) { __builtin_amdgcn_sched_barrier(0); issue_piece(hq, tn, BUF ^ 1, 2 * e + 1); __builtin_amdgcn_sched_barrier(0); }
          });
#pragma unroll
          for (int e = 0; e < ET; ++e)
            qtf1[e] = __builtin_shufflevector(ds_tr16_b64_asm(tb[0][e], TO + 16 * 256), ds_tr16_b64_asm(tb[1][e], TO + 16 * 256),
                                              0, 1, 2, 3, 4, 5, 6, 7);
          if (qt == 0) {
            // both batches' worth of registers are free (their MFMAs were issued long ago): request batch 0 of the
            // second sub-tile now, its latency hides under the remaining dV / dK MFMAs
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fq[ks] = ds_read_b128_asm(rb[ks], SB + 8192);
            asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");  // dotf1 landed (16 younger reads may be in flight)
          } else {
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // dotf1 landed
          }
          __builtin_amdgcn_sched_barrier(0);
          static_for<ET>([&](auto ic) {
            if (ABL == 2) { asm volatile("" ::"v"(dotf1[decltype(ic)::value]), "v"(pf1)); return; }
            acc_mfma<4 + decltype(ic)::value>(dotf1[decltype(ic)::value], pf1);
          });
          if (qt == 0) { __builtin_amdgcn_sched_barrier(0); issue_piece(hq, tn, BUF ^ 1, 8); __builtin_amdgcn_sched_barrier(0); }
          if (qt == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // qtf1 landed (batch 0 of sub-tile 1 may be in flight)
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          static_for<ET>([&](auto ic) {
            if (ABL == 2) { asm volatile("" ::"v"(qtf1[decltype(ic)::value]), "v"(dsf1)); return; }
            acc_mfma<decltype(ic)::value>(qtf1[decltype(ic)::value], dsf1);
          });
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      const uint32_t flip = buf ? (uint32_t)(-STAGE) : (uint32_t)STAGE;       // bases follow the buffer of the NEXT tile
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) rb[ks] += flip;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int et = 0; et < ET; ++et) tb[u][et] += flip;
      lh += flip;
    };

    for (int gq = 0; gq < kv_group; ++gq) {
      const int hq = h * kv_group + gq;
      issue_tile(hq, t_begin, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int t = t_begin; t < nt; ++t) tile(hq, t, (t - t_begin) & 1);
      if ((nt - t_begin) & 1) {                    // an odd number of tiles leaves the bases in buffer 1: back to buffer 0
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) rb[ks] -= STAGE;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int et = 0; et < ET; ++et) tb[u][et] -= STAGE;
        lh -= STAGE;
      }
    }  // query heads of the group
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // last asm MFMA -> v_accvgpr_read (hipcc pads nothing for asm)

    bf16_t* kp_out = dqkv + (tok0 + keyc) * lddq + h * HD;
    if (rope_cos) {       // dK leaves through the inverse rotation (pairs = tiles e, e + 2)
      const long pos = rope_pos ? rope_pos[tok0 + keyc] : keyc;
      static_for<2>([&](auto ic) {
        constexpr int e = decltype(ic)::value;
        f32x16_t lo, hi;
        acc_read<e>(lo);
        acc_read<e + 2>(hi);
        if (key < L) store_rope_bwd_pair(lo, hi, scale, rope_cos + pos * 64, rope_sin + pos * 64, e, half, kp_out + k_col0);
      });
    }
    static_for<ET>([&](auto ic) {
      constexpr int e = decltype(ic)::value;
      f32x16_t dk_e, dv_e;
      if (!rope_cos) acc_read<e>(dk_e);
      acc_read<4 + e>(dv_e);
      if (key < L) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          uint2 w;
          if (!rope_cos) {
            w.x = pack2bf(dk_e[rg * 4 + 0] * scale, dk_e[rg * 4 + 1] * scale);
            w.y = pack2bf(dk_e[rg * 4 + 2] * scale, dk_e[rg * 4 + 3] * scale);
            *(uint2*)(kp_out + k_col0 + e * 32 + rg * 8 + 4 * half) = w;
          }
          w.x = pack2bf(dv_e[rg * 4 + 0], dv_e[rg * 4 + 1]);
          w.y = pack2bf(dv_e[rg * 4 + 2], dv_e[rg * 4 + 3]);
          *(uint2*)(kp_out + v_col0 + e * 32 + rg * 8 + 4 * half) = w;
        }
      }
    });
  }  // pass
}


// =============================================================================================
// backward, dK/dV, version 5 (round 4): version 4's layouts and arithmetic under the schedule its phase stamps asked for
// (tools/gen_attn_dkv5.py has the analysis and generates the three straight-line bodies included below): sub-tile A runs one
// phase ahead of sub-tile B so every phase has VALU work of its own, the row terms enter through the MFMA accumulator inputs
// (S^T starts from -lse / scale, dP^T from -delta), ONE barrier per tile in front of the last phase with the next tile's first
// operands read behind it, a FOUR-stage LDS ring (133 KB) fetched two tiles ahead, transposed fragments two groups ahead.
// =============================================================================================
#define BF(x) __builtin_bit_cast(bf16x8_t, x)
// RV_DKV5_PROF (experiment builds): s_memtime stamps at the phase boundaries of a tile, accumulated per wave 0 of every workgroup
// into rv_dkv5_prof[] (ticks): 0 skipped-tile / loop glue, 1 P1, 2 P2, 3 P3, 4 P4, 5 end-of-tile waits, 6 barrier, 7 prologue +
// epilogue of a pass, 14 whole kernel per workgroup, 15 workgroups.  Each stamp drains lgkmcnt (s_memtime is an SMEM read).
#ifdef RV_DKV5_PROF
__device__ unsigned long long rv_dkv5_prof[16];
#define PROF(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pr[i] += (unsigned)(t_ - pr_last); pr_last = t_; }
#else
#define PROF(i)
#endif
template <bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv5_kernel(const bf16_t* __restrict__ qkv, long ld, int q_col0,
                                                               int k_col0, int v_col0,
                                                               const bf16_t* __restrict__ dO, long lddo,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta,
                                                               bf16_t* __restrict__ dqkv, long lddq, int Lmax, int H,
                                                               int nx, float scale, const int* __restrict__ seg_sh,
                                                               const int* __restrict__ seg_e1, int kv_group,
        const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, const int* __restrict__ rope_pos,
        const int* __restrict__ row_off, const int* __restrict__ row_len) {
  constexpr int HD = 128, KS = 8, ET = 4;
  constexpr int STAGE = 2 * 64 * 256 + 512;          // Q tile + dO tile + lse[64] + delta[64] = 0x8200
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, half = lane >> 5;
  int bx, h, s;
  attn_block_coords(nx, H, (int)gridDim.x / ((nx & 0xffff) * H), bx, h, s);
  const int L = row_len ? row_len[s] : Lmax;                   // pad-free rows: see attn_fwd2_kernel
  const long tok0 = row_off ? (long)row_off[s] : (long)s * Lmax;
  const int sh = seg_sh ? seg_sh[s] : 0, e1 = seg_e1 ? seg_e1[s] : 0;
  const int nkb = (L + 127) / 128;
  if (varlen_done(bx, nkb, CAUSAL && !((nx >> 21) & 1))) return;
  const float c = scale * LOG2E;
  const int HQ = H * kv_group;
  const long ws_plane = (long)((int)gridDim.x / ((nx & 0xffff) * H)) * HQ * Lmax;    // one [S, HQ, Lmax] plane of the delta workspace

  int d_row[4], d_chunk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    d_row[i] = (wave * 4 + i) * 4 + (lane >> 4);
    d_chunk[i] = (lane & 15) ^ (((d_row[i] & 3) << 2) | ((d_row[i] >> 2) & 3));
  }
  const uint32_t ldqb = (uint32_t)(ld * 2), lddob = (uint32_t)(lddo * 2);
  // One LDS-DMA piece of tile t into ring buffer buf: j = 0..7 -> Q piece j>>1 (even j) / dO piece j>>1 (odd j); j = 8: the tile's
  // lse (wave 0) or delta (wave 1).  Waves 0 and 1 therefore issue NP = 9 vector-memory operations per tile, waves 2 and 3 eight.
  auto issue_piece = [&](int hq, int t, int buf, int j) {
    uint8_t* st = smem + buf * STAGE;
    int qs0 = t * 64;
    asm volatile("" : "+s"(qs0));          // address arithmetic computed HERE, not hoisted to the top of the tile (spills)
    if (j < 8) {
      const int i = j >> 1;
      const uint32_t r = (uint32_t)min(qs0 + d_row[i], L - 1);
      if ((j & 1) == 0) {
        const char* base = (const char*)(qkv + tok0 * ld + q_col0 + hq * HD);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (__umul24(r, ldqb) + d_chunk[i] * 16u)),
                                         (__attribute__((address_space(3))) void*)(st + (wave * 4 + i) * 1024), 16, 0, 0);
      } else {
        const char* base = (const char*)(dO + tok0 * lddo + hq * HD);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (__umul24(r, lddob) + d_chunk[i] * 16u)),
                                         (__attribute__((address_space(3))) void*)(st + 16384 + (wave * 4 + i) * 1024), 16, 0, 0);
      }
    } else if (wave < 2) {         // plane 2 of the workspace = -lse / scale (wave 0), plane 1 = -delta (wave 1): written by the dQ kernel
      const int qq = min(qs0 + lane, L - 1);
      const float* src = delta + (wave == 0 ? 2 : 1) * ws_plane + ((long)s * HQ + hq) * Lmax + qq;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + 32768 + wave * 256), 4, 0, 0);
    }
  };
  auto issue_tile = [&](int hq, int t, int buf) {
#pragma unroll
    for (int j = 0; j < 9; ++j) issue_piece(hq, t, buf, j);
  };
  // everything of this wave but the NEWEST tile's pieces has landed
  auto wait_all_but_newest = [&]() {
    if (wave < 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  };

  const uint32_t lds0 = lds_addr_of(smem);
  uint32_t rb[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) rb[ks] = lds0 + qtile_off(fr, 2 * ks + half);
  uint32_t lh = lds0 + 32768u + (uint32_t)(half * 16);
  const int g4 = lane >> 4, s16 = lane & 15;
  uint32_t tb[2][ET];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int et = 0; et < ET; ++et)
      tb[u][et] = lds0 + qtile_off(8 * u + 4 * (g4 >> 1) + (s16 >> 2), et * 4 + 2 * (g4 & 1) + ((s16 & 3) >> 1)) +
                  (uint32_t)((s16 & 1) * 8);

  int blk_first, blk_second;
  pair_blocks<true>(bx, nkb, L, sh, e1, lane, CAUSAL && ((nx >> 20) & 1), CAUSAL && ((nx >> 21) & 1), blk_first, blk_second);
#ifdef RV_DKV5_PROF
  unsigned pr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pr_t0 = __builtin_amdgcn_s_memtime();
  unsigned long long pr_last = pr_t0;
#endif
  // Per-pass state (a causal workgroup owns TWO key blocks: blk_first, then blk_second).  The set-up of a pass - first two tiles
  // of its LDS ring, K / V fragments of its 32 keys per wave - is issued BEFORE the epilogue of the previous pass (accumulator
  // read-out, inverse rotation, stores), so the two memory latencies and the ~600 epilogue instructions overlap; version 3 / 4
  // spend 18 % of the kernel in these prologues and epilogues (phase stamps, profiles/r04_attn_dkv4_phase_profile.log).
  int kvb = blk_first, kv0 = 0, kv0w = 0, key = 0, keyc = 0, t_begin = 0, nt = 0;
  KeyLaneMask<CAUSAL> kmask;
  auto pass_begin = [&](int blk) {        // geometry of the pass + its K / V fragments -> AGPRs + the first two tiles of query head 0
    kvb = blk;
    kv0 = kvb * 128;
    kv0w = kv0 + wave * 32;
    key = kv0w + fr;
    keyc = min(key, L - 1);
    kmask.init(key, L, sh, e1);
    t_begin = CAUSAL ? (kv0 / 64) : 0;
    nt = (kv0 >= sh && kv0 + 127 < e1) ? min((L + 63) / 64, (e1 + 63) / 64) : (L + 63) / 64;
    issue_tile(h * kv_group, t_begin, 0);
    issue_tile(h * kv_group, min(t_begin + 1, nt - 1), 1);
    const bf16_t* kp = qkv + (tok0 + keyc) * ld + h * HD + 8 * half;
    static_for<KS>([&](auto ic) {
      constexpr int ks = decltype(ic)::value;
      frag_load<ks>(kp + k_col0 + 16 * ks);
      frag_load<8 + ks>(kp + v_col0 + 16 * ks);
    });
  };
  PROF(11);
  pass_begin(blk_first);
  static_for<8>([&](auto ic) { acc_zero<decltype(ic)::value>(); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int npass = CAUSAL ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    for (int gq = 0; gq < kv_group; ++gq) {
      const int hq = h * kv_group + gq;
      if (gq > 0) {                     // (query head 0: issued by pass_begin, landed and published before this point)
        issue_tile(hq, t_begin, 0);
        issue_tile(hq, min(t_begin + 1, nt - 1), 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      PROF(7);
      // carried from tile to tile: sub-tile A's accumulators (pre-loaded with -lse / scale and -delta), its first dO row
      // fragments (the matching Q fragments sit in AGPRs a[192:207])
      f32x16_t sA, pA;
      bf16x8_t foA[KS];
      // quarter j (rows 4 j .. 4 j + 3) of an accumulator input <- four consecutive floats of the tile's -lse / scale or -delta
      auto acc_quarter = [&](f32x16_t& ax, int j, const f32x4_t v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ax[4 * j + i] = v[i];
      };
      {
#include "attn_dkv5_prefetch.inc"
      }
      int buf = 0;
      for (int t = t_begin; t < nt; ++t) {
        const int qs0 = t * 64;
        const int tn = min(t + 2, nt - 1);               // fetched meanwhile (clamped: the DMA issues stay unconditional)
        const int bufn = (buf + 2) & 3;
        const uint32_t flip = buf == 3 ? (uint32_t)(-3 * STAGE) : (uint32_t)STAGE;
        auto flip_rows = [&]() {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) rb[ks] += flip;
          lh += flip;
        };
        auto flip_tr = [&]() {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int et = 0; et < ET; ++et) tb[u][et] += flip;
        };
        if ((CAUSAL && qs0 + 63 < kv0w) || (qs0 >= e1 && kv0w >= sh && kv0w + 31 < e1)) {
#include "attn_dkv5_skip.inc"
        } else {
          bool need_mask_a, need_mask_b;
          auto mask_flags = [&]() {            // computed under the first MFMAs (an opaque copy of qs0 keeps the SALU chain here)
            int q2 = qs0;
            asm volatile("" : "+s"(q2));
            need_mask_a = (q2 + 31 >= L) || (kv0w + 31 >= L) || (CAUSAL && kv0w + 31 > q2) ||
                          (q2 + 31 >= e1 && kv0w + 31 >= sh && kv0w < e1);
            need_mask_b = (q2 + 63 >= L) || (kv0w + 31 >= L) || (CAUSAL && kv0w + 31 > q2 + 32) ||
                          (q2 + 63 >= e1 && kv0w + 31 >= sh && kv0w < e1);
          };
          f32x16_t sB, pB;
          bf16x8_t foB[KS], tr[3][ET];
          u32x4_t pfA0, pfA1, pfB0, pfB1, dsA0, dsA1, dsB0, dsB1;
          // Each helper PINS what it produced with an empty volatile asm (pure VALU values have no ordering against the
          // volatile asm MFMAs / reads around them: hipcc otherwise sinks whole stages to their first use).
          auto exps = [&](f32x16_t& sx, int r0, int r1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (r >= r0 && r < r1) {
                float p = __builtin_amdgcn_exp2f(sx[r] * c);          // masked: exp2(-inf) = 0
                asm volatile("" : "+v"(p));
                sx[r] = p;
              }
          };
          auto dsmul = [&](f32x16_t& px, const f32x16_t& sx, int r0, int r1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (r >= r0 && r < r1) {
                float d = sx[r] * px[r];
                asm volatile("" : "+v"(d));
                px[r] = d;
              }
          };
          auto pack4 = [&](u32x4_t& pf, const f32x16_t& a, int base) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              uint32_t w = pack2bf(a[base + 2 * d], a[base + 2 * d + 1]);
              asm volatile("" : "+v"(w));
              pf[d] = w;
            }
          };
#include "attn_dkv5_body.inc"
        }
        buf = (buf + 1) & 3;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped re-fetches of the last tiles: nothing may land later
      __syncthreads();
      PROF(8);
      const uint32_t back = (uint32_t)(buf * STAGE);       // the bases point into ring buffer `buf`: back to buffer 0
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) rb[ks] -= back;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int et = 0; et < ET; ++et) tb[u][et] -= back;
      lh -= back;
    }  // query heads of the group
    // the finished pass's output rows, then the NEXT pass's set-up (its LDS ring is free: barrier above; its K / V AGPRs were last
    // read by MFMAs issued a whole phase ago), then the read-out of this pass underneath those loads
    const long pos_row = tok0 + keyc;
    const int out_row0 = kv0w;                  // first key row of this wave's finished tile
    const int next_blk = (pass + 1 < npass) ? blk_second : -1;
    if (next_blk >= 0) pass_begin(next_blk);
    PROF(9);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // last asm MFMA -> v_accvgpr_read (hipcc pads nothing for asm)
    // Read-out through LDS.  A lane owns ONE key row and 4 consecutive columns per accumulator row group: storing from registers
    // is 32 eight-byte stores per wave instruction into 32 different 24 KB-strided rows (phase stamps: 10 % of the kernel).  Each
    // wave stages its [32 keys][128] dK and dV tiles (bf16, 16-byte chunks XOR-swizzled by the row) in ring buffers 2 / 3 - the
    // next pass's first tiles go to buffers 0 / 1 - and writes them out as whole 256-byte rows, 16 bytes per lane.
    uint8_t* stg = smem + 2 * STAGE + wave * 16384;
    auto stage = [&](int m, int e, int rg, uint32_t w0, uint32_t w1) {       // 4 columns e*32 + rg*8 + 4*half .. of key row fr
      uint2 w;
      w.x = w0;
      w.y = w1;
      *(uint2*)(stg + m * 8192 + fr * 256 + (((e * 4 + rg) ^ (fr & 15)) << 4) + half * 8) = w;
    };
    if (rope_cos) {       // dK leaves through the inverse rotation (pairs = tiles e, e + 2): dx1 = dy1 cos + dy2 sin, dx2 = dy2 cos - dy1 sin
      const long pos = rope_pos ? rope_pos[pos_row] : (long)(pos_row - tok0);
      const float* cr = rope_cos + pos * 64;
      const float* sr = rope_sin + pos * 64;
      static_for<2>([&](auto ic) {
        constexpr int e = decltype(ic)::value;
        f32x16_t lo, hi;
        acc_read<e>(lo);
        acc_read<e + 2>(hi);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int i0 = e * 32 + rg * 8 + 4 * half;
          const f32x4_t cc = *(const f32x4_t*)(cr + i0), ss = *(const f32x4_t*)(sr + i0);
          float o1[4], o2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x = bf2f(f2bf(lo[rg * 4 + j] * scale)), y = bf2f(f2bf(hi[rg * 4 + j] * scale));
            o1[j] = x * cc[j] + y * ss[j];
            o2[j] = y * cc[j] - x * ss[j];
          }
          stage(0, e, rg, pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3]));
          stage(0, e + 2, rg, pack2bf(o2[0], o2[1]), pack2bf(o2[2], o2[3]));
        }
      });
    }
    static_for<ET>([&](auto ic) {
      constexpr int e = decltype(ic)::value;
      f32x16_t dk_e, dv_e;
      if (!rope_cos) acc_read<e>(dk_e);
      acc_read<4 + e>(dv_e);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        if (!rope_cos)
          stage(0, e, rg, pack2bf(dk_e[rg * 4 + 0] * scale, dk_e[rg * 4 + 1] * scale), pack2bf(dk_e[rg * 4 + 2] * scale, dk_e[rg * 4 + 3] * scale));
        stage(1, e, rg, pack2bf(dv_e[rg * 4 + 0], dv_e[rg * 4 + 1]), pack2bf(dv_e[rg * 4 + 2], dv_e[rg * 4 + 3]));
      }
    });
    const bool all_rows = out_row0 + 31 < L;      // wave-uniform: every row of the tile exists -> exactly 16 store instructions
    {
      bf16_t* out0 = dqkv + (tok0 + out_row0) * lddq + h * HD;
      const int pc = lane & 15;
      auto write_rows = [&](auto pred) {
        u32x4_t v[2][8];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int i = 0; i < 8; ++i) v[m][i] = *(const u32x4_t*)(stg + m * 8192 + (4 * i + (lane >> 4)) * 256 + (pc << 4));
        // hipcc's waitcnt insertion does not look at inline-asm operands: without this wait the asm stores below sent registers
        // whose LDS read had not landed yet (seen as garbage in the last rows of dK, non-deterministically)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + (lane >> 4);
            bf16_t* dst = out0 + (long)row * lddq + (m ? v_col0 : k_col0) + ((pc ^ (row & 15)) << 3);
            if constexpr (decltype(pred)::value) {
              if (out_row0 + row < L) *(u32x4_t*)dst = v[m][i];
            } else {      // asm: EXACTLY one store instruction each - the counted vmcnt(16) at the end of the pass relies on it
              // (s_nop: a store of more than 8 bytes reads its data registers late - they must not be rewritten in the next
              // cycle; hipcc pads its own stores, not inline asm: the address arithmetic of the following store reused them)
              asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v[m][i]) : "memory");
            }
          }
      };
      if (all_rows) write_rows(std::false_type{});
      else write_rows(std::true_type{});
    }
    PROF(10);
    if (next_blk < 0) break;
    static_for<8>([&](auto ic) { acc_zero<decltype(ic)::value>(); });
    // the next pass's K / V fragments and first tiles have landed; this pass's 16 row stores - the NEWEST vector-memory operations -
    // may stay in flight (a wave with a ragged last tile issued an unknown number of them: it waits for everything)
    if (all_rows) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    PROF(7);
  }  // pass
#ifdef RV_DKV5_PROF
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) atomicAdd(&rv_dkv5_prof[i], (unsigned long long)pr[i]);
    atomicAdd(&rv_dkv5_prof[14], __builtin_amdgcn_s_memtime() - pr_t0);
    atomicAdd(&rv_dkv5_prof[15], 1ull);
  }
#endif
}
#undef BF
#undef PROF

#include "attn_fwd3.inc"

}  // namespace

static int g_dkv_override = 0;
static int g_fwd_override = 0;

extern "C" {

int rv_set_attn_dkv_version(int version) {
  RV_REQUIRE(version == 0 || version == 3 || version == 5, "rv_set_attn_dkv_version: 0 (environment / default), 3 or 5");
  g_dkv_override = version;
  return 0;
}

int rv_set_attn_fwd_version(int version) {
  RV_REQUIRE(version == 0 || version == 2 || version == 3, "rv_set_attn_fwd_version: 0 (environment / default), 2 or 3");
  g_fwd_override = version;
  return 0;
}

long rv_attn_bwd_workspace_floats(int S, int H, int L) { return 3L * S * H * L; }

#ifdef RV_ATTN_PROF
// experiment builds only: read and clear the phase counters of the forward / dQ kernels (16 x u64)
int rv_debug_attn_prof(unsigned long long* out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rv_attn_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(rv_attn_prof), z, sizeof(z)) != hipSuccess;
}
#endif
#ifdef RV_DKV5_PROF
// experiment builds only: read and clear the phase counters of attn_bwd_dkv5_kernel (16 x u64)
int rv_debug_dkv5_prof(unsigned long long* out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rv_dkv5_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(rv_dkv5_prof), z, sizeof(z)) != hipSuccess;
}
#endif

int rv_attn_fwd(const void* qkv, long ld, int q_col0, int k_col0, int v_col0, void* out, long ldo, float* lse, int S,
                int L, int H, int hd, int causal, float scale, const int* seg_sh, const int* seg_e1, int kv_group,
                const int* row_off, const int* row_len, void* stream) {
  RV_REQUIRE(hd == 64 || hd == 128, "rv_attn_fwd: head dim must be 64 or 128");
  RV_REQUIRE((row_off == nullptr) == (row_len == nullptr), "rv_attn_fwd: row_off and row_len go together");
  RV_REQUIRE(kv_group >= 1 && H % kv_group == 0, "rv_attn_fwd: kv_group must divide the number of query heads");
  RV_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && q_col0 % 8 == 0 && k_col0 % 8 == 0 && v_col0 % 8 == 0,
             "rv_attn_fwd: alignment");
  RV_REQUIRE((double)L * (double)ld * 2.0 < 4.0e9, "rv_attn_fwd: one sequence must span < 4 GB of the qkv buffer (32-bit DMA offsets)");
  if (S == 0 || L == 0) return 0;
  static int map_mode = -1;
  if (map_mode < 0) { const char* e = getenv("RV_ATTN_MAP"); map_mode = e ? atoi(e) : 1; }
  hipStream_t st = (hipStream_t)stream;
  static int fwd_nw = -1;         // RV_ATTN_FWD_NW: 4 = 128-query workgroups (two per CU), 8 = 256-query workgroups on one K / V ring (hd 128)
  if (fwd_nw < 0) { const char* e = getenv("RV_ATTN_FWD_NW"); fwd_nw = (e && atoi(e) == 8) ? 8 : RV_ATTN_FWD_NW_DEFAULT; }
  static int fwd_env = -1;        // RV_ATTN_FWD: 3 = attn_fwd3_kernel (hd 128; csrc/attn_fwd3.inc), 2 = attn_fwd2_kernel
  if (fwd_env < 0) { const char* e = getenv("RV_ATTN_FWD"); fwd_env = (e && (atoi(e) == 2 || atoi(e) == 3)) ? atoi(e) : RV_ATTN_FWD_DEFAULT; }
  const int fwd_ver = g_fwd_override ? g_fwd_override : fwd_env;
  const bool v3 = hd == 128 && fwd_ver == 3;
  const int nw = v3 ? 8 : (hd == 128) ? fwd_nw : 4;             // version 3: 256-query blocks (4 waves x 64 rows)
  const int nb = (L + 32 * nw - 1) / (32 * nw);
  static int pair_mode = -1;      // RV_ATTN_PAIR: 0 = the plain (x, n-1-x) pairing, 1 = pair_blocks (ranked pairs), 2 = one ranked block per workgroup
  if (pair_mode < 0) { const char* e = getenv("RV_ATTN_PAIR"); pair_mode = e ? atoi(e) : 1; }
  const int nxr = (causal && pair_mode != 2) ? (nb + 1) / 2 : nb;
  const int nx = nxr | (map_mode << 16) | ((pair_mode ? 1 : 0) << 20) | ((causal && pair_mode == 2 ? 1 : 0) << 21);
  dim3 grid(nxr * H * S), block(64 * nw);
  static bool attr_done = false;
  if (v3) {
    static bool attr3_done = false;
    if (!attr3_done) {
      hipFuncSetAttribute((const void*)attn_fwd3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 64 * 128 * 2);
      hipFuncSetAttribute((const void*)attn_fwd3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 64 * 128 * 2);
      attr3_done = true;
    }
    if (causal)
      hipLaunchKernelGGL((attn_fwd3_kernel<true>), grid, dim3(256), 7 * 64 * 128 * 2, st, (const bf16_t*)qkv, ld, q_col0, k_col0, v_col0,
                         (bf16_t*)out, ldo, lse, L, H, nx, scale, seg_sh, seg_e1, kv_group, row_off, row_len);
    else
      hipLaunchKernelGGL((attn_fwd3_kernel<false>), grid, dim3(256), 7 * 64 * 128 * 2, st, (const bf16_t*)qkv, ld, q_col0, k_col0, v_col0,
                         (bf16_t*)out, ldo, lse, L, H, nx, scale, seg_sh, seg_e1, kv_group, row_off, row_len);
    RV_CHECK_LAUNCH();
    return 0;
  }
  if (!attr_done) {
    hipFuncSetAttribute((const void*)attn_fwd2_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)attn_fwd2_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)attn_fwd2_kernel<128, true, 0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)attn_fwd2_kernel<128, false, 0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr_done = true;
  }
#define LAUNCH_FWD(HD_, C_)                                                                                      \
  hipLaunchKernelGGL((attn_fwd2_kernel<HD_, C_>), grid, block, 4 * 64 * HD_ * 2, st, (const bf16_t*)qkv, ld, q_col0, \
                     k_col0, v_col0, (bf16_t*)out, ldo, lse, L, H, nx, scale, seg_sh, seg_e1, kv_group, row_off, row_len)
#ifdef RV_ATTN_EXPERIMENTS
  static int fwd_ablate = -1;
  if (fwd_ablate < 0) { const char* a = getenv("RV_FWD_ABLATE"); fwd_ablate = a ? atoi(a) : 0; }
#define LAUNCH_FWD_ABL(N)                                                                                                  \
  if (fwd_ablate == N && hd == 128 && causal) {                                                                            \
    hipFuncSetAttribute((const void*)attn_fwd2_kernel<128, true, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);     \
    hipLaunchKernelGGL((attn_fwd2_kernel<128, true, N>), grid, block, 4 * 64 * 128 * 2, st, (const bf16_t*)qkv, ld, q_col0, \
                       k_col0, v_col0, (bf16_t*)out, ldo, lse, L, H, nx, scale, seg_sh, seg_e1, kv_group, row_off, row_len);                   \
    RV_CHECK_LAUNCH();                                                                                                     \
    return 0;                                                                                                              \
  }
  LAUNCH_FWD_ABL(1) LAUNCH_FWD_ABL(2) LAUNCH_FWD_ABL(3) LAUNCH_FWD_ABL(6)
#undef LAUNCH_FWD_ABL
#endif
  if (hd == 128 && nw == 8) {
    if (causal)
      hipLaunchKernelGGL((attn_fwd2_kernel<128, true, 0, 8>), grid, block, 4 * 64 * 128 * 2, st, (const bf16_t*)qkv, ld, q_col0, k_col0,
                         v_col0, (bf16_t*)out, ldo, lse, L, H, nx, scale, seg_sh, seg_e1, kv_group, row_off, row_len);
    else
      hipLaunchKernelGGL((attn_fwd2_kernel<128, false, 0, 8>), grid, block, 4 * 64 * 128 * 2, st, (const bf16_t*)qkv, ld, q_col0, k_col0,
                         v_col0, (bf16_t*)out, ldo, lse, L, H, nx, scale, seg_sh, seg_e1, kv_group, row_off, row_len);
  } else if (hd == 128) { if (causal) LAUNCH_FWD(128, true); else LAUNCH_FWD(128, false); }
  else { if (causal) LAUNCH_FWD(64, true); else LAUNCH_FWD(64, false); }
#undef LAUNCH_FWD
  RV_CHECK_LAUNCH();
  return 0;
}

int rv_attn_bwd(const void* qkv, long ld, int q_col0, int k_col0, int v_col0, const void* dO, long lddo,
                const void* O, long ldo, const float* lse, float* delta, void* dqkv, long lddq, int S, int L, int H,
                int hd, int causal, float scale, const int* seg_sh, const int* seg_e1, int kv_group,
                const float* rope_cos, const float* rope_sin, const int* rope_pos, const int* row_off, const int* row_len,
                void* stream) {
  RV_REQUIRE(hd == 128, "rv_attn_bwd: head dim must be 128");
  RV_REQUIRE((row_off == nullptr) == (row_len == nullptr), "rv_attn_bwd: row_off and row_len go together");
  RV_REQUIRE(row_off == nullptr || rope_cos == nullptr || rope_pos != nullptr,
             "rv_attn_bwd: pad-free rows with the fused inverse rotation need the position table (rope_pos)");
  RV_REQUIRE((rope_cos == nullptr) == (rope_sin == nullptr), "rv_attn_bwd: rope_cos and rope_sin go together");
  RV_REQUIRE(O != nullptr && delta != nullptr && ldo % 8 == 0, "rv_attn_bwd: O (forward output) and the delta workspace are required");
  RV_REQUIRE(kv_group >= 1 && H % kv_group == 0, "rv_attn_bwd: kv_group must divide the number of query heads");
  RV_REQUIRE((seg_sh == nullptr) == (seg_e1 == nullptr), "rv_attn_bwd: seg_sh and seg_e1 go together");
  RV_REQUIRE(ld % 8 == 0 && lddo % 8 == 0 && lddq % 8 == 0, "rv_attn_bwd: alignment");
  RV_REQUIRE((double)L * (double)(ld > lddo ? ld : lddo) * 2.0 < 4.0e9, "rv_attn_bwd: one sequence must span < 4 GB of the qkv / dO buffers");
  if (S == 0 || L == 0) return 0;
  const int nb = (L + 127) / 128;
  static int map_mode = -1;
  if (map_mode < 0) { const char* e = getenv("RV_ATTN_MAP"); map_mode = e ? atoi(e) : 1; }
  static int pair_mode = -1;
  if (pair_mode < 0) { const char* e = getenv("RV_ATTN_PAIR"); pair_mode = e ? atoi(e) : 1; }
  const int nxr = (causal && pair_mode != 2) ? (nb + 1) / 2 : nb;
  const int nx = nxr | (map_mode << 16) | ((pair_mode ? 1 : 0) << 20) | ((causal && pair_mode == 2 ? 1 : 0) << 21);
  dim3 grid(nxr * H * S), block(256);
  hipStream_t st = (hipStream_t)stream;
  constexpr int DQ_LDS = 4 * 64 * 256;
  constexpr int DKV_LDS = 2 * (2 * 64 * 256 + 512);
  constexpr int DKV5_LDS = 4 * (2 * 64 * 256 + 512);     // version 5: four-stage ring
  static bool attr_done = false;
  static int dkv_env = 5;            // round 4: version 5 (tools/gen_attn_dkv5.py); RV_ATTN_DKV=3 selects the round-2/3 kernel
  static int dkv_ablate = 0;
  (void)dkv_ablate;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)attn_bwd_dq2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS);
    hipFuncSetAttribute((const void*)attn_bwd_dq2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS);
    hipFuncSetAttribute((const void*)attn_bwd_dkv3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
    hipFuncSetAttribute((const void*)attn_bwd_dkv3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
#ifdef RV_ATTN_EXPERIMENTS
    hipFuncSetAttribute((const void*)attn_bwd_dkv3_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
    hipFuncSetAttribute((const void*)attn_bwd_dkv3_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
    hipFuncSetAttribute((const void*)attn_bwd_dkv3_kernel<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
    hipFuncSetAttribute((const void*)attn_bwd_dkv3_kernel<true, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
    hipFuncSetAttribute((const void*)attn_bwd_dkv3_kernel<true, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
    { const char* a = getenv("RV_DKV_ABLATE"); if (a) dkv_ablate = atoi(a); }
#endif
    hipFuncSetAttribute((const void*)attn_bwd_dkv5_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV5_LDS);
    hipFuncSetAttribute((const void*)attn_bwd_dkv5_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV5_LDS);
    const char* e = getenv("RV_ATTN_DKV");        // A/B knob: 3 = the round-2/3 kernel, 5 = the round-4 kernel (versions 2 and 4: history)
    if (e && (atoi(e) == 3 || atoi(e) == 5)) dkv_env = atoi(e);
    attr_done = true;
  }
  const int dkv_version = g_dkv_override ? g_dkv_override : dkv_env;      // rv_set_attn_dkv_version (tests: both kernels in one process)
  // dQ: one workgroup per (query head, query block); dK/dV: per (KEY/VALUE head, key block), looping over its query heads
  const int Hkv = H / kv_group;
  dim3 grid_kv(nxr * Hkv * S);
#define BWD_HEAD (const bf16_t*)qkv, ld, q_col0, k_col0, v_col0, (const bf16_t*)dO, lddo
#define BWD_TAIL(HH) (bf16_t*)dqkv, lddq, L, HH, nx, scale, seg_sh, seg_e1, kv_group, rope_cos, rope_sin, rope_pos, row_off, row_len
  if (causal) {
    hipLaunchKernelGGL((attn_bwd_dq2_kernel<true>), grid, block, DQ_LDS, st, BWD_HEAD, (const bf16_t*)O, ldo, lse, delta, BWD_TAIL(H));
    RV_CHECK_LAUNCH();
    if (dkv_version == 5)
      hipLaunchKernelGGL((attn_bwd_dkv5_kernel<true>), grid_kv, block, DKV5_LDS, st, BWD_HEAD, lse, (const float*)delta, BWD_TAIL(Hkv));
#ifdef RV_ATTN_EXPERIMENTS
#define RV_ABL_LAUNCH(N) else if (dkv_ablate == N) hipLaunchKernelGGL((attn_bwd_dkv3_kernel<true, N>), grid_kv, block, DKV_LDS, st, BWD_HEAD, lse, (const float*)delta, BWD_TAIL(Hkv));
    RV_ABL_LAUNCH(1) RV_ABL_LAUNCH(2) RV_ABL_LAUNCH(3) RV_ABL_LAUNCH(5) RV_ABL_LAUNCH(6)
#undef RV_ABL_LAUNCH
#endif
    else
      hipLaunchKernelGGL((attn_bwd_dkv3_kernel<true>), grid_kv, block, DKV_LDS, st, BWD_HEAD, lse, (const float*)delta, BWD_TAIL(Hkv));
  } else {
    hipLaunchKernelGGL((attn_bwd_dq2_kernel<false>), grid, block, DQ_LDS, st, BWD_HEAD, (const bf16_t*)O, ldo, lse, delta, BWD_TAIL(H));
    RV_CHECK_LAUNCH();
    if (dkv_version == 5)
      hipLaunchKernelGGL((attn_bwd_dkv5_kernel<false>), grid_kv, block, DKV5_LDS, st, BWD_HEAD, lse, (const float*)delta, BWD_TAIL(Hkv));
    else
      hipLaunchKernelGGL((attn_bwd_dkv3_kernel<false>), grid_kv, block, DKV_LDS, st, BWD_HEAD, lse, (const float*)delta, BWD_TAIL(Hkv));
  }
#undef BWD_HEAD
#undef BWD_TAIL
  RV_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
