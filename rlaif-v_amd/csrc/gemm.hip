// C-ABI launchers for the bf16 NT GEMM and the fused LM-head log-prob kernels (gemm.hpp).
#include "gemm.hpp"
#include "rlaifv_hip.h"

#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";
void rv_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

// variants: 0 = 128x128x64 register staging, 1 = 128x128x64 global_load_lds, 2 = 256x256x32 ping-pong,
// -1 (default) = pick 2 when the problem fills the chip with 256x256 tiles, else 1.
static int g_default_variant = -1;
static int g_group = 0;     // experiment knob: RV_GEMM_GROUP
static int epi_narrow() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("RV_EPI_WIDE"); v = (e && atoi(e) == 0) ? 1 : 0; }
  return v;
}
static void read_group_env() {
  static bool env_done = false;
  if (!env_done) { const char* e = getenv("RV_GEMM_GROUP"); if (e) g_group = atoi(e); env_done = true; }
}

template <class Epi, bool DMA_IN_MSEG, int ABLATE = 0, int DIST = 3, int SPLIT = 0>
static int launch_gemm256(const GemmShape& g, const Epi& epi, hipStream_t st) {
  constexpr int LDS = (DMA_IN_MSEG ? DIST + 1 : 4) * G2_STAGE_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_nt_256_kernel<Epi, DMA_IN_MSEG, ABLATE, DIST, SPLIT>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  const int tiles_m = (g.M + G2_BM - 1) / G2_BM, tiles_n = (g.N + G2_BN - 1) / G2_BN;
  hipLaunchKernelGGL((gemm_nt_256_kernel<Epi, DMA_IN_MSEG, ABLATE, DIST, SPLIT>), dim3(tiles_m * tiles_n), dim3(G2_THREADS),
                     LDS, st, g, epi);
  RV_CHECK_LAUNCH();
  return 0;
}

template <class Epi>
static int launch_gemm256x64(const GemmShape& g, const Epi& epi, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_nt_256x64_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES);
    attr_done = true;
  }
  const int tiles_m = (g.M + G2_BM - 1) / G2_BM, tiles_n = (g.N + G2_BN - 1) / G2_BN;
  hipLaunchKernelGGL((gemm_nt_256x64_kernel<Epi>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G3_LDS_BYTES, st, g, epi);
  RV_CHECK_LAUNCH();
  return 0;
}

// RV_GEMM_MI16 (default 1): the 16x16x32-MFMA main loops of the 64-deep-A NN kernel and the TN kernel (gemm.hpp "MI16"); 0 = 32x32x16
static int g_mi16 = -1;
static int nn_mi16() {
  if (g_mi16 < 0) { const char* e = getenv("RV_GEMM_MI16"); g_mi16 = e ? atoi(e) : 1; }
  return g_mi16;
}

static int g_tn_dist = 3;   // prefetch distance of the TN kernel (RV_GEMM_TN_DIST = 3 | 4; measured equal, 3 = 128 KiB LDS)

template <class Epi, int DIST>
static int launch_gemm_tn_d(const bf16_t* P, long ldp, const bf16_t* Q, long ldq, int R, int I, int J, const Epi& epi,
                            hipStream_t st, int splits, int r_chunk, long split_stride, int bid0, int nblocks) {
  constexpr int LDS = (DIST + 1) * G2_STAGE_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_tn_256_kernel<Epi, DIST>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)gemm_tn_256_kernel<Epi, DIST, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  const int tiles_i = (I + 255) / 256, tiles_j = (J + 255) / 256;
  const int gx = nblocks > 0 ? nblocks : tiles_i * tiles_j;          // launch-order positions bid0 .. bid0 + gx - 1
  if (nn_mi16())
    hipLaunchKernelGGL((gemm_tn_256_kernel<Epi, DIST, true>), dim3(gx, splits), dim3(G2_THREADS), LDS, st, P,
                       ldp, Q, ldq, R, I, J, epi, r_chunk, split_stride, g_group, bid0);
  else
    hipLaunchKernelGGL((gemm_tn_256_kernel<Epi, DIST>), dim3(gx, splits), dim3(G2_THREADS), LDS, st, P,
                       ldp, Q, ldq, R, I, J, epi, r_chunk, split_stride, g_group, bid0);
  RV_CHECK_LAUNCH();
  return 0;
}

template <class Epi>
static int launch_gemm_tn(const bf16_t* P, long ldp, const bf16_t* Q, long ldq, int R, int I, int J, const Epi& epi,
                          hipStream_t st, int splits = 1, int r_chunk = 0, long split_stride = 0, int bid0 = 0,
                          int nblocks = 0) {
  static bool env_done = false;
  read_group_env();
  if (!env_done) { const char* e = getenv("RV_GEMM_TN_DIST"); if (e && atoi(e) == 4) g_tn_dist = 4; env_done = true; }
  if (g_tn_dist == 3) return launch_gemm_tn_d<Epi, 3>(P, ldp, Q, ldq, R, I, J, epi, st, splits, r_chunk, split_stride, bid0, nblocks);
  return launch_gemm_tn_d<Epi, 4>(P, ldp, Q, ldq, R, I, J, epi, st, splits, r_chunk, split_stride, bid0, nblocks);
}

// Tail split of a weight-gradient GEMM.  All output tiles cost the same (the contraction runs over every token), so a
// launch of T tiles takes ceil(T / 256) rounds on 256 CUs and the last round is as long as a full one however few tiles it
// holds (wgu: 1376 tiles = 5.375 rounds, wdown: 688 = 2.69).  (In practice a thin last round runs somewhat faster than a full
// one - fewer CUs share the fabric - so the measured gain is a third of this model's: wgu 3.76 -> 3.65 ms, step -0.5 %.)  Plan: the first floor(T / 256) * 256 launch-order positions
// run as they are; the `tail` remaining tiles are split s ways over the token axis (s x tail workgroups of 1 / s length,
// fp32 tile-dense slabs) and summed in fixed order by tn_tail_reduce_kernel.  The tail then takes ceil(tail * s / 256) / s of
// a round instead of 1.  s = 1: no split pays (or the problem has no full round).
struct TnTailPlan { int full_blocks, tail, splits, r_chunk; };
static TnTailPlan tn_tail_plan(int R, int I, int J) {
  static int enabled = -1;
  static double penalty = 0.04;       // rounds charged per split for its fixed costs (prologue, slab store, reduce pass); measured:
                                      // 8 splits of the wgu tail gain nothing, 2 splits -2.7 % (profiles/r03_tn_tail_split.log)
  if (enabled < 0) {
    const char* e = getenv("RV_TN_TAIL_SPLIT");
    enabled = e ? atoi(e) : 1;
    const char* q = getenv("RV_TN_TAIL_PENALTY");
    if (q) penalty = atof(q);
  }
  const int T = ((I + 255) / 256) * ((J + 255) / 256);
  TnTailPlan p{T, 0, 1, 0};
  const int full = T / 256, tail = T % 256;
  if (!enabled || full < 1 || tail == 0) return p;
  double best = 1.0;
  int best_s = 1;
  const int cand[5] = {2, 3, 4, 6, 8};
  for (int c = 0; c < 5; ++c) {
    const int s = cand[c];
    if (R / s < 2048) break;                                   // keep every split a long contraction
    const double t = (double)((tail * s + 255) / 256) / s + penalty * s;
    if (t < best - 1e-9) { best = t; best_s = s; }
  }
  if (1.0 - best < 0.15) return p;                             // not worth two more launches
  p.full_blocks = full * 256;
  p.tail = tail;
  p.splits = best_s;
  p.r_chunk = ((R + best_s - 1) / best_s + 31) / 32 * 32;
  p.splits = (R + p.r_chunk - 1) / p.r_chunk;
  return p;
}

// out tile = bf16(sum over splits of the fp32 slabs), fixed order; one workgroup per tail tile, same tile walk as the GEMM
__global__ __launch_bounds__(256) void tn_tail_reduce_kernel(const float* __restrict__ ws, int splits, int tail, int bid0, int I,
                                                              int J, int group, bf16_t* __restrict__ C, long ldc) {
  int i0, j0;
  tn_tile_origin((int)blockIdx.x + bid0, I, J, group, i0, j0);
  const float* base = ws + (long)blockIdx.x * 65536;
  for (int q = threadIdx.x; q < 256 * 64; q += 256) {          // 256 rows x 64 float4
    const int r = q >> 6, c4 = (q & 63) * 4;
    if (i0 + r >= I || j0 + c4 >= J) continue;
    float4 a = *(const float4*)(base + r * 256 + c4);
    for (int s = 1; s < splits; ++s) {
      const float4 b = *(const float4*)(base + (long)s * tail * 65536 + r * 256 + c4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    uint2 o;
    o.x = pack2bf(a.x, a.y);
    o.y = pack2bf(a.z, a.w);
    *(uint2*)(C + (long)(i0 + r) * ldc + j0 + c4) = o;
  }
}

template <int STAGE, class Epi>
static int launch_gemm(const GemmShape& g, const Epi& epi, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_nt_kernel<STAGE, Epi>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        GEMM_LDS_BYTES);
    attr_done = true;
  }
  const int tiles_m = (g.M + GEMM_BM - 1) / GEMM_BM, tiles_n = (g.N + GEMM_BN - 1) / GEMM_BN;
  hipLaunchKernelGGL((gemm_nt_kernel<STAGE, Epi>), dim3(tiles_m * tiles_n), dim3(GEMM_THREADS), GEMM_LDS_BYTES, st, g,
                     epi);
  RV_CHECK_LAUNCH();
  return 0;
}

// GEMMs with a second contraction segment (GemmShape::A2/B2/K2): the 256x256 ping-pong kernel when the problem
// fills the chip and the column groups are tile aligned, else the 128x128 kernel.
template <class Epi>
static int dispatch_ext(const GemmShape& g, const Epi& epi, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const long t256 = (long)((g.M + G2_BM - 1) / G2_BM) * ((g.N + G2_BN - 1) / G2_BN);
  if (t256 >= 192 && g.group_cols % G2_BN == 0 && g.group0 % G2_BN == 0) {
    constexpr int LDS = 4 * G2_STAGE_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
      hipFuncSetAttribute((const void*)gemm_nt_256_kernel<Epi, true, 0, 3, 0, true>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      attr_done = true;
    }
    hipLaunchKernelGGL((gemm_nt_256_kernel<Epi, true, 0, 3, 0, true>),
                       dim3(((g.M + G2_BM - 1) / G2_BM) * ((g.N + G2_BN - 1) / G2_BN)), dim3(G2_THREADS), LDS, st, g, epi);
    RV_CHECK_LAUNCH();
    return 0;
  }
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_nt_kernel<1, Epi, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        GEMM_LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<1, Epi, true>),
                     dim3(((g.M + GEMM_BM - 1) / GEMM_BM) * ((g.N + GEMM_BN - 1) / GEMM_BN)), dim3(GEMM_THREADS),
                     GEMM_LDS_BYTES, st, g, epi);
  RV_CHECK_LAUNCH();
  return 0;
}

// out[i][j] = bf16(alpha * sum_s ws[s][i][j])   (deterministic second pass of the split-K TN GEMM)
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int I, int J, float alpha,
                                     bf16_t* __restrict__ out, long ldc) {
  const long n4 = (long)I * J / 4;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
    const long e = q * 4;
    float4 a = *(const float4*)(ws + e);
    for (int s = 1; s < splits; ++s) {
      const float4 b = *(const float4*)(ws + (long)s * I * J + e);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const long i = e / J, j = e % J;
    uint2 o;
    o.x = pack2bf(a.x * alpha, a.y * alpha);
    o.y = pack2bf(a.z * alpha, a.w * alpha);
    *(uint2*)(out + i * ldc + j) = o;
  }
}

template <class Epi>
static int dispatch(const GemmShape& g, const Epi& epi, int variant, void* stream) {
  read_group_env();
  if (variant < 0) variant = g_default_variant;
  if (variant < 0) {
    const long t256 = (long)((g.M + G2_BM - 1) / G2_BM) * ((g.N + G2_BN - 1) / G2_BN);
    variant = (t256 >= 192) ? 3 : 1;
  }
#ifdef RV_GEMM_EXPERIMENTS
  // Experiment builds only (rlaif-v_amd/build.py --experiments -> librlaifv_hip_exp.so): the ablations return WRONG
  // results by construction and the schedule variants lost their A/B runs; none of them is in the shipped library.
  if (variant == 101) return launch_gemm256<Epi, true, 1>(g, epi, (hipStream_t)stream);   // ablation: no DMA
  if (variant == 102) return launch_gemm256<Epi, true, 2>(g, epi, (hipStream_t)stream);   // ablation: stale ds_reads
  if (variant == 103) return launch_gemm256<Epi, true, 3>(g, epi, (hipStream_t)stream);   // ablation: L2-resident DMA
  if (variant == 6) return launch_gemm256<Epi, true, 0, 3, 1>(g, epi, (hipStream_t)stream);
  if (variant == 5) return launch_gemm256x64<Epi>(g, epi, (hipStream_t)stream);
  if (variant == 4) return launch_gemm256<Epi, true, 0, 4>(g, epi, (hipStream_t)stream);
#else
  if (variant > 3) { rv_set_error("GEMM variant: 0..3 (experiment variants need an RV_GEMM_EXPERIMENTS build)"); return 1; }
#endif
  if (variant == 3) return launch_gemm256<Epi, true>(g, epi, (hipStream_t)stream);
  if (variant == 2) return launch_gemm256<Epi, false>(g, epi, (hipStream_t)stream);
  if (variant == 1) return launch_gemm<1, Epi>(g, epi, (hipStream_t)stream);
  return launch_gemm<0, Epi>(g, epi, (hipStream_t)stream);
}

static int check_shape(const GemmShape& g, const char* who) {
  if (g.K % GEMM_BK != 0 || g.K <= 0) { rv_set_error("GEMM: K must be a positive multiple of 64"); return 1; }
  if (g.N % 4 != 0) { rv_set_error("GEMM: N must be a multiple of 4"); return 1; }
  if (g.lda % 8 != 0 || g.ldb % 8 != 0) { rv_set_error("GEMM: lda/ldb must be multiples of 8 elements"); return 1; }
  if (((uintptr_t)g.A | (uintptr_t)g.B) & 15) { rv_set_error("GEMM: A/B must be 16-byte aligned"); return 1; }
  (void)who;
  return 0;
}

#ifdef RV_GEMM_H128
// experiment builds only: RV_H128=1 routes the NN GEMMs (plain and SwiGLU epilogues) to the two-workgroups-per-CU 128 x 256 kernel
static unsigned* g_h128_dbg = nullptr;
extern "C" void rv_debug_h128_dbg(unsigned* p) { g_h128_dbg = p; }
template <class Epi>
static bool launch_nn_h128(const GemmShape& g, const Epi& epi, hipStream_t st) {
  static int on = -1, stagger_per_phase = 0;
  if (on < 0) {
    const char* e = getenv("RV_H128"); on = e ? atoi(e) : 0;
    const char* s2 = getenv("RV_H128_STAGGER"); stagger_per_phase = s2 ? atoi(s2) : 0;     // s_memtime ticks per 32-deep phase
  }
  if (!on || g.K % 64 != 0 || g.K < 128) return false;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_nn_h128_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, H128_LDS_BYTES);
    attr_done = true;
  }
  const int tiles = ((g.M + 127) / 128) * ((g.N + G2_BN - 1) / G2_BN);
  hipLaunchKernelGGL((gemm_nn_h128_kernel<Epi>), dim3(tiles), dim3(256), H128_LDS_BYTES, st, g, epi,
                     stagger_per_phase * (g.K / G2_BK), g_h128_dbg);
  return true;
}
#endif

// EXPERIMENT switch (RV_GU_TILE_MAJOR=1, read once): the kept gate|up tensor of the fused SwiGLU epilogues in tile-major layout
// (EpiSwiGLU::gu_index).  The caller must hand a buffer of ceil(M / 256) * 256 rows (rlaif-v_amd/ops.py does when the variable is
// set); every consumer of that tensor other than rv_gemm_nn_swiglu_bwd_bf16 would read garbage - not for production use.
static int gu_tile_major() {
#ifdef RV_GEMM_EXPERIMENTS          // experiment library only (python rlaif-v_amd/build.py --experiments): measured, no gain - see
  static int v = -1;               // profiles/r06_swiglu_bwd_tile_major_negative_result.log
  if (v < 0) { const char* e = getenv("RV_GU_TILE_MAJOR"); v = (e && atoi(e) == 1) ? 1 : 0; }
  return v;
#else
  return 0;
#endif
}

// NN GEMM with one of the SwiGLU epilogues: the 64-deep-A kernel when K allows, else the 32-deep one
template <class Epi>
static int launch_nn_epi(const GemmShape& g, const Epi& epi, hipStream_t st) {
#ifdef RV_GEMM_H128
  if (launch_nn_h128(g, epi, st)) { RV_CHECK_LAUNCH(); return 0; }
#endif
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_nn_256_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<Epi, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    attr_done = true;
  }
  const int tiles_m = (g.M + G2_BM - 1) / G2_BM, tiles_n = (g.N + G2_BN - 1) / G2_BN;
  if (g.K % 64 == 0 && g.K >= 512 && nn_mi16())
    hipLaunchKernelGGL((gemm_nn_a64_kernel<Epi, false, true>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G4_LDS_BYTES, st, g, epi);
  else if (g.K % 64 == 0 && g.K >= 512)
    hipLaunchKernelGGL((gemm_nn_a64_kernel<Epi>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G4_LDS_BYTES, st, g, epi);
  else
    hipLaunchKernelGGL((gemm_nn_256_kernel<Epi>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G2_LDS_BYTES, st, g, epi);
  RV_CHECK_LAUNCH();
  return 0;
}

extern "C" {

const char* rv_last_error(void) { return g_err; }

int rv_set_gemm_variant(int variant) {
  RV_REQUIRE(variant >= -1 && variant <= 3, "rv_set_gemm_variant: -1 (auto), 0..3");
  g_default_variant = variant;
  return 0;
}

int rv_abi_version(void) { return RV_ABI_VERSION; }

int rv_set_gemm_mi16(int on) {
  g_mi16 = on ? 1 : 0;
  return 0;
}

int rv_gemm_nt_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                    const void* bias, const void* residual, long ldr, int act, float alpha, int variant,
                    void* stream) {
  if (M == 0 || N == 0) return 0;
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group};
  if (check_shape(g, "rv_gemm_nt_bf16")) return 1;
  RV_REQUIRE(ldc % 4 == 0 && (residual == nullptr || ldr % 4 == 0), "rv_gemm_nt_bf16: ldc/ldr must be multiples of 4");
  // many rows, at most 256 columns, plain store: the streaming kernel (the rank-r side of LoRA; RV_GEMM_NT_SKINNY=0: A/B knob)
  static int skinny = -1;
  if (skinny < 0) { const char* e = getenv("RV_GEMM_NT_SKINNY"); skinny = e ? atoi(e) : 1; }
  if (skinny && variant < 0 && g_default_variant < 0 && N % GSK_BN == 0 && N <= 256 && M >= 4096 && bias == nullptr &&
      residual == nullptr && act == RV_ACT_NONE && ((uintptr_t)C & 7) == 0) {
    static bool attr_done = false;
    if (!attr_done) {
      hipFuncSetAttribute((const void*)gemm_nt_skinny_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GSK_LDS_BYTES);
      attr_done = true;
    }
    hipLaunchKernelGGL(gemm_nt_skinny_kernel, dim3(((M + GSK_BM - 1) / GSK_BM) * (N / GSK_BN)), dim3(256), GSK_LDS_BYTES,
                       (hipStream_t)stream, g, (bf16_t*)C, ldc, alpha);
    RV_CHECK_LAUNCH();
    return 0;
  }
  EpiStore epi{(bf16_t*)C, ldc, (const bf16_t*)bias, (const bf16_t*)residual, ldr, act, alpha};
  epi.narrow = epi_narrow();
  return dispatch(g, epi, variant, stream);
}

int rv_gemm_tn_bf16(const void* P, long ldp, const void* Q, long ldq, void* C, long ldc, int R, int I, int J,
                    const void* residual, long ldr, float alpha, void* stream) {
  if (I == 0 || J == 0) return 0;
  RV_REQUIRE(R > 0, "rv_gemm_tn_bf16: R must be > 0");
  RV_REQUIRE(I % 8 == 0 && J % 8 == 0 && I >= 8 && J >= 8, "rv_gemm_tn_bf16: I and J must be multiples of 8");
  RV_REQUIRE(ldp % 8 == 0 && ldq % 8 == 0 && ldc % 4 == 0 && (residual == nullptr || ldr % 4 == 0),
             "rv_gemm_tn_bf16: leading dimensions must be multiples of 8 (inputs) / 4 (output)");
  RV_REQUIRE((((uintptr_t)P | (uintptr_t)Q) & 15) == 0, "rv_gemm_tn_bf16: P/Q must be 16-byte aligned");
  EpiStore epi{(bf16_t*)C, ldc, nullptr, (const bf16_t*)residual, ldr, RV_ACT_NONE, alpha};
  epi.narrow = epi_narrow();
  return launch_gemm_tn((const bf16_t*)P, ldp, (const bf16_t*)Q, ldq, R, I, J, epi, (hipStream_t)stream);
}

int rv_gemm_tn_workspace_floats(int R, int I, int J) {
  if (R <= 0 || I <= 0 || J <= 0) return 0;
  const TnTailPlan p = tn_tail_plan(R, I, J);
  return p.splits > 1 ? p.splits * p.tail * 65536 : 0;
}

int rv_gemm_tn_bf16_ws(const void* P, long ldp, const void* Q, long ldq, void* C, long ldc, int R, int I, int J,
                       float* workspace, long workspace_floats, void* stream) {
  if (I == 0 || J == 0) return 0;
  RV_REQUIRE(R > 0, "rv_gemm_tn_bf16_ws: R must be > 0");
  RV_REQUIRE(I % 8 == 0 && J % 8 == 0 && I >= 8 && J >= 8, "rv_gemm_tn_bf16_ws: I and J must be multiples of 8");
  RV_REQUIRE(ldp % 8 == 0 && ldq % 8 == 0 && ldc % 4 == 0, "rv_gemm_tn_bf16_ws: leading dimensions must be multiples of 8 (inputs) / 4 (output)");
  RV_REQUIRE((((uintptr_t)P | (uintptr_t)Q | (uintptr_t)workspace) & 15) == 0, "rv_gemm_tn_bf16_ws: P/Q/workspace must be 16-byte aligned");
  RV_REQUIRE(((uintptr_t)C & 7) == 0, "rv_gemm_tn_bf16_ws: C must be 8-byte aligned (the tail reduce stores 4 bf16 at a time)");
  const TnTailPlan p = tn_tail_plan(R, I, J);
  EpiStore epi{(bf16_t*)C, ldc, nullptr, nullptr, 0, RV_ACT_NONE, 1.0f};
  epi.narrow = epi_narrow();
  if (p.splits <= 1 || workspace == nullptr || workspace_floats < (long)p.splits * p.tail * 65536)
    return launch_gemm_tn((const bf16_t*)P, ldp, (const bf16_t*)Q, ldq, R, I, J, epi, (hipStream_t)stream);
  if (launch_gemm_tn((const bf16_t*)P, ldp, (const bf16_t*)Q, ldq, R, I, J, epi, (hipStream_t)stream, 1, 0, 0, 0, p.full_blocks))
    return 1;
  EpiStoreF32 epf{workspace, 256};
  if (launch_gemm_tn((const bf16_t*)P, ldp, (const bf16_t*)Q, ldq, R, I, J, epf, (hipStream_t)stream, p.splits, p.r_chunk, -1,
                     p.full_blocks, p.tail))
    return 1;
  read_group_env();
  hipLaunchKernelGGL(tn_tail_reduce_kernel, dim3(p.tail), dim3(256), 0, (hipStream_t)stream, workspace, p.splits, p.tail,
                     p.full_blocks, I, J, g_group, (bf16_t*)C, ldc);
  RV_CHECK_LAUNCH();
  return 0;
}

int rv_gemm_nt_lora_bf16(const void* A, long lda, const void* B, long ldb, const void* A2, long lda2, const void* B2,
                         long ldb2, int K2, int group_cols, int group0, void* C, long ldc, int M, int N, int K,
                         const void* residual, long ldr, void* stream) {
  if (M == 0 || N == 0) return 0;
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group,
              (const bf16_t*)A2, (const bf16_t*)B2, lda2, ldb2, K2, group_cols, group0};
  if (check_shape(g, "rv_gemm_nt_lora_bf16")) return 1;
  RV_REQUIRE(K2 > 0 && K2 % GEMM_BK == 0, "rv_gemm_nt_lora_bf16: K2 must be a positive multiple of 64");
  RV_REQUIRE(lda2 % 8 == 0 && ldb2 % 8 == 0 && ((((uintptr_t)A2 | (uintptr_t)B2) & 15) == 0),
             "rv_gemm_nt_lora_bf16: A2/B2 must be 16-byte aligned with leading dimensions multiples of 8");
  RV_REQUIRE(group_cols == 0 || (group_cols % GEMM_BN == 0 && group0 % GEMM_BN == 0 && group0 >= 0 && group0 <= N &&
                                 (N - (group0 ? group0 : group_cols)) % group_cols == 0),
             "rv_gemm_nt_lora_bf16: group_cols / group0 must be multiples of 128 that tile N (group0 + k * group_cols = N)");
  RV_REQUIRE(ldc % 4 == 0 && (residual == nullptr || ldr % 4 == 0), "rv_gemm_nt_lora_bf16: ldc/ldr must be multiples of 4");
  EpiStore epi{(bf16_t*)C, ldc, nullptr, (const bf16_t*)residual, ldr, RV_ACT_NONE, 1.0f};
  epi.narrow = epi_narrow();
  return dispatch_ext(g, epi, stream);
}

static int nn_bias_act(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, const void* bias,
                       const void* residual, long ldr, int act, float alpha, void* stream);

int rv_gemm_nn_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                    const void* residual, long ldr, float alpha, void* stream) {
  return nn_bias_act(A, lda, B, ldb, C, ldc, M, N, K, nullptr, residual, ldr, RV_ACT_NONE, alpha, stream);
}

int rv_gemm_nn_bias_act_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                             const void* bias, const void* residual, long ldr, int act, float alpha, void* stream) {
  RV_REQUIRE(act == RV_ACT_NONE || act == RV_ACT_QUICK_GELU || act == RV_ACT_GELU, "rv_gemm_nn_bias_act_bf16: unknown activation");
  RV_REQUIRE(bias == nullptr || ((uintptr_t)bias & 15) == 0, "rv_gemm_nn_bias_act_bf16: bias must be 16-byte aligned");
  return nn_bias_act(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, act, alpha, stream);
}

static int nn_bias_act(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, const void* bias,
                       const void* residual, long ldr, int act, float alpha, void* stream) {
  if (M == 0 || N == 0) return 0;
  RV_REQUIRE(K > 0 && K % G2_BK == 0, "rv_gemm_nn_bf16: K must be a positive multiple of 32");
  RV_REQUIRE(N % 8 == 0 && N >= 8, "rv_gemm_nn_bf16: N must be a multiple of 8");
  RV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && (residual == nullptr || ldr % 4 == 0),
             "rv_gemm_nn_bf16: leading dimensions must be multiples of 8 (inputs) / 4 (output)");
  RV_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0, "rv_gemm_nn_bf16: A/B must be 16-byte aligned");
  read_group_env();
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group};
  EpiStore epi{(bf16_t*)C, ldc, (const bf16_t*)bias, (const bf16_t*)residual, ldr, act, alpha};
  epi.narrow = epi_narrow();
  static bool attr_done = false;
  static int use_a64 = 1;          // RV_GEMM_NN_A64=0: 32-deep A tiles (gemm_nn_256_kernel) also when K % 64 == 0
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_nn_256_kernel<EpiStore>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiStore>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiStore, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    const char* e = getenv("RV_GEMM_NN_A64");
    if (e) use_a64 = atoi(e);
    attr_done = true;
  }
  const int tiles_m = (M + G2_BM - 1) / G2_BM, tiles_n = (N + G2_BN - 1) / G2_BN;
#ifdef RV_GEMM_H128
  if (launch_nn_h128(g, epi, (hipStream_t)stream)) { RV_CHECK_LAUNCH(); return 0; }
#endif
  if (use_a64 && K % 64 == 0 && K >= 512 && nn_mi16())
    hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiStore, false, true>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G4_LDS_BYTES,
                       (hipStream_t)stream, g, epi);
  else if (use_a64 && K % 64 == 0 && K >= 512)
    hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiStore>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G4_LDS_BYTES,
                       (hipStream_t)stream, g, epi);
  else
    hipLaunchKernelGGL((gemm_nn_256_kernel<EpiStore>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G2_LDS_BYTES,
                       (hipStream_t)stream, g, epi);
  RV_CHECK_LAUNCH();
  return 0;
}

int rv_gemm_nn_rope_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                         const float* cos_tab, const float* sin_tab, const int* pos, int L, int rope_cols, int hd, void* stream) {
  if (M == 0 || N == 0) return 0;
  RV_REQUIRE(hd == 128, "rv_gemm_nn_rope_bf16: head dim 128 only");
  RV_REQUIRE(K >= 512 && K % 64 == 0, "rv_gemm_nn_rope_bf16: K must be a multiple of 64 and >= 512 (64-deep-A kernel)");
  RV_REQUIRE(N % 256 == 0 && rope_cols % 256 == 0 && rope_cols >= 0 && rope_cols <= N,
             "rv_gemm_nn_rope_bf16: N and rope_cols must be multiples of 256 (a 256-column tile holds q / k heads or v heads)");
  RV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "rv_gemm_nn_rope_bf16: leading dimensions must be multiples of 8");
  RV_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)cos_tab | (uintptr_t)sin_tab) & 15) == 0,
             "rv_gemm_nn_rope_bf16: operands and tables must be 16-byte aligned");
  RV_REQUIRE(cos_tab && sin_tab && L > 0, "rv_gemm_nn_rope_bf16: tables / L");
  read_group_env();
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group};
  EpiStoreRope epi{(bf16_t*)C, ldc, cos_tab, sin_tab, pos, L, rope_cols};
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiStoreRope>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiStoreRope, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    attr_done = true;
  }
  const int tiles_m = (M + G2_BM - 1) / G2_BM, tiles_n = (N + G2_BN - 1) / G2_BN;
  if (nn_mi16())
    hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiStoreRope, false, true>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G4_LDS_BYTES,
                       (hipStream_t)stream, g, epi);
  else
    hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiStoreRope>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G4_LDS_BYTES,
                       (hipStream_t)stream, g, epi);
  RV_CHECK_LAUNCH();
  return 0;
}

int rv_gemm_nn_swiglu_bf16(const void* A, long lda, const void* B, long ldb, void* GU, long ldgu, void* ACT, long ldact,
                           int M, int N, int K, void* stream) {
  if (M == 0 || N == 0) return 0;
  RV_REQUIRE(K > 0 && K % G2_BK == 0, "rv_gemm_nn_swiglu_bf16: K must be a positive multiple of 32");
  RV_REQUIRE(N % 16 == 0, "rv_gemm_nn_swiglu_bf16: N (= 2 x ffn, interleaved gate/up columns) must be a multiple of 16");
  RV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldgu % 8 == 0 && ldact % 4 == 0, "rv_gemm_nn_swiglu_bf16: bad leading dimension");
  RV_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)GU) & 15) == 0 && ((uintptr_t)ACT & 7) == 0,
             "rv_gemm_nn_swiglu_bf16: operands must be 16-byte aligned (activation: 8)");
  read_group_env();
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group};
  EpiSwiGLU epi{(bf16_t*)GU, ldgu, (bf16_t*)ACT, ldact};
  if (gu_tile_major()) { epi.tile_major = 1; epi.tiles_n = (N + 255) / 256; }
  return launch_nn_epi(g, epi, (hipStream_t)stream);
}

int rv_gemm_nn_swiglu_bwd_bf16(const void* A, long lda, const void* B, long ldb, const void* GU, long ldgu, void* DGU,
                               long lddgu, int M, int N, int K, void* stream) {
  if (M == 0 || N == 0) return 0;
  RV_REQUIRE(K > 0 && K % G2_BK == 0, "rv_gemm_nn_swiglu_bwd_bf16: K must be a positive multiple of 32");
  RV_REQUIRE(N % 8 == 0, "rv_gemm_nn_swiglu_bwd_bf16: N (= ffn) must be a multiple of 8");
  RV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldgu % 8 == 0 && lddgu % 8 == 0, "rv_gemm_nn_swiglu_bwd_bf16: bad leading dimension");
  RV_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)GU | (uintptr_t)DGU) & 15) == 0,
             "rv_gemm_nn_swiglu_bwd_bf16: operands must be 16-byte aligned");
  read_group_env();
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group};
  EpiSwiGLUBwd epi{(const bf16_t*)GU, ldgu, (bf16_t*)DGU, lddgu};
  if (gu_tile_major()) { epi.tile_major = 1; epi.tiles_n = (2 * N + 255) / 256; }
  return launch_nn_epi(g, epi, (hipStream_t)stream);
}

int rv_gemm_nn_lora_bf16(const void* A, long lda, const void* B, long ldb, const void* A2, long lda2, const void* B2,
                         long ldb2, int K2, int group_cols, int group0, void* C, long ldc, int M, int N, int K,
                         const void* residual, long ldr, void* stream) {
  if (M == 0 || N == 0) return 0;
  RV_REQUIRE(K > 0 && K % G2_BK == 0 && K2 > 0 && K2 % G2_BK == 0, "rv_gemm_nn_lora_bf16: K and K2 must be positive multiples of 32");
  RV_REQUIRE(N % 8 == 0 && N >= 8, "rv_gemm_nn_lora_bf16: N must be a multiple of 8");
  RV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && ldc % 4 == 0 && (residual == nullptr || ldr % 4 == 0),
             "rv_gemm_nn_lora_bf16: leading dimensions must be multiples of 8 (inputs) / 4 (output)");
  RV_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)A2 | (uintptr_t)B2) & 15) == 0,
             "rv_gemm_nn_lora_bf16: operands must be 16-byte aligned");
  RV_REQUIRE(group_cols == 0 || (group_cols % G2_BN == 0 && group0 % G2_BN == 0 && group0 >= 0 && group0 <= N &&
                                 (N - (group0 ? group0 : group_cols)) % group_cols == 0),
             "rv_gemm_nn_lora_bf16: group_cols / group0 must be multiples of 256 that tile N (group0 + k * group_cols = N)");
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group,
              (const bf16_t*)A2, (const bf16_t*)B2, lda2, ldb2, K2, group_cols, group0};
  EpiStore epi{(bf16_t*)C, ldc, nullptr, (const bf16_t*)residual, ldr, RV_ACT_NONE, 1.0f};
  epi.narrow = epi_narrow();
  static bool attr_done = false;
  static int use_a64 = 1;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_nn_256_kernel<EpiStore, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiStore, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiStore, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    const char* e = getenv("RV_GEMM_NN_A64");
    if (e) use_a64 = atoi(e);
    attr_done = true;
  }
  const int tiles_m = (M + G2_BM - 1) / G2_BM, tiles_n = (N + G2_BN - 1) / G2_BN;
  if (use_a64 && K % 64 == 0 && K2 % 64 == 0 && K >= 512 && nn_mi16())
    hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiStore, true, true>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G4_LDS_BYTES,
                       (hipStream_t)stream, g, epi);
  else if (use_a64 && K % 64 == 0 && K2 % 64 == 0 && K >= 512)
    hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiStore, true>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G4_LDS_BYTES,
                       (hipStream_t)stream, g, epi);
  else
    hipLaunchKernelGGL((gemm_nn_256_kernel<EpiStore, true>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G2_LDS_BYTES,
                       (hipStream_t)stream, g, epi);
  RV_CHECK_LAUNCH();
  return 0;
}

// gate|up projection of an ADAPTER model with SwiGLU in the epilogue (round 6): the fused-LoRA K loop
// y = [x | t] [W | B2]^T of rv_gemm_nn_lora_bf16 with the epilogue of rv_gemm_nn_swiglu_bf16.  Interleaved gate / up columns; B2 is the
// EXPANDED adapter matrix [K2 = 2 r][N]: row k < r carries lora_B(gate) on the even columns and zeros on the odd ones, row r + k
// lora_B(up) on the odd columns - peft's adapters are per module, the zeros keep t_gate out of the up columns and vice versa.
int rv_gemm_nn_lora_swiglu_bf16(const void* A, long lda, const void* B, long ldb, const void* A2, long lda2, const void* B2,
                                long ldb2, int K2, void* GU, long ldgu, void* ACT, long ldact, void* ACTD, float p, int seed,
                                int M, int N, int K, void* stream) {
  if (M == 0 || N == 0) return 0;
  RV_REQUIRE(K >= 512 && K % 64 == 0 && K2 > 0 && K2 % 64 == 0, "rv_gemm_nn_lora_swiglu_bf16: K % 64, K >= 512, K2 % 64");
  RV_REQUIRE(N % 16 == 0, "rv_gemm_nn_lora_swiglu_bf16: N (= 2 x ffn, interleaved gate/up columns) must be a multiple of 16");
  RV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && ldgu % 8 == 0 && ldact % 4 == 0,
             "rv_gemm_nn_lora_swiglu_bf16: bad leading dimension");
  RV_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)A2 | (uintptr_t)B2 | (uintptr_t)GU) & 15) == 0 && ((uintptr_t)ACT & 7) == 0 &&
             ((uintptr_t)ACTD & 7) == 0, "rv_gemm_nn_lora_swiglu_bf16: operands must be 16-byte aligned (activations: 8)");
  RV_REQUIRE(p >= 0.f && p < 1.f && (ACTD == nullptr || ldact == N / 2), "rv_gemm_nn_lora_swiglu_bf16: 0 <= p < 1; ACTD needs a contiguous ACT");
  read_group_env();
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group, (const bf16_t*)A2, (const bf16_t*)B2, lda2, ldb2, K2, 0, 0};
  EpiSwiGLU epi{(bf16_t*)GU, ldgu, (bf16_t*)ACT, ldact};
  if (ACTD != nullptr && p > 0.f) {
    epi.ACTD = (bf16_t*)ACTD;
    epi.drop_thresh16 = (uint32_t)((double)p * 65536.0 + 0.5);
    epi.drop_key = (uint32_t)seed * 0x9e3779b9u + 0x85ebca6bu;
    epi.drop_inv_keep = 1.f / (1.f - p);
  }
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiSwiGLU, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiSwiGLU, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    attr_done = true;
  }
  const int tiles_m = (M + G2_BM - 1) / G2_BM, tiles_n = (N + G2_BN - 1) / G2_BN;
  if (nn_mi16())
    hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiSwiGLU, true, true>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G4_LDS_BYTES,
                       (hipStream_t)stream, g, epi);
  else
    hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiSwiGLU, true>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G4_LDS_BYTES,
                       (hipStream_t)stream, g, epi);
  RV_CHECK_LAUNCH();
  return 0;
}

// Input gradient of an ADAPTER model's down projection with the SwiGLU backward in the epilogue: acc = d act = dy W + [mask](dt A) / (1 - p)
// (p > 0: the adapter-first form of rv_gemm_nn_lora_pre_bf16, mask of rv_dropout for the [M][N] activation; p = 0: the in-ring form
// of rv_gemm_nn_lora_bf16), then d(gate|up) from the kept interleaved gate|up tile as rv_gemm_nn_swiglu_bwd_bf16 does.
int rv_gemm_nn_lora_swiglu_bwd_bf16(const void* A, long lda, const void* B, long ldb, const void* A2, long lda2, const void* B2,
                                    long ldb2, int K2, float p, int seed, const void* GU, long ldgu, void* DGU, long lddgu,
                                    int M, int N, int K, void* stream) {
  if (M == 0 || N == 0) return 0;
  RV_REQUIRE(K >= 512 && K % 64 == 0 && K2 > 0 && K2 % 64 == 0, "rv_gemm_nn_lora_swiglu_bwd_bf16: K % 64, K >= 512, K2 % 64");
  RV_REQUIRE(N % 8 == 0, "rv_gemm_nn_lora_swiglu_bwd_bf16: N (= ffn) must be a multiple of 8");
  RV_REQUIRE(p >= 0.f && p < 1.f, "rv_gemm_nn_lora_swiglu_bwd_bf16: 0 <= p < 1");
  RV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && ldgu % 8 == 0 && lddgu % 8 == 0,
             "rv_gemm_nn_lora_swiglu_bwd_bf16: bad leading dimension");
  RV_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)A2 | (uintptr_t)B2 | (uintptr_t)GU | (uintptr_t)DGU) & 15) == 0,
             "rv_gemm_nn_lora_swiglu_bwd_bf16: operands must be 16-byte aligned");
  read_group_env();
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group, (const bf16_t*)A2, (const bf16_t*)B2, lda2, ldb2, K2, 0, 0};
  EpiSwiGLUBwd epi{(const bf16_t*)GU, ldgu, (bf16_t*)DGU, lddgu};
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiSwiGLUBwd, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiSwiGLUBwd, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiSwiGLUBwd, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiSwiGLUBwd, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    attr_done = true;
  }
  const int tiles_m = (M + G2_BM - 1) / G2_BM, tiles_n = (N + G2_BN - 1) / G2_BN;
  const dim3 grid(tiles_m * tiles_n), block(G2_THREADS);
  if (p > 0.f) {
    g.pre_thresh16 = (uint32_t)((double)p * 65536.0 + 0.5);
    g.pre_key = (uint32_t)seed * 0x9e3779b9u + 0x85ebca6bu;
    g.pre_inv_keep = 1.f / (1.f - p);
    if (nn_mi16()) hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiSwiGLUBwd, false, true, true>), grid, block, G4_LDS_BYTES, (hipStream_t)stream, g, epi);
    else hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiSwiGLUBwd, false, false, true>), grid, block, G4_LDS_BYTES, (hipStream_t)stream, g, epi);
  } else {
    if (nn_mi16()) hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiSwiGLUBwd, true, true>), grid, block, G4_LDS_BYTES, (hipStream_t)stream, g, epi);
    else hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiSwiGLUBwd, true>), grid, block, G4_LDS_BYTES, (hipStream_t)stream, g, epi);
  }
  RV_CHECK_LAUNCH();
  return 0;
}

int rv_gemm_nn_lora_pre_bf16(const void* A, long lda, const void* B, long ldb, const void* A2, long lda2, const void* B2,
                             long ldb2, int K2, int group_cols, int group0, float p, int seed, void* C, long ldc, int M, int N,
                             int K, const void* residual, long ldr, void* stream) {
  if (M == 0 || N == 0) return 0;
  RV_REQUIRE(K >= 512 && K % 64 == 0 && K2 > 0 && K2 % 64 == 0,
             "rv_gemm_nn_lora_pre_bf16: K must be a multiple of 64 and >= 512, K2 a positive multiple of 64");
  RV_REQUIRE(N % 8 == 0 && N >= 8, "rv_gemm_nn_lora_pre_bf16: N must be a multiple of 8 (mask layout of rv_dropout)");
  RV_REQUIRE(p >= 0.f && p < 1.f, "rv_gemm_nn_lora_pre_bf16: 0 <= p < 1");
  RV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && ldc % 4 == 0 && (residual == nullptr || ldr % 4 == 0),
             "rv_gemm_nn_lora_pre_bf16: leading dimensions must be multiples of 8 (inputs) / 4 (output)");
  RV_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)A2 | (uintptr_t)B2) & 15) == 0,
             "rv_gemm_nn_lora_pre_bf16: operands must be 16-byte aligned");
  RV_REQUIRE(group_cols == 0 || (group_cols % G2_BN == 0 && group0 % G2_BN == 0 && group0 >= 0 && group0 <= N &&
                                 (N - (group0 ? group0 : group_cols)) % group_cols == 0),
             "rv_gemm_nn_lora_pre_bf16: group_cols / group0 must be multiples of 256 that tile N (group0 + k * group_cols = N)");
  read_group_env();
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group,
              (const bf16_t*)A2, (const bf16_t*)B2, lda2, ldb2, K2, group_cols, group0};
  g.pre_thresh16 = (uint32_t)((double)p * 65536.0 + 0.5);
  g.pre_key = (uint32_t)seed * 0x9e3779b9u + 0x85ebca6bu;
  g.pre_inv_keep = 1.f / (1.f - p);
  EpiStore epi{(bf16_t*)C, ldc, nullptr, (const bf16_t*)residual, ldr, RV_ACT_NONE, 1.0f};
  epi.narrow = epi_narrow();
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiStore, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_nn_a64_kernel<EpiStore, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
    attr_done = true;
  }
  const int tiles_m = (M + G2_BM - 1) / G2_BM, tiles_n = (N + G2_BN - 1) / G2_BN;
  if (nn_mi16())
    hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiStore, false, true, true>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G4_LDS_BYTES,
                       (hipStream_t)stream, g, epi);
  else
    hipLaunchKernelGGL((gemm_nn_a64_kernel<EpiStore, false, false, true>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G4_LDS_BYTES,
                       (hipStream_t)stream, g, epi);
  RV_CHECK_LAUNCH();
  return 0;
}

int rv_gemm_nt_dropout_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                            const void* residual, long ldr, float alpha, float p, int seed, void* stream) {
  if (M == 0 || N == 0) return 0;
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group};
  if (check_shape(g, "rv_gemm_nt_dropout_bf16")) return 1;
  RV_REQUIRE(p >= 0.f && p < 1.f, "rv_gemm_nt_dropout_bf16: 0 <= p < 1");
  RV_REQUIRE(N % 8 == 0, "rv_gemm_nt_dropout_bf16: N must be a multiple of 8 (mask layout of rv_dropout)");
  RV_REQUIRE(ldc % 4 == 0 && (residual == nullptr || ldr % 4 == 0), "rv_gemm_nt_dropout_bf16: ldc/ldr must be multiples of 4");
  EpiStore epi{(bf16_t*)C, ldc, nullptr, (const bf16_t*)residual, ldr, RV_ACT_NONE, alpha};
  epi.narrow = epi_narrow();
  epi.drop_thresh16 = (uint32_t)((double)p * 65536.0 + 0.5);
  epi.drop_key = (uint32_t)seed * 0x9e3779b9u + 0x85ebca6bu;
  epi.drop_inv_keep = 1.f / (1.f - p);
  return dispatch(g, epi, -1, stream);
}

int rv_gemm_tn_bf16_splitk(const void* P, long ldp, const void* Q, long ldq, void* C, long ldc, int R, int I, int J,
                           float alpha, int splits, float* workspace, void* stream) {
  if (I == 0 || J == 0) return 0;
  RV_REQUIRE(R > 0 && splits >= 1 && splits <= 1024, "rv_gemm_tn_bf16_splitk: R > 0, 1 <= splits <= 1024");
  RV_REQUIRE(I % 8 == 0 && J % 8 == 0, "rv_gemm_tn_bf16_splitk: I and J must be multiples of 8");
  RV_REQUIRE(ldp % 8 == 0 && ldq % 8 == 0 && ldc % 4 == 0, "rv_gemm_tn_bf16_splitk: bad leading dimension");
  RV_REQUIRE((((uintptr_t)P | (uintptr_t)Q | (uintptr_t)workspace) & 15) == 0,
             "rv_gemm_tn_bf16_splitk: P/Q/workspace must be 16-byte aligned");
  RV_REQUIRE(workspace != nullptr, "rv_gemm_tn_bf16_splitk: workspace of splits*I*J floats required");
  int r_chunk = ((R + splits - 1) / splits + 31) / 32 * 32;
  splits = (R + r_chunk - 1) / r_chunk;            // no empty chunk
  EpiStoreF32 epi{workspace, (long)J};
  if (launch_gemm_tn((const bf16_t*)P, ldp, (const bf16_t*)Q, ldq, R, I, J, epi, (hipStream_t)stream, splits, r_chunk,
                     (long)I * J))
    return 1;
  const long n4 = (long)I * J / 4;
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, workspace, splits, I, J, alpha,
                     (bf16_t*)C, ldc);
  RV_CHECK_LAUNCH();
  return 0;
}

int rv_gemm_nt_bf16_f32out(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                           int variant, void* stream) {
  if (M == 0 || N == 0) return 0;
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group};
  if (check_shape(g, "rv_gemm_nt_bf16_f32out")) return 1;
  RV_REQUIRE(ldc % 4 == 0, "rv_gemm_nt_bf16_f32out: ldc must be a multiple of 4");
  EpiStoreF32 epi{C, ldc};
  return dispatch(g, epi, variant, stream);
}

int rv_gemm_nt_bf16_f32res(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                           const void* bias, const float* residual, long ldr, int variant, void* stream) {
  if (M == 0 || N == 0) return 0;
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group};
  if (check_shape(g, "rv_gemm_nt_bf16_f32res")) return 1;
  RV_REQUIRE(ldc % 4 == 0 && ldr % 4 == 0, "rv_gemm_nt_bf16_f32res: ldc / ldr must be multiples of 4");
  EpiStoreF32 epi{C, ldc, (const bf16_t*)bias, residual, ldr};
  return dispatch(g, epi, variant, stream);
}

int rv_lmhead_logp_fwd(const void* h, long ldh, const void* W, long ldw, const int* tgt, int M, int V, int V_valid, int K,
                       float* pmax, float* psum, float* tgt_logit, int variant, void* stream) {
  if (M == 0) return 0;
  GemmShape g{(const bf16_t*)h, (const bf16_t*)W, M, V, K, ldh, ldw, g_group};
  if (check_shape(g, "rv_lmhead_logp_fwd")) return 1;
  RV_REQUIRE(V % 64 == 0, "rv_lmhead_logp_fwd: the STORED vocabulary (rows of W) must be a multiple of 64; pad with zero rows");
  RV_REQUIRE(V_valid > V - 64 && V_valid <= V, "rv_lmhead_logp_fwd: V - 64 < V_valid <= V");
  EpiLogpFwd epi{tgt, pmax, psum, tgt_logit, V_valid};
  return dispatch(g, epi, variant, stream);
}

int rv_lmhead_logp_bwd(const void* h, long ldh, const void* W, long ldw, const int* tgt, const float* lse,
                       const float* coef, void* dlogits, long ldd, int M, int V, int V_valid, int K, int variant,
                       void* stream) {
  if (M == 0) return 0;
  GemmShape g{(const bf16_t*)h, (const bf16_t*)W, M, V, K, ldh, ldw, g_group};
  if (check_shape(g, "rv_lmhead_logp_bwd")) return 1;
  RV_REQUIRE(ldd % 4 == 0, "rv_lmhead_logp_bwd: ldd must be a multiple of 4");
  RV_REQUIRE(V_valid > V - 64 && V_valid <= V, "rv_lmhead_logp_bwd: V - 64 < V_valid <= V");
  EpiLogpBwd epi{tgt, lse, coef, (bf16_t*)dlogits, ldd, V_valid};
  return dispatch(g, epi, variant, stream);
}

}  // extern "C"
