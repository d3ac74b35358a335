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

template <class Epi>
static int launch_gemm_tn(const bf16_t* P, long ldp, const bf16_t* Q, long ldq, int R, int I, int J, const Epi& epi,
                          hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_tn_256_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES);
    attr_done = true;
  }
  const int tiles_i = (I + 255) / 256, tiles_j = (J + 255) / 256;
  hipLaunchKernelGGL((gemm_tn_256_kernel<Epi>), dim3(tiles_i * tiles_j), dim3(G2_THREADS), G2_LDS_BYTES, st, P, ldp, Q,
                     ldq, R, I, J, epi);
  RV_CHECK_LAUNCH();
  return 0;
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

template <class Epi>
static int dispatch(const GemmShape& g, const Epi& epi, int variant, void* stream) {
  static bool env_done = false;
  if (!env_done) { const char* e = getenv("RV_GEMM_GROUP"); if (e) g_group = atoi(e); env_done = true; }
  if (variant < 0) variant = g_default_variant;
  if (variant < 0) {
    const long t256 = (long)((g.M + G2_BM - 1) / G2_BM) * ((g.N + G2_BN - 1) / G2_BN);
    variant = (t256 >= 192) ? 3 : 1;
  }
  if (variant == 101) return launch_gemm256<Epi, true, 1>(g, epi, (hipStream_t)stream);   // ablation: no DMA
  if (variant == 102) return launch_gemm256<Epi, true, 2>(g, epi, (hipStream_t)stream);   // ablation: stale ds_reads
  if (variant == 103) return launch_gemm256<Epi, true, 3>(g, epi, (hipStream_t)stream);   // ablation: L2-resident DMA
  if (variant == 6) return launch_gemm256<Epi, true, 0, 3, 1>(g, epi, (hipStream_t)stream);
  if (variant == 5) return launch_gemm256x64<Epi>(g, epi, (hipStream_t)stream);
  if (variant == 4) return launch_gemm256<Epi, true, 0, 4>(g, epi, (hipStream_t)stream);
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

extern "C" {

const char* rv_last_error(void) { return g_err; }

int rv_set_gemm_variant(int variant) {
  RV_REQUIRE(variant >= -1 && variant <= 6, "rv_set_gemm_variant: -1 (auto), 0..6");
  g_default_variant = variant;
  return 0;
}

int rv_abi_version(void) { return RV_ABI_VERSION; }

int rv_gemm_nt_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                    const void* bias, const void* residual, long ldr, int act, float alpha, int variant,
                    void* stream) {
  if (M == 0 || N == 0) return 0;
  GemmShape g{(const bf16_t*)A, (const bf16_t*)B, M, N, K, lda, ldb, g_group};
  if (check_shape(g, "rv_gemm_nt_bf16")) return 1;
  RV_REQUIRE(ldc % 4 == 0 && (residual == nullptr || ldr % 4 == 0), "rv_gemm_nt_bf16: ldc/ldr must be multiples of 4");
  EpiStore epi{(bf16_t*)C, ldc, (const bf16_t*)bias, (const bf16_t*)residual, ldr, act, alpha};
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
  return launch_gemm_tn((const bf16_t*)P, ldp, (const bf16_t*)Q, ldq, R, I, J, epi, (hipStream_t)stream);
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

int rv_lmhead_logp_fwd(const void* h, long ldh, const void* W, long ldw, const int* tgt, int M, int V, int K,
                       float* pmax, float* psum, float* tgt_logit, int variant, void* stream) {
  if (M == 0) return 0;
  GemmShape g{(const bf16_t*)h, (const bf16_t*)W, M, V, K, ldh, ldw, g_group};
  if (check_shape(g, "rv_lmhead_logp_fwd")) return 1;
  RV_REQUIRE(V % 64 == 0, "rv_lmhead_logp_fwd: vocabulary must be a multiple of 64");
  EpiLogpFwd epi{tgt, pmax, psum, tgt_logit};
  return dispatch(g, epi, variant, stream);
}

int rv_lmhead_logp_bwd(const void* h, long ldh, const void* W, long ldw, const int* tgt, const float* lse,
                       const float* coef, void* dlogits, long ldd, int M, int V, int K, int variant, void* stream) {
  if (M == 0) return 0;
  GemmShape g{(const bf16_t*)h, (const bf16_t*)W, M, V, K, ldh, ldw, g_group};
  if (check_shape(g, "rv_lmhead_logp_bwd")) return 1;
  RV_REQUIRE(ldd % 4 == 0, "rv_lmhead_logp_bwd: ldd must be a multiple of 4");
  EpiLogpBwd epi{tgt, lse, coef, (bf16_t*)dlogits, ldd};
  return dispatch(g, epi, variant, stream);
}

}  // extern "C"
