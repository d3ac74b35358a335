/* rlaifv_hip.h - C ABI of librlaifv_hip.so, the MI355X (gfx950) kernels behind the RLAIF-V LLaVA-1.5
 * DPO training step.
 *
 * Boundary: the reference has no FFI/plugin layer - its hot path is Python
 * calling HuggingFace/torch modules.  Each entry point below replaces the third-party / reference
 * arithmetic named in its comment (paths relative to the reference checkout).  The Python host
 * (rlaif_v_amd.hip) binds these with ctypes; INTEGRATION.md shows the stub a reference maintainer adds.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller and borrowed for the call;
 *    bf16 tensors are raw uint16 bit patterns (`void*`), statistics are fp32, indices are int32;
 *  - `ld*` are row strides in ELEMENTS; bf16 rows must be 16-byte aligned;
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises;
 *  - return 0 = ok, 1 = argument error, 2 = launch error; text via rv_last_error() (thread local);
 *  - no exceptions cross the ABI, no internal threads, no allocation.
 */
#ifndef RLAIFV_HIP_H
#define RLAIFV_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RV_ABI_VERSION 7

const char* rv_last_error(void);
int rv_abi_version(void);
/* ---- TEST-ONLY knobs (the four rv_set_* below).  They write process-global state without synchronisation: call them from ONE
 * thread, with no launch of this library in flight on another thread, and never from production code - the product path
 * (the rlaif-v_amd Python modules) does not call them; tests use them to run two kernel generations against each other inside one process.
 * Everything else in this library is stateless (per-call arguments only).
 * A/B and test knob: 1 (default, also RV_GEMM_MI16) = the NN-A64 / fused-LoRA NN / TN main loops issue 16x16x32 MFMAs
 * (more flops per joule under the package power cap, profiles/r02_mfma_shape_power_probe.log); 0 = the 32x32x16 loops.
 * Same results to fp32 accumulation order. */
int rv_set_gemm_mi16(int on);
/* dK/dV kernel of rv_attn_bwd: 5 = the round-4 kernel (default), 3 = the round-2/3 kernel, 0 = back to the default / RV_ATTN_DKV.
 * A/B and test knob (both kernels against each other in one process); replaces nothing in the reference. */
int rv_set_attn_dkv_version(int version);
/* forward kernel of rv_attn_fwd at head dim 128: 3 = the round-6 kernel (one wave per SIMD, two 32-query blocks per wave
 * software-pipelined across key tiles, lazy rescale; csrc/attn_fwd3.inc), 2 = the round-2..5 kernel, 0 = back to the default /
 * RV_ATTN_FWD.  A/B and test knob; replaces nothing in the reference. */
int rv_set_attn_fwd_version(int version);
/* GEMM kernel selection: -1 = auto (default), 0 = 128x128x64 register-staged, 1 = 128x128x64 global_load_lds,
 * 2 = 256x256x32 ping-pong (two wave groups alternating MFMA / load segments). */
int rv_set_gemm_variant(int variant);

/* ---- dense contractions (replace torch.nn.Linear inside HF LlamaForCausalLM / CLIPVisionModel /
 *      mm_projector: llava/model/language_model/llava_llama.py:91-102,
 *      llava/model/multimodal_encoder/clip_encoder.py:55, llava/model/multimodal_projector/builder.py:39-46)
 *   C[m][n] = act(alpha * sum_k A[m][k] * B[n][k] + bias[n]) + R[m][n]        (bf16 in, fp32 accumulate)
 *   act: 0 none, 1 quick_gelu (CLIP MLP), 2 gelu(erf) (projector).  K % 64 == 0, N % 4 == 0.
 *   variant: -1 = process default, else as rv_set_gemm_variant. */
int rv_gemm_nt_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                    const void* bias, const void* residual, long ldr, int act, float alpha, int variant,
                    void* stream);
/* C[m][n] = alpha * sum_k A[m][k] * B[k][n] (+ residual): NN form - B is [K][N] row-major (N contiguous).  The same
 * nn.Linear forward / input-gradient products as rv_gemm_nt_bf16, fed with the other orientation of the weight
 * (forward: B = W^T copy [in][out]; input gradient: B = W [out][in]) so that the weight tile is fetched in full
 * 512-byte row segments.  Large problems only (256x256 tiles); K % 32 == 0, N % 8 == 0. */
int rv_gemm_nn_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                    const void* residual, long ldr, float alpha, void* stream);
/* The same with the bias / activation epilogue of rv_gemm_nt_bf16: C = act(alpha * A B + bias[n]) + residual (ABI 7).  For frozen,
 * forward-only towers whose nn.Linear weights (with bias; fc1 + GELU) are kept ONLY in the [in][out] orientation: the EVA02 blocks
 * behind omnilmm/model/omnilmm.py:31-43 (timm Attention.qkv / proj, Mlp.fc1 / fc2). */
int rv_gemm_nn_bias_act_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                             const void* bias, const void* residual, long ldr, int act, float alpha, void* stream);

/* q|k|v projection with RoPE in the epilogue (ABI 7; HF apply_rotary_pos_emb on the outputs of q_proj / k_proj, reached through
 * llava/model/language_model/llava_llama.py:91-102): C = A B with B the W^T copy [K][N] of the fused q|k|v weight; the first
 * rope_cols columns (the q and k heads, head dim hd = 128) leave rotated - y1 = x1 cos - x2 sin, y2 = x2 cos + x1 sin for the
 * half-split pairs (i, i + 64), tables [position][64] fp32 as rv_rope_inplace takes them, position = pos[token] or token % L - from
 * the fp32 accumulators (one rounding; replaces rv_gemm_nn_bf16 + rv_rope_inplace).  N, rope_cols multiples of 256, K % 64 == 0. */
int rv_gemm_nn_rope_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                         const float* cos_tab, const float* sin_tab, const int* pos, int L, int rope_cols, int hd, void* stream);

/* The two fused SwiGLU GEMMs for ADAPTER models (ABI 7; peft lora.Linear.forward of gate_proj / up_proj + LlamaMLP's act_fn(gate) * up,
 * muffin/train/train_llava15_lora.py:304-318).  Forward: GU = [A | A2] [B | B2]^T with interleaved gate / up columns, ACT = silu(g) u,
 * optionally ACTD = rv_dropout(ACT; p, seed) (the dropped adapter input of the down projection).  B2 = the expanded adapter matrix
 * [K2 = 2 r][N]: rows 0..r-1 = lora_B(gate)^T on the even columns (zeros on the odd ones), rows r..2r-1 = lora_B(up)^T on the odd
 * columns; A2 = t = (alpha / r) dropout(x) [lora_A(gate); lora_A(up)]^T [M][2 r].
 * Backward: d(gate|up) = SwiGLU'(GU) o (A B + mask_{p,seed}(A2 B2) / (1 - p)) - the down projection's input gradient with the
 * adapter term (A = dy, B = W_down [d][f], A2 = dt [M][r], B2 = lora_A(down) [r][f]; p = 0: no mask). */
int rv_gemm_nn_lora_swiglu_bf16(const void* A, long lda, const void* B, long ldb, const void* A2, long lda2, const void* B2,
                                long ldb2, int K2, void* GU, long ldgu, void* ACT, long ldact, void* ACTD, float p, int seed,
                                int M, int N, int K, void* stream);
int rv_gemm_nn_lora_swiglu_bwd_bf16(const void* A, long lda, const void* B, long ldb, const void* A2, long lda2, const void* B2,
                                    long ldb2, int K2, float p, int seed, const void* GU, long ldgu, void* DGU, long lddgu,
                                    int M, int N, int K, void* stream);

/* gate|up projection with SwiGLU in the epilogue (HF LlamaMLP.forward: down_proj(act_fn(gate_proj(x)) * up_proj(x)),
 * reached through llava_llama.py:91-102).  B = the W^T copy [K][N] of the fused weight whose rows are INTERLEAVED
 * (row 2j = gate_j, row 2j+1 = up_j), N = 2 x ffn.  Writes GU [M][N] (kept for backward) and ACT [M][N/2] = silu(g) * u. */
int rv_gemm_nn_swiglu_bf16(const void* A, long lda, const void* B, long ldb, void* GU, long ldgu, void* ACT, long ldact,
                           int M, int N, int K, void* stream);
/* Input gradient of down_proj with the SwiGLU backward in the epilogue: d act = A [M][K] x B [K][N] (B = W_down, N = ffn)
 * never reaches memory; DGU [M][2N] = d(gate|up) interleaved like GU [M][2N]. */
int rv_gemm_nn_swiglu_bwd_bf16(const void* A, long lda, const void* B, long ldb, const void* GU, long ldgu, void* DGU,
                               long lddgu, int M, int N, int K, void* stream);

/* Fused LoRA GEMM (peft LoraLayer.forward as used by muffin/train/train_llava15_lora.py:304-318):
 *   C[m][n] = sum_{k<K} A[m][k] B[n][k] + sum_{q<K2} A2[m][c0(n)+q] B2[n][q] (+ residual[m][n]),
 *   c0(n) = group_cols ? group(n) * K2 : 0,  group(n) = n < group0 ? 0 : 1 + (n - group0) / group_cols  (group0 = 0: = group_cols;
 *   grouped-query attention: group0 = hidden (q), group_cols = kv_dim (k, v)).
 * The adapter contribution is two extra steps of the SAME K loop (operand tiles fetched from A2/B2), so
 * the base output is never re-read.  Forward: A = x, B = W, A2 = t = (alpha/r) x A_lora^T [M, G*r], B2 = stacked
 * lora_B [N, r], group_cols = rows of one fused projection (q|k|v, gate|up).  Input gradient: A = dy, B = W^T,
 * A2 = dt [M, G*r], B2 = stacked lora_A^T [N, G*r], group_cols = 0.  K2 multiple of 64; group_cols multiple of 128. */
int rv_gemm_nt_lora_bf16(const void* A, long lda, const void* B, long ldb, const void* A2, long lda2, const void* B2,
                         long ldb2, int K2, int group_cols, int group0, void* C, long ldc, int M, int N, int K,
                         const void* residual, long ldr, void* stream);

/* NN form of rv_gemm_nt_lora_bf16: B is [K][N], B2 is [K2][N] (both row-major); A2 as there.  Forward: B = W^T copy,
 * B2 = stacked lora_B^T [r][N]; input gradient: B = W, B2 = stacked lora_A [G*r][N], group_cols = 0.  K, K2 % 32 == 0;
 * group_cols a multiple of 256. */
int rv_gemm_nn_lora_bf16(const void* A, long lda, const void* B, long ldb, const void* A2, long lda2, const void* B2,
                         long ldb2, int K2, int group_cols, int group0, void* C, long ldc, int M, int N, int K,
                         const void* residual, long ldr, void* stream);

/* C = dropout_{p,seed}(alpha * A B^T) + residual: rv_gemm_nt_bf16 whose result is masked with exactly the mask
 * rv_dropout(p, seed) draws for a contiguous [M][N] tensor, before the residual is added.  Backward of the LoRA branch
 * dropout: dx = dy W + mask * (dt A) / (1 - p) without materialising dt A.  N % 8 == 0. */
int rv_gemm_nt_dropout_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                            const void* residual, long ldr, float alpha, float p, int seed, void* stream);

/* C = A B + dropout_{p,seed}(A2 B2) (+ residual) in ONE pass over C: the LoRA input gradient under adapter dropout
 * (peft lora.Linear: result += lora_B(lora_A(dropout(x))) * scaling, so dx = dy W + mask * (dt A) / (1 - p); replaces
 * rv_gemm_nn_bf16 followed by rv_gemm_nt_dropout_bf16 with C as its own residual - two more passes over [M][N]).  A [M][K],
 * B [K][N], A2 = dt [M][K2], B2 = stacked lora_A [K2][N], all row-major; the adapter segment runs FIRST, the mask rv_dropout(p,
 * seed) draws for a contiguous [M][N] tensor is applied to the fp32 accumulators, the main segment accumulates on top (one
 * rounding to bf16 instead of two).  K % 64 == 0, K >= 512, K2 % 64 == 0, N % 8 == 0.  group_cols / group0 as in
 * rv_gemm_nn_lora_bf16 (A2 then has one K2-wide column block per group); with p = 0 this is the adapter-first form of that
 * function (the forward of a fused projection: one 64-deep adapter step per column tile instead of seam phases in the ring). */
int rv_gemm_nn_lora_pre_bf16(const void* A, long lda, const void* B, long ldb, const void* A2, long lda2, const void* B2,
                             long ldb2, int K2, int group_cols, int group0, float p, int seed, void* C, long ldc, int M, int N,
                             int K, const void* residual, long ldr, void* stream);

/* Split-K form of rv_gemm_tn_bf16 for skinny outputs (LoRA weight gradients: I or J = r): `splits` chunks of the
 * contraction rows are reduced by separate workgroups into fp32 slabs (workspace: splits*I*J floats, caller owned)
 * which a second pass sums in a fixed order: C = bf16(alpha * sum).  Deterministic. */
int rv_gemm_tn_bf16_splitk(const void* P, long ldp, const void* Q, long ldq, void* C, long ldc, int R, int I, int J,
                           float alpha, int splits, float* workspace, void* stream);

/* Weight-gradient contraction over the ROW index of both operands (no transposed copies in HBM):
 *   C[i][j] = alpha * sum_r P[r][i] * Q[r][j] + residual[i][j]      P [R][I], Q [R][J], C [I][J]; I, J % 8 == 0
 * e.g. dW = dY^T X (autograd of nn.Linear); operands are transposed on the fly by ds_read_b64_tr_b16. */
int rv_gemm_tn_bf16(const void* P, long ldp, const void* Q, long ldq, void* C, long ldc, int R, int I, int J,
                    const void* residual, long ldr, float alpha, void* stream);
/* rv_gemm_tn_bf16 (no residual, alpha 1) with a TAIL SPLIT: all output tiles of a weight gradient cost the same, so the last,
 * partly filled round of 256 tiles takes as long as a full one; with a workspace the tiles of that round are split over the
 * token axis into fp32 slabs and summed in fixed order (deterministic), e.g. wgu 1376 tiles: 6 -> 5.5 rounds.
 * rv_gemm_tn_workspace_floats: floats the plan for (R, I, J) needs (0: no split pays - then rv_gemm_tn_bf16_ws is
 * rv_gemm_tn_bf16).  A NULL / too small workspace also falls back to the plain launch. */
int rv_gemm_tn_workspace_floats(int R, int I, int J);
int rv_gemm_tn_bf16_ws(const void* P, long ldp, const void* Q, long ldq, void* C, long ldc, int R, int I, int J,
                       float* workspace, long workspace_floats, void* stream);
int rv_gemm_nt_bf16_f32out(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                           int variant, void* stream);
/* C (fp32) = sum_k A[m][k] B[n][k] + bias[n] + residual[m][n] (fp32; may alias C): the out_proj / fc2 products of the frozen CLIP
 * tower (llava/model/multimodal_encoder/clip_encoder.py:46-58, HF CLIPEncoderLayer) with the tower's residual stream carried in fp32
 * (RV_CLIP_FP32_RESID).  bias / residual may be NULL. */
int rv_gemm_nt_bf16_f32res(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                           const void* bias, const float* residual, long ldr, int variant, void* stream);

/* ---- fused LM head + log-softmax + label gather (replaces lm_head + get_batch_logps,
 *      muffin/eval/muffin_inference_logp.py:82-115; logits [rows, V] never reach HBM).
 *   fwd : per selected row m and 64-column block j:  pmax[m][j], psum[m][j] = max / sum exp(x - max);
 *         tgt_logit[m] = logit of tgt[m].   V = rows of W = a multiple of 64; columns [V_valid, V) are vocabulary
 *         padding (a tokenizer with added tokens, e.g. OmniLMM's 32000 + 9): left out of the softmax, zero in dlogits.
 *   finish: lse[m], logp[m] = tgt_logit[m] - lse[m].
 *   bwd : dlogits[m][n] = coef[m] * ((n == tgt[m]) - exp(logit - lse[m]))  (bf16, recomputed logits). */
int rv_lmhead_logp_fwd(const void* h, long ldh, const void* W, long ldw, const int* tgt, int M, int V, int V_valid, int K,
                       float* pmax, float* psum, float* tgt_logit, int variant, void* stream);
int rv_logp_finish(const float* pmax, const float* psum, const float* tgt_logit, int nblk, int M, float* lse,
                   float* logp, void* stream);
int rv_lmhead_logp_bwd(const void* h, long ldh, const void* W, long ldw, const int* tgt, const float* lse,
                       const float* coef, void* dlogits, long ldd, int M, int V, int V_valid, int K, int variant,
                       void* stream);
/* (per_token_logps * loss_mask).sum(-1) and loss_mask.sum(-1) of get_batch_logps (:103-104); rows of
 * sequence s are seq_off[s] .. seq_off[s+1]; optional per-row weight (compute_weighted_logp,
 * muffin/train/trainers.py:128-137). Fixed summation order. */
int rv_seq_sum(const float* logp, const float* weight, const int* seq_off, int n_seq, float* out_sum, float* out_cnt,
               void* stream);
/* dpo_loss + loss mix + metrics (muffin/train/trainers.py:91-126, :292-309) and its closed-form gradient.
 *   per_pair[5][B] = losses, chosen_rewards, rejected_rewards, policy_win_logp, policy_rej_logp
 *   scalars[8]     = loss, mean chosen reward, mean rejected reward, accuracy, margin, mean win logp, mean rej logp, 0
 *   coef[2B]       = d loss / d (sequence log-prob sum) */
int rv_dpo_loss(const float* seq_sum, const float* seq_cnt, const float* ref_win, const float* ref_rej, int B,
                float beta, int use_average, float sft_weight, float dpo_weight, float* per_pair, float* scalars,
                float* coef, void* stream);
int rv_row_coef(const float* coef, const int* seq_of_row, const float* weight, float* out, int n, void* stream);

/* ---- attention (replaces HF LlamaAttention eager/SDPA math and CLIPAttention).
 *   qkv: [S*L][ld] with q heads at q_col0 + h*hd, k heads at k_col0 + h*hd, v heads at v_col0 + h*hd.
 *   out: [S*L][ldo] (head h at column h*hd); lse: [S][H][L] natural-log of sum exp(scale * q.k).
 *   causal=1: pure causal mask, no padding mask (muffin/train/trainers.py:199). hd in {64,128}.
 *   K/V tiles are staged by LDS-DMA and transposed on the fly (ds_read_b64_tr_b16): no side copies. */
/*   seg_sh / seg_e1 (int32 [S], both NULL = plain causal): packed preference pairs.  Row s holds
 *   [shared prefix | chosen branch | rejected branch]; queries at index >= seg_e1[s] (rejected branch) do not attend
 *   keys in [seg_sh[s], seg_e1[s]) (the chosen branch): the image / prompt prefix is computed ONCE per pair. */
/*   row_off / row_len (int32 [S], both NULL = S rectangular rows of L tokens; ABI 6): PAD-FREE rows.  Sequence s occupies token
 *   rows [row_off[s], row_off[s] + row_len[s]) of qkv / out (1 <= row_len[s] <= L); the reference right-pads every row to the batch
 *   maximum (llava/model/llava_arch.py:305-313) and SURVEY 8a property (i) makes skipping those pads exact.  L stays the MAXIMUM row
 *   length: it sizes the grid and the [S][H][L] stride of lse (and of the backward's delta planes); seg_sh / seg_e1 are per-row
 *   offsets as before.  Results of a row do not depend on where it starts (tiles are laid relative to the row's first token). */
int rv_attn_fwd(const void* qkv, long ld, int q_col0, int k_col0, int v_col0, void* out, long ldo, float* lse, int S,
                int L, int H, int hd, int causal, float scale, const int* seg_sh, const int* seg_e1, int kv_group,
                const int* row_off, const int* row_len, void* stream);
/* kv_group (both calls): grouped-query attention as in HF Mistral / Llama-3 (`repeat_kv`): H query heads share
 * H / kv_group key/value heads, query head h reads kv head h / kv_group (K at k_col0 + (h / kv_group) * hd); 1 = MHA.
 * backward (hd = 128): O = the forward output (rv_attn_fwd's `out`), delta = [3][S, H, L] fp32 WORKSPACE (ABI 5; it was [S, H, L]): the dQ
 * kernel fills it with rowsum(dO * O) (what rv_attn_delta computes), its negative and -lse / scale on the fly and the dK/dV kernel
 * reads them.  Writes dQ, dK, dV into
 * dqkv at the column offsets of qkv.  Deterministic (no atomics): one kernel per 128-query block for dQ, one per 128-key
 * block for dK/dV.  Packed rows: key tiles / query tiles that a whole block cannot see are never fetched.
 * rope_cos / rope_sin (fp32 [positions][64], both NULL = off) + rope_pos (int32 [S * L] position of every row, NULL = row
 * index inside its sequence): dQ and dK are written ALREADY rotated back (apply_rotary_pos_emb's autograd, HF
 * modeling_llama.py) - the separate rv_rope_inplace(backward) pass over dqkv is not needed.  rope_pos is indexed by the token's row
 * in the buffer (row_off[s] + i under pad-free rows, where it is required). */
int rv_attn_bwd(const void* qkv, long ld, int q_col0, int k_col0, int v_col0, const void* dO, long lddo,
                const void* O, long ldo, const float* lse, float* delta, void* dqkv, long lddq, int S, int L, int H,
                int hd, int causal, float scale, const int* seg_sh, const int* seg_e1, int kv_group,
                const float* rope_cos, const float* rope_sin, const int* rope_pos, const int* row_off, const int* row_len,
                void* stream);
/* Number of fp32 elements rv_attn_bwd's `delta` workspace must hold for (S, H, L) under THIS library's ABI (3 * S * H * L since
 * ABI 5).  Callers size the buffer with this query instead of hard-coding the plane count (the signature of rv_attn_bwd carries
 * no size argument). */
long rv_attn_bwd_workspace_floats(int S, int H, int L);
int rv_attn_delta(const void* dO, long lddo, const void* O, long ldo, float* delta, int S, int L, int H, int hd,
                  void* stream);

/* ---- norms / rotary / activations (replace HF LlamaRMSNorm, apply_rotary_pos_emb, LlamaMLP, CLIP LayerNorm) */
int rv_rmsnorm_fwd(const void* x, long ldx, const int* row_idx, const void* w, void* y, long ldy, float* rstd,
                   int rows, int d, float eps, void* stream);
int rv_rmsnorm_bwd_nblocks(int rows);   /* rows of the fp32 dw_partial scratch [nblocks][d] */
int rv_rmsnorm_bwd(const void* dy, long lddy, const void* x, long ldx, const int* row_idx, const void* w,
                   const float* rstd, const void* dres, long lddres, void* dx, long lddx, float* dw_partial,
                   void* dw, int dw_accumulate, int rows, int d, void* stream);
/* Opt-in fp32 RESIDUAL STREAM of the decoder (RV_RESID_FP32=1, round 5; HF's LlamaDecoderLayer keeps hidden_states in the model dtype,
 * llava_llama.py:91-102).  The projection GEMMs write their branch in bf16 WITHOUT the residual operand; these kernels carry the stream:
 *   rv_rmsnorm_fwd_f32:  add != NULL: xout = x + add (fp32 rows, same row indexing), y = rmsnorm(xout) * w (bf16), rstd;
 *                        add == NULL: y = rmsnorm(x) * w on the fp32 rows (row_idx gathers rows like rv_rmsnorm_fwd);
 *                        y == NULL with add: only the sum is written (no norm).
 *   rv_rmsnorm_bwd_f32x: rv_rmsnorm_bwd reading an fp32 x (gradients stay bf16);
 *   rv_add_f32_bf16:     out = x + b for the last branch of the stack. */
int rv_rmsnorm_fwd_f32(const float* x, long ldx, const int* row_idx, const void* add, long ldadd, float* xout, long ldxout,
                       const void* w, void* y, long ldy, float* rstd, int rows, int d, float eps, void* stream);
int rv_rmsnorm_bwd_f32x(const void* dy, long lddy, const float* x, long ldx, const int* row_idx, const void* w,
                        const float* rstd, const void* dres, long lddres, void* dx, long lddx, float* dw_partial,
                        void* dw, int dw_accumulate, int rows, int d, void* stream);
int rv_add_f32_bf16(const float* x, const void* b, float* out, long n, void* stream);
int rv_layernorm_fwd(const void* x, long ldx, const void* w, const void* b, void* y, long ldy, int rows, int d,
                     float eps, void* stream);
/* the same LayerNorm reading an fp32 input (the fp32 residual stream above); output bf16 (it feeds a bf16 MFMA operand) */
int rv_layernorm_fwd_f32in(const float* x, long ldx, const void* w, const void* b, void* y, long ldy, int rows, int d,
                           float eps, void* stream);
/* LayerNorm backward for the OmniLMM Resampler's trainable ln_q / ln_kv / ln_post (omnilmm/model/resampler.py:127-129,
 * autograd of F.layer_norm): dx (NULL = not wanted), dw / db written or accumulated.  x_period > 0: rows r and r + x_period
 * share the x row r % x_period (the learned queries, identical for every image).  partial: fp32 scratch
 * [rv_rmsnorm_bwd_nblocks(rows)][2 d]. */
int rv_layernorm_bwd(const void* dy, long lddy, const void* x, long ldx, int x_period, const void* w, void* dx, long lddx,
                     float* partial, void* dw, void* db, int accumulate, int rows, int d, float eps, void* stream);
/* y[r] = x[r] + p[r % period]: the Resampler's position tables added per image (resampler.py:150-155) */
int rv_add_rows(const void* x, long ldx, const void* p, long ldp, int period, void* y, long ldy, long rows, int d,
                void* stream);
/* y[q] = sum_b x[b * period + q]: gradient of a row block that was broadcast over the batch (resampler.py:166-167 _repeat) */
int rv_sum_rows_periodic(const void* x, long ldx, int period, int reps, void* y, long ldy, int d, void* stream);
/* in-place half-split RoPE over n_heads_total adjacent heads (q then k); position = pos[token] when pos != NULL
 * (packed pairs), else token % L (position_ids are dropped: llava/model/language_model/llava_llama.py:94);
 * backward = inverse rotation. */
int rv_rope_inplace(void* x, long ld, const float* cos_tab, const float* sin_tab, const int* pos, long n_tok, int L,
                    int n_heads_total, int hd, int backward, void* stream);
/* SwiGLU on a stored gate|up tensor [rows][2f] and its backward.  interleaved = 0: columns [gate | up];  1: column 2j = gate_j,
 * 2j+1 = up_j (the layout of the fused weight whose GEMM epilogues do this work - rv_gemm_nn_swiglu_bf16 - so the standalone
 * kernels then only serve the recompute paths). */
int rv_swiglu_fwd(const void* gu, long ldgu, void* act, long lda, long rows, int f, int interleaved, void* stream);
int rv_swiglu_bwd(const void* dact, long ldd, const void* gu, long ldgu, void* dgu, long lddgu, long rows, int f,
                  int interleaved, void* stream);
int rv_gelu_fwd(const void* x, void* y, long n, void* stream);
int rv_gelu_bwd(const void* dy, const void* x, void* dx, long n, void* stream);
/* Dropout of the LoRA branch input (peft lora_dropout, muffin/train/train_llava15_lora.py:114,309): element e is kept
 * iff hash16(seed, e) >= p * 2^16 and scaled by 1/(1-p); the same (seed, n) regenerates the same mask in backward.
 * y = dropped x (may alias x, may be NULL); acc (optional) += dropped x (gradient accumulation).  Contiguous, n % 8 == 0. */
int rv_dropout(const void* x, void* y, void* acc, long n, float p, int seed, void* stream);
/* Producer-side forms: rv_rmsnorm_fwd / rv_swiglu_fwd (block layout) that ALSO write yd = rv_dropout(y; p, seed) as a contiguous
 * [rows][d] / [rows][f] tensor, bit-identical to running rv_dropout on the output - the LoRA branch input of the projection that
 * follows (q|k|v, gate|up after the norms; down after SwiGLU) without re-reading the activation. */
int rv_rmsnorm_fwd_dropout(const void* x, long ldx, const int* row_idx, const void* w, void* y, long ldy, float* rstd,
                           int rows, int d, float eps, void* yd, float p, int seed, void* stream);
int rv_swiglu_fwd_dropout(const void* gu, long ldgu, void* act, long lda, long rows, int f, void* actd, float p, int seed,
                          void* stream);

/* ---- CLIP image preprocessing: CLIPImageProcessor of openai/clip-vit-large-patch14-336 as the reference applies it in
 * its DataLoader workers (muffin/train/train_llava15.py:244, muffin/train/train_utils.py:208) - PIL BICUBIC resize
 * (Pillow Resample.c: 22-bit fixed-point taps, uint8 after each pass), center crop, x 1/255, (x - mean) / std.
 * Two passes over the CROPPED window only; tap tables (bounds [n][2] = first source index, tap count; kk [n][ksize])
 * are built on the host exactly as Pillow's precompute_coeffs does.
 *   rv_resize_h_u8      : tmp[r][x][c] (uint8, rows x out_w x 3) from src [H][W][3] rows y0 .. y0+rows-1
 *   rv_resize_v_norm_u8 : out[c][y][x] (float32, 3 x out_h x out_w) = table[c][vertical pass], table = [3][256] */
int rv_resize_h_u8(const void* src, int H, int W, int y0, int rows, const int* bounds, const int* kk, int ksize, int out_w,
                   void* tmp, void* stream);
int rv_resize_v_norm_u8(const void* tmp, int rows, int out_w, const int* bounds, const int* kk, int ksize, int out_h,
                        const float* table, float* out, void* stream);

/* ---- data movement */
int rv_transpose(const void* in, long ld_in, void* out, long ldo, int R, int C, void* stream);
/* prepare_inputs_labels_for_multimodal (llava/model/llava_arch.py:237-315): out row n = embed[src[n]] if
 * src[n] >= 0, zero if -1, image feature row (-2 - src[n]) otherwise. */
int rv_splice_fwd(const int* src, const void* embed, const void* feats, void* out, long n_rows, int d, void* stream);
int rv_embed_bwd(const int* uniq_ids, const int* seg_off, const int* pos_sorted, int n_uniq, const void* dx, void* dW,
                 int d, void* stream);
int rv_feat_grad(const int* src_a, const int* src_b, const void* dx, void* dfeat, long n_rows, int d, void* stream);
int rv_gather_rows(const void* in, long ld_in, const int* idx, void* out, long ld_out, long n_rows, int d, int scatter,
                   void* stream);
int rv_colsum(const void* dy, long ld, void* db, int M, int N, void* stream);
/* CLIPVisionEmbeddings (patch conv as GEMM): fp32 pixels -> bf16 patch rows [B*P][Kp], then CLS + pos-emb */
int rv_im2col_patches(const float* pixels, void* out, int B, int image_size, int patch, int Kp, void* stream);
int rv_clip_assemble(const void* patch, const void* cls, const void* pos, void* x, int B, int P, int d, void* stream);
int rv_cast_f32_to_bf16(const float* in, void* out, long n, void* stream);
int rv_cast_bf16_to_f32(const void* in, float* out, long n, void* stream);

/* ---- optimizer (replaces torch.optim.AdamW + clip_grad_norm_ selected by optim="adamw_torch",
 *      muffin/train/train_llava15.py:75).  The buffer may hold a SUM over ranks: pre_scale = 1/world.
 *      out2[0] = ||pre_scale*g||, out2[1] = pre_scale * min(1, max_norm/(out2[0]+1e-6)) = the factor
 *      rv_adamw_step multiplies raw gradients by (pass out2 as `clip`; NULL = 1). */
int rv_sumsq_nblocks(void);
int rv_grad_norm(const void* g, long n, float* partial, float max_norm, float pre_scale, float* out2,
                 void* stream);
int rv_adamw_step(void* p, float* master, float* m, float* v, const void* g, long n, float lr, float beta1, float beta2,
                  float eps, float wd, int step, const float* clip, void* stream);
/* The two halves of rv_grad_norm for a SHARDED optimizer (opt-in ZeRO-1, rlaif-v_amd/dist.py ShardedGradReducer; the reference
 * shards its optimizer the same way under DeepSpeed ZeRO-2, script/zero2.json:16-22): each rank holds the reduced gradient of its
 * shard only.  rv_grad_sumsq: out1[0] (+)= sum g^2 over a bf16 slice (partial: rv_sumsq_nblocks floats of scratch); the caller
 * all-reduces out1 over the ranks; rv_clip_from_sumsq: out2 = [pre_scale * sqrt(sumsq), pre_scale * min(1, max_norm / (that + 1e-6))],
 * the pair rv_grad_norm produces. */
int rv_grad_sumsq(const void* g, long n, float* partial, float* out1, int accumulate, void* stream);
int rv_clip_from_sumsq(const float* sumsq, float max_norm, float pre_scale, float* out2, void* stream);
/* --gradient_accumulation_steps (HF Trainer, script/train/llava15_train.sh:23): fp32 accumulation of the bf16 micro-batch
 * gradients.  mode 0: acc = g;  1: acc += g;  2: g = bf16((acc + g) * scale)  (scale = 1 / accumulation steps). */
int rv_grad_accum(float* acc, void* g, long n, int mode, float scale, void* stream);

/* ---- measurement helper (bench.py's single-GPU data-parallel probe; replaces nothing in the reference): dst = a + b
 *      (bf16, n elements) streamed by n_wg persistent 256-thread workgroups - the CU footprint of n_wg RCCL channels
 *      doing the receive-reduce-send of a ring all-reduce step (script/zero2.json:16-22 is what the real exchange replaces). */
int rv_reduce_copy_persistent(const void* a, const void* b, void* dst, long n, int n_wg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RLAIFV_HIP_H */
