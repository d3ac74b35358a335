"""Thin tensor-level wrappers over the C ABI (rlaif_v_amd.hip).  torch is used only for device
memory and streams; every FLOP / byte moved below runs in a hand-written gfx950 kernel.

All matrices are 2-D row-major views with unit column stride (``ld = stride(0)``) so callers can pass
column slices of fused buffers (q/k/v inside qkv, gate/up inside gu) without copies.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Tuple

import torch

from . import hip

BF16 = torch.bfloat16
ACT_NONE, ACT_QUICK_GELU, ACT_GELU = 0, 1, 2


def _chk2d(t: torch.Tensor, name: str):
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype != BF16 or not t.is_cuda:
        raise ValueError(f"{name}: expected a 2-D bf16 CUDA tensor with unit column stride, got "
                         f"{tuple(t.shape)} {t.dtype} strides {t.stride()} on {t.device}")


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------- GEMM
def gemm_nt(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, act: int = ACT_NONE, alpha: float = 1.0,
            variant: int = -1) -> torch.Tensor:
    """out[m][n] = act(alpha * sum_k a[m][k] b[n][k] + bias[n]) + residual[m][n]."""
    _chk2d(a, "a"), _chk2d(b, "b")
    M, K = a.shape
    N, Kb = b.shape
    if K != Kb:
        raise ValueError(f"gemm_nt: K mismatch {K} vs {Kb}")
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=a.device)
    _chk2d(out, "out")
    if residual is not None:
        _chk2d(residual, "residual")
    hip.call("rv_gemm_nt_bf16", a, a.stride(0), b, b.stride(0), out, out.stride(0), M, N, K, bias, residual,
             residual.stride(0) if residual is not None else 0, act, float(alpha), variant)
    return out


def gemm_nn(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, alpha: float = 1.0, bias: Optional[torch.Tensor] = None,
            act: int = ACT_NONE) -> torch.Tensor:
    """out[m][n] = act(alpha * sum_k a[m][k] b[k][n] + bias[n]) + residual[m][n]   (b row-major [K, N]: rv_gemm_nn_bf16;
    with bias / activation: rv_gemm_nn_bias_act_bf16)."""
    _chk2d(a, "a"), _chk2d(b, "b")
    M, K = a.shape
    Kb, N = b.shape
    if K != Kb:
        raise ValueError(f"gemm_nn: inner dimensions differ: {K} vs {Kb}")
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=a.device)
    _chk2d(out, "out")
    if bias is not None or act != ACT_NONE:
        hip.call("rv_gemm_nn_bias_act_bf16", a, a.stride(0), b, b.stride(0), out, out.stride(0), M, N, K, bias, residual,
                 residual.stride(0) if residual is not None else 0, int(act), float(alpha))
        return out
    hip.call("rv_gemm_nn_bf16", a, a.stride(0), b, b.stride(0), out, out.stride(0), M, N, K, residual,
             residual.stride(0) if residual is not None else 0, float(alpha))
    return out


def linear(x: torch.Tensor, w: torch.Tensor, wT: Optional[torch.Tensor], residual: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x @ w^T (+ residual) for a weight held in both orientations (w [out, in], wT [in, out]).  Problems that fill
    the chip with 256x256 tiles take the NN kernel on wT - its weight tile is fetched in full 512-byte segments
    (+6-12 % over the NT kernel on the 7B shapes); small ones take the NT kernels on w."""
    M, N = x.shape[0], w.shape[0]
    if wT is not None and ((M + 255) // 256) * ((N + 255) // 256) >= 192 and x.shape[1] % 32 == 0 and N % 8 == 0:
        return gemm_nn(x, wT[:, :N], out=out, residual=residual)
    return gemm_nt(x, w, out=out, residual=residual)


def linear_rope_ok(M: int, N: int, K: int, rope_cols: int, hd: int) -> bool:
    """Whether ``linear_rope`` serves this q|k|v projection (else: ``linear`` + ``rope_inplace``)."""
    return (hd == 128 and N % 256 == 0 and rope_cols % 256 == 0 and K % 64 == 0 and K >= 512
            and ((M + 255) // 256) * (N // 256) >= 192)


def linear_rope(x: torch.Tensor, wT: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos: Optional[torch.Tensor], L: int,
                rope_cols: int, hd: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """qkv = x @ W_qkv^T with the q and k heads (the first ``rope_cols`` columns) leaving ROTATED: HF apply_rotary_pos_emb in the
    epilogue of the projection (rv_gemm_nn_rope_bf16), from the fp32 accumulators - one rounding instead of two and no separate
    pass over [tokens, 2 d].  wT = the [in, out] copy of the fused weight; cos / sin = rope_tables; pos = position per token or None."""
    _chk2d(x, "x"), _chk2d(wT, "wT")
    M, K = x.shape
    N = wT.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=x.device)
    hip.call("rv_gemm_nn_rope_bf16", x, x.stride(0), wT, wT.stride(0), out, out.stride(0), M, N, K, cos, sin, pos, L, rope_cols, hd)
    return out


def gemm_tn(p: torch.Tensor, q: torch.Tensor, out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
            alpha: float = 1.0) -> torch.Tensor:
    """out[i][j] = alpha * sum_r p[r][i] q[r][j] + residual[i][j]  (weight gradient dW = dY^T X)."""
    _chk2d(p, "p"), _chk2d(q, "q")
    R, I = p.shape
    Rq, J = q.shape
    if R != Rq:
        raise ValueError(f"gemm_tn: row mismatch {R} vs {Rq}")
    if out is None:
        out = torch.empty(I, J, dtype=BF16, device=p.device)
    _chk2d(out, "out")
    if residual is None and alpha == 1.0:
        # weight gradients: the partly filled last round of 256 tiles is split over the token axis (rv_gemm_tn_bf16_ws)
        need = _TN_WS_NEED.get((R, I, J))
        if need is None:
            need = _TN_WS_NEED[(R, I, J)] = int(hip.lib().lib.rv_gemm_tn_workspace_floats(R, I, J))
        if need > 0 and out.data_ptr() % 8 == 0:
            # scratch from the caching allocator per call: it is tied to the CURRENT stream there, freed blocks are recycled
            # stream-correctly, and nothing outlives the launch (a per-stream-handle cache could hand a new stream that reuses
            # a destroyed stream's handle a buffer the allocator still associates with the old one - ADVICE r3)
            ws = torch.empty(need, dtype=torch.float32, device=p.device)
            hip.call("rv_gemm_tn_bf16_ws", p, p.stride(0), q, q.stride(0), out, out.stride(0), R, I, J, ws, ws.numel())
            return out
    hip.call("rv_gemm_tn_bf16", p, p.stride(0), q, q.stride(0), out, out.stride(0), R, I, J, residual,
             residual.stride(0) if residual is not None else 0, float(alpha))
    return out


_TN_WS_NEED = {}     # (R, I, J) -> floats rv_gemm_tn_workspace_floats asks for (RV_TN_TAIL_SPLIT / RV_TN_TAIL_PENALTY are read
                     # ONCE per process, here and in the library)


def _lora_groups(N: int, group_cols: int, group0: int) -> int:
    """Number of adapter column groups of an N-wide fused projection: the first group0 columns (0 = group_cols), then
    groups of group_cols (grouped-query attention: q is hidden wide, k and v kv_dim wide)."""
    if not group_cols:
        return 1
    g0 = group0 or group_cols
    if g0 > N or (N - g0) % group_cols:
        raise ValueError(f"fused LoRA GEMM: groups {g0} + k x {group_cols} do not tile N = {N}")
    return 1 + (N - g0) // group_cols


def gemm_nt_lora(a: torch.Tensor, b: torch.Tensor, a2: torch.Tensor, b2: torch.Tensor, group_cols: int = 0,
                 out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, group0: int = 0) -> torch.Tensor:
    """out = a @ b^T + a2[:, c0(n) : c0(n)+K2] @ b2^T (+ residual): the fused LoRA GEMM (rv_gemm_nt_lora_bf16).
    K2 = b2.shape[1]; c0(n) = group(n) * K2 when group_cols > 0 (fused q|k|v, gate|up; _lora_groups), else 0."""
    _chk2d(a, "a"), _chk2d(b, "b"), _chk2d(a2, "a2"), _chk2d(b2, "b2")
    M, K = a.shape
    N, K2 = b2.shape
    if b.shape != (N, K) or a2.shape[0] != M:
        raise ValueError(f"gemm_nt_lora: shape mismatch a{tuple(a.shape)} b{tuple(b.shape)} a2{tuple(a2.shape)} b2{tuple(b2.shape)}")
    groups = _lora_groups(N, group_cols, group0)
    if a2.shape[1] != groups * K2:
        raise ValueError(f"gemm_nt_lora: a2 must have {groups} x {K2} columns, got {a2.shape[1]}")
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=a.device)
    _chk2d(out, "out")
    hip.call("rv_gemm_nt_lora_bf16", a, a.stride(0), b, b.stride(0), a2, a2.stride(0), b2, b2.stride(0), K2,
             int(group_cols), int(group0), out, out.stride(0), M, N, K, residual, residual.stride(0) if residual is not None else 0)
    return out


def gemm_nn_lora(a: torch.Tensor, b: torch.Tensor, a2: torch.Tensor, b2: torch.Tensor, group_cols: int = 0,
                 out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, group0: int = 0) -> torch.Tensor:
    """NN form of gemm_nt_lora: b [K, N] and b2 [K2, N] row-major (rv_gemm_nn_lora_bf16)."""
    _chk2d(a, "a"), _chk2d(b, "b"), _chk2d(a2, "a2"), _chk2d(b2, "b2")
    M, K = a.shape
    K2, N = b2.shape
    if b.shape != (K, N) or a2.shape[0] != M:
        raise ValueError(f"gemm_nn_lora: shape mismatch a{tuple(a.shape)} b{tuple(b.shape)} a2{tuple(a2.shape)} b2{tuple(b2.shape)}")
    groups = _lora_groups(N, group_cols, group0)
    if a2.shape[1] != groups * K2:
        raise ValueError(f"gemm_nn_lora: a2 must have {groups} x {K2} columns, got {a2.shape[1]}")
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=a.device)
    _chk2d(out, "out")
    hip.call("rv_gemm_nn_lora_bf16", a, a.stride(0), b, b.stride(0), a2, a2.stride(0), b2, b2.stride(0), K2,
             int(group_cols), int(group0), out, out.stride(0), M, N, K, residual, residual.stride(0) if residual is not None else 0)
    return out


def linear_lora(x: torch.Tensor, w: torch.Tensor, wT: torch.Tensor, a2: torch.Tensor, b2: torch.Tensor, b2T: torch.Tensor,
                group_cols: int = 0, residual: Optional[torch.Tensor] = None, group0: int = 0) -> torch.Tensor:
    """y = x @ w^T + a2[:, group block] @ b2^T (+ residual) with both weight orientations at hand (w [N, K], wT [K, N],
    b2 [N, K2], b2T [K2, N]): the NN kernel for chip-filling problems with 256-aligned groups, else the NT kernels."""
    M, N = x.shape[0], w.shape[0]
    if ((M + 255) // 256) * ((N + 255) // 256) >= 192 and group_cols % 256 == 0 and group0 % 256 == 0 and N % 8 == 0:
        K, K2 = x.shape[1], b2T.shape[0]
        # (adapter-first instead of in-ring: measured equal in time - 918.0 vs 917.9 ms per config-5 step - so the in-ring form,
        #  whose summation order the committed LoRA fixtures were checked with, stays the default: profiles/r04_lora_fwd_pre_ab.log)
        if K % 64 == 0 and K >= 512 and K2 % 64 == 0 and os.environ.get("RV_LORA_FWD_PRE", "0") == "1":
            return gemm_nn_lora_pre(x, wT[:, :N], a2, b2T[:, :N], out=None, residual=residual, group_cols=group_cols, group0=group0)
        return gemm_nn_lora(x, wT[:, :N], a2, b2T[:, :N], group_cols=group_cols, residual=residual, group0=group0)
    return gemm_nt_lora(x, w, a2, b2, group_cols=group_cols, residual=residual, group0=group0)


def gemm_nt_dropout(a: torch.Tensor, b: torch.Tensor, p: float, seed: int, out: Optional[torch.Tensor] = None,
                    residual: Optional[torch.Tensor] = None, alpha: float = 1.0) -> torch.Tensor:
    """out = dropout(alpha * a @ b^T; p, seed) + residual, the mask being the one ``dropout`` draws for an [M, N] tensor."""
    _chk2d(a, "a"), _chk2d(b, "b")
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=a.device)
    _chk2d(out, "out")
    hip.call("rv_gemm_nt_dropout_bf16", a, a.stride(0), b, b.stride(0), out, out.stride(0), M, N, K, residual,
             residual.stride(0) if residual is not None else 0, float(alpha), float(p), int(seed) & 0x7FFFFFFF)
    return out


def lora_dgrad_dropout(dy: torch.Tensor, w: torch.Tensor, wT: torch.Tensor, dt: torch.Tensor, a: torch.Tensor, aT: torch.Tensor,
                       p: float, seed: int) -> torch.Tensor:
    """dx = dy @ w + dropmask_{p,seed}(dt @ a) / (1 - p): the LoRA input gradient under adapter dropout (w [out, in] frozen base
    weight, wT its [in, out] copy, a = stacked lora_A [G r, in], aT its transpose).  One pass (rv_gemm_nn_lora_pre_bf16: adapter
    segment first, mask on the accumulators) when the shape fits the 256-tile NN kernel and fills the chip, else the plain input
    gradient followed by rv_gemm_nt_dropout_bf16 with dx as its own residual."""
    _chk2d(dy, "dy"), _chk2d(w, "w"), _chk2d(dt, "dt"), _chk2d(a, "a")
    M, K = dy.shape
    K2, N = a.shape
    if w.shape[0] != K or w.shape[1] < N or dt.shape != (M, K2):
        raise ValueError(f"lora_dgrad_dropout: shape mismatch dy{tuple(dy.shape)} w{tuple(w.shape)} dt{tuple(dt.shape)} a{tuple(a.shape)}")
    if (K % 64 == 0 and K >= 512 and K2 % 64 == 0 and N % 8 == 0 and ((M + 255) // 256) * ((N + 255) // 256) >= 192
            and os.environ.get("RV_LORA_DGRAD_PRE", "1") != "0"):
        return gemm_nn_lora_pre(dy, w[:, :N], dt, a, p, seed)
    dx = linear(dy, wT, w)
    gemm_nt_dropout(dt, aT, p, seed, out=dx, residual=dx)
    return dx


def gemm_nn_lora_pre(a: torch.Tensor, b: torch.Tensor, a2: torch.Tensor, b2: torch.Tensor, p: float = 0.0, seed: int = 0,
                     out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, group_cols: int = 0,
                     group0: int = 0) -> torch.Tensor:
    """out = a @ b + dropmask_{p,seed}(a2[:, group block] @ b2) / (1 - p) (+ residual); b [K, N], b2 [K2, N] row-major
    (rv_gemm_nn_lora_pre_bf16: the adapter segment runs first)."""
    _chk2d(a, "a"), _chk2d(b, "b"), _chk2d(a2, "a2"), _chk2d(b2, "b2")
    M, K = a.shape
    K2, N = b2.shape
    groups = _lora_groups(N, group_cols, group0)
    if b.shape != (K, N) or a2.shape != (M, groups * K2):
        raise ValueError(f"gemm_nn_lora_pre: shape mismatch a{tuple(a.shape)} b{tuple(b.shape)} a2{tuple(a2.shape)} b2{tuple(b2.shape)}")
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=a.device)
    _chk2d(out, "out")
    hip.call("rv_gemm_nn_lora_pre_bf16", a, a.stride(0), b, b.stride(0), a2, a2.stride(0), b2, b2.stride(0), K2, int(group_cols),
             int(group0), float(p), int(seed) & 0x7FFFFFFF, out, out.stride(0), M, N, K, residual,
             residual.stride(0) if residual is not None else 0)
    return out


_SPLITK_WS = {}


def gemm_tn_skinny(p: torch.Tensor, q: torch.Tensor, out: Optional[torch.Tensor] = None, alpha: float = 1.0,
                   splits: int = 0) -> torch.Tensor:
    """gemm_tn for outputs with few 256x256 tiles (LoRA weight gradients): split-K over the token rows so the whole
    chip works on it; deterministic fp32 second pass (rv_gemm_tn_bf16_splitk)."""
    _chk2d(p, "p"), _chk2d(q, "q")
    R, I = p.shape
    Rq, J = q.shape
    if R != Rq:
        raise ValueError(f"gemm_tn_skinny: row mismatch {R} vs {Rq}")
    if out is None:
        out = torch.empty(I, J, dtype=BF16, device=p.device)
    _chk2d(out, "out")
    tiles = ((I + 255) // 256) * ((J + 255) // 256)
    if splits <= 0:
        splits = max(1, min(256 // tiles, (R + 511) // 512))
    key = (p.device, torch.cuda.current_stream(p.device).cuda_stream)
    need = splits * I * J
    ws = _SPLITK_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=p.device)
        _SPLITK_WS[key] = ws
    hip.call("rv_gemm_tn_bf16_splitk", p, p.stride(0), q, q.stride(0), out, out.stride(0), R, I, J, float(alpha),
             int(splits), ws)
    return out


def gemm_nt_f32(a, b, variant: int = -1) -> torch.Tensor:
    _chk2d(a, "a"), _chk2d(b, "b")
    out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
    hip.call("rv_gemm_nt_bf16_f32out", a, a.stride(0), b, b.stride(0), out, out.stride(0), a.shape[0], b.shape[0],
             a.shape[1], variant)
    return out


def transpose(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[R, C] -> [C, roundup(R, 64)] (zero padded columns): the K-contiguous operand of a wgrad GEMM."""
    _chk2d(x, "x")
    R, C = x.shape
    Rp = round_up(R, 64)
    if out is None:
        out = torch.empty(C, Rp, dtype=BF16, device=x.device)
    hip.call("rv_transpose", x, x.stride(0), out, out.stride(0), R, C)
    return out


# ------------------------------------------------------------------------------------- norms etc.
def _chk_stream(x, name):
    """a residual-stream operand: 2-D bf16, or 2-D fp32 with unit inner stride (the opt-in fp32 stream)"""
    if x.dtype == torch.float32:
        if not (x.is_cuda and x.dim() == 2 and x.stride(1) == 1):
            raise ValueError(f"{name}: expected a 2-D fp32 CUDA tensor with unit column stride")
    else:
        _chk2d(x, name)


def rmsnorm_fwd(x, w, eps: float, row_idx: Optional[torch.Tensor] = None, out=None, want_rstd: bool = True):
    _chk_stream(x, "x")
    rows = x.shape[0] if row_idx is None else row_idx.numel()
    d = x.shape[1]
    if out is None:
        out = torch.empty(rows, d, dtype=BF16, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if want_rstd else None
    if x.dtype == torch.float32:       # the opt-in fp32 residual stream (RV_RESID_FP32): same call, fp32 rows in
        hip.call("rv_rmsnorm_fwd_f32", x, x.stride(0), row_idx, None, 0, None, 0, w, out, out.stride(0), rstd, rows, d, float(eps))
        return out, rstd
    hip.call("rv_rmsnorm_fwd", x, x.stride(0), row_idx, w, out, out.stride(0), rstd, rows, d, float(eps))
    return out, rstd


def add_rmsnorm_fwd(x32: torch.Tensor, branch: torch.Tensor, w, eps: float, want_norm: bool = True):
    """fp32 residual stream: (x32 + branch [fp32, new buffer], rmsnorm(sum) * w [bf16], rstd) in one pass (rv_rmsnorm_fwd_f32 with
    ``add``).  ``branch``: the bf16 output of o_proj / down_proj WITHOUT the residual operand.  want_norm=False: only the sum."""
    if x32.dtype != torch.float32 or branch.dtype != BF16 or x32.shape != branch.shape:
        raise ValueError("add_rmsnorm_fwd: fp32 stream and bf16 branch of equal shape required")
    rows, d = x32.shape
    xout = torch.empty_like(x32)
    y = torch.empty(rows, d, dtype=BF16, device=x32.device) if want_norm else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x32.device) if want_norm else None
    hip.call("rv_rmsnorm_fwd_f32", x32, x32.stride(0), None, branch, branch.stride(0), xout, xout.stride(0), w, y,
             y.stride(0) if y is not None else 0, rstd, rows, d, float(eps))
    return xout, y, rstd


def rmsnorm_fwd_dropout(x, w, eps: float, p: float, seed: int, row_idx: Optional[torch.Tensor] = None, want_rstd: bool = True):
    """(y, rstd, yd): rmsnorm_fwd plus yd = dropout(y, p, seed) written by the same kernel (rv_rmsnorm_fwd_dropout)."""
    _chk2d(x, "x")
    rows = x.shape[0] if row_idx is None else row_idx.numel()
    d = x.shape[1]
    out = torch.empty(rows, d, dtype=BF16, device=x.device)
    outd = torch.empty(rows, d, dtype=BF16, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if want_rstd else None
    hip.call("rv_rmsnorm_fwd_dropout", x, x.stride(0), row_idx, w, out, out.stride(0), rstd, rows, d, float(eps), outd, float(p),
             int(seed) & 0x7FFFFFFF)
    return out, rstd, outd


def rmsnorm_bwd(dy, x, w, rstd, dw: torch.Tensor, dres: Optional[torch.Tensor] = None,
                row_idx: Optional[torch.Tensor] = None, dx: Optional[torch.Tensor] = None,
                dw_accumulate: bool = False) -> torch.Tensor:
    """dx (same row indexing as x) = rmsnorm backward (+ dres); dw (bf16 [d]) written or accumulated."""
    _chk2d(dy, "dy"), _chk_stream(x, "x")
    rows, d = dy.shape
    if dx is None:      # (bf16 whatever the stream's dtype: gradients stay bf16 under the fp32 residual stream)
        dx = (torch.empty if row_idx is None else torch.zeros)(x.shape, dtype=BF16, device=x.device)
    nb = hip.lib().lib.rv_rmsnorm_bwd_nblocks(rows)
    partial = torch.empty(nb, d, dtype=torch.float32, device=x.device)
    if x.dtype == torch.float32:       # fp32 residual stream: x is fp32, the gradient stream stays bf16
        hip.call("rv_rmsnorm_bwd_f32x", dy, dy.stride(0), x, x.stride(0), row_idx, w, rstd, dres,
                 dres.stride(0) if dres is not None else 0, dx, dx.stride(0), partial, dw, int(dw_accumulate), rows, d)
        return dx
    hip.call("rv_rmsnorm_bwd", dy, dy.stride(0), x, x.stride(0), row_idx, w, rstd, dres,
             dres.stride(0) if dres is not None else 0, dx, dx.stride(0), partial, dw, int(dw_accumulate), rows, d)
    return dx


def layernorm_fwd(x, w, b, eps: float, out=None):
    _chk2d(x, "x")
    if out is None:
        out = torch.empty_like(x)
    hip.call("rv_layernorm_fwd", x, x.stride(0), w, b, out, out.stride(0), x.shape[0], x.shape[1], float(eps))
    return out


def layernorm_fwd_f32in(x: torch.Tensor, w, b, eps: float):
    """LayerNorm of an fp32 [rows, d] input -> bf16 (rv_layernorm_fwd_f32in)."""
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("layernorm_fwd_f32in: fp32 [rows, d] input with unit inner stride required")
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    hip.call("rv_layernorm_fwd_f32in", x, x.stride(0), w, b, out, out.stride(0), x.shape[0], x.shape[1], float(eps))
    return out


def gemm_nt_f32res(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor], residual: torch.Tensor, variant: int = -1):
    """residual (fp32 [M, N]) += a @ b^T + bias, IN PLACE (rv_gemm_nt_bf16_f32res with C aliasing the residual); returns it."""
    _chk2d(a, "a"), _chk2d(b, "b")
    if residual.dtype != torch.float32 or residual.shape != (a.shape[0], b.shape[0]) or residual.stride(1) != 1:
        raise ValueError("gemm_nt_f32res: fp32 residual [M, N] required")
    hip.call("rv_gemm_nt_bf16_f32res", a, a.stride(0), b, b.stride(0), residual, residual.stride(0), a.shape[0], b.shape[0],
             a.shape[1], bias, residual, residual.stride(0), variant)
    return residual


def layernorm_bwd(dy, x, w, eps: float, dw: torch.Tensor, db: torch.Tensor, want_dx: bool = True, x_period: int = 0,
                  accumulate: bool = False) -> Optional[torch.Tensor]:
    """F.layer_norm backward: returns dx [rows, d] (None if not wanted); dw / db (bf16 [d]) written or accumulated.
    ``x_period``: x has only that many rows and row r of dy belongs to x row r % x_period."""
    _chk2d(dy, "dy"), _chk2d(x, "x")
    rows, d = dy.shape
    dx = torch.empty(rows, d, dtype=dy.dtype, device=dy.device) if want_dx else None
    nb = hip.lib().lib.rv_rmsnorm_bwd_nblocks(rows)
    partial = torch.empty(nb, 2 * d, dtype=torch.float32, device=dy.device)
    hip.call("rv_layernorm_bwd", dy, dy.stride(0), x, x.stride(0), int(x_period), w, dx, dx.stride(0) if want_dx else 0,
             partial, dw, db, int(accumulate), rows, d, float(eps))
    return dx


def add_rows(x: torch.Tensor, p: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r] = x[r] + p[r % len(p)]"""
    _chk2d(x, "x"), _chk2d(p, "p")
    if out is None:
        out = torch.empty_like(x)
    hip.call("rv_add_rows", x, x.stride(0), p, p.stride(0), p.shape[0], out, out.stride(0), x.shape[0], x.shape[1])
    return out


def sum_rows_periodic(x: torch.Tensor, period: int) -> torch.Tensor:
    """out[q] = sum_b x[b * period + q]"""
    _chk2d(x, "x")
    assert x.shape[0] % period == 0
    out = torch.empty(period, x.shape[1], dtype=x.dtype, device=x.device)
    hip.call("rv_sum_rows_periodic", x, x.stride(0), period, x.shape[0] // period, out, out.stride(0), x.shape[1])
    return out


def rope_tables(L: int, hd: int, theta: float, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 cos/sin [L, hd/2] exactly as HF LlamaRotaryEmbedding computes them (fp32, before any cast)."""
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos().contiguous().to(device), fr.sin().contiguous().to(device)


def rope_inplace(x: torch.Tensor, cos, sin, L: int, n_heads_total: int, hd: int, backward: bool = False,
                 pos: Optional[torch.Tensor] = None):
    _chk2d(x, "x")
    hip.call("rv_rope_inplace", x, x.stride(0), cos, sin, pos, x.shape[0], L, n_heads_total, hd, int(backward))
    return x


def swiglu_fwd(gu: torch.Tensor, out=None, interleaved: bool = False):
    """act = silu(gate) * up from gu = [gate | up] columns, or interleaved (column 2j = gate_j, 2j+1 = up_j)."""
    _chk2d(gu, "gu")
    rows, f2 = gu.shape
    f = f2 // 2
    if out is None:
        out = torch.empty(rows, f, dtype=BF16, device=gu.device)
    hip.call("rv_swiglu_fwd", gu, gu.stride(0), out, out.stride(0), rows, f, int(interleaved))
    return out


def swiglu_fwd_dropout(gu: torch.Tensor, p: float, seed: int):
    """(act, actd): swiglu_fwd on the block layout plus actd = dropout(act, p, seed) from the same kernel (rv_swiglu_fwd_dropout)."""
    _chk2d(gu, "gu")
    rows, f2 = gu.shape
    f = f2 // 2
    out = torch.empty(rows, f, dtype=BF16, device=gu.device)
    outd = torch.empty(rows, f, dtype=BF16, device=gu.device)
    hip.call("rv_swiglu_fwd_dropout", gu, gu.stride(0), out, out.stride(0), rows, f, outd, float(p), int(seed) & 0x7FFFFFFF)
    return out, outd


def swiglu_bwd(dact, gu, out=None, interleaved: bool = False):
    rows, f2 = gu.shape
    if out is None:
        out = torch.empty_like(gu)
    hip.call("rv_swiglu_bwd", dact, dact.stride(0), gu, gu.stride(0), out, out.stride(0), rows, f2 // 2, int(interleaved))
    return out


def linear_swiglu(x: torch.Tensor, wguT: torch.Tensor):
    """(gu, act): gu = x @ W_gu^T with the INTERLEAVED fused weight (wguT = its [in, 2f] copy), act = silu(gate) * up computed
    in the GEMM epilogue (rv_gemm_nn_swiglu_bf16) - no separate SwiGLU pass."""
    _chk2d(x, "x"), _chk2d(wguT, "wguT")
    M, K = x.shape
    N = wguT.shape[1]
    if wguT.shape[0] != K:
        raise ValueError(f"linear_swiglu: inner dimensions differ: {K} vs {wguT.shape[0]}")
    rows = M if os.environ.get("RV_GU_TILE_MAJOR", "0") != "1" else (M + 255) // 256 * 256     # (experiment: tile-major gate|up, gemm.hip)
    gu = torch.empty(rows, N, dtype=BF16, device=x.device)[:M]
    act = torch.empty(M, N // 2, dtype=BF16, device=x.device)
    hip.call("rv_gemm_nn_swiglu_bf16", x, x.stride(0), wguT, wguT.stride(0), gu, gu.stride(0), act, act.stride(0), M, N, K)
    return gu, act


def linear_lora_swiglu(x: torch.Tensor, wguT: torch.Tensor, t: torch.Tensor, bexp: torch.Tensor, p: float = 0.0, seed: int = 0):
    """(gu, act, actd): the gate|up projection of an ADAPTER model with SwiGLU in the epilogue (rv_gemm_nn_lora_swiglu_bf16):
    gu = x W_gu^T + t bexp (interleaved gate / up columns; t = (alpha / r) dropout(x) [A_gate; A_up]^T [M, 2 r_pad], bexp = the
    expanded transposed adapter [2 r_pad, 2 f], ParamStore.gu_bexp), act = silu(gate) * up, actd = dropout(act; p, seed) when
    p > 0 (the dropped adapter input of the down projection; None otherwise)."""
    _chk2d(x, "x"), _chk2d(wguT, "wguT"), _chk2d(t, "t"), _chk2d(bexp, "bexp")
    M, K = x.shape
    N, K2 = wguT.shape[1], bexp.shape[0]
    if wguT.shape[0] != K or t.shape != (M, K2) or bexp.shape[1] != N:
        raise ValueError(f"linear_lora_swiglu: shapes x{tuple(x.shape)} wguT{tuple(wguT.shape)} t{tuple(t.shape)} bexp{tuple(bexp.shape)}")
    gu = torch.empty(M, N, dtype=BF16, device=x.device)
    act = torch.empty(M, N // 2, dtype=BF16, device=x.device)
    actd = torch.empty_like(act) if p > 0.0 else None
    hip.call("rv_gemm_nn_lora_swiglu_bf16", x, x.stride(0), wguT, wguT.stride(0), t, t.stride(0), bexp, bexp.stride(0), K2,
             gu, gu.stride(0), act, act.stride(0), actd, float(p), int(seed) & 0x7FFFFFFF, M, N, K)
    return gu, act, actd


def linear_lora_swiglu_ok(M: int, f: int, d: int, r_pad: int) -> bool:
    """Whether the fused adapter SwiGLU GEMMs serve this shape (chip-filling, 64-deep-A kernel)."""
    return d % 64 == 0 and d >= 512 and r_pad % 64 == 0 and f % 8 == 0 and ((M + 255) // 256) * ((2 * f + 255) // 256) >= 192


def linear_lora_swiglu_bwd(dy: torch.Tensor, w_down: torch.Tensor, dt: torch.Tensor, a_down: torch.Tensor, gu: torch.Tensor,
                           p: float = 0.0, seed: int = 0):
    """d(gate|up) [M, 2f] (interleaved like gu) of an ADAPTER model: SwiGLU'(gu) applied to d act = dy @ W_down +
    mask_{p,seed}(dt @ A_down) / (1 - p) in the epilogue of that GEMM (rv_gemm_nn_lora_swiglu_bwd_bf16); d act never reaches HBM.
    w_down [d, f] frozen base weight, a_down = lora_A of the down projection [r_pad, f], dt [M, r_pad]."""
    _chk2d(dy, "dy"), _chk2d(w_down, "w_down"), _chk2d(dt, "dt"), _chk2d(a_down, "a_down"), _chk2d(gu, "gu")
    M, K = dy.shape
    f, K2 = w_down.shape[1], a_down.shape[0]
    if w_down.shape[0] != K or gu.shape != (M, 2 * f) or dt.shape != (M, K2) or a_down.shape[1] != f:
        raise ValueError("linear_lora_swiglu_bwd: shapes")
    dgu = torch.empty_like(gu)
    hip.call("rv_gemm_nn_lora_swiglu_bwd_bf16", dy, dy.stride(0), w_down, w_down.stride(0), dt, dt.stride(0), a_down, a_down.stride(0),
             K2, float(p), int(seed) & 0x7FFFFFFF, gu, gu.stride(0), dgu, dgu.stride(0), M, f, K)
    return dgu


def linear_swiglu_bwd(dy: torch.Tensor, w_down: torch.Tensor, gu: torch.Tensor):
    """d(gate|up) [M, 2f] (interleaved like gu) = SwiGLU'(gu) applied to d act = dy @ W_down - the input gradient of the down
    projection with the SwiGLU backward in its epilogue (rv_gemm_nn_swiglu_bwd_bf16); d act never reaches HBM."""
    _chk2d(dy, "dy"), _chk2d(w_down, "w_down"), _chk2d(gu, "gu")
    M, K = dy.shape
    f = w_down.shape[1]
    if w_down.shape[0] != K or gu.shape != (M, 2 * f):
        raise ValueError(f"linear_swiglu_bwd: shapes dy{tuple(dy.shape)} w_down{tuple(w_down.shape)} gu{tuple(gu.shape)}")
    dgu = torch.empty_like(gu)
    hip.call("rv_gemm_nn_swiglu_bwd_bf16", dy, dy.stride(0), w_down, w_down.stride(0), gu, gu.stride(0), dgu, dgu.stride(0),
             M, f, K)
    return dgu


def gelu_fwd(x):
    y = torch.empty_like(x)
    hip.call("rv_gelu_fwd", x, y, x.numel())
    return y


_NO_OUT = object()


def dropout(x: torch.Tensor, p: float, seed: int, out=_NO_OUT, accumulate_into: Optional[torch.Tensor] = None):
    """y = keep(seed, e) ? x / (1 - p) : 0  (counter-based mask, regenerated - never stored - in backward).
    out=None with accumulate_into=t performs t += dropout(x) without materialising y."""
    if not x.is_contiguous() or (accumulate_into is not None and not accumulate_into.is_contiguous()):
        raise ValueError("dropout: contiguous tensors required")
    if out is _NO_OUT:
        out = torch.empty_like(x)
    hip.call("rv_dropout", x, out, accumulate_into, x.numel(), float(p), int(seed) & 0x7FFFFFFF)
    return out if out is not None else accumulate_into


def gelu_bwd(dy, x):
    dx = torch.empty_like(x)
    hip.call("rv_gelu_bwd", dy, x, dx, x.numel())
    return dx


# ------------------------------------------------------------------------------------- attention
def attn_fwd(qkv: torch.Tensor, S: int, L: int, H: int, hd: int, causal: bool, q_col0: int, k_col0: int,
             v_col0: int, out: Optional[torch.Tensor] = None, seg: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
             kv_group: int = 1, rows: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """Returns (out [tokens, H*hd], lse [S,H,L]).  kv_group > 1: grouped-query attention (H / kv_group key/value heads).
    ``rows`` = (row_off, row_len) int32 [S]: PAD-FREE rows - sequence s occupies token rows [row_off[s], row_off[s] + row_len[s])
    of qkv, L is the maximum row length; None: S rectangular rows of L tokens."""
    _chk2d(qkv, "qkv")
    if out is None:
        out = torch.empty(qkv.shape[0] if rows is not None else S * L, H * hd, dtype=BF16, device=qkv.device)
    lse = torch.empty(S, H, L, dtype=torch.float32, device=qkv.device)
    hip.call("rv_attn_fwd", qkv, qkv.stride(0), q_col0, k_col0, v_col0, out, out.stride(0), lse, S, L, H, hd,
             int(causal), 1.0 / math.sqrt(hd), seg[0] if seg else None, seg[1] if seg else None, int(kv_group),
             rows[0] if rows else None, rows[1] if rows else None)
    return out, lse


def attn_bwd(qkv, o, do, lse, S, L, H, hd, causal, q_col0, k_col0, v_col0, dqkv: Optional[torch.Tensor] = None,
             seg: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, kv_group: int = 1, rope=None,
             rows: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """Returns dqkv with dQ/dK/dV written at the qkv column offsets.  ``rope`` = (cos, sin, pos | None): dQ and dK come out
    already rotated back (the backward of apply_rotary_pos_emb fused into the stores).  ``rows``: pad-free rows (attn_fwd)."""
    _chk2d(qkv, "qkv"), _chk2d(o, "o"), _chk2d(do, "do")
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    # workspace: filled by the dQ kernel (delta, -delta, -lse / scale); sized by the library's own query
    delta = torch.empty(int(hip.lib().lib.rv_attn_bwd_workspace_floats(S, H, L)), dtype=torch.float32, device=qkv.device)
    hip.call("rv_attn_bwd", qkv, qkv.stride(0), q_col0, k_col0, v_col0, do, do.stride(0), o, o.stride(0), lse, delta,
             dqkv, dqkv.stride(0), S, L, H, hd, int(causal), 1.0 / math.sqrt(hd), seg[0] if seg else None,
             seg[1] if seg else None, int(kv_group), rope[0] if rope else None, rope[1] if rope else None,
             rope[2] if rope else None, rows[0] if rows else None, rows[1] if rows else None)
    return dqkv


# ------------------------------------------------------------------------------------- LM head / loss
def lmhead_logp_fwd(h: torch.Tensor, w: torch.Tensor, tgt: torch.Tensor, n_rows: int, v_valid: Optional[int] = None):
    """h: [rows_padded, d] (rows >= n_rows), w: [V, d], tgt int32 [n_rows] -> (logp, lse) fp32 [n_rows].
    ``v_valid``: the tokenizer's vocabulary when w carries padding rows up to a multiple of 64."""
    _chk2d(h, "h"), _chk2d(w, "w")
    V = w.shape[0]
    v_valid = V if v_valid is None else v_valid
    nblk = V // 64
    dev = h.device
    pmax = torch.empty(n_rows, nblk, dtype=torch.float32, device=dev)
    psum = torch.empty(n_rows, nblk, dtype=torch.float32, device=dev)
    tl = torch.zeros(n_rows, dtype=torch.float32, device=dev)
    lse = torch.empty(n_rows, dtype=torch.float32, device=dev)
    logp = torch.empty(n_rows, dtype=torch.float32, device=dev)
    hip.call("rv_lmhead_logp_fwd", h, h.stride(0), w, w.stride(0), tgt, n_rows, V, v_valid, h.shape[1], pmax, psum, tl, -1)
    hip.call("rv_logp_finish", pmax, psum, tl, nblk, n_rows, lse, logp)
    return logp, lse


def lmhead_logp_bwd(h, w, tgt, lse, coef, n_rows: int, out: Optional[torch.Tensor] = None, v_valid: Optional[int] = None):
    """dlogits bf16 [rows_padded, V]; rows >= n_rows and columns >= v_valid are zero."""
    V = w.shape[0]
    v_valid = V if v_valid is None else v_valid
    if out is None:
        # the kernel writes every row < n_rows completely (padding columns as zeros); only the < 64 padding ROWS need clearing -
        # not a 1.4 GB fill of the whole tensor
        out = torch.empty(h.shape[0], V, dtype=BF16, device=h.device)
        if h.shape[0] > n_rows:
            out[n_rows:].zero_()
    hip.call("rv_lmhead_logp_bwd", h, h.stride(0), w, w.stride(0), tgt, lse, coef, out, out.stride(0), n_rows, V, v_valid,
             h.shape[1], -1)
    return out


def seq_sum(logp, seq_off, n_seq: int, weight=None):
    s = torch.empty(n_seq, dtype=torch.float32, device=logp.device)
    c = torch.empty(n_seq, dtype=torch.float32, device=logp.device)
    hip.call("rv_seq_sum", logp, weight, seq_off, n_seq, s, c)
    return s, c


def dpo_loss(seq_sum_t, seq_cnt_t, ref_win, ref_rej, beta: float, use_average: bool, sft_weight: float,
             dpo_weight: float):
    B = ref_win.numel()
    dev = ref_win.device
    per_pair = torch.empty(5, B, dtype=torch.float32, device=dev)
    scalars = torch.empty(8, dtype=torch.float32, device=dev)
    coef = torch.empty(2 * B, dtype=torch.float32, device=dev)
    hip.call("rv_dpo_loss", seq_sum_t, seq_cnt_t, ref_win, ref_rej, B, float(beta), int(use_average),
             float(sft_weight), float(dpo_weight), per_pair, scalars, coef)
    return per_pair, scalars, coef


def row_coef(coef, seq_of_row, weight=None):
    n = seq_of_row.numel()
    out = torch.empty(n, dtype=torch.float32, device=coef.device)
    hip.call("rv_row_coef", coef, seq_of_row, weight, out, n)
    return out


# ------------------------------------------------------------------------------------- data movement
def splice_fwd(src: torch.Tensor, embed: torch.Tensor, feats: torch.Tensor, d: int):
    out = torch.empty(src.numel(), d, dtype=BF16, device=embed.device)
    hip.call("rv_splice_fwd", src, embed, feats, out, src.numel(), d)
    return out


def embed_bwd(uniq_ids, seg_off, pos_sorted, dx, dW):
    hip.call("rv_embed_bwd", uniq_ids, seg_off, pos_sorted, uniq_ids.numel(), dx, dW, dx.shape[1])
    return dW


def feat_grad(src_a, src_b, dx, d: int):
    out = torch.empty(src_a.numel(), d, dtype=BF16, device=dx.device)
    hip.call("rv_feat_grad", src_a, src_b, dx, out, src_a.numel(), d)
    return out


def gather_rows(x, idx, out=None):
    _chk2d(x, "x")
    if out is None:
        out = torch.empty(idx.numel(), x.shape[1], dtype=BF16, device=x.device)
    hip.call("rv_gather_rows", x, x.stride(0), idx, out, out.stride(0), idx.numel(), x.shape[1], 0)
    return out


def scatter_rows(x, idx, out):
    hip.call("rv_gather_rows", x, x.stride(0), idx, out, out.stride(0), idx.numel(), x.shape[1], 1)
    return out


def colsum(dy, out=None):
    _chk2d(dy, "dy")
    if out is None:
        out = torch.empty(dy.shape[1], dtype=BF16, device=dy.device)
    hip.call("rv_colsum", dy, dy.stride(0), out, dy.shape[0], dy.shape[1])
    return out


def im2col_patches(pixels: torch.Tensor, patch: int, Kp: int):
    B, _, H, _ = pixels.shape
    P = (H // patch) ** 2
    out = torch.empty(B * P, Kp, dtype=BF16, device=pixels.device)
    hip.call("rv_im2col_patches", pixels, out, B, H, patch, Kp)
    return out


def clip_assemble(patch_emb, cls, pos, B: int, P: int):
    d = patch_emb.shape[1]
    out = torch.empty(B * (P + 1), d, dtype=BF16, device=patch_emb.device)
    hip.call("rv_clip_assemble", patch_emb, cls, pos, out, B, P, d)
    return out


def cast_bf16_to_f32(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    hip.call("rv_cast_bf16_to_f32", x.contiguous(), out, x.numel())
    return out


def cast_f32_to_bf16(x: torch.Tensor, out: Optional[torch.Tensor] = None):
    if out is None:
        out = torch.empty(x.shape, dtype=BF16, device=x.device)
    hip.call("rv_cast_f32_to_bf16", x, out, x.numel())
    return out


def reduce_copy_persistent(a: torch.Tensor, b: torch.Tensor, dst: torch.Tensor, n_wg: int):
    """dst = a + b (flat bf16) streamed by ``n_wg`` persistent workgroups: the single-GPU stand-in for the CU footprint of an
    RCCL ring step (bench.py's data-parallel probe; not on the training path)."""
    n = a.numel()
    if a.dtype != BF16 or b.dtype != BF16 or dst.dtype != BF16 or b.numel() < n or dst.numel() < n or not (
            a.is_contiguous() and b.is_contiguous() and dst.is_contiguous()):
        raise ValueError("reduce_copy_persistent: contiguous bf16 buffers, b and dst at least as long as a")
    hip.call("rv_reduce_copy_persistent", a, b, dst, n, int(n_wg))
    return dst


# ------------------------------------------------------------------------------------- optimizer
def grad_norm(g: torch.Tensor, max_norm: float, out2: Optional[torch.Tensor] = None, pre_scale: float = 1.0):
    """out2 = [||pre_scale*g||, pre_scale * clip coefficient] on device (no host sync)."""
    nb = hip.lib().lib.rv_sumsq_nblocks()
    partial = torch.empty(nb, dtype=torch.float32, device=g.device)
    if out2 is None:
        out2 = torch.empty(2, dtype=torch.float32, device=g.device)
    hip.call("rv_grad_norm", g, g.numel(), partial, float(max_norm), float(pre_scale), out2)
    return out2


def add_f32_bf16(x32: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """x32 + b (fp32 stream + bf16 branch) -> new fp32 buffer (rv_add_f32_bf16)."""
    out = torch.empty_like(x32)
    hip.call("rv_add_f32_bf16", x32, b, out, x32.numel())
    return out


def grad_sumsq(g: torch.Tensor, out1: torch.Tensor, accumulate: bool = False):
    """out1[0] (+)= sum of squares of the bf16 slice g (device scalar, no host sync): the local half of a sharded grad norm."""
    if g.numel() == 0:
        if not accumulate:
            out1.zero_()
        return out1
    partial = torch.empty(hip.lib().lib.rv_sumsq_nblocks(), dtype=torch.float32, device=g.device)
    hip.call("rv_grad_sumsq", g, g.numel(), partial, out1, int(accumulate))
    return out1


def clip_from_sumsq(sumsq: torch.Tensor, max_norm: float, out2: torch.Tensor, pre_scale: float = 1.0):
    """out2 = [||pre_scale g||, pre_scale * clip coefficient] from the (all-reduced) sum of squares: grad_norm's second half."""
    hip.call("rv_clip_from_sumsq", sumsq, float(max_norm), float(pre_scale), out2)
    return out2


def grad_accum(acc: torch.Tensor, g: torch.Tensor, mode: int, scale: float = 1.0):
    """fp32 gradient accumulation over micro-batches: mode 0 acc = g, 1 acc += g, 2 g = bf16((acc + g) * scale)."""
    if acc.dtype != torch.float32 or g.dtype != BF16 or acc.numel() != g.numel() or not (acc.is_contiguous() and g.is_contiguous()):
        raise ValueError("grad_accum: fp32 accumulator and bf16 gradient slice of equal length required")
    hip.call("rv_grad_accum", acc, g, g.numel(), int(mode), float(scale))


def adamw_step(p, master, m, v, g, lr, beta1, beta2, eps, wd, step: int, clip: Optional[torch.Tensor] = None):
    hip.call("rv_adamw_step", p, master, m, v, g, p.numel(), float(lr), float(beta1), float(beta2), float(eps),
             float(wd), int(step), clip)
