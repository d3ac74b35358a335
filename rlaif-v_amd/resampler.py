"""OmniLMM perceiver ``Resampler`` (omnilmm/model/resampler.py:96-168) on the HIP kernels: forward and the manual backward of
every trainable tensor (kv_proj, ln_q / ln_kv / ln_post, the learned queries, nn.MultiheadAttention's in / out projections,
``proj``).  SURVEY.md section 8 row f4.

The single cross-attention layer has ``num_query`` (64) learned queries against the N tower tokens of an image (1024 at
448 px).  It runs on the self-attention kernels of the decoder (`rv_attn_fwd/bwd`, non-causal, head dim 128) by laying each
image out as ONE sequence of N rows in a fused [q | k | v] buffer whose q columns hold the projected queries in rows 0..nq-1
and zeros below: rows >= nq produce values nobody reads, and because their upstream gradient is zero they contribute exactly
zero to dK / dV (dS = P * (dP - delta) with dP = dO V^T = 0 and delta = dO . O = 0).  16x redundant query rows on a layer that
is 0.02 % of the step's FLOPs, and no new attention kernel.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import ops

BF16 = torch.bfloat16
PREFIX = "model.resampler."


def sincos_2d(embed_dim: int, grid: int) -> torch.Tensor:
    """get_2d_sincos_pos_embed (resampler.py:42-93): [grid*grid, embed_dim] fp32, frozen."""
    def one_d(dim, pos):
        omega = np.arange(dim // 2, dtype=np.float32)
        omega /= dim / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gh = np.arange(grid, dtype=np.float32)
    gw = np.arange(grid, dtype=np.float32)
    g = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid, grid])
    return torch.from_numpy(np.concatenate([one_d(embed_dim // 2, g[0]), one_d(embed_dim // 2, g[1])], axis=1)).float()


def abs_pos(table: torch.Tensor, n_tokens: int) -> torch.Tensor:
    """get_abs_pos (resampler.py:22-39): the frozen table resized bicubically to sqrt(n_tokens)^2 rows.  Host-side
    construction of a constant (once per token count), not part of the step."""
    src, tgt = int(math.sqrt(table.shape[0])), int(math.sqrt(n_tokens))
    if src == tgt:
        return table
    return F.interpolate(table.float().reshape(1, src, src, -1).permute(0, 3, 1, 2), size=(tgt, tgt), mode="bicubic",
                         align_corners=False).permute(0, 2, 3, 1).flatten(0, 2)


def param_shapes(d: int, kv_dim: int, num_query: int) -> Dict[str, tuple]:
    s = {"query": (num_query, d), "kv_proj.weight": (d, kv_dim), "attn.in_proj_weight": (3 * d, d),
         "attn.in_proj_bias": (3 * d,), "attn.out_proj.weight": (d, d), "attn.out_proj.bias": (d,), "proj": (d, d)}
    for n in ("ln_q", "ln_kv", "ln_post"):
        s[n + ".weight"] = (d,)
        s[n + ".bias"] = (d,)
    return s


class Resampler:
    def __init__(self, d: int, kv_dim: int, num_query: int, device, eps: float = 1e-6):
        if d % 128 != 0:
            raise ValueError("Resampler: embed_dim must be a multiple of 128 (num_heads = embed_dim // 128, omnilmm.py:49)")
        g = int(math.sqrt(num_query))
        if g * g != num_query:
            raise ValueError("Resampler: num_query must be a square (grid_size = sqrt(num_query), omnilmm.py:47)")
        self.d, self.kv_dim, self.nq, self.heads, self.eps, self.device = d, kv_dim, num_query, d // 128, eps, device
        self.pos_table = sincos_2d(d, g)                      # fp32, CPU
        self.pos_q = self.pos_table.to(device=device, dtype=BF16)
        self._pos_k: Dict[int, torch.Tensor] = {}

    def pos_k(self, n_tokens: int) -> torch.Tensor:
        if n_tokens not in self._pos_k:
            self._pos_k[n_tokens] = abs_pos(self.pos_table, n_tokens).to(device=self.device, dtype=BF16).contiguous()
        return self._pos_k[n_tokens]

    # P(name) -> parameter view, G(name) -> gradient view (both bf16, names without the "model.resampler." prefix)
    def forward(self, x: torch.Tensor, B: int, N: int, P: Callable[[str], torch.Tensor], ctx: Optional[dict] = None) -> torch.Tensor:
        """x: [B*N, kv_dim] tower tokens (prefix tokens stripped, omnilmm.py:113-118) -> [B*num_query, d]."""
        d, nq, H = self.d, self.nq, self.heads
        if N < nq:
            raise ValueError("Resampler: fewer tower tokens than queries is not supported by the padded-query layout")
        xp = ops.gemm_nt(x, P("kv_proj.weight"))
        xk = ops.layernorm_fwd(xp, P("ln_kv.weight"), P("ln_kv.bias"), self.eps)
        kin = ops.add_rows(xk, self.pos_k(N))
        qn = ops.layernorm_fwd(P("query"), P("ln_q.weight"), P("ln_q.bias"), self.eps)
        qin = ops.add_rows(qn, self.pos_q)
        wi, bi = P("attn.in_proj_weight"), P("attn.in_proj_bias")
        qp = ops.gemm_nt(qin, wi[:d], bias=bi[:d])
        qkv = torch.zeros(B * N, 3 * d, dtype=BF16, device=x.device)
        qkv.view(B, N, 3 * d)[:, :nq, :d] = qp                                   # the same projected queries for every image
        ops.gemm_nt(kin, wi[d:2 * d], bias=bi[d:2 * d], out=qkv[:, d:2 * d])
        ops.gemm_nt(xk, wi[2 * d:], bias=bi[2 * d:], out=qkv[:, 2 * d:])
        attn, lse = ops.attn_fwd(qkv, B, N, H, 128, False, 0, d, 2 * d)
        o_sel = attn.view(B, N, d)[:, :nq].reshape(B * nq, d)
        o = ops.gemm_nt(o_sel, P("attn.out_proj.weight"), bias=P("attn.out_proj.bias"))
        y = ops.layernorm_fwd(o, P("ln_post.weight"), P("ln_post.bias"), self.eps)
        z = ops.gemm_nn(y, P("proj"))
        if ctx is not None:
            ctx.update(rs=dict(x=x, xp=xp, xk=xk, kin=kin, qin=qin, qkv=qkv, attn=attn, lse=lse, o_sel=o_sel, o=o, y=y, B=B, N=N))
        return z

    def backward(self, dz: torch.Tensor, ctx: dict, P: Callable[[str], torch.Tensor], G: Callable[[str], torch.Tensor]):
        """Fills every resampler gradient (overwrites).  The tower is frozen: no gradient flows into x."""
        c = ctx["rs"]
        d, nq, H, B, N = self.d, self.nq, self.heads, c["B"], c["N"]
        ops.gemm_tn(c["y"], dz, out=G("proj"))                                   # y^T dz  [in, out]
        dy = ops.gemm_nt(dz, P("proj"))                                          # dz proj^T
        do = ops.layernorm_bwd(dy, c["o"], P("ln_post.weight"), self.eps, G("ln_post.weight"), G("ln_post.bias"))
        ops.gemm_tn(do, c["o_sel"], out=G("attn.out_proj.weight"))
        ops.colsum(do, out=G("attn.out_proj.bias"))
        dsel = ops.gemm_nn(do, P("attn.out_proj.weight"))
        dO = torch.zeros(B * N, d, dtype=BF16, device=dz.device)
        dO.view(B, N, d)[:, :nq] = dsel.view(B, nq, d)
        dqkv = ops.attn_bwd(c["qkv"], c["attn"], dO, c["lse"], B, N, H, 128, False, 0, d, 2 * d)
        dq_sum = ops.sum_rows_periodic(dqkv.view(B, N, 3 * d)[:, :nq, :d].reshape(B * nq, d), nq)
        dk, dv = dqkv[:, d:2 * d], dqkv[:, 2 * d:]
        wi = P("attn.in_proj_weight")
        gw, gb = G("attn.in_proj_weight"), G("attn.in_proj_bias")
        ops.gemm_tn(dq_sum, c["qin"], out=gw[:d])
        ops.gemm_tn(dk, c["kin"], out=gw[d:2 * d])
        ops.gemm_tn(dv, c["xk"], out=gw[2 * d:])
        ops.colsum(dq_sum, out=gb[:d])
        ops.colsum(dk, out=gb[d:2 * d])
        ops.colsum(dv, out=gb[2 * d:])
        dqin = ops.gemm_nn(dq_sum, wi[:d])
        dquery = ops.layernorm_bwd(dqin, P("query"), P("ln_q.weight"), self.eps, G("ln_q.weight"), G("ln_q.bias"))
        G("query").copy_(dquery)
        dxk = ops.gemm_nn(dk, wi[d:2 * d])
        dxk = ops.gemm_nn(dv, wi[2 * d:], residual=dxk)
        dxp = ops.layernorm_bwd(dxk, c["xp"], P("ln_kv.weight"), self.eps, G("ln_kv.weight"), G("ln_kv.bias"))
        ops.gemm_tn(dxp, c["x"], out=G("kv_proj.weight"))
        ctx["rs"] = None
