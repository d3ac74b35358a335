"""OmniLMM's vision tower: timm ``eva02_enormous_patch14_clip_224`` as ``create_vision_module`` builds it
(omnilmm/model/omnilmm.py:31-43: ``dynamic_img_size``, last block replaced by Identity, ``forward_features`` minus the
prefix token, :107-119) - forward only (the tower is frozen in this path), on the same HIP kernels as the CLIP tower.

PARITY UNPINNED.  timm is not vendored by the reference and absent offline, so this file restates the published
architecture from the timm 0.9.10 ``models/eva.py`` definition as the author knows it:
  * ``Eva(embed_dim=1792, depth=64, num_heads=16, mlp_ratio=15360/1792, use_post_norm=True, global_pool='token')``:
    Conv patch embedding with bias, class token, learned absolute positions (resampled bicubically, with antialiasing, from
    the 16 x 16 pre-training grid to the input's grid), no rotary embedding, no SwiGLU;
  * ``EvaBlockPostNorm``:  x = x + LN1(attn(x));  x = x + LN2(mlp(x))  (LayerNorm eps 1e-6);
  * ``EvaAttention`` with fused qkv: weight without bias + separate q_bias / v_bias (key bias is a zero buffer), softmax
    scale head_dim ** -0.5 with head_dim = 112, output projection with bias;  ``Mlp``: fc1 - GELU (erf) - fc2;
  * final ``norm`` LayerNorm applied inside forward_features.
tests/test_omnilmm_gpu.py checks the kernels against oracle/omnilmm_oracle.py::eva_forward_features - a restatement by the
same author, i.e. self-consistency, not parity with timm.  Use ``OmniLMMDPOModel.set_vision_tower(EvaTower(...))``.

Weight orientation (round 6): the tower is frozen and forward-only, so each block's four nn.Linear weights are kept ONLY as
W^T [in][out] and go through the NN GEMM with the bias / GELU epilogue (rv_gemm_nn_bias_act_bf16: the weight tile arrives in full
512-byte row segments; RV_EVA_NN=0 keeps [out][in] and the NT kernels).

Head dim 112 on the 128-wide attention kernels: every head is stored zero-padded to 128 (zero q / k columns add nothing to
the scores, zero v columns produce zero outputs that meet zero rows of the output projection) and the q rows carry the
factor sqrt(128 / 112) so that the kernels' 1 / sqrt(128) becomes 1 / sqrt(112).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F

from . import ops

BF16 = torch.bfloat16


@dataclass
class EvaConfig:
    width: int = 1792
    depth: int = 64
    heads: int = 16
    mlp: int = 15360
    patch: int = 14
    pretrain_grid: int = 16          # 224 / 14
    eps: float = 1e-6
    drop_last_block: bool = True     # omnilmm.py:43 "use 2nd last layer's output"

    @property
    def head_dim(self) -> int:
        return self.width // self.heads

    @property
    def blocks_used(self) -> int:
        return self.depth - 1 if self.drop_last_block else self.depth


def resample_pos_embed(pos: torch.Tensor, old_grid: int, new_grid: int) -> torch.Tensor:
    """[1 + old*old, C] -> [1 + new*new, C]: prefix token kept, grid part resized (bicubic, antialias) - timm
    ``resample_abs_pos_embed``.  Host-side construction of a constant."""
    if old_grid == new_grid:
        return pos
    prefix, grid = pos[:1], pos[1:]
    g = grid.float().reshape(1, old_grid, old_grid, -1).permute(0, 3, 1, 2)
    g = F.interpolate(g, size=(new_grid, new_grid), mode="bicubic", antialias=True, align_corners=False)
    return torch.cat([prefix.float(), g.permute(0, 2, 3, 1).reshape(new_grid * new_grid, -1)], 0).to(pos.dtype)


class EvaTower:
    def __init__(self, cfg: EvaConfig, device="cuda:0"):
        if cfg.head_dim > 128 or cfg.head_dim % 8:
            raise ValueError("EvaTower: head_dim must be <= 128 and a multiple of 8")
        self.cfg, self.device = cfg, torch.device(device)
        self.w: Dict[str, torch.Tensor] = {}
        self._pos: Dict[int, torch.Tensor] = {}
        self.nn_form = False             # the block weights are stored transposed ([in][out]) for the NN GEMM

    # ---- weights (timm names under ``prefix``)
    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = ""):
        c, dev = self.cfg, self.device
        d, H, hd = c.width, c.heads, c.head_dim
        Kp = ops.round_up(3 * c.patch * c.patch, 64)

        def t(name):
            return sd[prefix + name].detach().float()

        def dev16(x):
            return x.to(BF16).to(dev).contiguous()

        pw = torch.zeros(d, Kp)
        pw[:, :3 * c.patch * c.patch] = t("patch_embed.proj.weight").reshape(d, -1)
        w = self.w
        w["patch_w"], w["patch_b"] = dev16(pw), dev16(t("patch_embed.proj.bias"))
        w["cls"] = dev16(t("cls_token").reshape(d))
        self._pos_raw = t("pos_embed").reshape(-1, d)
        qscale = math.sqrt(128.0 / hd)
        for i in range(c.blocks_used):
            p = f"blocks.{i}."
            qkv = t(p + "attn.qkv.weight").reshape(3, H, hd, d)
            qb = torch.stack([t(p + "attn.q_bias").reshape(H, hd), torch.zeros(H, hd), t(p + "attn.v_bias").reshape(H, hd)])
            wp = torch.zeros(3, H, 128, d)
            bp = torch.zeros(3, H, 128)
            wp[:, :, :hd], bp[:, :, :hd] = qkv, qb
            wp[0] *= qscale
            bp[0] *= qscale
            w[f"{i}.wqkv"], w[f"{i}.bqkv"] = dev16(wp.reshape(3 * H * 128, d)), dev16(bp.reshape(-1))
            wo = torch.zeros(d, H, 128)
            wo[:, :, :hd] = t(p + "attn.proj.weight").reshape(d, H, hd)
            w[f"{i}.wo"], w[f"{i}.bo"] = dev16(wo.reshape(d, H * 128)), dev16(t(p + "attn.proj.bias"))
            for n in ("norm1", "norm2"):
                w[f"{i}.{n}.w"], w[f"{i}.{n}.b"] = dev16(t(p + n + ".weight")), dev16(t(p + n + ".bias"))
            w[f"{i}.fc1.w"], w[f"{i}.fc1.b"] = dev16(t(p + "mlp.fc1.weight")), dev16(t(p + "mlp.fc1.bias"))
            w[f"{i}.fc2.w"], w[f"{i}.fc2.b"] = dev16(t(p + "mlp.fc2.weight")), dev16(t(p + "mlp.fc2.bias"))
        w["norm.w"], w["norm.b"] = dev16(t("norm.weight")), dev16(t("norm.bias"))
        self._pos = {}
        self._finalize_layout()

    def init_random(self, seed: int = 0, std: float = 0.02):
        """Random weights of the full architecture drawn directly on the device (benchmarks: no checkpoint exists offline;
        4.4 B parameters = 8.8 GB bf16 at the EVA02-E defaults), in the padded-head layout load_state_dict builds."""
        c, dev = self.cfg, self.device
        d, H, hd = c.width, c.heads, c.head_dim
        g = torch.Generator(device=dev).manual_seed(seed)
        Kp = ops.round_up(3 * c.patch * c.patch, 64)

        def rn(*shape):
            return (torch.randn(*shape, device=dev, generator=g) * std).to(BF16)

        def ln():
            return torch.ones(d, dtype=BF16, device=dev), torch.zeros(d, dtype=BF16, device=dev)

        w = self.w
        w["patch_w"] = rn(d, Kp)
        w["patch_w"][:, 3 * c.patch * c.patch:] = 0
        w["patch_b"], w["cls"] = rn(d), rn(d)
        self._pos_raw = (torch.randn(1 + c.pretrain_grid ** 2, d, generator=torch.Generator().manual_seed(seed)) * std)
        head_mask = torch.zeros(3, H, 128, 1, device=dev, dtype=BF16)
        head_mask[:, :, :hd] = 1                                     # columns hd..127 of every head are zero padding
        for i in range(c.blocks_used):
            w[f"{i}.wqkv"] = (rn(3, H, 128, d) * head_mask).reshape(3 * H * 128, d)
            bq = rn(3, H, 128) * head_mask[..., 0]
            bq[1] = 0                                                # EvaAttention: no key bias
            w[f"{i}.bqkv"] = bq.reshape(-1)
            w[f"{i}.wo"] = (rn(d, H, 128) * head_mask[0, :, :, 0]).reshape(d, H * 128)
            w[f"{i}.bo"] = rn(d)
            for n in ("norm1", "norm2"):
                w[f"{i}.{n}.w"], w[f"{i}.{n}.b"] = ln()
            w[f"{i}.fc1.w"], w[f"{i}.fc1.b"] = rn(c.mlp, d), rn(c.mlp)
            w[f"{i}.fc2.w"], w[f"{i}.fc2.b"] = rn(d, c.mlp), rn(d)
        w["norm.w"], w["norm.b"] = ln()
        self._pos = {}
        self._finalize_layout()
        return self

    def _finalize_layout(self):
        """Block weights to [in][out] (one tensor at a time: no second copy of the 8.8 GB tower is ever alive)."""
        c = self.cfg
        self.nn_form = (os.environ.get("RV_EVA_NN", "1") != "0"
                        and all(k % 32 == 0 for k in (c.width, c.heads * 128, c.mlp)))      # K % 32 of the NN kernels
        if not self.nn_form:
            return
        for i in range(c.blocks_used):
            for key in (f"{i}.wqkv", f"{i}.wo", f"{i}.fc1.w", f"{i}.fc2.w"):
                self.w[key] = self.w[key].t().contiguous()

    def _linear(self, x: torch.Tensor, key: str, bias: torch.Tensor, act: int = ops.ACT_NONE) -> torch.Tensor:
        if self.nn_form:
            return ops.gemm_nn(x, self.w[key], bias=bias, act=act)
        return ops.gemm_nt(x, self.w[key], bias=bias, act=act)

    def n_params(self) -> int:
        """Parameters of the un-padded architecture (accounting)."""
        c = self.cfg
        d = c.width
        per_block = 3 * d * d + 2 * d + d * d + d + 4 * d + 2 * d * c.mlp + c.mlp + d
        return c.blocks_used * per_block + d * 3 * c.patch * c.patch + 2 * d + (1 + c.pretrain_grid ** 2) * d + 2 * d

    def flops_per_image(self, image_size: int) -> float:
        """Algorithmic forward FLOPs of one image (head dim 112, no padding counted)."""
        c = self.cfg
        T = (image_size // c.patch) ** 2 + 1
        d = c.width
        per_tok = 2 * d * 3 * d + 2 * d * d + 4 * d * c.mlp + 4 * T * d
        return c.blocks_used * T * per_tok + 2.0 * (T - 1) * d * 3 * c.patch * c.patch

    def pos(self, grid: int) -> torch.Tensor:
        if grid not in self._pos:
            self._pos[grid] = resample_pos_embed(self._pos_raw, self.cfg.pretrain_grid, grid).to(BF16).to(self.device).contiguous()
        return self._pos[grid]

    # ---- forward_features minus the prefix token
    @torch.no_grad()
    def __call__(self, pixels: torch.Tensor) -> torch.Tensor:
        c, w = self.cfg, self.w
        B, _, Hpx, Wpx = pixels.shape
        if Hpx != Wpx or Hpx % c.patch:
            raise ValueError("EvaTower: square images whose side is a multiple of the patch size (dynamic_img_pad is not restated)")
        grid = Hpx // c.patch
        P, T, d, H = grid * grid, grid * grid + 1, c.width, c.heads
        px = pixels.to(self.device, dtype=torch.float32).contiguous()
        cols = ops.im2col_patches(px, c.patch, w["patch_w"].shape[1])
        pe = ops.gemm_nt(cols, w["patch_w"], bias=w["patch_b"])
        x = ops.clip_assemble(pe, w["cls"], self.pos(grid), B, P)                  # [cls | patches] + positions
        if os.environ.get("RV_EVA_FP32_RESID", "1") != "0":
            # The frozen tower's residual stream in fp32 (round 5, like the CLIP tower's: DESIGN section 2): 2 x 63 bf16 roundings of
            # x become roundings of the (small) post-norm branch outputs; the GEMM operands are bf16 casts of the stream.
            x32 = ops.cast_bf16_to_f32(x)
            for i in range(c.blocks_used):
                qkv = self._linear(x, f"{i}.wqkv", w[f"{i}.bqkv"])
                a, _ = ops.attn_fwd(qkv, B, T, H, 128, False, 0, H * 128, 2 * H * 128)
                o = self._linear(a, f"{i}.wo", w[f"{i}.bo"])
                x32 = ops.add_f32_bf16(x32, ops.layernorm_fwd(o, w[f"{i}.norm1.w"], w[f"{i}.norm1.b"], c.eps))
                x = ops.cast_f32_to_bf16(x32)
                h = self._linear(x, f"{i}.fc1.w", w[f"{i}.fc1.b"], ops.ACT_GELU)
                m = self._linear(h, f"{i}.fc2.w", w[f"{i}.fc2.b"])
                x32 = ops.add_f32_bf16(x32, ops.layernorm_fwd(m, w[f"{i}.norm2.w"], w[f"{i}.norm2.b"], c.eps))
                x = ops.cast_f32_to_bf16(x32)
            x = ops.layernorm_fwd_f32in(x32, w["norm.w"], w["norm.b"], c.eps)
            idx = (torch.arange(B, device=self.device)[:, None] * T + 1 + torch.arange(P, device=self.device)[None]).reshape(-1)
            return ops.gather_rows(x, idx.to(torch.int32)).view(B, P, d)
        for i in range(c.blocks_used):
            qkv = self._linear(x, f"{i}.wqkv", w[f"{i}.bqkv"])
            a, _ = ops.attn_fwd(qkv, B, T, H, 128, False, 0, H * 128, 2 * H * 128)
            o = self._linear(a, f"{i}.wo", w[f"{i}.bo"])
            x = ops.add_rows(x, ops.layernorm_fwd(o, w[f"{i}.norm1.w"], w[f"{i}.norm1.b"], c.eps))
            h = self._linear(x, f"{i}.fc1.w", w[f"{i}.fc1.b"], ops.ACT_GELU)
            m = self._linear(h, f"{i}.fc2.w", w[f"{i}.fc2.b"])
            x = ops.add_rows(x, ops.layernorm_fwd(m, w[f"{i}.norm2.w"], w[f"{i}.norm2.b"], c.eps))
        x = ops.layernorm_fwd(x, w["norm.w"], w["norm.b"], c.eps)
        idx = (torch.arange(B, device=self.device)[:, None] * T + 1 + torch.arange(P, device=self.device)[None]).reshape(-1)
        return ops.gather_rows(x, idx.to(torch.int32)).view(B, P, d)
