"""LLaVA-1.5 policy for the DPO step, executed entirely by the gfx950 kernels of librlaifv_hip.so.

This is the host-side mirror of the reference's model surface for the DPO path
(/root/reference llava/model/language_model/llava_llama.py:41-102, llava/model/llava_arch.py:141-330,
llava/model/multimodal_encoder/clip_encoder.py:36-58, llava/model/multimodal_projector/builder.py:39-46)
with a hand-written forward AND backward (no autograd): torch only owns device memory and streams.

Data layout in HBM:
  * all trainable parameters live in ONE flat bf16 buffer (``flat_p``) in *backward-completion order*
    (lm_head, layer L-1 .. layer 0, embed_tokens, projector, then the no-decay tail: norm gains and
    biases); ``flat_g`` (bf16 grads), ``flat_master``/``flat_m``/``flat_v`` (fp32) mirror it, so AdamW is
    two launches (decay / no-decay range) and data-parallel all-reduce buckets are contiguous slices
    that become ready in address order while backward is still running;
  * q/k/v and gate/up projections are stored adjacent so each is ONE GEMM ([3d,d] and [2f,d]);
  * every weight that a dgrad GEMM needs K-contiguous has a transposed bf16 copy in ``flat_pT``
    (refreshed after each optimizer step by rv_transpose), so every contraction is the NT kernel;
  * activations are token-major [S*L, features] bf16; per layer we keep x, qkv (post-RoPE), attention
    output, x_mid, gate/up and the fp32 rstd/lse rows; normalised inputs and SwiGLU outputs are
    recomputed in backward (HBM-cheap) rather than stored.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .splice import SplicePlan, build_packed_plan, build_splice_plan

BF16 = torch.bfloat16
VT = "model.vision_tower.vision_tower.vision_model."


@dataclass
class LlavaConfig:
    """Same fields as the reference's checkpoint config (SURVEY.md section 8a notes)."""
    hidden: int = 4096
    layers: int = 32
    heads: int = 32
    ffn: int = 11008
    vocab: int = 32000
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    clip_hidden: int = 1024
    clip_layers: int = 24
    clip_heads: int = 16
    clip_ffn: int = 4096
    image_size: int = 336
    patch: int = 14
    clip_eps: float = 1e-5
    select_layer: int = -2
    model_max_length: int = 2048
    pad_token_id: int = 0

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def clip_head_dim(self) -> int:
        return self.clip_hidden // self.clip_heads

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch) ** 2

    @property
    def clip_layers_used(self) -> int:
        return self.select_layer if self.select_layer >= 0 else self.clip_layers + 1 + self.select_layer

    @property
    def patch_k(self) -> int:
        return ops.round_up(3 * self.patch * self.patch, 64)


def _is_decay(name: str) -> bool:
    return not (name.endswith("bias") or "norm" in name)


class ParamStore:
    """Flat parameter / gradient / optimizer-state buffers with named views."""

    def __init__(self, cfg: LlavaConfig, device, with_optimizer: bool = True):
        d, f, V, cd = cfg.hidden, cfg.ffn, cfg.vocab, cfg.clip_hidden
        # (key, shape, needs transposed copy)  in backward-completion order; fused keys map to HF names
        decay: List[Tuple[str, Tuple[int, ...], bool]] = [("lm_head.weight", (V, d), True)]
        for i in reversed(range(cfg.layers)):
            decay += [(f"layers.{i}.wdown", (d, f), True), (f"layers.{i}.wgu", (2 * f, d), True),
                      (f"layers.{i}.wo", (d, d), True), (f"layers.{i}.wqkv", (3 * d, d), True)]
        decay += [("model.embed_tokens.weight", (V, d), False),
                  ("model.mm_projector.2.weight", (d, d), True), ("model.mm_projector.0.weight", (d, cd), False)]
        nodecay: List[Tuple[str, Tuple[int, ...], bool]] = [("model.norm.weight", (d,), False)]
        for i in reversed(range(cfg.layers)):
            nodecay += [(f"layers.{i}.ln2", (d,), False), (f"layers.{i}.ln1", (d,), False)]
        nodecay += [("model.mm_projector.2.bias", (d,), False), ("model.mm_projector.0.bias", (d,), False)]
        self.entries = decay + nodecay
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for k, shp, _ in self.entries:
            n = math.prod(shp)
            assert n % 8 == 0, (k, shp)
            self.offsets[k] = (off, shp)
            off += n
            if k == decay[-1][0]:
                self.n_decay = off
        self.n_total = off
        self.device = device
        self.flat_p = torch.zeros(self.n_total, dtype=BF16, device=device)
        self.flat_g = torch.zeros(self.n_total, dtype=BF16, device=device)
        if with_optimizer:
            self.flat_master = torch.zeros(self.n_total, dtype=torch.float32, device=device)
            self.flat_m = torch.zeros(self.n_total, dtype=torch.float32, device=device)
            self.flat_v = torch.zeros(self.n_total, dtype=torch.float32, device=device)
        else:
            self.flat_master = self.flat_m = self.flat_v = None
        # transposed copies: W [out, in] -> W^T [in, out]   (out is a multiple of 64 for every entry)
        self.t_offsets: Dict[str, Tuple[int, Tuple[int, int]]] = {}
        toff = 0
        for k, shp, tr in self.entries:
            if tr:
                assert shp[0] % 64 == 0, (k, shp)
                self.t_offsets[k] = (toff, (shp[1], shp[0]))
                toff += shp[0] * shp[1]
        self.flat_pT = torch.zeros(toff, dtype=BF16, device=device)
        # gradient buckets for data parallelism: contiguous slices in the order backward finishes them
        self.buckets: Dict[str, Tuple[int, int]] = {"lm_head": self.grad_range("lm_head.weight", "lm_head.weight")}
        for i in reversed(range(cfg.layers)):
            self.buckets[f"layer{i}"] = self.grad_range(f"layers.{i}.wdown", f"layers.{i}.wqkv")
        self.buckets["embed_proj"] = self.grad_range("model.embed_tokens.weight", "model.mm_projector.0.weight")
        self.buckets["nodecay"] = (self.n_decay, self.n_total)

    def bucket_schedule(self) -> List[Tuple[str, int, int]]:
        """(name, start, end) in the order LlavaDPOModel.backward fires grad_ready_hook; covers flat_g exactly."""
        return [(k, a, b) for k, (a, b) in self.buckets.items()]

    def p(self, key: str) -> torch.Tensor:
        off, shp = self.offsets[key]
        return self.flat_p[off:off + math.prod(shp)].view(*shp)

    def g(self, key: str) -> torch.Tensor:
        off, shp = self.offsets[key]
        return self.flat_g[off:off + math.prod(shp)].view(*shp)

    def pT(self, key: str) -> torch.Tensor:
        off, shp = self.t_offsets[key]
        return self.flat_pT[off:off + shp[0] * shp[1]].view(*shp)

    def grad_range(self, first_key: str, last_key: str) -> Tuple[int, int]:
        a = self.offsets[first_key][0]
        off, shp = self.offsets[last_key]
        return a, off + math.prod(shp)

    def refresh_transposes(self):
        for k in self.t_offsets:
            ops.transpose(self.p(k), out=self.pT(k))

    def sync_master_from_params(self):
        if self.flat_master is not None:
            self.flat_master.copy_(self.flat_p)      # bf16 -> fp32 (device copy, plumbing)

    # ---- HF state-dict mapping ------------------------------------------------------------
    def hf_slices(self, cfg: LlavaConfig) -> Dict[str, Tuple[str, int, int]]:
        """HF name -> (store key, first row, n rows) for the language model + projector."""
        d, f = cfg.hidden, cfg.ffn
        m: Dict[str, Tuple[str, int, int]] = {}
        for i in range(cfg.layers):
            p = f"model.layers.{i}."
            m[p + "self_attn.q_proj.weight"] = (f"layers.{i}.wqkv", 0, d)
            m[p + "self_attn.k_proj.weight"] = (f"layers.{i}.wqkv", d, d)
            m[p + "self_attn.v_proj.weight"] = (f"layers.{i}.wqkv", 2 * d, d)
            m[p + "self_attn.o_proj.weight"] = (f"layers.{i}.wo", 0, d)
            m[p + "mlp.gate_proj.weight"] = (f"layers.{i}.wgu", 0, f)
            m[p + "mlp.up_proj.weight"] = (f"layers.{i}.wgu", f, f)
            m[p + "mlp.down_proj.weight"] = (f"layers.{i}.wdown", 0, d)
            m[p + "input_layernorm.weight"] = (f"layers.{i}.ln1", 0, d)
            m[p + "post_attention_layernorm.weight"] = (f"layers.{i}.ln2", 0, d)
        for k in ("lm_head.weight", "model.embed_tokens.weight", "model.norm.weight", "model.mm_projector.0.weight",
                  "model.mm_projector.0.bias", "model.mm_projector.2.weight", "model.mm_projector.2.bias"):
            m[k] = (k, 0, self.offsets[k][1][0])
        return m


@dataclass
class StepOutput:
    """What one DPO forward produced (all device tensors; nothing is synced to the host)."""
    loss: torch.Tensor                 # 0-d fp32
    scalars: torch.Tensor              # [8] see rv_dpo_loss
    per_pair: torch.Tensor             # [5, B]
    seq_logp: torch.Tensor             # [2B] sum of target log-probs (log_prob of get_batch_logps)
    seq_cnt: torch.Tensor              # [2B] number of targets
    per_token_logp: torch.Tensor       # [n_sel] fp32, selected rows only
    plan: SplicePlan = None
    ctx: dict = field(default_factory=dict)


class LlavaDPOModel:
    """Mirror of ``LlavaLlamaForCausalLM`` for the DPO call pattern of
    ``get_beta_and_logps`` (muffin/train/trainers.py:161-275): images -> CLIP (frozen) -> projector ->
    splice -> Llama stack -> fused LM-head log-probs -> DPO loss, plus the matching backward."""

    def __init__(self, cfg: LlavaConfig, device="cuda:0", with_optimizer: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError("LlavaDPOModel needs an MI355X (HIP) device; there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device(device)
        self.store = ParamStore(cfg, self.device, with_optimizer)
        self.clip: Dict[str, torch.Tensor] = {}
        self.training = True
        self._rope_cache: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.grad_ready_hook = None     # callable(name, start, end) fired when a slice of flat_g is final
        # compute the prefix shared by the chosen and rejected sequence of a pair once (splice.build_packed_plan)
        self.share_prefix = os.environ.get("RV_SHARE_PREFIX", "1") != "0"

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """HF-named fp32/bf16 CPU tensors (the reference's checkpoint layout, 4.35 CLIP key names)."""
        cfg, st = self.cfg, self.store
        self._clip_raw = {k: v.detach().to(BF16).cpu() for k, v in sd.items() if k.startswith(VT)}
        for name, (key, r0, n) in st.hf_slices(cfg).items():
            st.p(key)[r0:r0 + n].copy_(sd[name].to(BF16))
        st.sync_master_from_params()
        st.refresh_transposes()
        cd, Kp = cfg.clip_hidden, cfg.patch_k

        def dev(t):
            return t.to(BF16).to(self.device).contiguous()

        c = self.clip
        pw = torch.zeros(cd, Kp, dtype=BF16)
        pw[:, :3 * cfg.patch * cfg.patch] = sd[VT + "embeddings.patch_embedding.weight"].reshape(cd, -1).to(BF16)
        c["patch_w"] = pw.to(self.device)
        c["cls"] = dev(sd[VT + "embeddings.class_embedding"])
        c["pos"] = dev(sd[VT + "embeddings.position_embedding.weight"])
        c["pre_ln_w"], c["pre_ln_b"] = dev(sd[VT + "pre_layrnorm.weight"]), dev(sd[VT + "pre_layrnorm.bias"])
        for i in range(cfg.clip_layers_used):
            p = VT + f"encoder.layers.{i}."
            c[f"{i}.wqkv"] = dev(torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0))
            c[f"{i}.bqkv"] = dev(torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0))
            c[f"{i}.wo"], c[f"{i}.bo"] = dev(sd[p + "self_attn.out_proj.weight"]), dev(sd[p + "self_attn.out_proj.bias"])
            for ln in ("layer_norm1", "layer_norm2"):
                c[f"{i}.{ln}.w"], c[f"{i}.{ln}.b"] = dev(sd[p + ln + ".weight"]), dev(sd[p + ln + ".bias"])
            c[f"{i}.fc1.w"], c[f"{i}.fc1.b"] = dev(sd[p + "mlp.fc1.weight"]), dev(sd[p + "mlp.fc1.bias"])
            c[f"{i}.fc2.w"], c[f"{i}.fc2.b"] = dev(sd[p + "mlp.fc2.weight"]), dev(sd[p + "mlp.fc2.bias"])

    def init_random(self, seed: int = 0, std: float = 0.02):
        """HF-default style random init directly on the device (no checkpoints exist offline)."""
        cfg, st = self.cfg, self.store
        g = torch.Generator(device=self.device).manual_seed(seed)
        n = st.n_total
        chunk = 1 << 26
        for a in range(0, st.n_decay, chunk):
            b = min(st.n_decay, a + chunk)
            st.flat_p[a:b] = (torch.randn(b - a, device=self.device, generator=g) * std).to(BF16)
        st.flat_p[st.n_decay:n] = 1.0
        for k in ("model.mm_projector.2.bias", "model.mm_projector.0.bias"):
            st.p(k).zero_()
        st.sync_master_from_params()
        st.refresh_transposes()
        cd, Kp = cfg.clip_hidden, cfg.patch_k

        def rn(*shape, s=std):
            return (torch.randn(*shape, device=self.device, generator=g) * s).to(BF16)

        c = self.clip
        c["patch_w"] = rn(cd, Kp)
        c["patch_w"][:, 3 * cfg.patch * cfg.patch:] = 0
        c["cls"], c["pos"] = rn(cd), rn(cfg.n_patches + 1, cd)
        c["pre_ln_w"], c["pre_ln_b"] = torch.ones(cd, dtype=BF16, device=self.device), torch.zeros(cd, dtype=BF16, device=self.device)
        for i in range(cfg.clip_layers_used):
            c[f"{i}.wqkv"], c[f"{i}.bqkv"] = rn(3 * cd, cd), rn(3 * cd)
            c[f"{i}.wo"], c[f"{i}.bo"] = rn(cd, cd), rn(cd)
            for ln in ("layer_norm1", "layer_norm2"):
                c[f"{i}.{ln}.w"] = torch.ones(cd, dtype=BF16, device=self.device)
                c[f"{i}.{ln}.b"] = torch.zeros(cd, dtype=BF16, device=self.device)
            c[f"{i}.fc1.w"], c[f"{i}.fc1.b"] = rn(cfg.clip_ffn, cd), rn(cfg.clip_ffn)
            c[f"{i}.fc2.w"], c[f"{i}.fc2.b"] = rn(cd, cfg.clip_ffn), rn(cd)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """HF-named CPU bf16 tensors of the trainable part (safe_save_model_for_hf_trainer layout,
        muffin/train/train_llava15.py:102-112)."""
        out = {}
        for name, (key, r0, n) in self.store.hf_slices(self.cfg).items():
            out[name] = self.store.p(key)[r0:r0 + n].detach().cpu().clone()
        return out

    def clip_state_dict(self) -> Dict[str, torch.Tensor]:
        """The frozen tower under its checkpoint key names (kept from load_state_dict; empty after init_random)."""
        return dict(getattr(self, "_clip_raw", {}))

    def grads_state_dict(self) -> Dict[str, torch.Tensor]:
        out = {}
        for name, (key, r0, n) in self.store.hf_slices(self.cfg).items():
            out[name] = self.store.g(key)[r0:r0 + n].detach().float().cpu()
        return out

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    # ------------------------------------------------------------------ vision
    def _rope(self, L: int):
        if L not in self._rope_cache:
            self._rope_cache[L] = ops.rope_tables(L, self.cfg.head_dim, self.cfg.rope_theta, self.device)
        return self._rope_cache[L]

    def clip_features(self, pixels: torch.Tensor) -> torch.Tensor:
        """CLIPVisionTower.forward + feature_select('patch') (clip_encoder.py:36-58): [B*P, clip_hidden]."""
        cfg, c = self.cfg, self.clip
        B = pixels.shape[0]
        P, T, cd, H, hd = cfg.n_patches, cfg.n_patches + 1, cfg.clip_hidden, cfg.clip_heads, cfg.clip_head_dim
        px = pixels.to(self.device, dtype=torch.float32).contiguous()
        cols = ops.im2col_patches(px, cfg.patch, cfg.patch_k)
        pe = ops.gemm_nt(cols, c["patch_w"])
        x = ops.clip_assemble(pe, c["cls"], c["pos"], B, P)
        x = ops.layernorm_fwd(x, c["pre_ln_w"], c["pre_ln_b"], cfg.clip_eps)
        for i in range(cfg.clip_layers_used):
            h = ops.layernorm_fwd(x, c[f"{i}.layer_norm1.w"], c[f"{i}.layer_norm1.b"], cfg.clip_eps)
            qkv = ops.gemm_nt(h, c[f"{i}.wqkv"], bias=c[f"{i}.bqkv"])
            a, _ = ops.attn_fwd(qkv, B, T, H, hd, False, 0, cd, 2 * cd)
            x = ops.gemm_nt(a, c[f"{i}.wo"], bias=c[f"{i}.bo"], residual=x)
            h = ops.layernorm_fwd(x, c[f"{i}.layer_norm2.w"], c[f"{i}.layer_norm2.b"], cfg.clip_eps)
            h = ops.gemm_nt(h, c[f"{i}.fc1.w"], bias=c[f"{i}.fc1.b"], act=ops.ACT_QUICK_GELU)
            x = ops.gemm_nt(h, c[f"{i}.fc2.w"], bias=c[f"{i}.fc2.b"], residual=x)
        idx = (torch.arange(B, device=self.device)[:, None] * T + 1 + torch.arange(P, device=self.device)[None]).reshape(-1)
        return ops.gather_rows(x, idx.to(torch.int32))

    def encode_images(self, pixels: torch.Tensor, ctx: Optional[dict] = None) -> torch.Tensor:
        """llava_arch.py:141-148.  One pass per PAIR (the reference encodes [images, images],
        trainers.py:190; rows i and B+i are identical, so the pair shares one feature block)."""
        st = self.store
        f_clip = self.clip_features(pixels)
        z1 = ops.gemm_nt(f_clip, st.p("model.mm_projector.0.weight"), bias=st.p("model.mm_projector.0.bias"))
        h1 = ops.gelu_fwd(z1)
        feats = ops.gemm_nt(h1, st.p("model.mm_projector.2.weight"), bias=st.p("model.mm_projector.2.bias"))
        if ctx is not None:
            ctx.update(f_clip=f_clip, z1=z1, h1=h1)
        return feats

    # ------------------------------------------------------------------ forward
    def forward_logps(self, input_ids: torch.Tensor, labels: torch.Tensor, images: torch.Tensor,
                      save_for_backward: bool = True, all_rows: bool = False) -> StepOutput:
        """Everything of get_beta_and_logps up to ``get_batch_logps``: returns per-sequence log-prob sums
        and counts (muffin/eval/muffin_inference_logp.py:82-115) without materialising logits.
        ``all_rows`` (forward only, reference layout): evaluate EVERY position like
        ``get_batch_logps(return_all=True)`` does - masked positions get the log-prob of token id 0, exactly what
        the reference stores in its ``logps`` parquet column."""
        cfg, st = self.cfg, self.store
        d, H, hd, f = cfg.hidden, cfg.heads, cfg.head_dim, cfg.ffn
        B = images.shape[0]
        ctx: dict = {}
        w_rows = None
        if all_rows:
            if save_for_backward:
                raise ValueError("all_rows is a forward-only mode")
            plan = build_splice_plan(input_ids, labels, cfg.n_patches, B, cfg.model_max_length)
            nxt = plan.labels[:, 1:]
            S_, Lm1 = nxt.shape
            plan.sel_idx = (torch.arange(S_)[:, None] * plan.L + torch.arange(Lm1)[None]).reshape(-1).to(torch.int32)
            plan.tgt = torch.where(nxt != -100, nxt, torch.zeros_like(nxt)).reshape(-1).to(torch.int32)
            plan.seq_off = (torch.arange(S_ + 1) * Lm1).to(torch.int32)
            plan.seq_of_row = torch.arange(S_).repeat_interleave(Lm1).to(torch.int32)
            plan.n_sel = int(plan.sel_idx.numel())
            w_rows = (nxt != -100).reshape(-1).to(torch.float32).to(self.device)     # loss_mask of get_batch_logps
        elif self.share_prefix and input_ids.shape[0] == 2 * B:
            plan = build_packed_plan(input_ids, labels, cfg.n_patches, B, cfg.model_max_length, cfg.pad_token_id)
        else:
            plan = build_splice_plan(input_ids, labels, cfg.n_patches, B, cfg.model_max_length)
        plan = plan.to(self.device)
        S, L = plan.S, plan.L
        N = S * L
        feats = self.encode_images(images, ctx if save_for_backward else None)
        x = ops.splice_fwd(plan.src, st.p("model.embed_tokens.weight"), feats, d)
        cos, sin = self._rope(L)
        layers_ctx = []
        for i in range(cfg.layers):
            xn, rstd1 = ops.rmsnorm_fwd(x, st.p(f"layers.{i}.ln1"), cfg.rms_eps)
            qkv = ops.gemm_nt(xn, st.p(f"layers.{i}.wqkv"))
            ops.rope_inplace(qkv, cos, sin, L, 2 * H, hd, pos=plan.pos)
            attn, lse = ops.attn_fwd(qkv, S, L, H, hd, True, 0, d, 2 * d, seg=plan.seg)
            x_mid = ops.gemm_nt(attn, st.p(f"layers.{i}.wo"), residual=x)
            xn2, rstd2 = ops.rmsnorm_fwd(x_mid, st.p(f"layers.{i}.ln2"), cfg.rms_eps)
            gu = ops.gemm_nt(xn2, st.p(f"layers.{i}.wgu"))
            act = ops.swiglu_fwd(gu)
            x_next = ops.gemm_nt(act, st.p(f"layers.{i}.wdown"), residual=x_mid)
            if save_for_backward:
                layers_ctx.append(dict(x=x, rstd1=rstd1, qkv=qkv, attn=attn, lse=lse, x_mid=x_mid, rstd2=rstd2, gu=gu))
            x = x_next
        n_sel = plan.n_sel
        n_pad = max(64, ops.round_up(n_sel, 64))
        hsel = torch.zeros(n_pad, d, dtype=BF16, device=self.device)
        if n_sel > 0:
            _, rstd_f = ops.rmsnorm_fwd(x, st.p("model.norm.weight"), cfg.rms_eps, row_idx=plan.sel_idx, out=hsel[:n_sel])
            logp, lse_v = ops.lmhead_logp_fwd(hsel, st.p("lm_head.weight"), plan.tgt, n_sel)
        else:
            rstd_f = torch.empty(0, dtype=torch.float32, device=self.device)
            logp = torch.empty(0, dtype=torch.float32, device=self.device)
            lse_v = logp
        seq_logp, seq_cnt = ops.seq_sum(logp, plan.seq_off, plan.n_seq, weight=w_rows)
        out = StepOutput(loss=None, scalars=None, per_pair=None, seq_logp=seq_logp, seq_cnt=seq_cnt,
                         per_token_logp=logp, plan=plan)
        if save_for_backward:
            ctx.update(layers=layers_ctx, x_final=x, hsel=hsel, rstd_f=rstd_f, lse_v=lse_v, w_rows=w_rows, N=N, B=B)
            out.ctx = ctx
        return out

    # ------------------------------------------------------------------ backward
    def backward(self, out: StepOutput, coef: torch.Tensor):
        """coef[2B] = d loss / d seq_logp (rv_dpo_loss).  Fills ``store.flat_g`` (overwrites)."""
        cfg, st = self.cfg, self.store
        d, H, hd, f = cfg.hidden, cfg.heads, cfg.head_dim, cfg.ffn
        ctx, plan = out.ctx, out.plan
        S, L, N = plan.S, plan.L, ctx["N"]
        cos, sin = self._rope(L)
        hook = self.grad_ready_hook

        def wgrad(dy: torch.Tensor, xin: torch.Tensor, key: str, rows: Optional[Tuple[int, int]] = None):
            """dW[key] = dy^T @ xin: TN GEMM (operands transposed on the fly by ds_read_b64_tr_b16)."""
            tgt = st.g(key) if rows is None else st.g(key)[rows[0]:rows[1]]
            ops.gemm_tn(dy, xin, out=tgt)

        # ---- LM head + final norm
        n_sel = plan.n_sel
        dx = torch.zeros(N, d, dtype=BF16, device=self.device)
        if n_sel > 0:
            rc = ops.row_coef(coef, plan.seq_of_row, ctx["w_rows"])
            dlog = ops.lmhead_logp_bwd(ctx["hsel"], st.p("lm_head.weight"), plan.tgt, ctx["lse_v"], rc, n_sel)
            dh = ops.gemm_nt(dlog, st.pT("lm_head.weight"))
            wgrad(dlog, ctx["hsel"], "lm_head.weight")
            del dlog
            ops.rmsnorm_bwd(dh[:n_sel], ctx["x_final"], st.p("model.norm.weight"), ctx["rstd_f"],
                            st.g("model.norm.weight"), row_idx=plan.sel_idx, dx=dx)
        else:
            st.g("lm_head.weight").zero_()
            st.g("model.norm.weight").zero_()
        if hook:
            hook("lm_head", *st.buckets["lm_head"])

        # ---- decoder layers, last to first
        for i in reversed(range(cfg.layers)):
            c = ctx["layers"][i]
            act = ops.swiglu_fwd(c["gu"])
            dact = ops.gemm_nt(dx, st.pT(f"layers.{i}.wdown"))
            wgrad(dx, act, f"layers.{i}.wdown")
            del act
            dgu = ops.swiglu_bwd(dact, c["gu"])
            del dact
            xn2, _ = ops.rmsnorm_fwd(c["x_mid"], st.p(f"layers.{i}.ln2"), cfg.rms_eps, want_rstd=False)
            dxn2 = ops.gemm_nt(dgu, st.pT(f"layers.{i}.wgu"))
            wgrad(dgu, xn2, f"layers.{i}.wgu")
            del dgu, xn2
            dx_mid = ops.rmsnorm_bwd(dxn2, c["x_mid"], st.p(f"layers.{i}.ln2"), c["rstd2"], st.g(f"layers.{i}.ln2"),
                                     dres=dx)
            del dxn2
            dattn = ops.gemm_nt(dx_mid, st.pT(f"layers.{i}.wo"))
            wgrad(dx_mid, c["attn"], f"layers.{i}.wo")
            dqkv = ops.attn_bwd(c["qkv"], c["attn"], dattn, c["lse"], S, L, H, hd, True, 0, d, 2 * d, seg=plan.seg)
            del dattn
            ops.rope_inplace(dqkv, cos, sin, L, 2 * H, hd, backward=True, pos=plan.pos)
            xn, _ = ops.rmsnorm_fwd(c["x"], st.p(f"layers.{i}.ln1"), cfg.rms_eps, want_rstd=False)
            dxn = ops.gemm_nt(dqkv, st.pT(f"layers.{i}.wqkv"))
            wgrad(dqkv, xn, f"layers.{i}.wqkv")
            del dqkv, xn
            dx = ops.rmsnorm_bwd(dxn, c["x"], st.p(f"layers.{i}.ln1"), c["rstd1"], st.g(f"layers.{i}.ln1"),
                                 dres=dx_mid)
            del dxn, dx_mid
            ctx["layers"][i] = None          # free this layer's activations
            if hook:
                hook(f"layer{i}", *st.buckets[f"layer{i}"])

        # ---- embedding (deterministic segmented sum) and projector
        ge = st.g("model.embed_tokens.weight")
        ge.zero_()
        ops.embed_bwd(plan.uniq_ids, plan.seg_off, plan.pos_sorted, dx, ge)
        dfeat = ops.feat_grad(plan.feat_src_a, plan.feat_src_b, dx, d)
        ops.colsum(dfeat, out=st.g("model.mm_projector.2.bias"))
        wgrad(dfeat, ctx["h1"], "model.mm_projector.2.weight")
        dh1 = ops.gemm_nt(dfeat, st.pT("model.mm_projector.2.weight"))
        dz1 = ops.gelu_bwd(dh1, ctx["z1"])
        ops.colsum(dz1, out=st.g("model.mm_projector.0.bias"))
        wgrad(dz1, ctx["f_clip"], "model.mm_projector.0.weight")
        if hook:
            hook("embed_proj", *st.buckets["embed_proj"])
            hook("nodecay", *st.buckets["nodecay"])
        out.ctx = {}

    # ------------------------------------------------------------------ reference-style surface
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images):
        """Same 6-tuple contract as llava_arch.py:150-330 (embeds materialised by rv_splice_fwd)."""
        if attention_mask is not None:
            raise NotImplementedError("the DPO path passes attention_mask=None (trainers.py:199)")
        feats = self.encode_images(images)      # one feature block per row of `images`, like the reference
        n_img = images.shape[0]
        plan = build_splice_plan(input_ids, labels, self.cfg.n_patches, n_img, self.cfg.model_max_length).to(self.device)
        emb = ops.splice_fwd(plan.src, self.store.p("model.embed_tokens.weight"), feats, self.cfg.hidden)
        return None, None, None, past_key_values, emb.view(plan.S, plan.L, -1), plan.labels.to(self.device)
