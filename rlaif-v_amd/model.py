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
    kv_heads: Optional[int] = None      # num_key_value_heads (Mistral / Llama-3 style GQA); None = heads (LLaVA-1.5: MHA)

    arch = "llava"                      # class attribute: "llava" (CLIP + mlp2x_gelu projector) | "omnilmm" (omnilmm.py)

    @property
    def vocab_padded(self) -> int:
        """Rows of embed_tokens / lm_head as stored: the tokenizer's vocabulary rounded up to the fused LM head's 64-column
        blocks.  The padding rows are zero, are never indexed, never enter the softmax (rv_lmhead_logp_* V_valid) and get
        zero gradient, so they stay zero under AdamW; HF tensors map to the first ``vocab`` rows."""
        return ops.round_up(self.vocab, 64)

    @property
    def n_image_tokens(self) -> int:
        """Sequence positions one image occupies after the splice."""
        return self.n_patches

    def vision_param_entries(self):
        """(decay weights, no-decay tensors) of the trainable vision-to-language adapter, each (key, shape, transposed copy),
        in backward-completion order.  LLaVA-1.5: the mlp2x_gelu projector (llava/model/multimodal_projector/builder.py)."""
        d, cd = self.hidden, self.clip_hidden
        return ([("model.mm_projector.2.weight", (d, d), True), ("model.mm_projector.0.weight", (d, cd), False)],
                [("model.mm_projector.2.bias", (d,), False), ("model.mm_projector.0.bias", (d,), False)])

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def n_kv_heads(self) -> int:
        return self.heads if self.kv_heads is None else self.kv_heads

    @property
    def kv_dim(self) -> int:
        return self.n_kv_heads * self.head_dim

    @property
    def kv_group(self) -> int:
        return self.heads // self.n_kv_heads

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


@dataclass
class LoraConfig:
    """peft.LoraConfig as built by the reference's LoRA entry point (muffin/train/train_llava15_lora.py:304-318,
    flag defaults :111-116): every nn.Linear of the language model except lm_head (find_all_linear_names, :120-134)."""
    r: int = 64
    lora_alpha: int = 16
    lora_dropout: float = 0.05
    bias: str = "none"
    target_modules: Tuple[str, ...] = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")

    def __post_init__(self):
        if set(self.target_modules) != {"q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"}:
            raise NotImplementedError("LoRA is applied to all seven decoder projections (find_all_linear_names)")
        if self.bias != "none":
            raise NotImplementedError("lora_bias other than 'none' (the reference's default) is not supported")

    @property
    def scaling(self) -> float:
        return self.lora_alpha / self.r

    @property
    def r_pad(self) -> int:
        """Stored rank: r rounded up to the GEMM K step (64); the padding rows/columns are zero and stay zero."""
        return ops.round_up(self.r, 64)


# fused projection -> (store suffix, peft module names in stacking order)
LORA_GROUPS = {"qkv": ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"), "o": ("self_attn.o_proj",),
               "gu": ("mlp.gate_proj", "mlp.up_proj"), "down": ("mlp.down_proj",)}


def lora_group_rows(cfg: "LlavaConfig", grp: str) -> List[Tuple[int, int]]:
    """(first output row, rows) of every peft module inside a fused projection: q | k | v have widths hidden, kv_dim, kv_dim
    (grouped-query attention: kv_dim < hidden), gate | up are ffn each."""
    d, kvd, f = cfg.hidden, cfg.kv_dim, cfg.ffn
    return {"qkv": [(0, d), (d, kvd), (d + kvd, kvd)], "o": [(0, d)], "gu": [(0, f), (f, f)], "down": [(0, d)]}[grp]


def _is_decay(name: str) -> bool:
    return not (name.endswith("bias") or "norm" in name)


class ParamStore:
    """Flat parameter / gradient / optimizer-state buffers with named views."""

    def __init__(self, cfg: LlavaConfig, device, with_optimizer: bool = True, lora: Optional[LoraConfig] = None):
        d, f, V, cd, kvd = cfg.hidden, cfg.ffn, cfg.vocab_padded, cfg.clip_hidden, cfg.kv_dim
        if cfg.heads % cfg.n_kv_heads != 0:
            raise ValueError("kv_heads must divide heads")
        Entry = Tuple[str, Tuple[int, ...], bool]        # (key, shape, needs transposed copy); fused keys map to HF names
        self.lora = lora
        # Full fine-tune: the fused gate|up weight stores its rows INTERLEAVED (row 2j = gate_j, 2j+1 = up_j) so that the GEMM
        # epilogues can apply SwiGLU and its backward in registers (ops.linear_swiglu / linear_swiglu_bwd).  LoRA keeps
        # [gate | up] blocks (its adapter column groups address them).  RV_FUSE_SWIGLU=0: block layout + separate kernels.
        fuse = os.environ.get("RV_FUSE_SWIGLU", "1") != "0"
        # Round 6: adapter models can take the fused SwiGLU epilogues too (RV_LORA_FUSE_SWIGLU=1): the frozen gate|up weight AND the
        # rows of the gate|up adapter lora_B are stored interleaved, and the fused GEMM's adapter segment runs over BOTH modules'
        # t = x A^T columns against an EXPANDED lora_B^T whose zeros keep t_gate out of the up columns (``gu_bexp``).
        self.lora_il = lora is not None and fuse and os.environ.get("RV_LORA_FUSE_SWIGLU", "0") == "1" \
            and os.environ.get("RV_LORA_PEFT_MASKS", "0") == "0"      # (independent gate / up masks need two adapter inputs)
        self.interleave_gu = fuse and (lora is None or self.lora_il)
        self.gu_bexp: Dict[int, torch.Tensor] = {}      # layer -> [2 r_pad, 2 ffn] expanded transposed gate|up adapter (lora_il only)
        proj_w, proj_b = cfg.vision_param_entries()
        self.vision_keys = [k for k, _, _ in proj_w + proj_b]
        base_w: List[Entry] = [("lm_head.weight", (V, d), True)]
        for i in reversed(range(cfg.layers)):
            base_w += [(f"layers.{i}.wdown", (d, f), True), (f"layers.{i}.wgu", (2 * f, d), True),
                       (f"layers.{i}.wo", (d, d), True), (f"layers.{i}.wqkv", (d + 2 * kvd, d), True)]
        base_w += [("model.embed_tokens.weight", (V, d), False)]
        norms: List[Entry] = [("model.norm.weight", (d,), False)]
        for i in reversed(range(cfg.layers)):
            norms += [(f"layers.{i}.ln2", (d,), False), (f"layers.{i}.ln1", (d,), False)]
        if lora is None:
            # full fine-tune: everything trainable, laid out in backward-completion order
            frozen: List[Entry] = []
            decay = base_w + proj_w
            nodecay = norms + proj_b
        else:
            # LoRA: base language model frozen; adapters (backward-completion order) + projector trainable
            rp = lora.r_pad
            frozen = base_w + norms
            decay = []
            for i in reversed(range(cfg.layers)):
                decay += [(f"layers.{i}.lora_down.B", (d, rp), True), (f"layers.{i}.lora_down.A", (rp, f), True),
                          (f"layers.{i}.lora_gu.B", (2 * f, rp), True), (f"layers.{i}.lora_gu.A", (2 * rp, d), True),
                          (f"layers.{i}.lora_o.B", (d, rp), True), (f"layers.{i}.lora_o.A", (rp, d), True),
                          (f"layers.{i}.lora_qkv.B", (d + 2 * kvd, rp), True), (f"layers.{i}.lora_qkv.A", (3 * rp, d), True)]
            decay += proj_w
            nodecay = proj_b
        self.entries = frozen + decay + nodecay
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for k, shp, _ in self.entries:
            n = math.prod(shp)
            assert n % 8 == 0, (k, shp)
            self.offsets[k] = (off, shp)
            off += n
        self.n_total = off
        self.t0 = sum(math.prod(shp) for _, shp, _ in frozen)            # first trainable element of flat_p
        self.n_decay = sum(math.prod(shp) for _, shp, _ in decay)        # trainable elements with weight decay
        self.n_train = self.n_total - self.t0
        self.trainable = {k for k, _, _ in decay + nodecay}
        self.device = device
        self.flat_p = torch.zeros(self.n_total, dtype=BF16, device=device)
        self.train_p = self.flat_p[self.t0:]                             # the slice the optimizer owns
        self.flat_g = torch.zeros(self.n_train, dtype=BF16, device=device)
        if with_optimizer:
            self.flat_master = torch.zeros(self.n_train, dtype=torch.float32, device=device)
            self.flat_m = torch.zeros(self.n_train, dtype=torch.float32, device=device)
            self.flat_v = torch.zeros(self.n_train, dtype=torch.float32, device=device)
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
        # gradient buckets for data parallelism: contiguous slices of flat_g in the order backward finishes them
        self.buckets: Dict[str, Tuple[int, int]] = {}
        if lora is None:
            self.buckets["lm_head"] = self.grad_range("lm_head.weight", "lm_head.weight")
            for i in reversed(range(cfg.layers)):
                self.buckets[f"layer{i}"] = self.grad_range(f"layers.{i}.wdown", f"layers.{i}.wqkv")
            self.buckets["embed_proj"] = self.grad_range("model.embed_tokens.weight", proj_w[-1][0])
        else:
            for i in reversed(range(cfg.layers)):
                self.buckets[f"layer{i}"] = self.grad_range(f"layers.{i}.lora_down.B", f"layers.{i}.lora_qkv.A")
            self.buckets["embed_proj"] = self.grad_range(proj_w[0][0], proj_w[-1][0])
        self.buckets["nodecay"] = (self.n_decay, self.n_train)

    def bucket_schedule(self) -> List[Tuple[str, int, int]]:
        """(name, start, end) in the order LlavaDPOModel.backward fires grad_ready_hook; covers flat_g exactly."""
        return [(k, a, b) for k, (a, b) in self.buckets.items()]

    def p(self, key: str) -> torch.Tensor:
        off, shp = self.offsets[key]
        return self.flat_p[off:off + math.prod(shp)].view(*shp)

    def g(self, key: str) -> torch.Tensor:
        off, shp = self.offsets[key]
        if key not in self.trainable:
            raise KeyError(f"{key} is frozen: it has no gradient slot")
        off -= self.t0
        return self.flat_g[off:off + math.prod(shp)].view(*shp)

    def pT(self, key: str) -> torch.Tensor:
        off, shp = self.t_offsets[key]
        return self.flat_pT[off:off + shp[0] * shp[1]].view(*shp)

    def grad_range(self, first_key: str, last_key: str) -> Tuple[int, int]:
        """[start, end) inside flat_g (offsets relative to the first trainable element)."""
        a = self.offsets[first_key][0]
        off, shp = self.offsets[last_key]
        return a - self.t0, off + math.prod(shp) - self.t0

    def refresh_transposes(self, trainable_only: bool = False):
        """W^T copies; after an optimizer step only the trainable ones moved (LoRA: 160 MB instead of 13 GB)."""
        for k in self.t_offsets:
            if not trainable_only or k in self.trainable:
                ops.transpose(self.p(k), out=self.pT(k))
        if self.lora_il:
            self.refresh_gu_bexp()

    def refresh_gu_bexp(self):
        """lora_il: the expanded transposed gate|up adapter of every layer from its (row-interleaved) lora_B: rows 0..rp-1 =
        lora_B(gate)^T on the even columns, rows rp..2rp-1 = lora_B(up)^T on the odd columns, zeros elsewhere (plumbing: 64
        strided device copies of 2.8 MB per optimizer step)."""
        rp = self.lora.r_pad
        for key in self.t_offsets:
            if not key.endswith("lora_gu.B"):
                continue
            i = int(key.split(".")[1])
            BT = self.pT(key)                                   # [rp, 2f]: column 2j = lora_B(gate)[j], 2j + 1 = lora_B(up)[j]
            e = self.gu_bexp.get(i)
            if e is None:
                e = self.gu_bexp[i] = torch.zeros(2 * rp, BT.shape[1], dtype=BT.dtype, device=BT.device)
            e[:rp, 0::2].copy_(BT[:, 0::2])
            e[rp:, 1::2].copy_(BT[:, 1::2])

    def sync_master_from_params(self):
        if self.flat_master is not None:
            self.flat_master.copy_(self.train_p)     # bf16 -> fp32 (device copy, plumbing)
        if getattr(self, "sharded_optimizer", None) is not None:
            self.sharded_optimizer.sync_master_from_params()      # opt-in ZeRO-1: the fp32 masters live in the trainer's shards

    # ---- HF state-dict mapping ------------------------------------------------------------
    def hf_slices(self, cfg: LlavaConfig) -> Dict[str, Tuple[str, int, int, int]]:
        """HF name -> (store key, first row, n rows, row step) for the language model + projector: the HF tensor is
        ``rows(view(key), r0, n, step)`` (step 2 = the interleaved gate / up rows of the fused MLP weight)."""
        d, f, kvd = cfg.hidden, cfg.ffn, cfg.kv_dim
        m: Dict[str, Tuple[str, int, int, int]] = {}
        il = self.interleave_gu
        for i in range(cfg.layers):
            p = f"model.layers.{i}."
            m[p + "self_attn.q_proj.weight"] = (f"layers.{i}.wqkv", 0, d, 1)
            m[p + "self_attn.k_proj.weight"] = (f"layers.{i}.wqkv", d, kvd, 1)
            m[p + "self_attn.v_proj.weight"] = (f"layers.{i}.wqkv", d + kvd, kvd, 1)
            m[p + "self_attn.o_proj.weight"] = (f"layers.{i}.wo", 0, d, 1)
            m[p + "mlp.gate_proj.weight"] = (f"layers.{i}.wgu", 0, f, 2) if il else (f"layers.{i}.wgu", 0, f, 1)
            m[p + "mlp.up_proj.weight"] = (f"layers.{i}.wgu", 1, f, 2) if il else (f"layers.{i}.wgu", f, f, 1)
            m[p + "mlp.down_proj.weight"] = (f"layers.{i}.wdown", 0, d, 1)
            m[p + "input_layernorm.weight"] = (f"layers.{i}.ln1", 0, d, 1)
            m[p + "post_attention_layernorm.weight"] = (f"layers.{i}.ln2", 0, d, 1)
        for k in ["lm_head.weight", "model.embed_tokens.weight", "model.norm.weight"] + self.vision_keys:
            m[k] = (k, 0, cfg.vocab if k in ("lm_head.weight", "model.embed_tokens.weight") else self.offsets[k][1][0], 1)
        return m

    @staticmethod
    def rows(view: torch.Tensor, r0: int, n: int, step: int = 1) -> torch.Tensor:
        """Rows r0, r0 + step, ... (n of them) of a store view."""
        return view[r0:r0 + n * step:step]


    @staticmethod
    def lora_view(t: torch.Tensor, r0: int, n: int, ncol: int, step: int = 1) -> torch.Tensor:
        """The peft tensor inside a store view: rows r0, r0 + step, ... (n of them), first ncol columns."""
        return t[r0:r0 + n * step:step, :ncol]

    def lora_slices(self, cfg: LlavaConfig) -> Dict[str, Tuple[str, int, int, int, int]]:
        """peft adapter name (without the 'base_model.model.' prefix) -> (store key, first row, n rows, n cols, row step); the
        tensor is ``lora_view(view(key), r0, n, ncol, step)`` (step 2 = the interleaved gate / up rows of lora_il's lora_gu.B)."""
        m: Dict[str, Tuple[str, int, int, int, int]] = {}
        if self.lora is None:
            return m
        r, rp, d, f = self.lora.r, self.lora.r_pad, cfg.hidden, cfg.ffn
        in_cols = {"qkv": d, "o": d, "gu": d, "down": f}
        for i in range(cfg.layers):
            for grp, mods in LORA_GROUPS.items():
                for gi, (mod, (r0, rows)) in enumerate(zip(mods, lora_group_rows(cfg, grp))):
                    p = f"model.layers.{i}.{mod}."
                    m[p + "lora_A.weight"] = (f"layers.{i}.lora_{grp}.A", gi * rp, r, in_cols[grp], 1)
                    if grp == "gu" and self.lora_il:
                        m[p + "lora_B.weight"] = (f"layers.{i}.lora_{grp}.B", gi, rows, r, 2)
                    else:
                        m[p + "lora_B.weight"] = (f"layers.{i}.lora_{grp}.B", r0, rows, r, 1)
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

    def __init__(self, cfg: LlavaConfig, device="cuda:0", with_optimizer: bool = True,
                 lora: Optional[LoraConfig] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("LlavaDPOModel needs an MI355X (HIP) device; there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device(device)
        self.lora = lora
        self.store = ParamStore(cfg, self.device, with_optimizer, lora)
        self._dropout_step = 0          # advances once per forward: seeds the LoRA dropout masks
        # keep the dropped adapter inputs for backward (+1.3 GB per layer at 27 k tokens) instead of regenerating them
        self.keep_dropped_inputs = os.environ.get("RV_LORA_KEEP_DROPPED", "1") != "0"
        # keep normalised inputs and SwiGLU outputs for backward instead of recomputing them (RV_KEEP_RECOMPUTABLE=0:
        # the lean layout, 1.05 GB / layer less at 27 k tokens)
        self.keep_recomputable = os.environ.get("RV_KEEP_RECOMPUTABLE", "1") != "0"
        # --gradient_checkpointing (script/train/llava15_train.sh:39): keep only each decoder layer's input and re-run the
        # layer in backward.  Off by default: 288 GB holds 8 pairs x 2048 tokens without it.
        self.gradient_checkpointing = False
        self.clip: Dict[str, torch.Tensor] = {}
        self.training = True
        self._rope_cache: Optional[Tuple[int, torch.Tensor, torch.Tensor]] = None     # ONE table, grown geometrically
        self.dropout_rank = 0           # data-parallel rank: mixed into the LoRA dropout seeds (set by the trainer)
        self.grad_ready_hook = None     # callable(name, start, end) fired when a slice of flat_g is final
        # compute the prefix shared by the chosen and rejected sequence of a pair once (splice.build_packed_plan)
        self.share_prefix = os.environ.get("RV_SHARE_PREFIX", "1") != "0"
        # packed rows are concatenated without inter-row padding (splice.build_packed_plan pad_free; RV_PAD_FREE=0: every packed
        # row right-padded to the longest, the round-1..4 layout).  Log-probs are bit-identical either way (SURVEY 8a property (i))
        self.pad_free = os.environ.get("RV_PAD_FREE", "1") != "0"
        # The DECODER's residual stream in fp32 - DEFAULT for the full fine-tune since round 6 (RV_RESID_FP32=0 restores the bf16
        # stream; LoRA runs keep bf16: their fixtures were validated that way and the producer-side dropout kernels write bf16).
        # o_proj / down_proj write their branch in bf16 without the residual operand; rv_rmsnorm_fwd_f32 adds it to the fp32 stream
        # and normalises in one pass.  Decided on round 6's evidence (profiles/r06_cfg1_step_numerics.json): with
        # SwiGLU and RoPE taken from fp32 accumulators the stream is what is left - per-token RMS error 0.035 -> 0.027 at 32 layers
        # (the HF-style bf16 emulation: 0.053), and BASELINE config 1's literal batch - whose 1e-3 loss bar is a fraction of a sigma
        # of ANY bf16 forward - lands at 4.2e-4 with it and at 2.6e-3 ... 3.9e-3 without.  Price: +1.1 % step time, +14 GB.
        env = os.environ.get("RV_RESID_FP32")
        self.resid_fp32 = (env != "0") if env not in (None, "") else (lora is None)
        # OPT-IN: peft's dropout semantics to the letter (RV_LORA_PEFT_MASKS=1): peft wraps every nn.Linear in its own lora.Linear with
        # its OWN nn.Dropout, so q / k / v (and gate / up) draw INDEPENDENT masks of the same input.  The default fuses them: one
        # dropped input per fused projection (same marginals, one pass over the activation instead of three).  See _module_seeds.
        self.lora_peft_masks = os.environ.get("RV_LORA_PEFT_MASKS", "0") != "0"
        self.clip_fp32_resid = os.environ.get("RV_CLIP_FP32_RESID", "1") != "0"      # default ON since round 5, see clip_features
        self.fuse_rope_bwd = os.environ.get("RV_FUSE_ROPE_BWD", "1") != "0"
        self.fuse_rope_fwd = os.environ.get("RV_FUSE_ROPE_FWD", "1") != "0"      # RoPE in the q|k|v GEMM epilogue (ops.linear_rope)

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """HF-named fp32/bf16 CPU tensors (the reference's checkpoint layout, 4.35 CLIP key names)."""
        cfg, st = self.cfg, self.store
        for name, (key, r0, n, step) in st.hf_slices(cfg).items():
            st.rows(st.p(key), r0, n, step).copy_(sd[name].to(BF16))
        if self.lora is not None:
            self.load_lora_state_dict(sd, strict=False, _refresh=False)
        st.sync_master_from_params()
        st.refresh_transposes()
        self._load_tower(sd)

    def _load_tower(self, sd: Dict[str, torch.Tensor]):
        """The frozen CLIP-ViT-L/14 tower (llava/model/multimodal_encoder/clip_encoder.py) into device tensors."""
        cfg = self.cfg
        self._clip_raw = {k: v.detach().to(BF16).cpu() for k, v in sd.items() if k.startswith(VT)}
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

    def init_random(self, seed: int = 0, std: float = 0.02, lora_b_std: Optional[float] = None):
        """HF-default style random init directly on the device (no checkpoints exist offline)."""
        cfg, st = self.cfg, self.store
        g = torch.Generator(device=self.device).manual_seed(seed)
        chunk = 1 << 26
        for k, shp, _ in st.entries:
            off, n = st.offsets[k][0], math.prod(shp)
            if ".lora_" in k or k.endswith("bias"):
                continue                                        # adapters: reset_lora_parameters below; biases stay 0
            if len(shp) == 1:
                st.flat_p[off:off + n] = 1.0                    # norm gains
                continue
            for a in range(off, off + n, chunk):
                b = min(off + n, a + chunk)
                st.flat_p[a:b] = (torch.randn(b - a, device=self.device, generator=g) * std).to(BF16)
        if self.lora is not None:
            self.reset_lora_parameters(seed + 1, lora_b_std)
        if cfg.vocab_padded != cfg.vocab:                       # vocabulary padding rows are zero (LlavaConfig.vocab_padded)
            st.p("lm_head.weight")[cfg.vocab:].zero_()
            st.p("model.embed_tokens.weight")[cfg.vocab:].zero_()
        st.sync_master_from_params()
        st.refresh_transposes()
        self._init_tower(g, std)

    def _init_tower(self, g: torch.Generator, std: float):
        cfg = self.cfg
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
        for name, (key, r0, n, step) in self.store.hf_slices(self.cfg).items():
            out[name] = self.store.rows(self.store.p(key), r0, n, step).detach().cpu().clone()
        return out

    # ---- LoRA adapters (peft naming, muffin/train/train_llava15_lora.py:152-197) ----------------
    def reset_lora_parameters(self, seed: int = 1, b_std: Optional[float] = None):
        """peft LoraLayer.reset_lora_parameters: lora_A ~ kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in)),
        lora_B = 0 (b_std: draw B ~ N(0, b_std) instead - tests and benchmarks that want a non-trivial adapter)."""
        st, g = self.store, torch.Generator(device=self.device).manual_seed(seed)
        for name, (key, r0, n, ncol, step) in st.lora_slices(self.cfg).items():
            view = st.lora_view(st.p(key), r0, n, ncol, step)
            if name.endswith("lora_A.weight"):
                bound = 1.0 / math.sqrt(ncol)
                view.copy_(((torch.rand(n, ncol, device=self.device, generator=g) * 2 - 1) * bound).to(BF16))
            elif b_std is None:
                view.zero_()
            else:
                view.copy_((torch.randn(n, ncol, device=self.device, generator=g) * b_std).to(BF16))

    def load_lora_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True, _refresh: bool = True):
        """Adapter tensors under peft names, with or without the 'base_model.model.' prefix / '.default' infix."""
        st = self.store
        norm = {k.replace("base_model.model.", "").replace(".default.", "."): v for k, v in sd.items() if ".lora_" in k}
        found = 0
        for name, (key, r0, n, ncol, step) in st.lora_slices(self.cfg).items():
            if name in norm:
                st.lora_view(st.p(key), r0, n, ncol, step).copy_(norm[name].to(BF16))
                found += 1
            elif strict:
                raise KeyError(f"adapter tensor {name} missing")
        if found == 0 and not strict:
            self.reset_lora_parameters()
        if _refresh:
            st.sync_master_from_params()
            st.refresh_transposes(trainable_only=True)

    def lora_state_dict(self, grads: bool = False) -> Dict[str, torch.Tensor]:
        """What get_peft_state_maybe_zero_3(named_parameters(), 'none') collects: 'base_model.model.<module>.lora_X.weight'."""
        st, out = self.store, {}
        for name, (key, r0, n, ncol, step) in st.lora_slices(self.cfg).items():
            src = st.g(key) if grads else st.p(key)
            t = st.lora_view(src, r0, n, ncol, step).detach()
            out["base_model.model." + name] = t.float().cpu() if grads else t.cpu().clone()
        return out

    def non_lora_trainables(self) -> Dict[str, torch.Tensor]:
        """get_peft_state_non_lora_maybe_zero_3: the trainable non-adapter tensors = the projector."""
        return {"base_model.model." + k: self.store.p(k).detach().cpu().clone() for k in self.store.trainable
                if "mm_projector" in k}

    def merge_lora(self):
        """peft merge_and_unload (llava/model/builder.py:81-85): W += (alpha/r) B A for every adapted projection, on
        the device; the adapters are zeroed afterwards (B = 0 makes them the identity)."""
        if self.lora is None:
            return
        st, cfg, rp, sc = self.store, self.cfg, self.lora.r_pad, self.lora.scaling
        for i in range(cfg.layers):
            for grp in LORA_GROUPS:
                W, B, AT = st.p(f"layers.{i}.w{grp}"), st.p(f"layers.{i}.lora_{grp}.B"), st.pT(f"layers.{i}.lora_{grp}.A")
                if grp == "gu" and st.lora_il:
                    # interleaved rows: W[2j + gi] += sc B[2j + gi] A_gi - one GEMM per module on contiguous copies (a one-off)
                    for gi in range(2):
                        Wg = W[gi::2].contiguous()
                        ops.gemm_nt(B[gi::2].contiguous(), AT[:, gi * rp:(gi + 1) * rp].contiguous(), out=Wg, residual=Wg, alpha=sc)
                        W[gi::2].copy_(Wg)
                    B.zero_()
                    continue
                for gi, (r0, rows) in enumerate(lora_group_rows(cfg, grp)):
                    Wg = W[r0:r0 + rows]
                    ops.gemm_nt(B[r0:r0 + rows], AT[:, gi * rp:(gi + 1) * rp], out=Wg, residual=Wg, alpha=sc)
                B.zero_()
        st.sync_master_from_params()
        st.refresh_transposes()

    def clip_state_dict(self) -> Dict[str, torch.Tensor]:
        """The frozen tower under its checkpoint key names (kept from load_state_dict; empty after init_random)."""
        return dict(getattr(self, "_clip_raw", {}))

    def grads_state_dict(self) -> Dict[str, torch.Tensor]:
        out = {}
        for name, (key, r0, n, step) in self.store.hf_slices(self.cfg).items():
            if key in self.store.trainable:
                out[name] = self.store.rows(self.store.g(key), r0, n, step).detach().float().cpu()
        for name, t in self.lora_state_dict(grads=True).items():
            out[name.replace("base_model.model.", "")] = t
        return out

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    # ------------------------------------------------------------------ vision
    def _rope(self, L: int):
        """cos/sin rows 0..L-1.  The kernels index the table by position, so one table for the largest length seen so far
        serves every batch (a table per distinct L would grow without bound with ragged / packed batches)."""
        if self._rope_cache is None or self._rope_cache[0] < L:
            n = max(L, self.cfg.model_max_length, 2 * self._rope_cache[0] if self._rope_cache else 0)
            self._rope_cache = (n,) + ops.rope_tables(n, self.cfg.head_dim, self.cfg.rope_theta, self.device)
        return self._rope_cache[1], self._rope_cache[2]

    def clip_features(self, pixels: torch.Tensor) -> torch.Tensor:
        """CLIPVisionTower.forward + feature_select('patch') (clip_encoder.py:36-58): [B*P, clip_hidden]."""
        cfg, c = self.cfg, self.clip
        if isinstance(pixels, (list, tuple)) or pixels.dtype == torch.uint8:
            # raw decoded images (uint8 HWC, any sizes): CLIPImageProcessor's resize / crop / normalise on the device
            from .image import clip_preprocess_batch
            pixels = clip_preprocess_batch(pixels, size=cfg.image_size, device=self.device)
        B = pixels.shape[0]
        P, T, cd, H, hd = cfg.n_patches, cfg.n_patches + 1, cfg.clip_hidden, cfg.clip_heads, cfg.clip_head_dim
        px = pixels.to(self.device, dtype=torch.float32).contiguous()
        cols = ops.im2col_patches(px, cfg.patch, cfg.patch_k)
        pe = ops.gemm_nt(cols, c["patch_w"])
        x = ops.clip_assemble(pe, c["cls"], c["pos"], B, P)
        x = ops.layernorm_fwd(x, c["pre_ln_w"], c["pre_ln_b"], cfg.clip_eps)
        if self.clip_fp32_resid:
            # The frozen tower's RESIDUAL STREAM in fp32 (default since round 5; RV_CLIP_FP32_RESID=0 = the bf16 stream): the 47 bf16 roundings of x on the way
            # through 23 layers are 84 % of the vision front's share - 27 % of the whole - of the per-token log-prob error variance
            # (profiles/r05_rounding_attribution.json), and the tower is 0.6 % of the step.  Every MFMA operand stays
            # bf16 (LayerNorm outputs, q / k / v, attention output, quick_gelu(fc1)); only out_proj / fc2 accumulate into an fp32 x.
            x32 = ops.cast_bf16_to_f32(x)
            for i in range(cfg.clip_layers_used):
                h = ops.layernorm_fwd_f32in(x32, c[f"{i}.layer_norm1.w"], c[f"{i}.layer_norm1.b"], cfg.clip_eps)
                qkv = ops.gemm_nt(h, c[f"{i}.wqkv"], bias=c[f"{i}.bqkv"])
                a, _ = ops.attn_fwd(qkv, B, T, H, hd, False, 0, cd, 2 * cd)
                ops.gemm_nt_f32res(a, c[f"{i}.wo"], c[f"{i}.bo"], x32)
                h = ops.layernorm_fwd_f32in(x32, c[f"{i}.layer_norm2.w"], c[f"{i}.layer_norm2.b"], cfg.clip_eps)
                h = ops.gemm_nt(h, c[f"{i}.fc1.w"], bias=c[f"{i}.fc1.b"], act=ops.ACT_QUICK_GELU)
                ops.gemm_nt_f32res(h, c[f"{i}.fc2.w"], c[f"{i}.fc2.b"], x32)
            x = ops.cast_f32_to_bf16(x32)            # hidden_states[-2] as the projector's bf16 GEMM operand: one rounding
            idx = (torch.arange(B, device=self.device)[:, None] * T + 1 + torch.arange(P, device=self.device)[None]).reshape(-1)
            return ops.gather_rows(x, idx.to(torch.int32))
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

    def _vision_backward(self, dfeat: torch.Tensor, ctx: dict):
        """Autograd of the mlp2x_gelu projector (the CLIP tower is frozen: llava_arch.py:141-148, train_llava15.py:268)."""
        st = self.store
        ops.colsum(dfeat, out=st.g("model.mm_projector.2.bias"))
        ops.gemm_tn(dfeat, ctx["h1"], out=st.g("model.mm_projector.2.weight"))
        dh1 = ops.gemm_nt(dfeat, st.pT("model.mm_projector.2.weight"))
        dz1 = ops.gelu_bwd(dh1, ctx["z1"])
        ops.colsum(dz1, out=st.g("model.mm_projector.0.bias"))
        ops.gemm_tn(dz1, ctx["f_clip"], out=st.g("model.mm_projector.0.weight"))

    def _row_splicer(self):
        """Row-level splice rule handed to the planners (None = LLaVA's expanding <image> splice, splice._splice_rows)."""
        return None

    # ------------------------------------------------------------------ decoder projections (+ LoRA)
    def _lora_grouping(self, grp: str) -> Tuple[int, int]:
        """(group_cols, group0) of the fused LoRA GEMM for a fused projection (rv_gemm_nt_lora_bf16)."""
        cfg = self.cfg
        if grp == "qkv":
            return cfg.kv_dim, (cfg.hidden if cfg.kv_dim != cfg.hidden else 0)
        return (cfg.ffn, 0) if grp == "gu" else (0, 0)

    def _proj_fwd(self, x: torch.Tensor, i: int, grp: str, residual: Optional[torch.Tensor] = None,
                  drop_slot: int = 0, xd: Optional[torch.Tensor] = None):
        """y = x W^T (+ residual); with adapters  y = x W^T + t B^T,  t = (alpha/r) dropout(x) A^T  (peft
        lora.Linear.forward) - the adapter term rides in the same K loop (rv_gemm_nt_lora_bf16).  Returns (y, t, xd).
        ``xd``: dropout(x) with this slot's seed when the kernel that produced x wrote it already (_lora_drop)."""
        st = self.store
        W = st.p(f"layers.{i}.w{grp}")
        if self.lora is None:
            return ops.linear(x, W, st.pT(f"layers.{i}.w{grp}"), residual=residual), None, None
        if self.lora_peft_masks and self._lora_drop():
            # one independent mask per peft module: t_g = (alpha / r) dropout_g(x) A_g^T, a skinny GEMM per module (the dropped
            # copies are regenerated from their seeds in backward)
            rp, A = self.lora.r_pad, st.p(f"layers.{i}.lora_{grp}.A")
            seeds = self._module_seeds(i, grp, drop_slot)
            t = torch.empty(x.shape[0], len(seeds) * rp, dtype=BF16, device=self.device)
            for g, seed in enumerate(seeds):
                xg = ops.dropout(x, self.lora.lora_dropout, seed)
                ops.gemm_nt(xg, A[g * rp:(g + 1) * rp], out=t[:, g * rp:(g + 1) * rp], alpha=self.lora.scaling)
                del xg
            gc, g0 = self._lora_grouping(grp)
            y = ops.linear_lora(x, W, st.pT(f"layers.{i}.w{grp}"), t, st.p(f"layers.{i}.lora_{grp}.B"),
                                st.pT(f"layers.{i}.lora_{grp}.B"), group_cols=gc, residual=residual, group0=g0)
            return y, t, None
        if xd is None and self._lora_drop():
            xd = ops.dropout(x, self.lora.lora_dropout, self._dropout_seed(i, drop_slot))
        t = ops.gemm_nt(x if xd is None else xd, st.p(f"layers.{i}.lora_{grp}.A"), alpha=self.lora.scaling)
        gc, g0 = self._lora_grouping(grp)
        y = ops.linear_lora(x, W, st.pT(f"layers.{i}.w{grp}"), t, st.p(f"layers.{i}.lora_{grp}.B"),
                            st.pT(f"layers.{i}.lora_{grp}.B"), group_cols=gc, residual=residual, group0=g0)
        return y, t, (xd if self.keep_dropped_inputs else None)

    def _lora_drop(self) -> bool:
        """The adapter branch drops its input (peft lora_dropout, training mode only)."""
        return self.lora is not None and self.training and self.lora.lora_dropout > 0.0

    def _dropout_seed(self, layer: int, slot: int) -> int:
        return (self._cur_drop_step * 1000003 + self.dropout_rank * 7919 + layer * 8 + slot) & 0x7FFFFFFF

    def _module_seeds(self, layer: int, grp: str, slot: int) -> List[int]:
        """RV_LORA_PEFT_MASKS: the seed of every peft module of a fused projection.  The first module keeps the fused slot's seed
        (q: 0, o: 1, gate: 2, down: 3), the others take the layer's spare slots (k: 4, v: 5, up: 6)."""
        extra = {"qkv": (4, 5), "gu": (6,)}.get(grp, ())
        return [self._dropout_seed(layer, sl) for sl in (slot,) + extra]

    def _proj_bwd(self, dy: torch.Tensor, xin: torch.Tensor, t: Optional[torch.Tensor], i: int, grp: str,
                  drop_slot: int = 0, xd: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Input gradient of _proj_fwd; writes the weight gradients that exist (full fine-tune: dW = dy^T x by the TN
        GEMM; LoRA: dA = dt^T x_d, dB = dy^T t by the split-K TN GEMM, base weight frozen)."""
        st = self.store
        wkey = f"layers.{i}.w{grp}"
        if self.lora is None:
            dx = ops.linear(dy, st.pT(wkey), st.p(wkey))        # dx = dy @ W: "weight" = W^T, its transpose = W itself
            ops.gemm_tn(dy, xin, out=st.g(wkey))
            return dx
        rp, sc = self.lora.r_pad, self.lora.scaling
        akey, bkey = f"layers.{i}.lora_{grp}.A", f"layers.{i}.lora_{grp}.B"
        groups = lora_group_rows(self.cfg, grp)                       # (first output column, width) per peft module
        G = len(groups)
        BT = st.pT(bkey)                                              # [rp, total output width]
        il = grp == "gu" and st.lora_il                               # interleaved gate / up columns: the modules are column PARITIES
        if il:
            if self.training and self.lora.lora_dropout > 0.0 and self.lora_peft_masks:
                raise NotImplementedError("RV_LORA_FUSE_SWIGLU with RV_LORA_PEFT_MASKS")
            # dt = (alpha/r) dy bexp^T: the expanded adapter's zeros keep the gate columns of dy out of dt_up and vice versa
            dt = ops.gemm_nt(dy, st.gu_bexp[i], alpha=sc)
        else:
            dt = torch.empty(dy.shape[0], G * rp, dtype=BF16, device=self.device)
            for g, (c0, og) in enumerate(groups):                     # dt_g = (alpha/r) dy_g B_g
                ops.gemm_nt(dy[:, c0:c0 + og], BT[:, c0:c0 + og], out=dt[:, g * rp:(g + 1) * rp], alpha=sc)
        if self.training and self.lora.lora_dropout > 0.0 and self.lora_peft_masks:
            # per-module masks: dx = dy W + sum_g mask_g * (dt_g A_g) / (1 - p); dA_g = dt_g^T dropout_g(x)
            p_, A, AT, gA = self.lora.lora_dropout, st.p(akey), st.pT(akey), st.g(akey)
            dx = ops.linear(dy, st.pT(wkey), st.p(wkey))
            for g, seed in enumerate(self._module_seeds(i, grp, drop_slot)):
                dtg = dt[:, g * rp:(g + 1) * rp]
                ops.gemm_nt_dropout(dtg, AT[:, g * rp:(g + 1) * rp], p_, seed, out=dx, residual=dx)
                xg = ops.dropout(xin, p_, seed)
                ops.gemm_tn_skinny(dtg, xg, out=gA[g * rp:(g + 1) * rp])
                del xg
            gB = st.g(bkey)
            for g, (c0, og) in enumerate(groups):
                ops.gemm_tn_skinny(dy[:, c0:c0 + og], t[:, g * rp:(g + 1) * rp], out=gB[c0:c0 + og])
            return dx
        if self.training and self.lora.lora_dropout > 0.0:
            # dropout sits on the adapter branch only: dx = dy W + mask * (dt A) / (1 - p)
            dx = ops.lora_dgrad_dropout(dy, st.p(wkey), st.pT(wkey), dt, st.p(akey), st.pT(akey), self.lora.lora_dropout,
                                        self._dropout_seed(i, drop_slot))
            # the dropped adapter input: kept from forward (288 GB HBM) or regenerated from the seed
            xin = xd if xd is not None else ops.dropout(xin, self.lora.lora_dropout, self._dropout_seed(i, drop_slot))
        else:
            dx = ops.linear_lora(dy, st.pT(wkey), st.p(wkey), dt, st.pT(akey), st.p(akey), group_cols=0)
        ops.gemm_tn_skinny(dt, xin, out=st.g(akey))
        gB = st.g(bkey)
        if il:
            # dB of both modules from ONE product dy^T [t_gate | t_up] -> [2f, 2 rp]; row 2j keeps its first rp columns (gate),
            # row 2j + 1 its last rp (up): the other halves are the cross terms the expanded adapter's zeros stand for
            full = ops.gemm_tn_skinny(dy, t)
            gB[0::2].copy_(full[0::2, :rp])
            gB[1::2].copy_(full[1::2, rp:])
            return dx
        for g, (c0, og) in enumerate(groups):
            ops.gemm_tn_skinny(dy[:, c0:c0 + og], t[:, g * rp:(g + 1) * rp], out=gB[c0:c0 + og])
        return dx

    # ---- adapter models with interleaved gate|up layout (ParamStore.lora_il; RV_LORA_FUSE_SWIGLU=1, round 6) -----------------------
    def _lora_gu_fwd(self, xn2: torch.Tensor, i: int, xn2d: Optional[torch.Tensor], p_drop: float):
        """(gu, act, actd, t, xd) of the gate|up projection + SwiGLU of an adapter model: the fused-LoRA GEMM with the SwiGLU
        epilogue (and the dropped activation for the down projection's adapter written by the same epilogue) at chip-filling
        shapes, the unfused composition on the same layout otherwise."""
        st, cfg = self.store, self.cfg
        if self._lora_drop() and self.lora_peft_masks:
            raise NotImplementedError("RV_LORA_FUSE_SWIGLU with RV_LORA_PEFT_MASKS")
        xd = xn2d
        if xd is None and self._lora_drop():
            xd = ops.dropout(xn2, self.lora.lora_dropout, self._dropout_seed(i, 2))
        t = ops.gemm_nt(xn2 if xd is None else xd, st.p(f"layers.{i}.lora_gu.A"), alpha=self.lora.scaling)
        keep_xd = xd if self.keep_dropped_inputs else None
        p3, seed3 = (self.lora.lora_dropout, self._dropout_seed(i, 3)) if self._lora_drop() else (0.0, 0)
        bexp = st.gu_bexp[i]
        if ops.linear_lora_swiglu_ok(xn2.shape[0], cfg.ffn, cfg.hidden, self.lora.r_pad):
            # the producer-side dropped copy only when the fused-dropout path wants it (p_drop > 0), as swiglu_fwd_dropout does
            gu, act, actd = ops.linear_lora_swiglu(xn2, st.pT(f"layers.{i}.wgu"), t, bexp, p3 if p_drop > 0.0 else 0.0, seed3)
            return gu, act, actd, t, keep_xd
        gu = ops.linear(xn2, st.p(f"layers.{i}.wgu"), st.pT(f"layers.{i}.wgu"))
        ops.gemm_nn(t, bexp, out=gu, residual=gu)
        act = ops.swiglu_fwd(gu, interleaved=True)
        actd = ops.dropout(act, p3, seed3) if p_drop > 0.0 else None
        return gu, act, actd, t, keep_xd

    def _lora_down_bwd_swiglu(self, dy: torch.Tensor, act: torch.Tensor, t: torch.Tensor, i: int, xd: Optional[torch.Tensor],
                              gu: torch.Tensor) -> torch.Tensor:
        """d(gate|up) of an adapter model: the down projection's backward (_proj_bwd "down": dA, dB, input gradient) with the SwiGLU
        backward in the epilogue of its input-gradient GEMM."""
        st, cfg = self.store, self.cfg
        rp, sc = self.lora.r_pad, self.lora.scaling
        wkey, akey, bkey = f"layers.{i}.wdown", f"layers.{i}.lora_down.A", f"layers.{i}.lora_down.B"
        if (self.training and self.lora.lora_dropout > 0.0 and self.lora_peft_masks) or \
                not ops.linear_lora_swiglu_ok(dy.shape[0], cfg.ffn, cfg.hidden, rp):
            dact = self._proj_bwd(dy, act, t, i, "down", drop_slot=3, xd=xd)
            return ops.swiglu_bwd(dact, gu, interleaved=True)
        dt = ops.gemm_nt(dy, st.pT(bkey), alpha=sc)                   # dt = (alpha/r) dy B_down: [M, rp]
        drop = self.training and self.lora.lora_dropout > 0.0
        p_, seed = (self.lora.lora_dropout, self._dropout_seed(i, 3)) if drop else (0.0, 0)
        dgu = ops.linear_lora_swiglu_bwd(dy, st.p(wkey), dt, st.p(akey), gu, p_, seed)
        xin = act
        if drop:
            xin = xd if xd is not None else ops.dropout(act, p_, seed)
        ops.gemm_tn_skinny(dt, xin, out=st.g(akey))
        ops.gemm_tn_skinny(dy, t, out=st.g(bkey))
        return dgu

    def _layer_fwd(self, i: int, x: torch.Tensor, plan: SplicePlan, cos, sin, save: bool):
        """One decoder layer (HF LlamaDecoderLayer: RMSNorm, QKV, RoPE, causal attention, O + residual, RMSNorm, SwiGLU
        + residual).  Returns (x_next, context for backward or None)."""
        cfg, st = self.cfg, self.store
        d, H, hd = cfg.hidden, cfg.heads, cfg.head_dim
        S, L = plan.S, plan.L
        # LoRA under adapter dropout: the kernels that produce a projection's input also write its dropped copy (one pass less over
        # the activation per projection; the attention output keeps the stand-alone rv_dropout)
        drop = self._lora_drop() and os.environ.get("RV_LORA_FUSED_DROPOUT", "1") != "0" and not self.lora_peft_masks
        p_drop = self.lora.lora_dropout if drop else 0.0
        xnd = xn2d = actd = None
        pending = None
        if isinstance(x, tuple):      # RV_RESID_FP32: (fp32 stream, the previous layer's bf16 down-projection branch not yet added to it)
            x, pending = x
        if pending is not None:       # the add rides in this layer's first norm: x = stream + branch, xn = rmsnorm(x), one pass
            x, xn, rstd1 = ops.add_rmsnorm_fwd(x, pending, st.p(f"layers.{i}.ln1"), cfg.rms_eps)
            del pending
        elif drop:
            xn, rstd1, xnd = ops.rmsnorm_fwd_dropout(x, st.p(f"layers.{i}.ln1"), cfg.rms_eps, p_drop, self._dropout_seed(i, 0))
        else:
            xn, rstd1 = ops.rmsnorm_fwd(x, st.p(f"layers.{i}.ln1"), cfg.rms_eps)
        rope_cols = (H + cfg.n_kv_heads) * hd
        if (self.lora is None and self.fuse_rope_fwd
                and ops.linear_rope_ok(xn.shape[0], rope_cols + cfg.kv_dim, xn.shape[1], rope_cols, hd)):
            # RoPE in the epilogue of the q|k|v projection, from the fp32 accumulators (round 6): no rope pass, one rounding less on Q / K
            qkv = ops.linear_rope(xn, st.pT(f"layers.{i}.wqkv")[:, :rope_cols + cfg.kv_dim], cos, sin, plan.pos, L, rope_cols, hd)
            t_qkv = xd_qkv = None
        else:
            qkv, t_qkv, xd_qkv = self._proj_fwd(xn, i, "qkv", drop_slot=0, xd=xnd)
            ops.rope_inplace(qkv, cos, sin, L, H + cfg.n_kv_heads, hd, pos=plan.pos)      # q heads then k heads
        attn, lse = ops.attn_fwd(qkv, S, L, H, hd, True, 0, d, d + cfg.kv_dim, seg=plan.seg, kv_group=cfg.kv_group, rows=plan.rows)
        if self.resid_fp32:
            if drop:
                raise NotImplementedError("RV_RESID_FP32 with the producer-side LoRA dropout kernels (set RV_LORA_FUSED_DROPOUT=0)")
            o, t_o, xd_o = self._proj_fwd(attn, i, "o", residual=None, drop_slot=1)          # the BRANCH, bf16
            x_mid, xn2, rstd2 = ops.add_rmsnorm_fwd(x, o, st.p(f"layers.{i}.ln2"), cfg.rms_eps)   # fp32 stream + norm, one pass
            del o
        else:
            x_mid, t_o, xd_o = self._proj_fwd(attn, i, "o", residual=x, drop_slot=1)
            if drop:
                xn2, rstd2, xn2d = ops.rmsnorm_fwd_dropout(x_mid, st.p(f"layers.{i}.ln2"), cfg.rms_eps, p_drop, self._dropout_seed(i, 2))
            else:
                xn2, rstd2 = ops.rmsnorm_fwd(x_mid, st.p(f"layers.{i}.ln2"), cfg.rms_eps)
        if st.interleave_gu and self.lora is None:   # full fine-tune: SwiGLU in the epilogue of the gate|up GEMM (interleaved weight rows)
            gu, act = ops.linear_swiglu(xn2, st.pT(f"layers.{i}.wgu"))
            t_gu = xd_gu = None
        elif st.lora_il:              # adapter model, interleaved layout (RV_LORA_FUSE_SWIGLU=1): the same epilogue on the fused-LoRA GEMM
            gu, act, actd, t_gu, xd_gu = self._lora_gu_fwd(xn2, i, xn2d, p_drop)
        else:
            gu, t_gu, xd_gu = self._proj_fwd(xn2, i, "gu", drop_slot=2, xd=xn2d)
            if drop:
                act, actd = ops.swiglu_fwd_dropout(gu, p_drop, self._dropout_seed(i, 3))
            else:
                act = ops.swiglu_fwd(gu)
        if self.resid_fp32:
            dwn, t_down, xd_down = self._proj_fwd(act, i, "down", residual=None, drop_slot=3, xd=actd)
            x_next = (x_mid, dwn)     # the NEXT layer's first norm (or the caller, after the last layer) adds the branch to the stream
        else:
            x_next, t_down, xd_down = self._proj_fwd(act, i, "down", residual=x_mid, drop_slot=3, xd=actd)
        if not save:
            return x_next, None
        keep = self.keep_recomputable
        return x_next, dict(x=x, rstd1=rstd1, qkv=qkv, attn=attn, lse=lse, x_mid=x_mid, rstd2=rstd2, gu=gu,
                            t_qkv=t_qkv, t_o=t_o, t_gu=t_gu, t_down=t_down,
                            xd_qkv=xd_qkv, xd_o=xd_o, xd_gu=xd_gu, xd_down=xd_down,
                            # 288 GB of HBM: keep the cheap-to-recompute operands too (+1.05 GB / layer at 27 k tokens)
                            # instead of re-running RMSNorm / SwiGLU in backward
                            xn=xn if keep else None, xn2=xn2 if keep else None, act=act if keep else None)

    # ------------------------------------------------------------------ forward
    def forward_logps(self, input_ids: torch.Tensor, labels: torch.Tensor, images: torch.Tensor,
                      save_for_backward: bool = True, all_rows: bool = False, label_shift: int = 1) -> StepOutput:
        """Everything of get_beta_and_logps up to ``get_batch_logps``: returns per-sequence log-prob sums
        and counts (muffin/eval/muffin_inference_logp.py:82-115) without materialising logits.
        ``all_rows`` (forward only, reference layout): evaluate EVERY position like
        ``get_batch_logps(return_all=True)`` does - masked positions get the log-prob of token id 0, exactly what
        the reference stores in its ``logps`` parquet column."""
        cfg, st = self.cfg, self.store
        d, H, hd, f = cfg.hidden, cfg.heads, cfg.head_dim, cfg.ffn
        B = len(images)                  # tensor [B, 3, H, W] or a list of raw uint8 [H, W, 3] images
        ctx: dict = {}
        w_rows = None
        if self.training:            # evaluation passes between steps must not shift the dropout seeds of later steps
            self._dropout_step += 1
        self._cur_drop_step = ctx["dropout_step"] = self._dropout_step
        if all_rows:
            if save_for_backward:
                raise ValueError("all_rows is a forward-only mode")
            plan = build_splice_plan(input_ids, labels, cfg.n_image_tokens, B, cfg.model_max_length, splicer=self._row_splicer())
            nxt = plan.labels[:, 1:]
            S_, Lm1 = nxt.shape
            plan.sel_idx = (torch.arange(S_)[:, None] * plan.L + torch.arange(Lm1)[None]).reshape(-1).to(torch.int32)
            plan.tgt = torch.where(nxt != -100, nxt, torch.zeros_like(nxt)).reshape(-1).to(torch.int32)
            plan.seq_off = (torch.arange(S_ + 1) * Lm1).to(torch.int32)
            plan.seq_of_row = torch.arange(S_).repeat_interleave(Lm1).to(torch.int32)
            plan.n_sel = int(plan.sel_idx.numel())
            w_rows = (nxt != -100).reshape(-1).to(torch.float32).to(self.device)     # loss_mask of get_batch_logps
        elif label_shift != 1:
            # get_batch_logps_minicpm convention (labels pre-shifted, muffin_inference_logp.py:21-52): reference layout only
            plan = build_splice_plan(input_ids, labels, cfg.n_image_tokens, B, cfg.model_max_length, label_shift=label_shift,
                                     splicer=self._row_splicer())
        elif self.share_prefix and input_ids.shape[0] == 2 * B:
            plan = build_packed_plan(input_ids, labels, cfg.n_image_tokens, B, cfg.model_max_length, cfg.pad_token_id,
                                     splicer=self._row_splicer(), pad_free=self.pad_free)
        else:
            plan = build_splice_plan(input_ids, labels, cfg.n_image_tokens, B, cfg.model_max_length, splicer=self._row_splicer())
        plan = plan.to(self.device)
        S, L = plan.S, plan.L
        N = plan.n_tokens                # S * L in the rectangular layouts, the sum of the packed rows' lengths when pad-free
        feats = self.encode_images(images, ctx if save_for_backward else None)
        x = ops.splice_fwd(plan.src, st.p("model.embed_tokens.weight"), feats, d)
        if self.resid_fp32:
            x = ops.cast_bf16_to_f32(x)          # embedding / feature rows are bf16 values: exact
        cos, sin = self._rope(L)
        layers_ctx = []
        for i in range(cfg.layers):
            x_next, lctx = self._layer_fwd(i, x, plan, cos, sin, save_for_backward and not self.gradient_checkpointing)
            if save_for_backward:
                if lctx is None:      # --gradient_checkpointing: keep the layer's INPUT (fp32 stream: with its pending branch added)
                    lctx = dict(x=x if not isinstance(x, tuple) else ops.add_f32_bf16(*x), recompute=True)
                layers_ctx.append(lctx)
            x = x_next
        if isinstance(x, tuple):
            x = ops.add_f32_bf16(*x)          # the last layer's branch joins the fp32 stream
        n_sel = plan.n_sel
        n_pad = max(64, ops.round_up(n_sel, 64))
        hsel = torch.zeros(n_pad, d, dtype=BF16, device=self.device)
        if n_sel > 0:
            _, rstd_f = ops.rmsnorm_fwd(x, st.p("model.norm.weight"), cfg.rms_eps, row_idx=plan.sel_idx, out=hsel[:n_sel])
            logp, lse_v = ops.lmhead_logp_fwd(hsel, st.p("lm_head.weight"), plan.tgt, n_sel, v_valid=cfg.vocab)
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

        lora = self.lora is not None
        self._cur_drop_step = ctx.get("dropout_step", 0)
        scratch_dw = torch.empty(d, dtype=BF16, device=self.device) if lora else None   # gains are frozen under LoRA

        def wgrad(dy: torch.Tensor, xin: torch.Tensor, key: str):
            """dW[key] = dy^T @ xin: TN GEMM (operands transposed on the fly by ds_read_b64_tr_b16)."""
            ops.gemm_tn(dy, xin, out=st.g(key))

        def gain_grad(key: str) -> torch.Tensor:
            return scratch_dw if lora else st.g(key)

        # ---- LM head + final norm
        n_sel = plan.n_sel
        dx = torch.zeros(N, d, dtype=BF16, device=self.device)
        if n_sel > 0:
            rc = ops.row_coef(coef, plan.seq_of_row, ctx["w_rows"])
            dlog = ops.lmhead_logp_bwd(ctx["hsel"], st.p("lm_head.weight"), plan.tgt, ctx["lse_v"], rc, n_sel,
                                       v_valid=cfg.vocab)
            dh = ops.linear(dlog, st.pT("lm_head.weight"), st.p("lm_head.weight"))
            if not lora:
                wgrad(dlog, ctx["hsel"], "lm_head.weight")
            del dlog
            ops.rmsnorm_bwd(dh[:n_sel], ctx["x_final"], st.p("model.norm.weight"), ctx["rstd_f"],
                            gain_grad("model.norm.weight"), row_idx=plan.sel_idx, dx=dx)
        elif not lora:
            st.g("lm_head.weight").zero_()
            st.g("model.norm.weight").zero_()
        if hook and not lora:
            hook("lm_head", *st.buckets["lm_head"])

        # ---- decoder layers, last to first
        for i in reversed(range(cfg.layers)):
            c = ctx["layers"][i]
            if c.get("recompute"):      # --gradient_checkpointing: only the layer input was kept; run the layer again
                _, c = self._layer_fwd(i, c["x"], plan, cos, sin, True)
            act = c["act"] if c["act"] is not None else ops.swiglu_fwd(c["gu"], interleaved=st.interleave_gu)
            c["act"] = None
            if st.interleave_gu and self.lora is None:
                # d(gate|up) straight out of the down projection's input-gradient GEMM (SwiGLU backward in its epilogue)
                dgu = ops.linear_swiglu_bwd(dx, st.p(f"layers.{i}.wdown"), c["gu"])
                ops.gemm_tn(dx, act, out=st.g(f"layers.{i}.wdown"))
                del act
            elif st.lora_il:
                dgu = self._lora_down_bwd_swiglu(dx, act, c["t_down"], i, c["xd_down"], c["gu"])
                del act
            else:
                dact = self._proj_bwd(dx, act, c["t_down"], i, "down", drop_slot=3, xd=c["xd_down"])
                del act
                dgu = ops.swiglu_bwd(dact, c["gu"])
                del dact
            xn2 = c["xn2"] if c["xn2"] is not None else \
                ops.rmsnorm_fwd(c["x_mid"], st.p(f"layers.{i}.ln2"), cfg.rms_eps, want_rstd=False)[0]
            c["xn2"] = None
            dxn2 = self._proj_bwd(dgu, xn2, c["t_gu"], i, "gu", drop_slot=2, xd=c["xd_gu"])
            del dgu, xn2
            dx_mid = ops.rmsnorm_bwd(dxn2, c["x_mid"], st.p(f"layers.{i}.ln2"), c["rstd2"], gain_grad(f"layers.{i}.ln2"),
                                     dres=dx)
            del dxn2
            dattn = self._proj_bwd(dx_mid, c["attn"], c["t_o"], i, "o", drop_slot=1, xd=c["xd_o"])
            # dQ / dK leave the attention backward already rotated back (RV_FUSE_ROPE_BWD=0: separate rv_rope_inplace pass)
            dqkv = ops.attn_bwd(c["qkv"], c["attn"], dattn, c["lse"], S, L, H, hd, True, 0, d, d + cfg.kv_dim,
                                seg=plan.seg, kv_group=cfg.kv_group, rope=(cos, sin, plan.pos) if self.fuse_rope_bwd else None,
                                rows=plan.rows)
            del dattn
            if not self.fuse_rope_bwd:
                ops.rope_inplace(dqkv, cos, sin, L, H + cfg.n_kv_heads, hd, backward=True, pos=plan.pos)
            xn = c["xn"] if c["xn"] is not None else \
                ops.rmsnorm_fwd(c["x"], st.p(f"layers.{i}.ln1"), cfg.rms_eps, want_rstd=False)[0]
            c["xn"] = None
            dxn = self._proj_bwd(dqkv, xn, c["t_qkv"], i, "qkv", drop_slot=0, xd=c["xd_qkv"])
            del dqkv, xn
            dx = ops.rmsnorm_bwd(dxn, c["x"], st.p(f"layers.{i}.ln1"), c["rstd1"], gain_grad(f"layers.{i}.ln1"),
                                 dres=dx_mid)
            del dxn, dx_mid
            ctx["layers"][i] = None          # free this layer's activations
            if hook:
                hook(f"layer{i}", *st.buckets[f"layer{i}"])

        # ---- embedding (deterministic segmented sum; frozen under LoRA) and projector
        if not lora:
            ge = st.g("model.embed_tokens.weight")
            ge.zero_()
            ops.embed_bwd(plan.uniq_ids, plan.seg_off, plan.pos_sorted, dx, ge)
        dfeat = ops.feat_grad(plan.feat_src_a, plan.feat_src_b, dx, d)
        self._vision_backward(dfeat, ctx)
        if hook:
            hook("embed_proj", *st.buckets["embed_proj"])
            hook("nodecay", *st.buckets["nodecay"])
        out.ctx = {}

    # ------------------------------------------------------------------ reference-style surface
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, images=None, **kwargs):
        """``LlavaLlamaForCausalLM.forward(inputs_embeds=, labels=None).logits`` - the call get_beta_and_logps makes after
        the splice (muffin/train/trainers.py:221-225, llava/model/language_model/llava_llama.py:57-102).  Debug / evaluation
        surface for SMALL inputs: it materialises [S, L, V] logits, which the training path never does (forward_logps fuses
        the LM head with the log-softmax).  Forward only; pure causal mask, positions 0..L-1, like the reference call."""
        from types import SimpleNamespace
        if inputs_embeds is None:
            raise NotImplementedError("forward() takes inputs_embeds (prepare_inputs_labels_for_multimodal builds them)")
        if labels is not None or attention_mask is not None or past_key_values is not None:
            raise NotImplementedError("the DPO call site passes labels=None, attention_mask=None, no cache")
        cfg, st = self.cfg, self.store
        S, L, d = inputs_embeds.shape
        if S * L * cfg.vocab * 2 > (16 << 30):
            raise ValueError("forward(): logits would exceed 16 GB - use forward_logps (fused LM head) for training shapes")
        x = inputs_embeds.to(self.device, BF16).reshape(S * L, d).contiguous()
        if self.resid_fp32:
            x = ops.cast_bf16_to_f32(x)
        plan = SimpleNamespace(S=S, L=L, pos=None, seg=None, rows=None)
        cos, sin = self._rope(L)
        for i in range(cfg.layers):
            x, _ = self._layer_fwd(i, x, plan, cos, sin, False)
        if isinstance(x, tuple):
            x = ops.add_f32_bf16(*x)
        h, _ = ops.rmsnorm_fwd(x, st.p("model.norm.weight"), cfg.rms_eps, want_rstd=False)
        logits = ops.gemm_nt(h, st.p("lm_head.weight"))[:, :cfg.vocab]          # drop the vocabulary padding columns
        return SimpleNamespace(logits=logits.reshape(S, L, cfg.vocab), loss=None)

    __call__ = forward

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images):
        """Same 6-tuple contract as llava_arch.py:150-330 (embeds materialised by rv_splice_fwd)."""
        if attention_mask is not None:
            raise NotImplementedError("the DPO path passes attention_mask=None (trainers.py:199)")
        feats = self.encode_images(images)      # one feature block per row of `images`, like the reference
        n_img = len(images)
        plan = build_splice_plan(input_ids, labels, self.cfg.n_image_tokens, n_img, self.cfg.model_max_length,
                                 splicer=self._row_splicer()).to(self.device)
        emb = ops.splice_fwd(plan.src, self.store.p("model.embed_tokens.weight"), feats, self.cfg.hidden)
        return None, None, None, past_key_values, emb.view(plan.S, plan.L, -1), plan.labels.to(self.device)
