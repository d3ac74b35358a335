"""CPU ORACLE for the RLAIF-V LLaVA-1.5 DPO step.  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-CPU (fp32, optionally fp64) restatement of the reference's DPO hot
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product package (``rlaif_v_amd``) never does.

Pinning: the reference ships no tests or golden vectors, so this oracle is
pinned against outputs of the reference ITSELF executed in the build container
(``tests/golden/make_golden.py`` imports ``/root/reference`` + the installed transformers 5.15
Llama/CLIP modules and writes ``tests/golden/*.pt``); ``tests/test_oracle_golden.py`` replays
them.  The third-party arithmetic (HF ``LlamaForCausalLM`` / ``CLIPVisionModel``, pinned by the
reference at transformers==4.35.0, source not vendored in /root/reference) is restated here from
the published algorithm; the call sites it is anchored on are cited per function
(paths relative to /root/reference).

Weights travel as a flat ``dict[str, Tensor]`` keyed by the HF state-dict names of
``LlavaLlamaForCausalLM`` (llava/model/language_model/llava_llama.py:41-49).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100        # muffin/train/train_utils.py:21
IMAGE_TOKEN_INDEX = -200   # muffin/train/train_utils.py:20

VT = "model.vision_tower.vision_tower.vision_model."


@dataclass
class LlavaCfg:
    """Shape parameters (Vicuna-7B-v1.5 + CLIP-L/14-336 defaults; SURVEY.md section 8a notes)."""
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
    select_layer: int = -2          # mm_vision_select_layer (script/train/llava15_train.sh)
    model_max_length: int = 2048    # tokenizer_model_max_length (train_llava15.py:249)
    pad_token_id: int = 0           # tokenizer.pad_token = unk (train_llava15.py:228)
    kv_heads: Optional[int] = None  # num_key_value_heads of the HF Llama/Mistral config (None = heads: LLaVA-1.5 is MHA)

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def n_kv_heads(self) -> int:
        return self.heads if self.kv_heads is None else self.kv_heads

    @property
    def clip_head_dim(self) -> int:
        return self.clip_hidden // self.clip_heads

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch) ** 2

    @property
    def clip_layers_used(self) -> int:
        # hidden_states[select_layer]; hidden_states has clip_layers+1 entries
        # (llava/model/multimodal_encoder/clip_encoder.py:36-44)
        idx = self.select_layer if self.select_layer >= 0 else self.clip_layers + 1 + self.select_layer
        return idx


def tiny_gqa_cfg() -> LlavaCfg:
    """Grouped-query attention (4 query heads on 2 key/value heads), the Mistral / Llama-3 arrangement of the
    reference's other language models (OmniLMM's Zephyr, MiniCPM-Llama3-V); head_dim stays 128."""
    return LlavaCfg(hidden=512, layers=2, heads=4, kv_heads=2, ffn=768, vocab=512,
                    clip_hidden=128, clip_layers=3, clip_heads=2, clip_ffn=256,
                    image_size=56, patch=14, model_max_length=256)


def tiny_cfg() -> LlavaCfg:
    """Small config whose head dims (128 LLM / 64 CLIP) match the production kernels."""
    return LlavaCfg(hidden=256, layers=2, heads=2, ffn=512, vocab=512,
                    clip_hidden=128, clip_layers=3, clip_heads=2, clip_ffn=256,
                    image_size=56, patch=14, model_max_length=256)


# --------------------------------------------------------------------------------------------
# deterministic weights (shared by golden generation, oracle tests and GPU parity tests)
# --------------------------------------------------------------------------------------------

def weight_shapes(cfg: LlavaCfg) -> Dict[str, Tuple[int, ...]]:
    d, f, v = cfg.hidden, cfg.ffn, cfg.vocab
    cd, cf = cfg.clip_hidden, cfg.clip_ffn
    s: Dict[str, Tuple[int, ...]] = {}
    s["model.embed_tokens.weight"] = (v, d)
    for i in range(cfg.layers):
        p = f"model.layers.{i}."
        kvd = cfg.n_kv_heads * cfg.head_dim
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            s[p + f"self_attn.{n}.weight"] = (kvd if n in ("k_proj", "v_proj") else d, d)
        s[p + "mlp.gate_proj.weight"] = (f, d)
        s[p + "mlp.up_proj.weight"] = (f, d)
        s[p + "mlp.down_proj.weight"] = (d, f)
        s[p + "input_layernorm.weight"] = (d,)
        s[p + "post_attention_layernorm.weight"] = (d,)
    s["model.norm.weight"] = (d,)
    s["lm_head.weight"] = (v, d)
    s["model.mm_projector.0.weight"] = (d, cd)
    s["model.mm_projector.0.bias"] = (d,)
    s["model.mm_projector.2.weight"] = (d, d)
    s["model.mm_projector.2.bias"] = (d,)
    s[VT + "embeddings.class_embedding"] = (cd,)
    s[VT + "embeddings.patch_embedding.weight"] = (cd, 3, cfg.patch, cfg.patch)
    s[VT + "embeddings.position_embedding.weight"] = (cfg.n_patches + 1, cd)
    s[VT + "pre_layrnorm.weight"] = (cd,)
    s[VT + "pre_layrnorm.bias"] = (cd,)
    for i in range(cfg.clip_layers):
        p = VT + f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (cd, cd)
            s[p + f"self_attn.{n}.bias"] = (cd,)
        s[p + "layer_norm1.weight"] = (cd,)
        s[p + "layer_norm1.bias"] = (cd,)
        s[p + "layer_norm2.weight"] = (cd,)
        s[p + "layer_norm2.bias"] = (cd,)
        s[p + "mlp.fc1.weight"] = (cf, cd)
        s[p + "mlp.fc1.bias"] = (cf,)
        s[p + "mlp.fc2.weight"] = (cd, cf)
        s[p + "mlp.fc2.bias"] = (cd,)
    s[VT + "post_layernorm.weight"] = (cd,)
    s[VT + "post_layernorm.bias"] = (cd,)
    return s


def make_weights(cfg: LlavaCfg, seed: int = 0, std: float = 0.02, bf16_round: bool = True
                 ) -> Dict[str, torch.Tensor]:
    """Seeded random weights: N(0, std) matrices (HF default init), norm gains
    1 + N(0, 0.1), small biases.  One private generator per tensor so any subset reproduces.
    ``bf16_round`` rounds values to bf16-representable fp32 so the GPU (bf16 storage) and the
    oracle (fp32) start from bit-identical parameters."""
    out: Dict[str, torch.Tensor] = {}
    for idx, (k, shp) in enumerate(weight_shapes(cfg).items()):
        g = torch.Generator().manual_seed(seed * 1000003 + idx)
        if k.endswith("layernorm.weight") or k.endswith("norm.weight") or "layer_norm" in k and k.endswith("weight") \
                or k.endswith("pre_layrnorm.weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            t = 0.02 * torch.randn(shp, generator=g)
        else:
            t = std * torch.randn(shp, generator=g)
        if bf16_round:
            t = t.to(torch.bfloat16).to(torch.float32)
        out[k] = t
    return out


# --------------------------------------------------------------------------------------------
# synthetic preference batch  (BASELINE.md section 2 "Inputs")
# --------------------------------------------------------------------------------------------

def make_synthetic_batch(cfg: LlavaCfg, n_pairs: int, text_len: int, prompt_len: int = 64,
                         seed: int = 0, ragged: bool = True, image_pos: int = 35,
                         beta: float = 0.1, answer_lens: Optional[List[Tuple[int, int]]] = None) -> Dict[str, object]:
    """Batch dict with exactly the collator's schema (muffin/train/train_muffin.py:43-112,
    muffin/eval/muffin_inference_logp.py:187-208): all wins then all rejects, right-padded with
    pad id 0 / label -100, one -200 image placeholder inside the shared prompt.
    ``answer_lens``: explicit (chosen, rejected) answer lengths per pair instead of the ragged draw."""
    g = torch.Generator().manual_seed(seed)
    image_pos = min(image_pos, prompt_len - 2)
    prompt = torch.randint(3, cfg.vocab, (n_pairs, prompt_len), generator=g)
    prompt[:, 0] = 1
    prompt[:, image_pos] = IMAGE_TOKEN_INDEX
    wins, rejs = [], []
    for b in range(n_pairs):
        for dst in (wins, rejs):
            if answer_lens is not None:
                alen = int(answer_lens[b][0 if dst is wins else 1])
                assert 2 <= alen <= text_len - prompt_len
            elif ragged:
                lo = max(2, (text_len - prompt_len) // 2)
                alen = int(torch.randint(lo, text_len - prompt_len + 1, (1,), generator=g))
            else:
                alen = text_len - prompt_len
            ans = torch.randint(3, cfg.vocab, (alen,), generator=g)
            ans[-1] = 2
            ids = torch.cat([prompt[b], ans])
            lab = ids.clone()
            lab[:prompt_len] = IGNORE_INDEX
            dst.append(dict(input_ids=ids, labels=lab))
    if ragged and n_pairs > 0 and answer_lens is None:
        # make sure the batch really reaches text_len so shapes are deterministic
        w0 = wins[0]
        if w0["input_ids"].numel() < text_len:
            extra = torch.randint(3, cfg.vocab, (text_len - w0["input_ids"].numel(),), generator=g)
            ids = torch.cat([w0["input_ids"][:-1], extra, w0["input_ids"][-1:]])
            lab = ids.clone()
            lab[:prompt_len] = IGNORE_INDEX
            wins[0] = dict(input_ids=ids, labels=lab)
    images = torch.randn(n_pairs, 3, cfg.image_size, cfg.image_size, generator=g)

    def pad(seqs, val):
        return torch.nn.utils.rnn.pad_sequence(seqs, batch_first=True, padding_value=val)

    win_ids = pad([x["input_ids"] for x in wins], cfg.pad_token_id)
    rej_ids = pad([x["input_ids"] for x in rejs], cfg.pad_token_id)
    win_lab = pad([x["labels"] for x in wins], IGNORE_INDEX)
    rej_lab = pad([x["labels"] for x in rejs], IGNORE_INDEX)
    cat_ids = pad(list(win_ids) + list(rej_ids), cfg.pad_token_id)
    cat_lab = pad(list(win_lab) + list(rej_lab), IGNORE_INDEX)
    batch = dict(
        concatenated_input_ids=cat_ids, concatenated_labels=cat_lab,
        concatenated_attention_mask=cat_ids.ne(cfg.pad_token_id),
        win_input_ids=win_ids, rej_input_ids=rej_ids, win_labels=win_lab, rej_labels=rej_lab,
        win_attention_mask=win_ids.ne(cfg.pad_token_id), rej_attention_mask=rej_ids.ne(cfg.pad_token_id),
        images=images, beta=beta,
        ref_win_logp=torch.full((n_pairs,), -100.0) - torch.arange(n_pairs, dtype=torch.float32),
        ref_rej_logp=torch.full((n_pairs,), -101.0) - 0.5 * torch.arange(n_pairs, dtype=torch.float32),
        ref_win_avg_logp=torch.full((n_pairs,), -1.0), ref_rej_avg_logp=torch.full((n_pairs,), -1.1),
        ref_win_per_token_logp=torch.zeros(n_pairs, win_ids.shape[1] - 1),
        ref_rej_per_token_logp=torch.zeros(n_pairs, rej_ids.shape[1] - 1),
        win_token_weight=torch.ones(n_pairs, win_ids.shape[1] - 1),
        rej_token_weight=torch.ones(n_pairs, rej_ids.shape[1] - 1),
        concatenated_token_weight=torch.ones(2 * n_pairs, cat_ids.shape[1] - 1),
    )
    return batch


# --------------------------------------------------------------------------------------------
# vision: CLIP tower (frozen) + projector
# --------------------------------------------------------------------------------------------

def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def clip_vision_features(pixels: torch.Tensor, W: Dict[str, torch.Tensor], cfg: LlavaCfg) -> torch.Tensor:
    """CLIPVisionTower.forward + feature_select('patch')
    (llava/model/multimodal_encoder/clip_encoder.py:36-58) over HF CLIPVisionModel
    (transformers/models/clip/modeling_clip.py: embeddings = conv(no bias) + CLS + learned pos;
    pre_layrnorm; encoder layer = LN1, MHA(scale hd^-0.5, biases), +res, LN2, fc1, quick_gelu,
    fc2, +res).  Returns hidden_states[select_layer][:, 1:]  -> [B, n_patches, clip_hidden]."""
    B = pixels.shape[0]
    cd, H, hd = cfg.clip_hidden, cfg.clip_heads, cfg.clip_head_dim
    x = F.conv2d(pixels, W[VT + "embeddings.patch_embedding.weight"], None, stride=cfg.patch)
    x = x.flatten(2).transpose(1, 2)                                   # [B, P, cd]
    cls = W[VT + "embeddings.class_embedding"].expand(B, 1, cd)
    x = torch.cat([cls, x], dim=1) + W[VT + "embeddings.position_embedding.weight"][None]
    x = _ln(x, W[VT + "pre_layrnorm.weight"], W[VT + "pre_layrnorm.bias"], cfg.clip_eps)
    T = x.shape[1]
    for i in range(cfg.clip_layers_used):
        p = VT + f"encoder.layers.{i}."
        h = _ln(x, W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"], cfg.clip_eps)
        q = F.linear(h, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"])
        k = F.linear(h, W[p + "self_attn.k_proj.weight"], W[p + "self_attn.k_proj.bias"])
        v = F.linear(h, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"])
        q = q.view(B, T, H, hd).transpose(1, 2)
        k = k.view(B, T, H, hd).transpose(1, 2)
        v = v.view(B, T, H, hd).transpose(1, 2)
        a = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, T, cd)
        x = x + F.linear(a, W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"])
        h = _ln(x, W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"], cfg.clip_eps)
        h = F.linear(h, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"])
        h = h * torch.sigmoid(1.702 * h)                               # quick_gelu
        x = x + F.linear(h, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"])
    return x[:, 1:]


def mm_projector(feats: torch.Tensor, W: Dict[str, torch.Tensor]) -> torch.Tensor:
    """mlp2x_gelu (llava/model/multimodal_projector/builder.py:39-46)."""
    h = F.linear(feats, W["model.mm_projector.0.weight"], W["model.mm_projector.0.bias"])
    h = F.gelu(h)  # erf GELU
    return F.linear(h, W["model.mm_projector.2.weight"], W["model.mm_projector.2.bias"])


def encode_images(pixels, W, cfg):
    """llava/model/llava_arch.py:141-148; the tower runs under no_grad (clip_encoder.py:46)."""
    with torch.no_grad():
        f = clip_vision_features(pixels, W, cfg)
    return mm_projector(f, W)


# --------------------------------------------------------------------------------------------
# splice (token indexing must be bit exact)
# --------------------------------------------------------------------------------------------

def splice_plan(input_ids: torch.Tensor, labels: torch.Tensor, n_img_tokens: int,
                max_len: Optional[int]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Integer plan of prepare_inputs_labels_for_multimodal (llava/model/llava_arch.py:150-330)
    for attention_mask=None (trainers.py:199 -> mask all ones, pads kept, :220-231).
    Returns (src_kind[S,L], src_index[S,L], new_labels[S,L]) with src_kind 0 = zero pad,
    1 = embed_tokens[src_index], 2 = image feature row src_index of image #row."""
    S, T = input_ids.shape
    rows_kind, rows_idx, rows_lab = [], [], []
    for r in range(S):
        ids, lab = input_ids[r], labels[r]
        kind, idx, nl = [], [], []
        for t in range(T):
            tok = int(ids[t])
            if tok == IMAGE_TOKEN_INDEX:                       # :248-269
                kind += [2] * n_img_tokens
                idx += list(range(n_img_tokens))
                nl += [IGNORE_INDEX] * n_img_tokens
            else:
                kind.append(1)
                idx.append(tok)
                nl.append(int(lab[t]))
        if max_len is not None:                                # :280-283
            kind, idx, nl = kind[:max_len], idx[:max_len], nl[:max_len]
        rows_kind.append(kind), rows_idx.append(idx), rows_lab.append(nl)
    L = max(len(k) for k in rows_kind)                         # :286
    sk = torch.zeros(S, L, dtype=torch.int64)
    si = torch.zeros(S, L, dtype=torch.int64)
    nl = torch.full((S, L), IGNORE_INDEX, dtype=torch.int64)   # :305-313 right pad
    for r in range(S):
        n = len(rows_kind[r])
        sk[r, :n] = torch.tensor(rows_kind[r], dtype=torch.int64)
        si[r, :n] = torch.tensor(rows_idx[r], dtype=torch.int64)
        nl[r, :n] = torch.tensor(rows_lab[r], dtype=torch.int64)
    return sk, si, nl


def prepare_inputs_labels_for_multimodal(input_ids, labels, image_features, embed_weight, max_len):
    """Embeds [S,L,d] and labels [S,L] exactly as llava_arch.py:150-330 builds them."""
    S = input_ids.shape[0]
    sk, si, nl = splice_plan(input_ids, labels, image_features.shape[1], max_len)
    L = sk.shape[1]
    emb = torch.zeros(S, L, embed_weight.shape[1], dtype=image_features.dtype)
    for r in range(S):
        m1 = sk[r] == 1
        m2 = sk[r] == 2
        if m1.any():
            emb[r, m1] = F.embedding(si[r, m1], embed_weight).to(emb.dtype)
        if m2.any():
            emb[r, m2] = image_features[r][si[r, m2]]
    return emb, nl


# --------------------------------------------------------------------------------------------
# language model (HF LlamaForCausalLM restated; call site llava_llama.py:91-102)
# --------------------------------------------------------------------------------------------

def rms_norm(x, w, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps)).to(x.dtype)


def rope_tables(L: int, hd: int, theta: float, dtype=torch.float32):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


LORA_TARGETS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


def lora_linear(x: torch.Tensor, W: Dict[str, torch.Tensor], name: str, lora_scale: Optional[float],
                masks: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """peft 0.10 `lora.Linear.forward` (third-party, absent offline; pinned by pyproject.toml:16-23, configured at
    muffin/train/train_llava15_lora.py:304-318): result = base(x) + lora_B(lora_A(dropout(x))) * (lora_alpha / r).
    Dropout is the identity unless ``masks[name]`` (an explicit keep/(1-p) multiplier of x's shape) is supplied -
    that is how a test replays the exact mask a device kernel drew."""
    y = F.linear(x, W[name + ".weight"])
    a = W.get(name + ".lora_A.weight") if lora_scale is not None else None
    if a is not None:
        xa = (x * masks[name].view_as(x)).to(x.dtype) if masks is not None and name in masks else x    # (one rounding under bf16 emulation)
        y = y + lora_scale * F.linear(F.linear(xa, a), W[name + ".lora_B.weight"])
    return y


def make_lora_weights(cfg: LlavaCfg, r: int, seed: int = 1, b_std: Optional[float] = 0.02,
                      bf16_round: bool = True) -> Dict[str, torch.Tensor]:
    """Adapter tensors under peft's module names.  peft initialises lora_A with kaiming_uniform(a=sqrt(5)) and lora_B
    with zeros (`LoraLayer.reset_lora_parameters`); b_std=None reproduces that, a float draws B ~ N(0, b_std) so that
    parity tests exercise a non-trivial adapter."""
    g = torch.Generator().manual_seed(seed)
    d, f = cfg.hidden, cfg.ffn
    kvd = cfg.n_kv_heads * cfg.head_dim                     # grouped-query attention: k / v projections are [kv_dim, hidden]
    dims = {"self_attn.q_proj": (d, d), "self_attn.k_proj": (kvd, d), "self_attn.v_proj": (kvd, d), "self_attn.o_proj": (d, d),
            "mlp.gate_proj": (f, d), "mlp.up_proj": (f, d), "mlp.down_proj": (d, f)}
    out: Dict[str, torch.Tensor] = {}
    for i in range(cfg.layers):
        for t in LORA_TARGETS:
            o, n_in = dims[t]
            bound = 1.0 / math.sqrt(n_in)                       # kaiming_uniform_(a=sqrt(5)) on [r, n_in]
            a = (torch.rand(r, n_in, generator=g) * 2 - 1) * bound
            b = torch.zeros(o, r) if b_std is None else torch.randn(o, r, generator=g) * b_std
            if bf16_round:
                a, b = a.bfloat16().float(), b.bfloat16().float()
            out[f"model.layers.{i}.{t}.lora_A.weight"] = a
            out[f"model.layers.{i}.{t}.lora_B.weight"] = b
    return out


def merge_lora(W: Dict[str, torch.Tensor], lora_scale: float) -> Dict[str, torch.Tensor]:
    """peft merge_and_unload (llava/model/builder.py:81-85): W' = W + scale * B @ A; adapter keys dropped."""
    out = {k: v for k, v in W.items() if ".lora_" not in k}
    for k, a in W.items():
        if k.endswith(".lora_A.weight"):
            base = k[:-len(".lora_A.weight")]
            out[base + ".weight"] = W[base + ".weight"] + lora_scale * (W[base + ".lora_B.weight"] @ a)
    return out


def lora_trainable_names(W: Dict[str, torch.Tensor]) -> List[str]:
    """LoRA run: adapters + the projector, which initialize_vision_modules re-enables after peft froze it
    (llava/model/llava_arch.py:90-93 'In case it is frozen by LoRA'); saved as non_lora_trainables.bin."""
    return [k for k in W if ".lora_" in k or "mm_projector" in k]


def llama_layer(x: torch.Tensor, W: Dict[str, torch.Tensor], cfg: LlavaCfg, i: int, cos, sin, causal,
                lora_scale: Optional[float] = None, lora_masks: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """ONE HF LlamaDecoderLayer (input RMSNorm, q/k/v, rotary, causal softmax attention, o, residual, post-attention
    RMSNorm, SwiGLU MLP, residual).  ``llama_hidden`` is the loop over it; ``oracle/streamed.py`` calls it layer by
    layer so that the 32-layer model can be differentiated inside the build container's memory."""
    S, L, d = x.shape
    H, hd, Hkv = cfg.heads, cfg.head_dim, cfg.n_kv_heads
    p = f"model.layers.{i}."
    h = rms_norm(x, W[p + "input_layernorm.weight"], cfg.rms_eps)
    q = lora_linear(h, W, p + "self_attn.q_proj", lora_scale, lora_masks).view(S, L, H, hd).transpose(1, 2)
    k = lora_linear(h, W, p + "self_attn.k_proj", lora_scale, lora_masks).view(S, L, Hkv, hd).transpose(1, 2)
    v = lora_linear(h, W, p + "self_attn.v_proj", lora_scale, lora_masks).view(S, L, Hkv, hd).transpose(1, 2)
    q = q * cos + rotate_half(q) * sin
    k = k * cos + rotate_half(k) * sin
    if Hkv != H:        # HF repeat_kv: query head h attends key/value head h // (H / Hkv)
        k = k.repeat_interleave(H // Hkv, dim=1)
        v = v.repeat_interleave(H // Hkv, dim=1)
    att = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + causal
    att = torch.softmax(att.float(), dim=-1).to(q.dtype)
    a = (att @ v).transpose(1, 2).reshape(S, L, d)
    x = x + lora_linear(a, W, p + "self_attn.o_proj", lora_scale, lora_masks)
    h = rms_norm(x, W[p + "post_attention_layernorm.weight"], cfg.rms_eps)
    g = lora_linear(h, W, p + "mlp.gate_proj", lora_scale, lora_masks)
    u = lora_linear(h, W, p + "mlp.up_proj", lora_scale, lora_masks)
    return x + lora_linear(F.silu(g) * u, W, p + "mlp.down_proj", lora_scale, lora_masks)


def llama_tables(L: int, cfg: LlavaCfg, dtype):
    """(cos, sin, causal mask) shared by every layer: positions = arange(L) (position_ids dropped at
    llava_llama.py:94), pure causal mask, no pad mask (trainers.py:199)."""
    cos, sin = rope_tables(L, cfg.head_dim, cfg.rope_theta, dtype)
    return cos, sin, torch.full((L, L), float("-inf"), dtype=dtype).triu(1)


def llama_hidden(embeds: torch.Tensor, W: Dict[str, torch.Tensor], cfg: LlavaCfg,
                 n_layers: Optional[int] = None, lora_scale: Optional[float] = None,
                 lora_masks: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """Decoder stack + final norm."""
    cos, sin, causal = llama_tables(embeds.shape[1], cfg, embeds.dtype)
    x = embeds
    for i in range(cfg.layers if n_layers is None else n_layers):
        x = llama_layer(x, W, cfg, i, cos, sin, causal, lora_scale, lora_masks)
    return rms_norm(x, W["model.norm.weight"], cfg.rms_eps)


def llama_logits(embeds, W, cfg, lora_scale: Optional[float] = None, lora_masks=None):
    return F.linear(llama_hidden(embeds, W, cfg, lora_scale=lora_scale, lora_masks=lora_masks), W["lm_head.weight"]).float()


# --------------------------------------------------------------------------------------------
# log-probs and the DPO loss (the reference's own arithmetic)
# --------------------------------------------------------------------------------------------

def get_batch_logps(logits: torch.Tensor, labels: torch.Tensor, return_all: bool = False):
    """muffin/eval/muffin_inference_logp.py:82-115 (labels[:,1:] vs logits[:,:-1]; -100 masked;
    0/0 -> NaN average for a row without targets)."""
    labels = labels[:, 1:].clone()
    logits = logits[:, :-1, :]
    loss_mask = labels != IGNORE_INDEX
    labels[labels == IGNORE_INDEX] = 0
    per_token = torch.gather(logits.log_softmax(-1), 2, labels.unsqueeze(2)).squeeze(2)
    log_prob = (per_token * loss_mask).sum(-1)
    avg = log_prob / loss_mask.sum(-1)
    if return_all:
        return per_token, log_prob, avg
    return log_prob, avg


def get_batch_logps_minicpm(logits: torch.Tensor, labels: torch.Tensor, return_all: bool = False):
    """muffin/eval/muffin_inference_logp.py:21-52: the MiniCPM data pipeline pre-shifts its labels, so position t of
    the logits is scored against labels[:, t] (labels[:, :-1] vs logits[:, :-1])."""
    labels = labels[:, :-1].clone()
    logits = logits[:, :-1, :]
    loss_mask = labels != IGNORE_INDEX
    labels[labels == IGNORE_INDEX] = 0
    per_token = torch.gather(logits.log_softmax(-1), 2, labels.unsqueeze(2)).squeeze(2)
    log_prob = (per_token * loss_mask).sum(-1)
    avg = log_prob / loss_mask.sum(-1)
    if return_all:
        return per_token, log_prob, avg
    return log_prob, avg


def dpo_loss(pw, pr, rw, rr, beta: float, reference_free: bool = False):
    """muffin/train/trainers.py:91-126."""
    ref = 0 if reference_free else (rw - rr)
    z = (pw - pr) - ref
    losses = -F.logsigmoid(beta * z)
    return losses, beta * (pw - rw).detach(), beta * (pr - rr).detach()


def compute_weighted_logp(per_token_logp, labels, token_weight, use_average):
    """muffin/train/trainers.py:128-137."""
    loss_mask = labels[:, 1:].clone() != IGNORE_INDEX
    wm = token_weight * loss_mask
    logp = (per_token_logp * wm).sum(-1)
    return logp / wm.sum(-1) if use_average else logp


def dpo_step_forward(batch: Dict[str, object], W: Dict[str, torch.Tensor], cfg: LlavaCfg,
                     dpo_use_average: bool = False, sft_weight: Optional[float] = None,
                     dpo_weight: Optional[float] = None, lora_scale: Optional[float] = None,
                     lora_masks: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """get_beta_and_logps (trainers.py:161-275, is_llava15 branch) + compute_loss
    (trainers.py:281-311).  Images are encoded for [images, images] like the reference
    (trainers.py:190); rows i and B+i of the features are identical."""
    images = batch["images"]
    cat_images = torch.cat([images, images], dim=0)
    feats = encode_images(cat_images, W, cfg)
    embeds, new_labels = prepare_inputs_labels_for_multimodal(
        batch["concatenated_input_ids"], batch["concatenated_labels"], feats,
        W["model.embed_tokens.weight"], cfg.model_max_length)
    logits = llama_logits(embeds, W, cfg, lora_scale=lora_scale, lora_masks=lora_masks)
    per_token, log_prob, avg = get_batch_logps(logits, new_labels, return_all=True)
    cat = avg if dpo_use_average else log_prob
    B = batch["win_input_ids"].shape[0]
    pw, pr = cat.split([B, B])
    rw = batch["ref_win_avg_logp"] if dpo_use_average else batch["ref_win_logp"]
    rr = batch["ref_rej_avg_logp"] if dpo_use_average else batch["ref_rej_logp"]
    losses, cw, cr = dpo_loss(pw, pr, rw, rr, batch["beta"])
    acc = (cw > cr).float()
    sft = float(os.environ.get("SFT_weight", 0.0)) if sft_weight is None else sft_weight   # trainers.py:299-300
    dpo = float(os.environ.get("DPO_weight", 1.0)) if dpo_weight is None else dpo_weight
    loss = dpo * losses.mean() - sft * pw.mean()
    return dict(loss=loss, losses=losses, chosen_rewards=cw, rejected_rewards=cr,
                reward_accuracies=acc, policy_win_logp=pw, policy_rej_logp=pr,
                per_token_logps=per_token, log_prob=log_prob, average_log_prob=avg,
                labels=new_labels, embeds=embeds, image_features=feats)


def emulate_bf16(batch: Dict[str, object], W: Dict[str, torch.Tensor]):
    """Calibration leg: the SAME restatement evaluated the way the reference runs under ``--bf16`` (model loaded in
    bf16, train_llava15.py:203; tower forced to bf16, :242; HF 4.35 rounds every module output to bf16, keeps RMSNorm
    statistics / softmax / rotary tables in fp32 and upcasts the logits AFTER the bf16 lm_head GEMM): weights and pixels
    are cast to bf16, every function above is dtype-generic and follows its input dtype (CPU bf16 matmuls accumulate in
    fp32 like the GPU GEMMs).  Its distance from the fp32 run is what "bf16 tolerance" means for this model and batch;
    the HIP path (fp32 norms / RoPE / log-softmax, one rounding per op) is asserted to sit no further from fp32."""
    Wb = {k: v.detach().to(torch.bfloat16) for k, v in W.items()}
    bb = dict(batch)
    bb["images"] = batch["images"].to(torch.bfloat16)
    return bb, Wb


def preference_metrics(out: Dict[str, torch.Tensor], batch, task: str = "train") -> Dict[str, float]:
    """collect_preference_metrics (trainers.py:140-158), single process."""
    t = task
    m = {
        f"rewards_{t}/chosen": out["chosen_rewards"].mean().item(),
        f"rewards_{t}/rejected": out["rejected_rewards"].mean().item(),
        f"logps_{t}/rejected": out["policy_rej_logp"].mean().item(),
        f"logps_{t}/chosen": out["policy_win_logp"].mean().item(),
        f"logps_{t}/ref_rejected": batch["ref_rej_logp"].mean().item(),
        f"logps_{t}/ref_chosen": batch["ref_win_logp"].mean().item(),
        f"rewards_{t}/accuracies": out["reward_accuracies"].mean().item(),
    }
    m[f"rewards_{t}/margins"] = m[f"rewards_{t}/chosen"] - m[f"rewards_{t}/rejected"]
    return m


# --------------------------------------------------------------------------------------------
# optimizer (HF Trainer adamw_torch defaults; train_llava15.py:75, llava15_train.sh:31-34)
# --------------------------------------------------------------------------------------------

def is_decay_param(name: str) -> bool:
    """HF get_decay_parameter_names: no weight decay on norm gains (parameters of LayerNorm / RMSNorm modules - the Llama
    ``*norm`` gains and the OmniLMM Resampler's ``ln_q`` / ``ln_kv`` / ``ln_post``) and biases."""
    return not (name.endswith("bias") or "norm" in name or ".ln_" in name)


def trainable_names(cfg: LlavaCfg) -> List[str]:
    """fully_tune (train_llava15.py:268-269): everything except the CLIP tower, which runs under
    no_grad (clip_encoder.py:46) and is treated as frozen (SURVEY.md section 8a notes)."""
    return [k for k in weight_shapes(cfg) if not k.startswith(VT)]


def cosine_lr(step: int, total: int, base_lr: float, warmup_ratio: float = 0.05) -> float:
    """get_cosine_schedule_with_warmup; step = number of optimizer steps already taken."""
    warm = math.ceil(total * warmup_ratio)
    if step < warm:
        return base_lr * step / max(1, warm)
    prog = (step - warm) / max(1, total - warm)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


def adamw_reference(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor],
                    state: Dict[str, Dict[str, torch.Tensor]], lr: float, step: int,
                    betas=(0.9, 0.999), eps=1e-8, wd=0.01, max_grad_norm: Optional[float] = 1.0):
    """torch.optim.AdamW single step (decoupled decay) preceded by clip_grad_norm_.
    Returns the pre-clip global grad norm."""
    tot = math.sqrt(sum(float(g.double().pow(2).sum()) for g in grads.values()))
    scale = 1.0
    if max_grad_norm is not None:
        scale = min(1.0, max_grad_norm / (tot + 1e-6))
    b1, b2 = betas
    for k, p in params.items():
        if k not in grads:
            continue
        g = grads[k].float() * scale
        st = state.setdefault(k, dict(m=torch.zeros_like(p), v=torch.zeros_like(p)))
        if is_decay_param(k):
            p.mul_(1.0 - lr * wd)
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st["m"], denom, value=-lr / bc1)
    return tot


def dpo_train_step(batch, W: Dict[str, torch.Tensor], cfg: LlavaCfg, opt_state, lr: float, step: int,
                   timings: Optional[Dict[str, float]] = None, **kw):
    """One full optimisation step on CPU: forward, autograd backward, clip, AdamW.  This is the
    <=30-line shim loop of SURVEY.md section 8c around the reference functions."""
    names = lora_trainable_names(W) if kw.get("lora_scale") is not None else trainable_names(cfg)
    for k in names:
        W[k].requires_grad_(True)
        W[k].grad = None
    import time
    t0 = time.time()
    out = dpo_step_forward(batch, W, cfg, **kw)
    t1 = time.time()
    out["loss"].backward()
    t2 = time.time()
    grads = {k: W[k].grad.detach().clone() for k in names if W[k].grad is not None}
    with torch.no_grad():
        params = {k: W[k] for k in names}
        for k in names:
            W[k].requires_grad_(False)
        gn = adamw_reference(params, grads, opt_state, lr, step)
    if timings is not None:            # forward, backward and optimizer timed separately
        timings.update(fwd_s=t1 - t0, bwd_s=t2 - t1, opt_s=time.time() - t2)
    return out, grads, gn
