"""Where does a bf16 forward's per-token log-prob error come from?  TEST INFRASTRUCTURE ONLY (same rules as dpo_oracle.py).

VERDICT r4 weak 1: at 32 layers the HIP path's per-token log-probs sit 0.0435 nats (RMS) from the fp32 oracle, which puts one
sigma of BASELINE config 1's saturated loss at 4.9e-3 of its value - the north_star 1e-3 bar is met by one draw of rounding
noise.  This module evaluates the oracle's decoder layer (``dpo_oracle.llama_layer``: HF LlamaDecoderLayer, reached through
llava/model/language_model/llava_llama.py:91-102) in fp32 WITH EXPLICIT bf16 ROUNDINGS at the points where the HIP path stores
bf16 (rlaif-v_amd/model.py ``_layer_forward``), each group of points selectable, so that the error can be attributed on the
CPU before anything is built on the device:

    R  residual stream: x after the attention residual add and after the MLP residual add (the GEMM epilogues add the
       residual in fp32 and store bf16)
    N  RMSNorm outputs (fp32 statistics, bf16 store)
    Q  q / k / v projection outputs (bf16 store) and the rotated q / k (RoPE in fp32, second bf16 store)
    P  attention probabilities as the PV MFMA's bf16 operand
    A  attention output (bf16 store)
    G  gate / up projection outputs and silu(gate) * up (bf16 stores)
    F  the final norm's output in front of the LM head
    V  the vision front: CLIP tower, projector and embedding rows evaluated in bf16 (``emulate_bf16`` style)
       ... or, resolved further (lower-case letters, explicit roundings in an fp32 evaluation like the decoder's):
    c  the CLIP tower's residual stream (x after pre_layrnorm and after each of the 2 x 23 residual adds)
    o  the CLIP tower's other bf16 stores (patch embedding, LayerNorm outputs, q / k / v, attention output, quick_gelu(fc1))
    j  the projector's stores (first linear, GELU, output features)

Weights are bf16-representable in every variant (they are on both sides of every parity test), accumulation is fp32
everywhere, softmax / norm statistics / log-softmax are fp32 - exactly the HIP path's arithmetic.  ``tools/rounding_attribution.py``
runs the variants at 32 layers through ``oracle/streamed.py`` (``layer_fn`` / ``hidden_fn`` hooks).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import dpo_oracle as O

ALL_POINTS = "RNQPAGFV"
VISION_RESOLVED = "coj"          # V resolved: never combined with V itself


def _bf(x: torch.Tensor, on: bool) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32) if on else x


def make_layer_fn(points: str):
    """``dpo_oracle.llama_layer`` (same signature, same arithmetic, fp32) with bf16 roundings at the selected points."""
    R, N, Q, P, A, G = (c in points for c in "RNQPAG")

    def layer(x, W, cfg, i, cos, sin, causal, lora_scale=None, lora_masks=None):
        S, L, d = x.shape
        H, hd, Hkv = cfg.heads, cfg.head_dim, cfg.n_kv_heads
        p = f"model.layers.{i}."
        h = _bf(O.rms_norm(x, W[p + "input_layernorm.weight"], cfg.rms_eps), N)
        q = _bf(O.lora_linear(h, W, p + "self_attn.q_proj", lora_scale, lora_masks), Q).view(S, L, H, hd).transpose(1, 2)
        k = _bf(O.lora_linear(h, W, p + "self_attn.k_proj", lora_scale, lora_masks), Q).view(S, L, Hkv, hd).transpose(1, 2)
        v = _bf(O.lora_linear(h, W, p + "self_attn.v_proj", lora_scale, lora_masks), Q).view(S, L, Hkv, hd).transpose(1, 2)
        q = _bf(q * cos + O.rotate_half(q) * sin, Q)
        k = _bf(k * cos + O.rotate_half(k) * sin, Q)
        if Hkv != H:
            k = k.repeat_interleave(H // Hkv, dim=1)
            v = v.repeat_interleave(H // Hkv, dim=1)
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd) + causal, dim=-1)
        if P:
            # the kernel rounds the UN-normalised exp(s - running max) to bf16 and divides the fp32 accumulator by the fp32 row sum
            # at the end; rounding p / max(p) (in [0, 1] like exp(s - m)) and renormalising by the fp32 sum is that arithmetic
            mx = att.amax(-1, keepdim=True)
            a = (_bf(att / mx, True) @ v) * mx
        else:
            a = att @ v
        a = _bf(a.transpose(1, 2).reshape(S, L, d), A)
        x = _bf(x + O.lora_linear(a, W, p + "self_attn.o_proj", lora_scale, lora_masks), R)
        h = _bf(O.rms_norm(x, W[p + "post_attention_layernorm.weight"], cfg.rms_eps), N)
        g = _bf(O.lora_linear(h, W, p + "mlp.gate_proj", lora_scale, lora_masks), G)
        u = _bf(O.lora_linear(h, W, p + "mlp.up_proj", lora_scale, lora_masks), G)
        act = _bf(F.silu(g) * u, G)
        return _bf(x + O.lora_linear(act, W, p + "mlp.down_proj", lora_scale, lora_masks), R)

    return layer


def make_hidden_fn(points: str):
    on = "F" in points
    return lambda hidden: _bf(hidden, on)


class RoundedLlavaFront:
    """``streamed.LlavaFront`` with the V group: tower, projector and embedding rows evaluated in bf16 (module outputs rounded the way
    ``dpo_oracle.emulate_bf16`` does it: the functions follow their input dtype), handed to the fp32 decoder as fp32 values."""

    def __init__(self, batch, cfg: O.LlavaCfg, W: Dict[str, torch.Tensor], points: str):
        self.batch, self.cfg, self.v = batch, cfg, "V" in points
        images = batch["images"]
        with torch.no_grad():
            if self.v:
                Wb = {k: t.to(torch.bfloat16) for k, t in W.items() if k.startswith(O.VT) or "mm_projector" in k}
                Wb["model.embed_tokens.weight"] = W["model.embed_tokens.weight"].to(torch.bfloat16)
                tower = O.clip_vision_features(torch.cat([images, images], 0).to(torch.bfloat16), Wb, cfg)
                feats = O.mm_projector(tower, Wb)
                x, labels = O.prepare_inputs_labels_for_multimodal(batch["concatenated_input_ids"], batch["concatenated_labels"], feats,
                                                                   Wb["model.embed_tokens.weight"], cfg.model_max_length)
                self.out = (x.float(), labels, feats.float())
            else:
                tower = O.clip_vision_features(torch.cat([images, images], 0), W, cfg)
                feats = O.mm_projector(tower, W)
                x, labels = O.prepare_inputs_labels_for_multimodal(batch["concatenated_input_ids"], batch["concatenated_labels"], feats,
                                                                   W["model.embed_tokens.weight"], cfg.model_max_length)
                self.out = (x, labels, feats)
        self.names = []

    def __call__(self, W):
        return self.out


def clip_features_rounded(pixels: torch.Tensor, W: Dict[str, torch.Tensor], cfg: O.LlavaCfg, points: str) -> torch.Tensor:
    """``dpo_oracle.clip_vision_features`` (llava/model/multimodal_encoder/clip_encoder.py:36-58 over HF CLIPVisionModel) in fp32 with
    bf16 roundings at the HIP tower's store points (rlaif-v_amd/model.py ``clip_features``): ``c`` residual stream, ``o`` the rest."""
    c, o = "c" in points, "o" in points
    VT = O.VT
    B = pixels.shape[0]
    cd, H, hd = cfg.clip_hidden, cfg.clip_heads, cfg.clip_head_dim
    x = _bf(F.conv2d(pixels, W[VT + "embeddings.patch_embedding.weight"], None, stride=cfg.patch), o)
    x = x.flatten(2).transpose(1, 2)
    x = _bf(torch.cat([W[VT + "embeddings.class_embedding"].expand(B, 1, cd), x], 1) + W[VT + "embeddings.position_embedding.weight"][None], o)
    x = _bf(O._ln(x, W[VT + "pre_layrnorm.weight"], W[VT + "pre_layrnorm.bias"], cfg.clip_eps), c)
    T = x.shape[1]
    for i in range(cfg.clip_layers_used):
        p = VT + f"encoder.layers.{i}."
        h = _bf(O._ln(x, W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"], cfg.clip_eps), o)
        q = _bf(F.linear(h, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"]), o).view(B, T, H, hd).transpose(1, 2)
        k = _bf(F.linear(h, W[p + "self_attn.k_proj.weight"], W[p + "self_attn.k_proj.bias"]), o).view(B, T, H, hd).transpose(1, 2)
        v = _bf(F.linear(h, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"]), o).view(B, T, H, hd).transpose(1, 2)
        a = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1) @ v
        a = _bf(a.transpose(1, 2).reshape(B, T, cd), o)
        x = _bf(x + F.linear(a, W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"]), c)
        h = _bf(O._ln(x, W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"], cfg.clip_eps), o)
        h = F.linear(h, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"])
        h = _bf(h * torch.sigmoid(1.702 * h), o)
        x = _bf(x + F.linear(h, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"]), c)
    return _bf(x[:, 1:], c or o)          # the features handed to the projector are a bf16 GEMM operand either way


class ResolvedLlavaFront:
    """``streamed.LlavaFront`` with the vision front's rounding points resolved (c / o / j, see the module docstring)."""

    def __init__(self, batch, cfg: O.LlavaCfg, W: Dict[str, torch.Tensor], points: str):
        j = "j" in points
        images = batch["images"]
        with torch.no_grad():
            tower = clip_features_rounded(torch.cat([images, images], 0), W, cfg, points)
            h = _bf(F.linear(tower, W["model.mm_projector.0.weight"], W["model.mm_projector.0.bias"]), j)
            h = _bf(F.gelu(h), j)
            feats = _bf(F.linear(h, W["model.mm_projector.2.weight"], W["model.mm_projector.2.bias"]), j)
            x, labels = O.prepare_inputs_labels_for_multimodal(batch["concatenated_input_ids"], batch["concatenated_labels"], feats,
                                                               W["model.embed_tokens.weight"], cfg.model_max_length)
        self.out = (x, labels, feats)
        self.names = []

    def __call__(self, W):
        return self.out
