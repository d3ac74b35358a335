"""CPU restatement of the device's counter-based LoRA dropout mask.  TEST INFRASTRUCTURE ONLY.

peft's ``lora_dropout`` (muffin/train/train_llava15_lora.py:115, p = 0.05) is ``nn.Dropout`` on the adapter branch input;
the HIP path draws its mask from a counter hash of (seed, element index) (rlaif-v_amd/csrc/elementwise.hip ``dropout_chunk`` /
``rv_dropout``) so that backward can regenerate it.  The full-depth LoRA parity case (tests/full_depth.py ``cfg5_drop``) runs
the fp32 oracle in the build container - where no GPU exists - on EXACTLY the masks the device will draw, so the hash is
restated here in integer numpy; ``tests/test_lora_gpu.py::test_dropout_mask_restatement_bit_exact`` pins it bit-exactly
against ``rv_dropout`` on the GPU.

    element e of the contiguous [rows, width] tensor:  chunk i8 = e >> 3, j = (e & 7) >> 1
    h     = mix32((uint32(i8) * 4 + j) ^ (uint32(i8 >> 30) * 0x9e3779b9 + key)),  key = uint32(seed) * 0x9e3779b9 + 0x85ebca6b
    bits  = h & 0xffff for even e, h >> 16 for odd e;   keep  <=>  bits >= round(p * 65536)
"""
from __future__ import annotations

import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & _M32
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & _M32
    h ^= h >> np.uint64(16)
    return h


def keep_mask(n: int, p: float, seed: int, chunk: int = 1 << 24) -> np.ndarray:
    """bool[n]: True where rv_dropout(x, p, seed) keeps element e (n % 8 == 0 like the kernel requires)."""
    assert n % 8 == 0
    if n < (1 << 33):         # i8 >> 30 == 0 and the pair index fits 32 bits: the whole hash in wrapping uint32 array arithmetic (25 x faster)
        key32 = np.uint32(((int(seed) & 0xFFFFFFFF) * 0x9E3779B9 + 0x85EBCA6B) & 0xFFFFFFFF)
        h = np.arange(n // 2, dtype=np.uint32) ^ key32
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x7FEB352D)
        h ^= h >> np.uint32(15)
        h *= np.uint32(0x846CA68B)
        h ^= h >> np.uint32(16)
        t32 = np.uint32(int(p * 65536.0 + 0.5))
        out = np.empty(n, dtype=bool)
        out[0::2] = (h & np.uint32(0xFFFF)) >= t32
        out[1::2] = (h >> np.uint32(16)) >= t32
        return out
    thresh = np.uint64(int(p * 65536.0 + 0.5))
    key = np.uint64((int(seed) & 0xFFFFFFFF) * 0x9E3779B9 + 0x85EBCA6B) & _M32
    out = np.empty(n, dtype=bool)
    for q0 in range(0, n // 2, chunk):                    # one hash per PAIR of elements
        q = np.arange(q0, min(q0 + chunk, n // 2), dtype=np.uint64)
        i8 = q >> np.uint64(2)
        base = (((i8 >> np.uint64(30)) * np.uint64(0x9E3779B9)) + key) & _M32
        h = _mix32((((i8 & _M32) * np.uint64(4) + (q & np.uint64(3))) & _M32) ^ base)
        out[2 * q0:2 * q0 + 2 * q.size:2] = (h & np.uint64(0xFFFF)) >= thresh
        out[2 * q0 + 1:2 * q0 + 2 * q.size:2] = (h >> np.uint64(16)) >= thresh
    return out


def multiplier(rows: int, width: int, p: float, seed: int) -> torch.Tensor:
    """fp32 [rows, width] multiplier keep / (1 - p): what ``oracle.dpo_oracle.lora_linear`` takes as ``masks[name]``.
    The device multiplies by the fp32 value 1.f / (1.f - p); the same fp32 constant is used here."""
    inv_keep = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return torch.from_numpy(keep_mask(rows * width, p, seed).reshape(rows, width).astype(np.float32) * inv_keep)


def model_seed(step: int, rank: int, layer: int, slot: int) -> int:
    """rlaif-v_amd/model.py ``LlavaDPOModel._dropout_seed``: slot 0 = q|k|v input, 1 = o input, 2 = gate|up input, 3 = down input."""
    return (step * 1000003 + rank * 7919 + layer * 8 + slot) & 0x7FFFFFFF


SLOTS = ((("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"), "hidden"), (("self_attn.o_proj",), "hidden"),
         (("mlp.gate_proj", "mlp.up_proj"), "hidden"), (("mlp.down_proj",), "ffn"))


MODULE_SLOTS = {"self_attn.q_proj": 0, "self_attn.o_proj": 1, "mlp.gate_proj": 2, "mlp.down_proj": 3,
                "self_attn.k_proj": 4, "self_attn.v_proj": 5, "mlp.up_proj": 6}


def layer_masks(layer: int, rows: int, hidden: int, ffn: int, p: float, step: int = 1, rank: int = 0, per_module: bool = False):
    """{module name: multiplier [rows, in]} of one decoder layer as the HIP model draws them in its ``step``-th training forward.
    Default: q / k / v and gate / up share one mask (the fused projection drops its input once - DESIGN section 5, conscious
    deviation).  ``per_module`` (RV_LORA_PEFT_MASKS=1, ``LlavaDPOModel._module_seeds``): every peft module its own mask, like peft's
    one nn.Dropout per lora.Linear - k, v and up draw from the layer's spare seed slots 4, 5, 6."""
    out = {}
    if per_module:
        for mod, slot in MODULE_SLOTS.items():
            out[f"model.layers.{layer}.{mod}"] = multiplier(rows, ffn if mod == "mlp.down_proj" else hidden, p, model_seed(step, rank, layer, slot))
        return out
    for slot, (mods, w) in enumerate(SLOTS):
        m = multiplier(rows, hidden if w == "hidden" else ffn, p, model_seed(step, rank, layer, slot))
        for mod in mods:
            out[f"model.layers.{layer}.{mod}"] = m
    return out
