"""HF-layout checkpoint I/O.

The reference saves ``trainer.model.state_dict()`` through HF's ``_save`` (muffin/train/train_llava15.py:102-112) and
loads released checkpoints with ``from_pretrained`` (llava/model/builder.py:26-167).  Here the same key names
(``model.layers.N.self_attn.q_proj.weight`` ..., ``model.mm_projector.*``,
``model.vision_tower.vision_tower.vision_model.*``, ``lm_head.weight``) are written as sharded safetensors with the
standard ``model.safetensors.index.json`` + ``config.json``, so the output loads in the reference's chat / eval code,
and a released LLaVA-1.5 checkpoint directory (safetensors or ``pytorch_model*.bin``) loads into ``LlavaDPOModel``.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Iterable, Tuple

import torch


def hf_config_dict(cfg) -> Dict:
    """LlavaConfig (HF ``llava_llama``) fields of the checkpoint's config.json; for the OmniLMM branch the fields
    ``OmniLMMModel.initialize_vision_modules`` writes into its Mistral config (omnilmm/model/omnilmm.py:76-80)."""
    if getattr(cfg, "arch", "llava") == "omnilmm":
        return {
            "architectures": ["OmniLMMForCausalLM"], "model_type": "omnilmm",
            "hidden_size": cfg.hidden, "intermediate_size": cfg.ffn, "num_hidden_layers": cfg.layers,
            "num_attention_heads": cfg.heads, "num_key_value_heads": cfg.n_kv_heads, "vocab_size": cfg.vocab,
            "rms_norm_eps": cfg.rms_eps, "rope_theta": cfg.rope_theta, "max_position_embeddings": 32768,
            "sliding_window": 4096, "pad_token_id": cfg.pad_token_id, "bos_token_id": 1, "eos_token_id": 2,
            "hidden_act": "silu", "torch_dtype": "bfloat16", "tie_word_embeddings": False, "use_cache": True,
            "mm_vision_tower": "eva02_enormous_patch14_clip_224.laion2b_plus", "use_mm_proj": True,
            "num_query": cfg.num_query, "image_size": cfg.image_size,
            # not HF fields: what this package needs to rebuild the model without a tokenizer
            "rlaifv_vision_width": cfg.vision_width, "rlaifv_im_patch_token": cfg.im_patch_token,
            "rlaifv_im_start_token": cfg.im_start_token, "rlaifv_im_end_token": cfg.im_end_token,
            "tokenizer_model_max_length": cfg.model_max_length,
        }
    return {
        "architectures": ["LlavaLlamaForCausalLM"], "model_type": "llava_llama",
        "hidden_size": cfg.hidden, "intermediate_size": cfg.ffn, "num_hidden_layers": cfg.layers,
        "num_attention_heads": cfg.heads, "num_key_value_heads": cfg.n_kv_heads, "vocab_size": cfg.vocab,
        "rms_norm_eps": cfg.rms_eps, "rope_theta": cfg.rope_theta, "max_position_embeddings": 4096,
        "pad_token_id": cfg.pad_token_id, "bos_token_id": 1, "eos_token_id": 2, "hidden_act": "silu",
        "torch_dtype": "bfloat16", "tie_word_embeddings": False, "use_cache": True,
        "mm_vision_tower": "openai/clip-vit-large-patch14-336", "mm_projector_type": "mlp2x_gelu",
        "mm_hidden_size": cfg.clip_hidden, "mm_vision_select_layer": cfg.select_layer,
        "mm_vision_select_feature": "patch", "mm_patch_merge_type": "flat", "image_aspect_ratio": "pad",
        "tokenizer_model_max_length": cfg.model_max_length, "tokenizer_padding_side": "right",
        "mm_use_im_start_end": False, "mm_use_im_patch_token": False, "use_mm_proj": True,
    }


def config_from_hf(d: Dict, **overrides):
    from .model import LlavaConfig
    if d.get("model_type") == "omnilmm":
        from .omnilmm import OmniLMMConfig
        kw = dict(hidden=d["hidden_size"], layers=d["num_hidden_layers"], heads=d["num_attention_heads"],
                  kv_heads=d.get("num_key_value_heads", d["num_attention_heads"]), ffn=d["intermediate_size"],
                  vocab=d["vocab_size"], rms_eps=d.get("rms_norm_eps", 1e-5), rope_theta=d.get("rope_theta", 10000.0),
                  num_query=d.get("num_query", 64), image_size=d.get("image_size", 448),
                  vision_width=d.get("rlaifv_vision_width", 1792), im_patch_token=d.get("rlaifv_im_patch_token", 32000),
                  im_start_token=d.get("rlaifv_im_start_token", 32001), im_end_token=d.get("rlaifv_im_end_token", 32002),
                  model_max_length=d.get("tokenizer_model_max_length", 2048), pad_token_id=d.get("pad_token_id") or 0)
        kw.update(overrides)
        return OmniLMMConfig(**kw)
    kw = dict(hidden=d["hidden_size"], layers=d["num_hidden_layers"], heads=d["num_attention_heads"],
              ffn=d["intermediate_size"], vocab=d["vocab_size"], rms_eps=d.get("rms_norm_eps", 1e-5),
              rope_theta=d.get("rope_theta", 10000.0), clip_hidden=d.get("mm_hidden_size", 1024),
              select_layer=d.get("mm_vision_select_layer", -2),
              model_max_length=d.get("tokenizer_model_max_length", 2048), pad_token_id=d.get("pad_token_id") or 0)
    if d.get("num_key_value_heads", d["num_attention_heads"]) != d["num_attention_heads"]:
        kw["kv_heads"] = d["num_key_value_heads"]           # Mistral / Llama-3 style grouped-query attention
    kw.update(overrides)
    return LlavaConfig(**kw)


def save_state_dict_sharded(sd: Dict[str, torch.Tensor], output_dir: str, cfg=None,
                            max_shard_bytes: int = 5 << 30) -> Dict:
    """Write ``sd`` (CPU tensors, HF names) as model-XXXXX-of-YYYYY.safetensors + index (+ config.json)."""
    from safetensors.torch import save_file
    os.makedirs(output_dir, exist_ok=True)
    shards, cur, cur_bytes = [], {}, 0
    for k in sd:                                     # keep the caller's (layer) order inside shards
        t = sd[k].detach().contiguous().cpu()
        nbytes = t.numel() * t.element_size()
        if cur and cur_bytes + nbytes > max_shard_bytes:
            shards.append(cur)
            cur, cur_bytes = {}, 0
        cur[k] = t
        cur_bytes += nbytes
    if cur:
        shards.append(cur)
    index = {"metadata": {"total_size": sum(v.numel() * v.element_size() for s in shards for v in s.values())},
             "weight_map": {}}
    for i, shard in enumerate(shards):
        name = "model.safetensors" if len(shards) == 1 else f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(shard, os.path.join(output_dir, name), metadata={"format": "pt"})
        for k in shard:
            index["weight_map"][k] = name
    if len(shards) > 1:
        with open(os.path.join(output_dir, "model.safetensors.index.json"), "w") as f:
            json.dump(index, f, indent=2)
    if cfg is not None:
        with open(os.path.join(output_dir, "config.json"), "w") as f:
            json.dump(hf_config_dict(cfg), f, indent=2)
    return index


def load_state_dict_dir(ckpt_dir: str) -> Dict[str, torch.Tensor]:
    """All tensors of an HF checkpoint directory (safetensors shards preferred, else pytorch_model*.bin)."""
    sd: Dict[str, torch.Tensor] = {}
    st_files = sorted(glob.glob(os.path.join(ckpt_dir, "*.safetensors")))
    if st_files:
        from safetensors.torch import load_file
        for f in st_files:
            sd.update(load_file(f))
        return sd
    bin_files = sorted(glob.glob(os.path.join(ckpt_dir, "pytorch_model*.bin")))
    if not bin_files:
        raise FileNotFoundError(f"no *.safetensors or pytorch_model*.bin under {ckpt_dir}")
    for f in bin_files:
        sd.update(torch.load(f, map_location="cpu", weights_only=True))
    return sd


def save_pretrained(model, output_dir: str, max_shard_bytes: int = 5 << 30):
    """Trainable part (LLM + projector) + the frozen CLIP tower under the reference's key names."""
    sd = model.state_dict()
    sd.update(model.clip_state_dict())
    return save_state_dict_sharded(sd, output_dir, model.cfg, max_shard_bytes)


def save_lora_adapter(model, output_dir: str, base_model_name_or_path: str = ""):
    """What the LoRA entry point leaves in a checkpoint directory (safe_save_model_for_hf_trainer of
    muffin/train/train_llava15_lora.py:184-197 + script/train/llava15_train_lora.sh:51-70): the peft adapter
    (adapter_model.safetensors + adapter_config.json), non_lora_trainables.bin (the projector) and config.json - the
    layout llava/model/builder.py:52-85 loads and merges."""
    from safetensors.torch import save_file
    if model.lora is None:
        raise ValueError("save_lora_adapter: the model has no adapters")
    os.makedirs(output_dir, exist_ok=True)
    sd = {k: v.contiguous() for k, v in model.lora_state_dict().items()}
    save_file(sd, os.path.join(output_dir, "adapter_model.safetensors"), metadata={"format": "pt"})
    lc = model.lora
    with open(os.path.join(output_dir, "adapter_config.json"), "w") as f:
        json.dump({"peft_type": "LORA", "task_type": "CAUSAL_LM", "r": lc.r, "lora_alpha": lc.lora_alpha,
                   "lora_dropout": lc.lora_dropout, "bias": lc.bias, "target_modules": sorted(lc.target_modules),
                   "base_model_name_or_path": base_model_name_or_path, "inference_mode": True,
                   "fan_in_fan_out": False, "init_lora_weights": True, "modules_to_save": None}, f, indent=2)
    torch.save(model.non_lora_trainables(), os.path.join(output_dir, "non_lora_trainables.bin"))
    with open(os.path.join(output_dir, "config.json"), "w") as f:
        json.dump(hf_config_dict(model.cfg), f, indent=2)


def load_lora_adapter(model, adapter_dir: str, merge: bool = False):
    """llava/model/builder.py:52-85: non_lora_trainables.bin into the base model, then the adapter; ``merge`` applies
    peft's merge_and_unload (W += (alpha/r) B A) on the device."""
    from safetensors.torch import load_file
    nl = os.path.join(adapter_dir, "non_lora_trainables.bin")
    if os.path.exists(nl):
        extra = torch.load(nl, map_location="cpu", weights_only=True)
        for k, v in extra.items():
            k = k[len("base_model.model."):] if k.startswith("base_model.model.") else k       # builder.py:76-78
            if k in model.store.offsets:
                model.store.p(k).copy_(v.to(torch.bfloat16))
    f_st = os.path.join(adapter_dir, "adapter_model.safetensors")
    sd = load_file(f_st) if os.path.exists(f_st) else torch.load(os.path.join(adapter_dir, "adapter_model.bin"),
                                                                map_location="cpu", weights_only=True)
    model.load_lora_state_dict(sd, strict=True)
    model.store.refresh_transposes()
    if merge:
        model.merge_lora()
    return model


def lora_config_from_dir(adapter_dir: str):
    from .model import LoraConfig
    with open(os.path.join(adapter_dir, "adapter_config.json")) as f:
        d = json.load(f)
    return LoraConfig(r=d["r"], lora_alpha=d["lora_alpha"], lora_dropout=d.get("lora_dropout", 0.0),
                      bias=d.get("bias", "none"), target_modules=tuple(d["target_modules"]))


def from_pretrained(ckpt_dir: str, device="cuda:0", with_optimizer: bool = True, lora=None, **cfg_overrides):
    """Build a LlavaDPOModel from an HF LLaVA-1.5 checkpoint directory (vision tower weights may live in the same
    directory - as our own save_pretrained writes them - or in ``vision_tower_dir``).  ``lora``: a LoraConfig wraps
    the loaded base model with fresh adapters (get_peft_model, train_llava15_lora.py:304-318)."""
    from .model import LlavaDPOModel
    vt_dir = cfg_overrides.pop("vision_tower_dir", None)
    with open(os.path.join(ckpt_dir, "config.json")) as f:
        cfg = config_from_hf(json.load(f), **cfg_overrides)
    sd = load_state_dict_dir(ckpt_dir)
    if vt_dir is not None:
        vt = load_state_dict_dir(vt_dir)
        sd.update({("model.vision_tower.vision_tower." + k if not k.startswith("model.") else k): v for k, v in vt.items()})
    if getattr(cfg, "arch", "llava") == "omnilmm":
        # OmniLMMForCausalLM checkpoint: language model + model.resampler.*; tower tensors (model.vision_tower.*, timm names)
        # go to the unpinned EvaTower when they are present, else the caller feeds precomputed tower tokens
        from .omnilmm import OmniLMMDPOModel
        model = OmniLMMDPOModel(cfg, device=device, with_optimizer=with_optimizer, lora=lora)
        model.load_state_dict(sd)
        if any(k.startswith("model.vision_tower.blocks.") for k in sd):
            from .eva_tower import EvaConfig, EvaTower
            tower = EvaTower(EvaConfig(width=cfg.vision_width), device=device)
            tower.load_state_dict(sd, prefix="model.vision_tower.")
            model.set_vision_tower(tower)
        return model
    model = LlavaDPOModel(cfg, device=device, with_optimizer=with_optimizer, lora=lora)
    model.load_state_dict(sd)
    return model
