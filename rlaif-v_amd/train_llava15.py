"""Entry point with the reference's flag surface: muffin/train/train_llava15.py (+ train_llava15_lora.py), as launched by
script/train/llava15_train.sh / llava15_train_lora.sh.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        -m rlaif_v_amd.train_llava15 --model_name_or_path <llava-v1.5-7b dir> --data_dir ./RLAIF-V-Dataset_logps/ \
        --vision_tower <clip-vit-large-patch14-336 dir> --task DPO --bf16 True --model_max_length 2048 --dpo_beta 0.1 ...

Same three argument dataclasses (ModelArguments :33-46, DataArguments :49-69, TrainingArguments :72-100 + the LoRA flags
of train_llava15_lora.py:111-116), same flow: init_model (:198-290) -> make_dpo_data_module (:148-195) ->
LLaVA15DPOTrainer -> train / resume (:320-331) -> save_state + safe_save_model_for_hf_trainer (:332-334).  Flags that
only steer the HF / DeepSpeed control plane (--deepspeed, --report_to, --tf32, --save_total_limit ...) are accepted and
ignored; one process per GPU comes from torch.distributed.run instead of the deepspeed launcher."""
from __future__ import annotations

import argparse
import dataclasses
import glob
import os
import sys
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

from .trainer import LLaVA15DPOTrainer
from .trainer import TrainingArguments as _CoreTrainingArguments


@dataclass
class ModelArguments:
    model_name_or_path: Optional[str] = "facebook/opt-125m"
    version: Optional[str] = "llava_v1"
    freeze_backbone: bool = False
    tune_mm_mlp_adapter: bool = False
    vision_tower: Optional[str] = None
    mm_vision_select_layer: Optional[int] = -1
    pretrain_mm_mlp_adapter: Optional[str] = None
    mm_projector_type: Optional[str] = "linear"
    mm_use_im_start_end: bool = False
    mm_use_im_patch_token: bool = True
    mm_patch_merge_type: Optional[str] = "flat"
    mm_vision_select_feature: Optional[str] = "patch"


@dataclass
class DataArguments:
    lazy_preprocess: bool = False
    is_multimodal: bool = False
    image_token_len: int = 0
    image_folder: Optional[str] = None
    image_aspect_ratio: str = "square"
    parquet: bool = False
    data_source_names: str = "unimm-chat"
    data_source_weights: str = "100"
    eval_data_source_names: Optional[str] = None
    data_dir: str = "./RLAIF-V-Dataset/"
    kto_win_data_source_names: str = "100"
    kto_win_data_source_weights: str = "100"
    kto_rej_data_source_names: str = "100"
    kto_rej_data_source_weights: str = "100"
    dpo_beta: float = 0.5
    dpo_token_weight: float = 3.0
    shuffle_data: bool = True


@dataclass
class TrainingArguments(_CoreTrainingArguments):
    """The DPO-relevant fields live in trainer.TrainingArguments; these are the remaining flags of the shipped scripts."""
    task: str = "LM"                       # reference default; the scripts pass --task DPO
    model_max_length: int = 512
    max_steps: int = 1000
    cache_dir: Optional[str] = None
    optim: str = "adamw_torch"
    freeze_mm_mlp_adapter: bool = False
    mm_projector_lr: Optional[float] = None
    num_train_epochs: float = 3.0
    per_device_eval_batch_size: int = 8
    save_strategy: str = "steps"
    dataloader_num_workers: int = 0
    resume_from_checkpoint: Optional[str] = None


def _str2bool(v: str) -> bool:
    if v.lower() in ("true", "1", "yes"):
        return True
    if v.lower() in ("false", "0", "no"):
        return False
    raise argparse.ArgumentTypeError(f"boolean expected, got {v!r}")


def parse_args(argv=None):
    """HfArgumentParser semantics for the three dataclasses (`--flag value`, booleans as True/False); unknown flags of
    the HF / DeepSpeed control plane are reported and ignored."""
    parser = argparse.ArgumentParser(allow_abbrev=False)
    owners: Dict[str, int] = {}
    classes = (ModelArguments, DataArguments, TrainingArguments)
    for ci, cls in enumerate(classes):
        for f in dataclasses.fields(cls):
            if f.name in owners:
                continue
            owners[f.name] = ci
            t = f.type if isinstance(f.type, str) else getattr(f.type, "__name__", str(f.type))
            kind = _str2bool if "bool" in t else float if "float" in t else int if "int" in t else str
            if kind is _str2bool:
                parser.add_argument(f"--{f.name}", type=kind, nargs="?", const=True, default=f.default)
            else:
                parser.add_argument(f"--{f.name}", type=kind, default=f.default)
    ns, unknown = parser.parse_known_args(argv)
    if unknown and int(os.environ.get("RANK", "0")) == 0:
        print(f"[train_llava15] ignoring control-plane flags: {' '.join(unknown)}", file=sys.stderr)
    out = []
    for ci, cls in enumerate(classes):
        out.append(cls(**{f.name: getattr(ns, f.name) for f in dataclasses.fields(cls)}))
    return tuple(out)


def make_dpo_data_module(tokenizer, data_args: DataArguments, reference_model=None) -> Dict:
    """train_llava15.py:148-195: DPODataset over the `*logp*.parquet` rows + DataCollatorForDPODataset."""
    from .data import DataCollatorForDPODataset
    from .dataset import DPODataset
    mm_cfg = dict(is_multimodal=data_args.is_multimodal, image_token_len=data_args.image_token_len,
                  image_folder=data_args.image_folder, image_aspect_ratio=data_args.image_aspect_ratio,
                  use_im_start_end=getattr(data_args, "mm_use_im_start_end", False),
                  image_processor=getattr(data_args, "image_processor", None),
                  data_source_names=data_args.data_source_names, data_source_weights=data_args.data_source_weights,
                  shuffle_data=data_args.shuffle_data)
    train_dataset = DPODataset(tokenizer=tokenizer, data_dir=data_args.data_dir, multimodal_cfg=mm_cfg,
                               reference_model=reference_model)
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"Train data size is {len(train_dataset)}", flush=True)
    collator = DataCollatorForDPODataset(tokenizer=tokenizer, beta=data_args.dpo_beta, mod_token_weight=data_args.dpo_token_weight)
    return dict(train_dataset=train_dataset, eval_dataset=None, data_collator=collator)


def init_model(model_args: ModelArguments, data_args: DataArguments, training_args: TrainingArguments, tokenizer=None,
               device: Optional[str] = None):
    """train_llava15.py:198-290 / train_llava15_lora.py:286-384.  `tokenizer` may be injected (tests, pre-tokenised
    pipelines); otherwise the slow LLaMA tokenizer of the checkpoint directory is loaded exactly as the reference does."""
    from .checkpoint import from_pretrained
    from .image import RawImageProcessor
    if model_args.version != "llava_v1":
        raise NotImplementedError("only the llava_v1 conversation version is used by the shipped scripts")
    if training_args.task != "DPO":
        raise NotImplementedError("task must be DPO (train_llava15.py:303-304 raises for LM)")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = device or f"cuda:{local}"
    vt_dir = model_args.vision_tower if model_args.vision_tower and os.path.isdir(model_args.vision_tower) else None
    overrides = dict(model_max_length=training_args.model_max_length, vision_tower_dir=vt_dir)
    overrides.update(getattr(model_args, "config_overrides", None) or {})      # tests: tiny CLIP shapes
    model = from_pretrained(model_args.model_name_or_path, device=device, lora=training_args.lora_config(), **overrides)
    if model_args.mm_vision_select_layer is not None and model_args.mm_vision_select_layer != -1:
        assert model.cfg.select_layer == model_args.mm_vision_select_layer, "checkpoint and --mm_vision_select_layer disagree"
    if training_args.gradient_checkpointing:
        model.gradient_checkpointing = True
    if tokenizer is None:
        import transformers
        tokenizer = transformers.AutoTokenizer.from_pretrained(model_args.model_name_or_path, cache_dir=training_args.cache_dir,
                                                               model_max_length=training_args.model_max_length,
                                                               padding_side="right", use_fast=False, truncation_side="right")
        tokenizer.pad_token = tokenizer.unk_token                       # "for llava 1.5" (:226-228)
    # the workers only decode; CLIPImageProcessor's resize / crop / normalise run on the device (image.py)
    data_args.image_processor = RawImageProcessor(size=model.cfg.image_size)
    data_args.is_multimodal = True
    data_args.mm_use_im_start_end = model_args.mm_use_im_start_end
    data_module = make_dpo_data_module(tokenizer, data_args, reference_model=model)
    return model, data_module, tokenizer


def safe_save_model_for_hf_trainer(trainer: LLaVA15DPOTrainer, output_dir: str):
    """train_llava15.py:102-112 (full fine-tune: HF-named safetensors) / train_llava15_lora.py:184-197 (adapter +
    non_lora_trainables.bin) - both behind trainer._save."""
    torch.cuda.synchronize()
    trainer._save(output_dir)


def train(argv=None, tokenizer=None):
    from .dist import init_process_group_from_env, make_reducer
    model_args, data_args, training_args = parse_args(argv)
    rank, local, world = init_process_group_from_env()
    data_args.data_source_names = data_args.data_source_names.split("#")
    data_args.data_source_weights = [int(x) for x in data_args.data_source_weights.split("#")]
    model, data_module, tokenizer = init_model(model_args, data_args, training_args, tokenizer=tokenizer)
    reducer = make_reducer(model.store.flat_g) if world > 1 else None       # RV_ZERO1=1: opt-in sharded optimizer (dist.py)
    trainer = LLaVA15DPOTrainer(model=model, tokenizer=tokenizer, args=training_args, reducer=reducer, **data_module)
    ckpts = sorted(glob.glob(os.path.join(training_args.output_dir, "checkpoint-*")),
                   key=lambda p: int(p.rsplit("-", 1)[1]) if p.rsplit("-", 1)[1].isdigit() else -1)
    if training_args.resume_from_checkpoint or ckpts:
        print("Resume from checkpoint.")
        trainer.train(resume_from_checkpoint=training_args.resume_from_checkpoint or ckpts[-1])
    else:
        print("Train from start.")
        trainer.train()
    trainer.save_state()
    safe_save_model_for_hf_trainer(trainer, training_args.output_dir)
    return trainer


if __name__ == "__main__":
    train()
