"""OmniLMM branch of the DPO step (BASELINE config 4): the reference's
``OmniLMMForCausalLM`` (omnilmm/model/omnilmm.py:268-340) under ``forward_DPO`` (muffin/train/trainers.py:66-88):

    tower tokens -> Resampler (64 learned queries, resampler.py:96-168) -> REPLACE the <im_patch> embeddings between
    <im_start> and <im_end> (omnilmm.py:221-257) -> Mistral decoder (= the Llama kernels with grouped-query attention)
    -> get_batch_logps on the unchanged labels -> dpo_loss.

Everything downstream of the tower features is the same HIP path as LLaVA-1.5 (``LlavaDPOModel``): this module only swaps
the vision-to-language adapter (projector -> ``resampler.Resampler``), its parameters in the flat store, and the row rule of
the splice planner.  Pinned by tests/golden/omnilmm_tiny.pt (the reference's own classes).

The vision tower: timm's ``eva02_enormous_patch14_clip_224`` (omnilmm.py:31-43) is not vendored in the reference and timm is
absent offline, so no oracle of it can be pinned; the tower is frozen in this path (``OmniLMMConfig.tune_clip = False``, the
reference's initialize_vision_modules default; the constructor default tune_clip=True would train it - a documented
deviation), its output is a pure function of the
image, and ``images`` may therefore be handed over as PRECOMPUTED tower tokens [B, N, width] (3-D tensor) - which is also
what one would cache across the 4 epochs of a run.  Pixel input goes through a tower registered with ``set_vision_tower``
(any callable pixels -> [B, N, width]); ``rlaif-v_amd/eva_tower.py`` restates timm's EVA02-E/14 on the HIP kernels - PARITY
UNPINNED (checked only against the author's own torch restatement, tests/test_omnilmm_gpu.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch

from . import ops
from .model import BF16, LlavaConfig, LlavaDPOModel, LoraConfig
from .resampler import PREFIX, Resampler, param_shapes
from .splice import make_omnilmm_splicer


@dataclass
class OmniLMMConfig(LlavaConfig):
    """OmniLMM-12B defaults (Mistral-7B LM with 8 key/value heads and f = 14336, ``num_query`` 64,
    448-px images, EVA02-E/14 width 1792; three added tokens <im_patch>, <im_start>, <im_end>)."""
    ffn: int = 14336
    kv_heads: Optional[int] = 8
    vocab: int = 32009                  # 32000 + <im_patch> <im_start> <im_end> + 6 box / ref / quad tokens (omnilmm.py:388-420)
    image_size: int = 448
    num_query: int = 64
    vision_width: int = 1792
    im_patch_token: int = 32000
    im_start_token: int = 32001
    im_end_token: int = 32002
    # How the reference holds its tower (omnilmm/model/omnilmm.py): ``tune_clip=False`` - the default of
    # initialize_vision_modules (:74, :93) - keeps it in a plain list, OUT of the module's parameters: no optimizer ever
    # sees it although, lacking a no_grad (:107-119), gradients still flow into it; ``tune_clip=True`` - the CONSTRUCTOR
    # default (:58, :69-70; chat.py loads inference checkpoints that way) - registers it and a trainer built on
    # model.parameters() then trains its 4.3 B weights.  This path implements tune_clip=False: the tower is frozen, its
    # output is a pure function of the pixels (which is also what makes precomputed tower tokens legal input).
    tune_clip: bool = False

    arch = "omnilmm"

    @property
    def n_image_tokens(self) -> int:
        return self.num_query

    def vision_param_entries(self):
        """The Resampler's tensors in backward-completion order.  Weight decay follows HF Trainer's rule (all but biases
        and nn.LayerNorm weights): query, kv_proj, in_proj_weight, out_proj.weight and proj decay."""
        shp = param_shapes(self.hidden, self.vision_width, self.num_query)
        decay = ["proj", "attn.out_proj.weight", "attn.in_proj_weight", "query", "kv_proj.weight"]
        nodecay = ["ln_post.weight", "ln_post.bias", "attn.out_proj.bias", "attn.in_proj_bias", "ln_q.weight", "ln_q.bias",
                   "ln_kv.weight", "ln_kv.bias"]
        return ([(PREFIX + k, shp[k], False) for k in decay], [(PREFIX + k, shp[k], False) for k in nodecay])


class OmniLMMDPOModel(LlavaDPOModel):
    def __init__(self, cfg: OmniLMMConfig, device="cuda:0", with_optimizer: bool = True, lora: Optional[LoraConfig] = None):
        if cfg.tune_clip:
            raise NotImplementedError("OmniLMMConfig.tune_clip=True (the tower as a trained submodule, omnilmm/model/omnilmm.py:58,"
                                      "69-70) is not implemented: this path keeps the EVA02 tower frozen, i.e. the reference's "
                                      "tune_clip=False arrangement (initialize_vision_modules default, :74,:93)")
        super().__init__(cfg, device, with_optimizer, lora)
        self.resampler = Resampler(cfg.hidden, cfg.vision_width, cfg.num_query, self.device)
        self._tower: Optional[Callable[[torch.Tensor], torch.Tensor]] = None

    def set_vision_tower(self, fn: Callable[[torch.Tensor], torch.Tensor]):
        """fn(pixels [B, 3, H, W]) -> tower tokens [B, N, vision_width] with the prefix tokens already stripped
        (``get_vision_embedding``, omnilmm.py:107-119)."""
        self._tower = fn

    # ---- weights: the tower is external, the store holds LM + resampler
    def _load_tower(self, sd: Dict[str, torch.Tensor]):
        pass

    def _init_tower(self, g: torch.Generator, std: float):
        d = self.cfg.hidden
        proj = self.store.p(PREFIX + "proj")                  # (embed_dim ** -0.5) * randn, resampler.py:131-132
        proj.copy_((torch.randn(d, d, device=self.device, generator=g) * d ** -0.5).to(BF16))
        self.store.sync_master_from_params()

    def _P(self, name: str) -> torch.Tensor:
        return self.store.p(PREFIX + name)

    def _G(self, name: str) -> torch.Tensor:
        return self.store.g(PREFIX + name)

    # ---- vision-to-language adapter
    def encode_images(self, images, ctx: Optional[dict] = None) -> torch.Tensor:
        """``get_vision_embedding`` (omnilmm.py:107-119) for the B DISTINCT images of the batch (the reference encodes
        cat([images, images]), trainers.py:190: rows i and B+i are identical)."""
        cfg = self.cfg
        if torch.is_tensor(images) and images.dim() == 3:
            tok = images
        elif self._tower is not None:
            tok = self._tower(images)
        else:
            raise NotImplementedError("OmniLMM pixel input needs a vision tower (set_vision_tower); pass precomputed tower "
                                      "tokens [B, N, vision_width] instead - the tower is frozen in this path")
        B, N, w = tok.shape
        if w != cfg.vision_width:
            raise ValueError(f"tower width {w} != cfg.vision_width {cfg.vision_width}")
        x = tok.to(self.device, dtype=BF16).reshape(B * N, w).contiguous()
        return self.resampler.forward(x, B, N, self._P, ctx)

    def _vision_backward(self, dfeat: torch.Tensor, ctx: dict):
        self.resampler.backward(dfeat, ctx, self._P, self._G)

    def _row_splicer(self):
        c = self.cfg
        return make_omnilmm_splicer(c.im_patch_token, c.im_start_token, c.im_end_token)

    def non_lora_trainables(self) -> Dict[str, torch.Tensor]:
        return {"base_model.model." + k: self.store.p(k).detach().cpu().clone() for k in self.store.trainable
                if k.startswith(PREFIX)}
