"""Golden vectors for the OmniLMM branch, produced by the REFERENCE'S OWN classes on CPU:

    python tests/golden/make_omnilmm_golden.py          (build container only: needs /root/reference)

What runs: ``omnilmm.model.omnilmm.OmniLMMForCausalLM`` (its ``OmniLMMModel.forward`` splice, omnilmm.py:183-265, and its
``Resampler``, resampler.py:96-168, on top of the installed transformers Mistral with 4 query / 2 key-value heads), driven by
``muffin.train.trainers.forward_DPO`` / ``dpo_loss`` (trainers.py:66-126), then ``backward()``.

Shims (modules that do not exist offline and are imported at module scope by the reference): ``timm`` (the EVA02 tower
factory), ``torchvision``, ``cv2``, ``wandb``.  ``timm.create_model`` returns a small deterministic stand-in tower with the
attributes the reference touches (``embed_dim``, ``pos_embed``, ``num_prefix_tokens``, ``blocks``, ``attn_pool``,
``forward_features``); its output features are stored in the fixture, so everything the fixture pins starts AT the tower
features: resampler, splice, Mistral decoder, log-probs, DPO loss and their gradients.  The EVA02 tower stays unpinned.
"""
import math
import os
import sys
import types

import torch
import transformers  # noqa: F401
import accelerate  # noqa: F401
import transformers.generation.utils  # noqa: F401
import datasets  # noqa: F401  (probes torchvision with find_spec: must run before the stub exists)
from transformers import Trainer, MistralForCausalLM, MistralModel, MistralConfig, AutoModelForCausalLM  # noqa: F401,E402

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")


class StubTower(torch.nn.Module):
    """Stand-in for timm's Eva: patch embedding + cls token + learned positions + two residual MLP blocks."""

    def __init__(self, embed_dim=192, patch=14, img=84):
        super().__init__()
        g = torch.Generator().manual_seed(1234)
        self.embed_dim = embed_dim
        self.num_prefix_tokens = 1
        self.attn_pool = None
        n = (img // patch) ** 2
        self.patch = patch
        self.proj = torch.nn.Parameter(0.05 * torch.randn(embed_dim, 3 * patch * patch, generator=g))
        self.cls = torch.nn.Parameter(0.1 * torch.randn(1, 1, embed_dim, generator=g))
        self.pos_embed = torch.nn.Parameter(0.1 * torch.randn(1, n + 1, embed_dim, generator=g))
        self.blocks = torch.nn.ModuleList([torch.nn.Linear(embed_dim, embed_dim) for _ in range(3)])
        for b in self.blocks:
            torch.nn.init.normal_(b.weight, std=0.05, generator=g)
            torch.nn.init.zeros_(b.bias)

    def forward_features(self, px):
        B = px.shape[0]
        p = self.patch
        cols = px.unfold(2, p, p).unfold(3, p, p).permute(0, 2, 3, 1, 4, 5).reshape(B, -1, 3 * p * p)
        x = torch.cat([self.cls.expand(B, -1, -1), cols @ self.proj.t()], 1) + self.pos_embed
        for b in self.blocks:           # the reference replaces blocks[-1] by Identity (omnilmm.py:43)
            x = x + torch.tanh(b(x)) if isinstance(b, torch.nn.Linear) else b(x)
        return x


def install_stubs():
    timm = types.ModuleType("timm")
    timm.models = types.ModuleType("timm.models")
    timm.models.VisionTransformer = type("VisionTransformer", (), {})
    timm.create_model = lambda *a, **k: StubTower()
    timm.data = types.ModuleType("timm.data")
    timm.data.transforms = types.ModuleType("timm.data.transforms")
    timm.data.transforms.RandomResizedCropAndInterpolation = object
    timm.data.constants = types.ModuleType("timm.data.constants")
    timm.data.constants.IMAGENET_INCEPTION_MEAN = (0.5, 0.5, 0.5)
    timm.data.constants.IMAGENET_INCEPTION_STD = (0.5, 0.5, 0.5)
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.InterpolationMode = types.SimpleNamespace(BICUBIC="bicubic")
    for name, mod in (("timm", timm), ("timm.models", timm.models), ("timm.data", timm.data),
                      ("timm.data.transforms", timm.data.transforms), ("timm.data.constants", timm.data.constants),
                      ("torchvision", tv), ("torchvision.transforms", tv.transforms), ("cv2", types.ModuleType("cv2")),
                      ("wandb", types.ModuleType("wandb"))):
        sys.modules.setdefault(name, mod)


install_stubs()
from oracle import dpo_oracle as O  # noqa: E402
from oracle import omnilmm_oracle as OO  # noqa: E402
from omnilmm.model.omnilmm import OmniLMMForCausalLM, OmniLMMConfig  # noqa: E402
from omnilmm.model.resampler import Resampler  # noqa: E402
from muffin.train.trainers import forward_DPO, dpo_loss  # noqa: E402

NUM_QUERY, KV_DIM, IMG = 16, 192, 84


def compress_grads(grads):
    """Small tensors in full; large ones as (norm, seeded random projection, leading 8 x 64 block) - what the tests compare."""
    out = {}
    for k, g in grads.items():
        g = g.detach().float()
        if g.numel() <= 16384:
            out[k] = dict(full=g.clone())
        else:
            r = torch.randn(g.shape, generator=torch.Generator().manual_seed(99))
            g2 = g.reshape(g.shape[0], -1)
            out[k] = dict(norm=float(g.double().norm()), proj=float((g.double() * r.double()).sum()),
                          block=g2[:8, :64].clone())
    return out
TOKENS = (322, 323, 324)            # <im_patch>, <im_start>, <im_end>: the last ids of a 325-token toy vocabulary (NOT a multiple of 64: padded-head path)


def tiny_cfg() -> O.LlavaCfg:
    return O.LlavaCfg(hidden=512, layers=2, heads=4, kv_heads=2, ffn=768, vocab=325, model_max_length=256)


def lm_weights(cfg, seed=3):
    W = O.make_weights(cfg, seed=seed)
    return {k: v for k, v in W.items() if "vision_tower" not in k and "mm_projector" not in k}


def main():
    torch.manual_seed(0)
    cfg = tiny_cfg()
    ocfg = OmniLMMConfig(hidden_size=cfg.hidden, intermediate_size=cfg.ffn, num_hidden_layers=cfg.layers,
                         num_attention_heads=cfg.heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, max_position_embeddings=4096,
                         sliding_window=None, pad_token_id=None, attn_implementation="eager",
                         mm_vision_tower="stub", num_query=NUM_QUERY, image_size=IMG)
    model = OmniLMMForCausalLM(ocfg).float()
    W = lm_weights(cfg)
    W.update(OO.make_resampler_weights(cfg.hidden, KV_DIM, NUM_QUERY))
    sd = dict(W)
    # frozen buffers / tower of the reference module keep their own values
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("vision_tower" in m) or m.endswith("resampler.pos_embed") for m in missing), missing
    vc = model.model.vision_config
    vc.im_patch_token, vc.im_start_token, vc.im_end_token = TOKENS
    vc.use_im_start_end = True
    assert isinstance(model.model.resampler, Resampler)
    model.train()

    out = {}
    # ---- case 1: the Resampler alone (both position-table branches: 36 tower tokens -> interpolated, 16 -> as is)
    g = torch.Generator().manual_seed(11)
    for n_tok in (36, 16):
        x = torch.randn(3, n_tok, KV_DIM, generator=g).to(torch.bfloat16).float().requires_grad_(True)
        y = model.model.resampler(x)
        gy = torch.randn(y.shape, generator=g)
        model.zero_grad()
        (y * gy).sum().backward()
        out[f"resampler_{n_tok}"] = dict(x=x.detach(), y=y.detach(), gy=gy, dx=x.grad.clone(),
                                          grads=compress_grads({k: v.grad for k, v in model.model.resampler.named_parameters()
                                                                if v.grad is not None}))
    # ---- case 2: the whole DPO forward / backward through forward_DPO
    batch = OO.make_omnilmm_batch(cfg, n_pairs=2, text_len=72, num_query=NUM_QUERY, tokens=TOKENS, seed=5)
    images = torch.randn(2, 3, IMG, IMG, generator=g)
    with torch.no_grad():
        tf = model.model.vision_tower.forward_features(images)[:, 1:]
    model.zero_grad()
    cat_images = torch.cat([images, images], 0)                                   # trainers.py:190
    logp = forward_DPO(model, batch["concatenated_input_ids"], batch["concatenated_labels"], None, cat_images)
    B = 2
    pw, pr = logp.split([B, B])
    losses, cw, cr = dpo_loss(pw, pr, batch["ref_win_logp"], batch["ref_rej_logp"], beta=batch["beta"])
    loss = losses.mean()
    loss.backward()
    # What the reference does with its tower (omnilmm.py:58,69-70,107-119): under the constructor default tune_clip=True it is
    # a registered submodule and - there being no no_grad around forward_features - RECEIVES gradients; with tune_clip=False
    # (initialize_vision_modules' default) the same module sits in a plain list: still differentiated through, but invisible to
    # model.parameters() and therefore to any optimizer.  The product implements the second arrangement (frozen tower); the
    # fixture records both facts instead of silently dropping the tower's gradients.
    tower_named = [k for k, _ in model.named_parameters() if "vision_tower" in k]
    tower_gnorm = float(sum(float(v.grad.double().pow(2).sum()) for k, v in model.named_parameters()
                            if "vision_tower" in k and v.grad is not None) ** 0.5)
    frozen_style = OmniLMMForCausalLM(ocfg, tune_clip=False)
    tower_meta = dict(tune_clip_true=dict(tower_params_registered=len(tower_named) > 0, tower_receives_grad=tower_gnorm > 0,
                                          tower_grad_norm=tower_gnorm),
                      tune_clip_false=dict(tower_params_registered=any("vision_tower" in k for k, _ in frozen_style.named_parameters()),
                                           tower_is_list=isinstance(frozen_style.model.vision_tower, list)))
    del frozen_style
    grads = compress_grads({k: v.grad for k, v in model.named_parameters() if v.grad is not None and "vision_tower" not in k})
    with torch.no_grad():
        logits = model(input_ids=batch["concatenated_input_ids"], images=cat_images).logits
    out["dpo"] = dict(batch=batch, tower_features=tf, logp=logp.detach(), loss=loss.detach(), losses=losses.detach(),
                      chosen_rewards=cw.detach(), rejected_rewards=cr.detach(), grads=grads, logits=logits[:, :, :].clone())
    out["tower_trainability"] = tower_meta
    out["meta"] = dict(num_query=NUM_QUERY, kv_dim=KV_DIM, tokens=TOKENS, resampler_heads=cfg.hidden // 128,
                       pos_embed=model.model.resampler.pos_embed.detach().clone())
    path = os.path.join(REPO, "tests", "golden", "omnilmm_tiny.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; loss", float(loss))


if __name__ == "__main__":
    main()
