"""Generate golden vectors by executing the REFERENCE ITSELF (/root/reference, read-only) on CPU.

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

What runs: the reference's ``LlavaLlamaForCausalLM`` (llava/model/language_model/llava_llama.py)
with its ``CLIPVisionTower`` + ``mlp2x_gelu`` projector, driven by the reference's
``get_beta_and_logps`` / ``dpo_loss`` (muffin/train/trainers.py:91-275) and the loss mix of
``compute_loss`` (:297-301), then ``backward()``.  The dense math underneath is the installed
transformers 5.15 Llama/CLIP (the reference pins 4.35.0; same arithmetic).
Shims: ``wandb`` stub (trainers.py:6 imports it, never uses it).

Weights come from ``oracle.dpo_oracle.make_weights`` (seeded, bf16-representable) loaded into the
reference model with ``load_state_dict`` so the fixtures need not store them.
"""
import os
import sys
import types
import tempfile

import torch
import transformers  # noqa: F401  (must be imported before the wandb stub)
import accelerate  # noqa: F401  (its wandb probe must run before the stub exists)
from transformers import Trainer, LlamaForCausalLM, CLIPVisionModel  # noqa: F401,E402
import transformers.generation.utils  # noqa: F401,E402

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("wandb", types.ModuleType("wandb"))

from oracle import dpo_oracle as O  # noqa: E402

from transformers import CLIPVisionConfig, CLIPVisionModel  # noqa: E402
from llava.model.language_model.llava_llama import LlavaLlamaForCausalLM, LlavaConfig  # noqa: E402
from muffin.train.trainers import get_beta_and_logps, dpo_loss  # noqa: E402
from muffin.eval.muffin_inference_logp import get_batch_logps  # noqa: E402


def build_reference_model(cfg: O.LlavaCfg, W):
    tmp = tempfile.mkdtemp(prefix="clip_rand_")
    ccfg = CLIPVisionConfig(hidden_size=cfg.clip_hidden, intermediate_size=cfg.clip_ffn,
                            num_hidden_layers=cfg.clip_layers, num_attention_heads=cfg.clip_heads,
                            image_size=cfg.image_size, patch_size=cfg.patch, hidden_act="quick_gelu",
                            layer_norm_eps=cfg.clip_eps, projection_dim=cfg.clip_hidden)
    CLIPVisionModel(ccfg).save_pretrained(tmp)
    # image processor config (CLIPVisionTower.load_model reads it; values unused here)
    import json
    with open(os.path.join(tmp, "preprocessor_config.json"), "w") as f:
        json.dump({"crop_size": cfg.image_size, "size": cfg.image_size, "do_resize": True,
                   "do_center_crop": True, "do_normalize": True,
                   "image_mean": [0.48145466, 0.4578275, 0.40821073],
                   "image_std": [0.26862954, 0.26130258, 0.27577711],
                   "image_processor_type": "CLIPImageProcessor"}, f)
    lcfg = LlavaConfig(hidden_size=cfg.hidden, intermediate_size=cfg.ffn, num_hidden_layers=cfg.layers,
                       num_attention_heads=cfg.heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                       rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                       max_position_embeddings=4096, pad_token_id=None, attn_implementation="eager")
    model = LlavaLlamaForCausalLM(lcfg)
    margs = types.SimpleNamespace(vision_tower=tmp, mm_vision_select_layer=cfg.select_layer,
                                  mm_vision_select_feature="patch", pretrain_mm_mlp_adapter=None,
                                  mm_patch_merge_type="flat", mm_projector_type="mlp2x_gelu")
    model.get_model().initialize_vision_modules(model_args=margs, fsdp=None)
    model.config.tokenizer_model_max_length = cfg.model_max_length   # train_llava15.py:249
    model.config.tokenizer_padding_side = "right"
    sd = model.state_dict()
    # transformers 5.x dropped the `vision_model.` infix of CLIPVisionModel's state dict; the oracle
    # keeps the 4.35 / released-checkpoint names, so map them when loading into the installed version.
    if not any(".vision_model." in k for k in sd):
        W = {k.replace("vision_tower.vision_tower.vision_model.", "vision_tower.vision_tower."): v
             for k, v in W.items()}
    missing = [k for k in sd if k not in W and "rotary" not in k and "position_ids" not in k]
    extra = [k for k in W if k not in sd]
    assert not missing and not extra, (missing[:5], extra[:5])
    model.load_state_dict(W, strict=False)
    model.requires_grad_(True)      # fully_tune (train_llava15.py:268-269)
    model.train()
    return model


def run_case(name, cfg, n_pairs, text_len, prompt_len, seed, dpo_use_average=False, sft_weight=0.0):
    torch.manual_seed(0)
    W = O.make_weights(cfg, seed=seed)
    model = build_reference_model(cfg, {k: v.clone() for k, v in W.items()})
    batch = O.make_synthetic_batch(cfg, n_pairs, text_len, prompt_len, seed=seed)
    args = types.SimpleNamespace(dpo_use_average=dpo_use_average, task="DPO", dpo_token_weighted=False,
                                 past_index=-1)
    data = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}

    # capture the spliced tensors + logits by re-running the two reference calls get_beta_and_logps makes
    with torch.no_grad():
        cat_images = torch.cat([batch["images"], batch["images"]], 0)
        (_, _, _, _, emb, lab) = model.prepare_inputs_labels_for_multimodal(
            input_ids=batch["concatenated_input_ids"], position_ids=None, attention_mask=None,
            past_key_values=None, labels=batch["concatenated_labels"], images=cat_images)
        logits = model.forward(inputs_embeds=emb, labels=None).logits
        per_tok, lp, avg = get_batch_logps(logits, lab, return_all=True)

    pw, pr, rw, rr, beta = get_beta_and_logps(data, model, args, is_llava15=True)
    losses, cw, cr = dpo_loss(pw, pr, rw, rr, beta=beta)
    loss = 1.0 * losses.mean() - sft_weight * pw.mean()     # trainers.py:299-301 with env defaults
    loss.backward()

    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    keep_full = ["model.norm.weight", "model.mm_projector.2.bias", "model.mm_projector.0.bias",
                 "model.layers.0.input_layernorm.weight",
                 f"model.layers.{cfg.layers - 1}.post_attention_layernorm.weight"]
    out = dict(
        cfg=O.asdict(cfg), n_pairs=n_pairs, text_len=text_len, prompt_len=prompt_len, seed=seed,
        dpo_use_average=dpo_use_average, sft_weight=sft_weight,
        labels=lab, embeds_sum=emb.double().sum(-1).float(), image_features_row0=emb[0, :4].clone(),
        logits_lse=torch.logsumexp(logits.float(), -1), per_token_logps=per_tok, log_prob=lp,
        average_log_prob=avg, policy_win_logp=pw.detach(), policy_rej_logp=pr.detach(),
        losses=losses.detach(), chosen_rewards=cw, rejected_rewards=cr, loss=loss.detach(),
        grad_norms={k: float(g.double().norm()) for k, g in grads.items()},
        grad_full={k: grads[k] for k in keep_full if k in grads},
        grad_embed_rowsum=grads["model.embed_tokens.weight"].double().sum(-1).float(),
        clip_has_grad=any("vision_tower" in k for k in grads),
    )
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{name}.pt")
    torch.save(out, path)
    print(name, "loss", float(loss), "logp", lp.tolist(), "->", path, os.path.getsize(path), "bytes")


def fullwidth_cfg() -> "O.LlavaCfg":
    """LLaVA-1.5-7B widths (d 4096, f 11008, V 32000, 32 heads; CLIP-ViT-L/14-336: 1024 wide, 24 layers, 336 px -> 576
    patches) at TWO language-model layers: what the reference runs in about a minute on 8 vCPUs."""
    return O.LlavaCfg(layers=2, model_max_length=2048)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if "--full-width" in sys.argv:
        # production widths through the reference classes themselves: pins the oracle (and through it the HIP path) at
        # the tile shapes the 7B step really uses, not only at hidden 256 / 512
        run_case("fullwidth_l2_b2", fullwidth_cfg(), n_pairs=2, text_len=96, prompt_len=40, seed=21)
        sys.exit(0)
    # grouped-query attention (num_key_value_heads < heads) through the same reference classes
    run_case("tiny_b2_gqa", O.tiny_gqa_cfg(), n_pairs=2, text_len=40, prompt_len=12, seed=4)
    if "--gqa-only" in sys.argv:
        sys.exit(0)
    cfg = O.tiny_cfg()
    run_case("tiny_b2", cfg, n_pairs=2, text_len=40, prompt_len=12, seed=1)
    run_case("tiny_b3_avg_sft", cfg, n_pairs=3, text_len=56, prompt_len=16, seed=2,
             dpo_use_average=True, sft_weight=0.1)
    # truncation edge: model_max_length cuts the answers of the longest rows (llava_arch.py:280-283)
    cfg_t = O.tiny_cfg()
    cfg_t.model_max_length = 48
    run_case("tiny_b2_trunc", cfg_t, n_pairs=2, text_len=40, prompt_len=12, seed=3)
