"""End to end through the reference's entry-point surface (muffin/train/train_llava15.py flags): checkpoint directory ->
raw preference parquet -> reference-log-prob precompute -> DPODataset (PIL decode, device-side CLIP preprocessing) ->
collator -> LLaVA15DPOTrainer.train() -> save -> resume.  A toy tokenizer stands in for the LLaMA sentencepiece model
(no tokenizer files exist offline); everything else is the shipped code path."""
import io
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from toy_tokenizer import SAMPLES, ToyTokenizer  # noqa: E402
from oracle import dpo_oracle as O  # noqa: E402


def _png(h, w, seed):
    from PIL import Image
    rng = np.random.default_rng(seed)
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(buf, format="PNG")
    return buf.getvalue()


def test_train_entrypoint_end_to_end(tmp_path, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import pandas as pd
    from rlaif_v_amd import train_llava15 as T
    from rlaif_v_amd.checkpoint import from_pretrained, save_pretrained
    from rlaif_v_amd.model import LlavaConfig, LlavaDPOModel
    monkeypatch.setenv("SFT_weight", "0.0")
    monkeypatch.setenv("DPO_weight", "1.0")
    cfg = O.tiny_cfg()
    base = LlavaDPOModel(LlavaConfig(**O.asdict(cfg)), with_optimizer=False)
    base.load_state_dict(O.make_weights(cfg, seed=9))
    ckpt, data_dir, out_dir = str(tmp_path / "llava-tiny"), str(tmp_path / "data"), str(tmp_path / "out")
    save_pretrained(base, ckpt)
    del base
    os.makedirs(data_dir)
    rows = []
    for i in range(6):
        s = SAMPLES[i % len(SAMPLES)]
        rows.append(dict(image={"bytes": _png(40 + 7 * i, 64 - 5 * i, i)}, question=s["question"], chosen=s["chosen"],
                         rejected=s["rejected"], idx=i, origin_dataset="toy", origin_split=json.dumps({"k": i}),
                         image_path=f"img{i}.png"))
    pd.DataFrame(rows).to_parquet(os.path.join(data_dir, "raw_preferences.parquet"))

    tiny_clip = dict(clip_layers=cfg.clip_layers, clip_heads=cfg.clip_heads, clip_ffn=cfg.clip_ffn, image_size=cfg.image_size)
    orig_parse = T.parse_args

    def parse(argv=None):
        m, d, t = orig_parse(argv)
        m.config_overrides = tiny_clip          # the tiny CLIP shapes are not part of the HF llava config
        return m, d, t
    monkeypatch.setattr(T, "parse_args", parse)
    argv = (f"--deepspeed ./script/zero2.json --model_name_or_path {ckpt} --data_dir {data_dir} --image_folder not_used "
            f"--vision_tower openai/clip-vit-large-patch14-336 --mm_use_im_start_end False --fully_tune True --bf16 True "
            f"--mm_projector_type mlp2x_gelu --mm_vision_select_layer -2 --output_dir {out_dir} "
            f"--per_device_train_batch_size 2 --max_steps 3 --learning_rate 1e-4 --weight_decay 0.01 --warmup_ratio 0.05 "
            f"--lr_scheduler_type cosine --logging_steps 1 --save_steps 2 --model_max_length 256 --task DPO "
            f"--dpo_use_average False --dpo_token_weighted False --dpo_token_weight 1.0 --dpo_beta 0.1 --report_to none").split()
    tok = ToyTokenizer()
    trainer = T.train(argv, tokenizer=tok)
    # (1) the reference-log-prob precompute ran and cached its parquet in the reference's naming / column layout
    cached = [f for f in os.listdir(data_dir) if "logp" in f]
    assert cached == ["RLAIF-V-Dataset-withlogp_000-6.parquet"]
    rec = json.loads(pd.read_parquet(os.path.join(data_dir, cached[0])).iloc[0]["logps"])["logps"]
    assert len(rec) == 6 and np.isfinite(rec[0]) and len(rec[2]) > 10
    # (2) three optimizer steps were logged with the reference's metric names; the untrained policy equals the reference
    #     model at step 1, so the first DPO loss is log 2 and the first rewards are ~0
    hist = trainer.state["log_history"]
    assert [h["step"] for h in hist] == [1, 2, 3]
    assert abs(hist[0]["loss"] - float(np.log(2))) < 2e-2 and abs(hist[0]["rewards_train/chosen"]) < 2e-2
    assert {"rewards_train/chosen", "rewards_train/rejected", "rewards_train/accuracies", "rewards_train/margins",
            "logps_train/chosen", "logps_train/rejected", "logps_train/ref_chosen", "logps_train/ref_rejected"} <= set(hist[0])
    # (3) outputs: HF-named safetensors + config + trainer_state + a resumable checkpoint
    assert os.path.exists(os.path.join(out_dir, "config.json")) and os.path.exists(os.path.join(out_dir, "trainer_state.json"))
    assert os.path.isdir(os.path.join(out_dir, "checkpoint-2"))
    m2 = from_pretrained(out_dir, with_optimizer=False, **tiny_clip)
    sd1, sd2 = trainer.model.state_dict(), m2.state_dict()
    assert all(torch.equal(sd1[k], sd2[k]) for k in sd1)
    # (4) a second invocation resumes from the newest checkpoint and stops at max_steps
    t2 = T.train(argv, tokenizer=tok)
    assert t2.state["global_step"] == 3 and [h["step"] for h in t2.state["log_history"]][-1] == 3
