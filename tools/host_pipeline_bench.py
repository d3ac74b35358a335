#!/usr/bin/env python
"""Host side of the DPO input pipeline against what the GPUs consume (VERDICT r3 missing 7 / next 6).

One MI355X takes 8.1 pairs/s at BASELINE config 2, a node of eight 65 pairs/s.  The reference feeds that with 16 DataLoader
workers per process (script/train/llava15_train.sh:44) running, per pair:

    JPEG bytes -> PIL decode -> RGB                       muffin/data/datasets.py:59-91        rlaif_v_amd/dataset.py
    image processor (here: RawImageProcessor, uint8 HWC;  muffin/train/train_utils.py:198-263  rlaif_v_amd/image.py
      resize / crop / normalise run on the GPU)
    preprocess_v1 (conversation template, 2 + 2 x rounds  llava/train/train.py (via train_utils)  rlaif_v_amd/dataset.py
      tokenizer calls per answer)
    DataCollatorForDPODataset incl. get_diff_ids          muffin/train/train_muffin.py:43-112   rlaif_v_amd/data.py
      (difflib.SequenceMatcher over the two answers)
    build_packed_plan (splice tables of the batch)         llava/model/llava_arch.py:150-330     rlaif_v_amd/splice.py

CPU only (runs in the build container and on a GPU box's host).  Synthetic but shaped like RLAIF-V: JPEGs of COCO-like sizes
(quality 90, smooth content so the entropy decode is realistic), questions of 8-20 words, answers of 40-160 words, the
rejected answer = the chosen one with a few sentences rewritten (what get_diff_ids is for), a sentencepiece BPE tokenizer
trained on the synthetic corpus (the reference uses the slow LlamaTokenizer over sentencepiece).

    python tools/host_pipeline_bench.py [--pairs 256] [--workers 1,4,8,16] [--batch 8] [--out profiles/r04_host_pipeline.json]
"""
import argparse
import io
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORDS = ("the a an of on in at with and or but near under over behind beside man woman child dog cat horse bird car bus train street "
         "table chair plate food pizza cake cup bottle window door tree grass sky cloud water beach mountain building sign light "
         "red blue green yellow black white large small old young two three several many standing sitting walking holding looking "
         "eating riding playing wearing parked placed visible background foreground left right center image picture scene shows "
         "appears there is are has have which while also quite very slightly clearly probably").split()
SIZES = [(640, 480), (640, 427), (500, 375), (480, 640), (427, 640), (640, 640), (1024, 768), (333, 500)]


def make_corpus(n, rng):
    def sent(lo, hi):
        k = int(rng.integers(lo, hi))
        return " ".join(WORDS[int(i)] for i in rng.integers(0, len(WORDS), k)).capitalize() + " ."
    rows = []
    for i in range(n):
        q = sent(8, 20).replace(" .", " ?")
        sents = [sent(8, 22) for _ in range(int(rng.integers(4, 9)))]
        rej = list(sents)
        for j in rng.choice(len(sents), size=max(1, len(sents) // 3), replace=False):
            rej[int(j)] = sent(8, 22)
        rows.append(dict(question=q, chosen=" ".join(sents), rejected=" ".join(rej)))
    return rows


def make_jpeg(rng, size):
    from PIL import Image, ImageFilter
    w, h = size
    small = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8)
    img = Image.fromarray(small).resize((w, h), Image.BICUBIC).filter(ImageFilter.GaussianBlur(1.5))
    noise = rng.integers(-12, 13, (h, w, 3))
    img = Image.fromarray(np.clip(np.asarray(img).astype(np.int16) + noise, 0, 255).astype(np.uint8))
    buf = io.BytesIO()
    img.save(buf, format="JPEG", quality=90)
    return buf.getvalue()


class SpmTokenizer:
    """sentencepiece BPE behind the surface preprocess_v1 uses (llama-style: BOS first, no EOS)."""
    legacy = True

    def __init__(self, model_file, model_max_length=2048):
        import sentencepiece as spm
        self.sp = spm.SentencePieceProcessor(model_file=model_file)
        self.bos_token_id, self.eos_token_id, self.pad_token_id, self.unk_token_id = 1, 2, 0, 0
        self.model_max_length = model_max_length

    def _ids(self, text):
        # the HF slow tokenizer splits the text at special tokens first and runs sentencepiece on every piece (legacy mode)
        ids = [self.bos_token_id]
        parts = text.split("</s>")
        for i, part in enumerate(parts):
            if part:
                ids += self.sp.encode(part)
            if i + 1 < len(parts):
                ids.append(self.eos_token_id)
        return ids

    def __call__(self, text, return_tensors=None, padding=None, max_length=None, truncation=None):
        import types
        if isinstance(text, str):
            return types.SimpleNamespace(input_ids=self._ids(text))
        rows = [self._ids(t) for t in text]
        if truncation and max_length:
            rows = [r[:max_length] for r in rows]
        n = max(len(r) for r in rows)
        return types.SimpleNamespace(input_ids=torch.tensor([r + [self.pad_token_id] * (n - len(r)) for r in rows]))


def train_tokenizer(rows, tmp):
    import sentencepiece as spm
    path = os.path.join(tmp, "corpus.txt")
    with open(path, "w") as f:
        for r in rows:
            f.write(r["question"] + "\n" + r["chosen"] + "\n" + r["rejected"] + "\n")
        f.write("A chat between a curious human and an artificial intelligence assistant . USER : ASSISTANT : </s>\n")
    spm.SentencePieceTrainer.train(input=path, model_prefix=os.path.join(tmp, "tok"), vocab_size=400, model_type="bpe",
                                   pad_id=0, unk_id=3, bos_id=1, eos_id=2, user_defined_symbols=["<image>"], minloglevel=2)
    return os.path.join(tmp, "tok.model")


class Rows(torch.utils.data.Dataset):
    """RLAIFVDataset rows held in memory + DPODataset.__getitem__ (the parquet read happens once at start-up)."""

    def __init__(self, rows, tok_file):
        self.rows, self.tok_file, self.tok = rows, tok_file, None

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i, stages=None):
        from rlaif_v_amd.dataset import bytes_to_PIL_image, encode_multimodal_preference_sample, preprocess_v1
        from rlaif_v_amd.image import RawImageProcessor
        if self.tok is None:
            self.tok = SpmTokenizer(self.tok_file)
            self.proc = RawImageProcessor(336)
        s = self.rows[i]
        t0 = time.perf_counter()
        img = bytes_to_PIL_image(s["image"])
        t1 = time.perf_counter()
        src = dict(image=img, question={"from": "human", "value": f"<image>\n{s['question']}"},
                   chosen={"from": "gpt", "value": s["chosen"]}, rejected={"from": "gpt", "value": s["rejected"]},
                   ref_win_logp=-100.0, ref_win_avg_logp=-1.0, ref_win_per_token_logp=[0.0] * 2048,
                   ref_rej_logp=-101.0, ref_rej_avg_logp=-1.1, ref_rej_per_token_logp=[0.0] * 2048)
        cfg = dict(image_processor=(lambda im: (im, self.proc(im))[1]), is_multimodal=True, keep_image_tag=True)
        if stages is not None:
            arr = self.proc(img)
            t2 = time.perf_counter()
            cfg["image_processor"] = lambda im: arr
            out = encode_multimodal_preference_sample(src, self.tok, cfg, preprocess_func=lambda a, b: preprocess_v1(a, b, has_image=True))
            t3 = time.perf_counter()
            stages["decode"] += t1 - t0
            stages["raw_processor"] += t2 - t1
            stages["preprocess_v1_x2"] += t3 - t2
            return out
        return encode_multimodal_preference_sample(src, self.tok, cfg, preprocess_func=lambda a, b: preprocess_v1(a, b, has_image=True))


def collate_factory(tok_file):
    from rlaif_v_amd.data import DataCollatorForDPODataset
    import types

    def fix(instances):        # the collator stacks 'image' tensors; RawImageProcessor hands ragged uint8 arrays: keep them as a list
        imgs = [w["image"] for _, w in instances]
        inst = [({**r, "image": torch.zeros(1)}, {**w, "image": torch.zeros(1)}) for r, w in instances]
        batch = DataCollatorForDPODataset(types.SimpleNamespace(pad_token_id=0), 0.1, 1.0)(inst)
        batch["images"] = imgs
        return batch
    return fix


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--workers", default="1,4,8,16")
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r04_host_pipeline.json"))
    args = ap.parse_args()
    torch.set_num_threads(1)
    rng = np.random.default_rng(0)
    tmp = tempfile.mkdtemp(prefix="hostpipe_")
    rows = make_corpus(args.pairs, rng)
    for i, r in enumerate(rows):
        r["image"] = make_jpeg(rng, SIZES[i % len(SIZES)])
    tok_file = train_tokenizer(rows, tmp)
    ds = Rows(rows, tok_file)
    collate = collate_factory(tok_file)
    # ---- per-stage cost, one process
    stages = dict(decode=0.0, raw_processor=0.0, preprocess_v1_x2=0.0)
    items = [ds.__getitem__(i, stages) for i in range(args.pairs)]
    n_tok = [int(w["input_ids"].numel()) for _, w in items]
    t0 = time.perf_counter()
    batches = [collate(items[i:i + args.batch]) for i in range(0, args.pairs - args.batch + 1, args.batch)]
    stages["collate_incl_get_diff_ids"] = time.perf_counter() - t0
    from rlaif_v_amd.splice import build_packed_plan
    t0 = time.perf_counter()
    for b in batches:
        build_packed_plan(b["concatenated_input_ids"], b["concatenated_labels"], 576, args.batch, 2048)
    plan_s = time.perf_counter() - t0
    per_pair_ms = {k: 1e3 * v / args.pairs for k, v in stages.items()}
    per_pair_ms["build_packed_plan (training process, not the workers)"] = 1e3 * plan_s / (len(batches) * args.batch)
    worker_ms = sum(v for k, v in per_pair_ms.items() if "build_packed" not in k)
    rep = dict(host=dict(cpus=os.cpu_count()), pairs=args.pairs, batch=args.batch, jpeg_kb_mean=float(np.mean([len(r["image"]) for r in rows]) / 1024),
               text_tokens_mean=float(np.mean(n_tok)), per_pair_ms=per_pair_ms, one_worker_pairs_per_s_from_stages=1e3 / worker_ms,
               build_packed_plan_us_per_batch=1e6 * plan_s / len(batches), consumers=dict(one_gpu_pairs_per_s=8.1, node_pairs_per_s=65.0))
    print(json.dumps(rep, indent=1), flush=True)
    # ---- DataLoader end to end
    rep["dataloader"] = {}
    for w in [int(x) for x in args.workers.split(",") if x]:
        if w > 2 * (os.cpu_count() or 1):
            continue
        dl = torch.utils.data.DataLoader(ds, batch_size=args.batch, shuffle=False, num_workers=w, collate_fn=collate, drop_last=True,
                                         persistent_workers=w > 0, prefetch_factor=4 if w > 0 else None)
        for _ in dl:            # first pass: worker start-up, tokenizer load
            pass
        t0 = time.perf_counter()
        n = 0
        for _ in range(2):
            for b in dl:
                n += args.batch
        dt = time.perf_counter() - t0
        rep["dataloader"][str(w)] = dict(pairs_per_s=n / dt, per_worker=n / dt / max(w, 1))
        print(f"workers {w}: {n / dt:.1f} pairs/s", flush=True)
        del dl
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(rep, fh, indent=1)


if __name__ == "__main__":
    main()
