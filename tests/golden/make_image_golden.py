"""Golden vectors for the image preprocessing: outputs of the REAL chain (PIL + transformers CLIPImageProcessor as
configured for openai/clip-vit-large-patch14-336) on seeded synthetic images.  Run in the build container:
    python tests/golden/make_image_golden.py
Writes tests/golden/image_preprocess.npz (inputs are regenerated from the seeds; the SHA-256 of every output, and for two
cases the full float output + the PIL-resized uint8, are stored)."""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(480, 640), (640, 480), (336, 336), (200, 150), (97, 411), (1000, 352), (337, 900), (768, 1024)]   # (H, W)


def make_image(h, w, seed):
    """Smooth structure + noise so that both the low-pass and the clipping paths of the filter are exercised."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 120 * np.sin(xx / 7.0 + seed)[..., None] * np.cos(yy / 11.0)[..., None] * np.array([1.0, 0.7, -0.9])
    img = base + rng.normal(0, 40, (h, w, 3))
    img[: h // 5, : w // 5] = rng.integers(0, 2, (h // 5, w // 5, 3)) * 255          # hard edges -> overshoot -> clip8
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336})
    out = {}
    for i, (h, w) in enumerate(CASES):
        img = make_image(h, w, i)
        px = proc(Image.fromarray(img))["pixel_values"][0]
        px = np.ascontiguousarray(np.asarray(px, dtype=np.float32))
        out[f"sha256_{i}"] = np.frombuffer(hashlib.sha256(px.tobytes()).digest(), dtype=np.uint8)
        if i in (0, 3):
            out[f"pixel_values_{i}"] = px
            sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
            from oracle.clip_preprocess_oracle import resize_output_size
            oh, ow = resize_output_size(h, w)
            out[f"resized_{i}"] = np.asarray(Image.fromarray(img).resize((ow, oh), resample=Image.BICUBIC))
    out["cases"] = np.array(CASES)
    np.savez_compressed(os.path.join(HERE, "image_preprocess.npz"), **out)
    print("wrote", os.path.join(HERE, "image_preprocess.npz"))


if __name__ == "__main__":
    main()
