"""CPU restatement of the image preprocessing on the DPO input path - TEST INFRASTRUCTURE ONLY (imported by tests/,
__graft_entry__.smoke() and nothing under rlaif-v_amd/).

The reference hands every PIL image to ``vision_tower.image_processor`` (muffin/train/train_llava15.py:244,
muffin/train/train_utils.py:208), a third-party ``transformers.CLIPImageProcessor`` (pin 4.35.0, pyproject.toml:16-23;
not vendored) configured by openai/clip-vit-large-patch14-336: convert RGB -> resize shortest edge to 336 with
PIL BICUBIC -> center crop 336x336 -> x 1/255 -> (x - mean) / std, float32 CHW.  The resize is Pillow's two-pass
antialiased convolution in 22-bit fixed point with uint8 rounding after EACH pass (src/libImaging/Resample.c:
precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc), restated here with numpy.

Pinned against the real thing in the build container (PIL 12.2 + the installed CLIPImageProcessor) by
tests/golden/make_image_golden.py -> tests/golden/image_preprocess.npz; tests/test_image_preprocess.py replays it
bit for bit."""
import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # transformers OPENAI_CLIP_MEAN / OPENAI_CLIP_STD
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, in0: float, in1: float, out_size: int) -> Tuple[int, np.ndarray, np.ndarray]:
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter (support 2).
    Returns (ksize, bounds[out_size, 2] = (first source index, tap count), kk[out_size, ksize] int32)."""
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)            # C (int) cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize_bicubic_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """ImagingResample (two passes, horizontal first, uint8 between the passes) on an [H, W, C] uint8 image."""
    H, W, C = img.shape
    need_h, need_v = out_w != W, out_h != H
    _, bh, kh = precompute_coeffs(W, 0.0, float(W), out_w)
    _, bv, kv = precompute_coeffs(H, 0.0, float(H), out_h)
    src = img
    if need_h:
        y0 = int(bv[0, 0])
        y1 = int(bv[out_h - 1, 0] + bv[out_h - 1, 1])
        bv = bv.copy()
        bv[:, 0] -= y0
        rows = src[y0:y1].astype(np.int64)
        tmp = np.empty((y1 - y0, out_w, C), dtype=np.uint8)
        for xx in range(out_w):
            x0, n = int(bh[xx, 0]), int(bh[xx, 1])
            acc = (rows[:, x0:x0 + n, :] * kh[xx, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            tmp[:, xx, :] = _clip8(acc)
        src = tmp
    if need_v:
        s64 = src.astype(np.int64)
        out = np.empty((out_h, src.shape[1], C), dtype=np.uint8)
        for yy in range(out_h):
            r0, n = int(bv[yy, 0]), int(bv[yy, 1])
            acc = (s64[r0:r0 + n] * kv[yy, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        src = out
    return src.copy() if not (need_h or need_v) else src


def resize_output_size(h: int, w: int, size: int = 336) -> Tuple[int, int]:
    """transformers get_resize_output_image_size(default_to_square=False): shortest edge -> size. Returns (h, w)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def normalize_table(mean=CLIP_MEAN, std=CLIP_STD, scale: float = 1 / 255) -> np.ndarray:
    """[3, 256] float32: transformers rescale (float64 multiply, cast to float32) then normalize (float32 arithmetic)."""
    v = (np.arange(256, dtype=np.uint8).astype(np.float64) * scale).astype(np.float32)
    m, s = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32)
    return ((v[None, :] - m[:, None]) / s[:, None]).astype(np.float32)


def clip_preprocess(img: np.ndarray, size: int = 336) -> np.ndarray:
    """[H, W, 3] uint8 RGB -> [3, size, size] float32, the whole CLIPImageProcessor chain."""
    H, W, _ = img.shape
    oh, ow = resize_output_size(H, W, size)
    r = pil_resize_bicubic_u8(img, ow, oh)
    top, left = (oh - size) // 2, (ow - size) // 2
    crop = r[top:top + size, left:left + size]
    tab = normalize_table()
    return np.stack([tab[c][crop[:, :, c]] for c in range(3)], 0)
