"""CLIP image preprocessing on the GPU.

The reference runs ``vision_tower.image_processor`` - transformers' CLIPImageProcessor for
openai/clip-vit-large-patch14-336 - on every PIL image inside its DataLoader workers
(muffin/train/train_llava15.py:244, muffin/train/train_utils.py:208): RGB -> PIL BICUBIC resize of the shortest edge
to 336 -> center crop 336 x 336 -> x 1/255 -> (x - mean) / std.  Here the workers only decode (JPEG -> uint8 HWC,
``RawImageProcessor``); resize + crop + normalise run as two small HIP kernels on the training stream
(rv_resize_h_u8, rv_resize_v_norm_u8), bit-identical to PIL + transformers: Pillow's antialiased two-pass convolution
in 22-bit fixed point with uint8 rounding after each pass, and the float32 rescale/normalise arithmetic reproduced
through a 3 x 256 lookup table built with the same numpy operations transformers executes.

Only the host-side tap tables are computed here (float64, same operation order as Pillow's precompute_coeffs so the
integer taps are identical); they depend on (source size, output size) only and are cached."""
from __future__ import annotations

import functools
import math
from typing import List, Sequence, Tuple, Union

import numpy as np
import torch

from . import hip

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class RawImageProcessor:
    """Drop-in for ``multimodal_cfg['image_processor']``: returns the decoded RGB image as uint8 [H, W, 3]; the rest
    of CLIPImageProcessor happens on the device (clip_preprocess_batch)."""

    def __init__(self, size: int = 336):
        self.size = {"shortest_edge": size}
        self.crop_size = {"height": size, "width": size}
        self.image_mean, self.image_std = CLIP_MEAN, CLIP_STD

    def __call__(self, image) -> np.ndarray:
        if hasattr(image, "convert"):
            image = image.convert("RGB")
        arr = np.asarray(image)
        if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[2] != 3:
            raise ValueError(f"RawImageProcessor: expected an RGB uint8 image, got {arr.dtype} {arr.shape}")
        return np.array(arr, dtype=np.uint8, order="C")          # own, writable copy (PIL hands out a read-only view)


def resize_output_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """Shortest edge -> size, the other int(size * long / short)  (get_resize_output_image_size). Returns (h, w)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


@functools.lru_cache(maxsize=512)
def _taps(in_size: int, out_size: int) -> Tuple[int, np.ndarray, np.ndarray]:
    """Pillow precompute_coeffs + normalize_coeffs_8bpc (bicubic, a = -0.5, support 2) for a full-range resize.
    Vectorised over output positions; the tap loop stays sequential so every float64 operation happens in Pillow's
    order.  Returns (ksize, bounds [out, 2] int32 = (first source index, n taps), kk [out, ksize] int32)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    w = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):
        t = np.abs((x + xmin - center + 0.5) * ss)
        a = -0.5
        f = np.where(t < 1.0, ((a + 2.0) * t - (a + 3.0)) * t * t + 1,
                     np.where(t < 2.0, (((t - 5) * t + 8) * t - 4) * a, 0.0))
        f = np.where(x < xmax, f, 0.0)
        w[:, x] = f
        ww = ww + f
    nz = ww != 0.0
    w[nz] = w[nz] / ww[nz, None]
    fixed = w * float(1 << PRECISION_BITS)
    kk = np.where(w < 0, np.trunc(-0.5 + fixed), np.trunc(0.5 + fixed)).astype(np.int32)
    kk[np.arange(ksize)[None, :] >= xmax[:, None]] = 0
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return ksize, bounds, kk


@functools.lru_cache(maxsize=8)
def _norm_table(mean: Tuple[float, ...], std: Tuple[float, ...], scale: float) -> np.ndarray:
    """[3, 256] float32: transformers rescale (float64 product cast to float32) then normalize (float32 arithmetic)."""
    v = (np.arange(256, dtype=np.uint8).astype(np.float64) * scale).astype(np.float32)
    m, s = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32)
    return np.ascontiguousarray(((v[None, :] - m[:, None]) / s[:, None]).astype(np.float32))


_DEV_CACHE = {}


def _dev(key, arr: np.ndarray, device) -> torch.Tensor:
    k = (key, str(device))
    t = _DEV_CACHE.get(k)
    if t is None:
        if len(_DEV_CACHE) > 2048:
            _DEV_CACHE.clear()
        t = torch.from_numpy(arr).to(device)
        _DEV_CACHE[k] = t
    return t


def clip_preprocess_batch(images: Union[Sequence[np.ndarray], torch.Tensor], size: int = 336, device="cuda:0",
                          mean=CLIP_MEAN, std=CLIP_STD, rescale_factor: float = 1 / 255) -> torch.Tensor:
    """uint8 RGB images (list of [H, W, 3] arrays of any sizes, or one [B, H, W, 3] tensor) -> float32
    [B, 3, size, size] on ``device``: exactly CLIPImageProcessor's output."""
    device = torch.device(device)
    n = len(images)
    out = torch.empty(n, 3, size, size, dtype=torch.float32, device=device)
    table = _dev(("tab", mean, std, rescale_factor), _norm_table(tuple(mean), tuple(std), float(rescale_factor)), device)
    for i in range(n):
        img = images[i]
        src = (img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))).to(device).contiguous()
        if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3:
            raise ValueError(f"clip_preprocess_batch: image {i} must be uint8 [H, W, 3], got {src.dtype} {tuple(src.shape)}")
        H, W = int(src.shape[0]), int(src.shape[1])
        oh, ow = resize_output_size(H, W, size)
        top, left = (oh - size) // 2, (ow - size) // 2
        ks_h, b_h, k_h = _taps(W, ow)
        ks_v, b_v, k_v = _taps(H, oh)
        # only the cropped window of the resized image is ever consumed: its columns / rows and the source rows they tap
        bv = b_v[top:top + size]
        y0 = int(bv[0, 0])
        y1 = int(bv[-1, 0] + bv[-1, 1])
        bh_d = _dev(("bh", W, ow, left, size), b_h[left:left + size], device)
        kh_d = _dev(("kh", W, ow, left, size), k_h[left:left + size], device)
        bv_rel = bv.copy()
        bv_rel[:, 0] -= y0
        bv_d = _dev(("bv", H, oh, top, size), bv_rel, device)
        kv_d = _dev(("kv", H, oh, top, size), k_v[top:top + size], device)
        tmp = torch.empty(y1 - y0, size, 3, dtype=torch.uint8, device=device)
        hip.call("rv_resize_h_u8", src, H, W, y0, y1 - y0, bh_d, kh_d, ks_h, size, tmp)
        hip.call("rv_resize_v_norm_u8", tmp, y1 - y0, size, bv_d, kv_d, ks_v, size, table, out[i])
    return out
